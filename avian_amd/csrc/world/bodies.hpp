// world/bodies.hpp -- fragment of the body of `template <class T> struct World` (avn_world.hip includes it inside the class):
// SolverBodyPlugin's component side: avn_bodies_upload / _download, avn_solver_bodies_download.

    // ---- bodies ------------------------------------------------------------------------------------------
    static constexpr uint32_t DUMMY_SLOTS = 2 * AVN_JOINT_TYPE_COUNT;  // joint_damping::<T>: two fresh DUMMY SolverBodies per joint type
    avn_status bodies_upload(const avn_bodies* b) override {
        slp_world_asleep = slp_world_idle = false;
        bodies_prepared_early = false; constraints_prepared_early = false; slot_clear_pending = false;   // (whatever an aborted step left behind: the new bodies are prepared by the next solver front)
        if (!b || (b->count && (!b->position || !b->rotation || !b->linear_velocity || !b->angular_velocity || !b->inv_mass || !b->inv_inertia_local || !b->rb_type))) {
            error = "bodies_upload: null array"; return AVN_ERR_BAD_ARG;
        }
        uint32_t n = b->count;
        bool moved = false;
        if (despawn_needs_bodies && n != despawn_expected_bodies) { error = "bodies_upload: after avn_despawn exactly the remaining bodies must be uploaded"; return AVN_ERR_STATE; }
        const bool after_despawn = despawn_needs_bodies;   // (the library has renumbered everything it holds for exactly this upload)
        if (pipe_on && have_bodies && n < dw.n_bodies && !after_despawn) {
            // a despawn renumbers bodies: every contact row, colour mask and handle list of the closed loop names body indices.  Nothing is
            // touched; the host ends the loop (avn_pipeline_enable(0)), uploads the new bodies and colliders and starts it again.
            error = "bodies_upload: fewer bodies than before while the closed loop is on: call avn_despawn for the bodies that leave (or avn_pipeline_enable(0), upload, enable again)";
            return AVN_ERR_STATE;
        }
        if (n + DUMMY_SLOTS > cap_bodies || !have_bodies) {
            HIPCHK(hipStreamSynchronize(stream));
            size_t c = (size_t)std::max<uint32_t>(n + DUMMY_SLOTS, cap_bodies + cap_bodies / 2);  // + the virtual DUMMY bodies of joint_damping
            GROW(b_pos, c, dw.pos); GROW(b_rot, c, dw.rot); GROW(b_lvel, c, dw.lvel); GROW(b_avel, c, dw.avel); GROW(b_com, c, dw.com);
            GROW(b_iloc_a, c, dw.iloc_a); GROW(b_iloc_b, c, dw.iloc_b); GROW(b_acc_l, c, dw.acc_l); GROW(b_acc_a, c, dw.acc_a); GROW(b_bmeta, c, dw.bmeta);
            GROW(b_sb_vel, 2 * c, dw.sb_lin.p); dw.sb_ang.p = dw.sb_lin.p + 1;   // Pair2 slots (avn_device.h)
            GROW(b_sb_delta, 2 * c, dw.sb_dp.p); dw.sb_dq.p = dw.sb_dp.p + 1;
            GROW(b_si, 2 * c, dw.si_a.p); dw.si_b.p = dw.si_a.p + 1;
            GROW(b_vid_l, c, dw.vid_l); GROW(b_vid_a, c, dw.vid_a);
            GROW(b_pre_dp, c, dw.pre_dp); GROW(b_pre_dq, c, dw.pre_dq); GROW(b_sb_flags, c, dw.sb_flags);
            cap_bodies = (uint32_t)c;
            if (pipe_dev) { avn_status sg = pg_bcol_grow(); if (sg != AVN_OK) return sg; }   // bodies spawned inside the closed loop: their colour masks start empty
        }
        if (dw.n_bodies != n && dw.lacc_l) { dw.lacc_l = dw.lacc_a = nullptr; graph_valid = false; }   // (header: an upload with another body count drops the local accelerations)
        if (moved || dw.n_bodies != n) graph_valid = false;
        if (have_bodies && n < dw.n_bodies && !after_despawn) {
            // fewer bodies than before: everything that may still index a body >= n is dropped (the host re-uploads it; nothing
            // may gather or schedule out of range meanwhile) -- uploaded manifolds, joints, colliders whose body is gone
            bool bad_m = false, bad_j = false, bad_c = false;
            if (!use_handles) for (size_t i = 0; i < h_m_body1.size() && !bad_m; ++i) bad_m = (uint32_t)h_m_body1[i] >= n || (uint32_t)h_m_body2[i] >= n;
            for (size_t i = 0; i < h_j_body1.size() && !bad_j; ++i) bad_j = (h_j_body1[i] >= 0 && (uint32_t)h_j_body1[i] >= n) || (h_j_body2[i] >= 0 && (uint32_t)h_j_body2[i] >= n);
            for (size_t i = 0; i < h_col_body.size() && !bad_c; ++i) bad_c = h_col_body[i] >= 0 && (uint32_t)h_col_body[i] >= n;
            if (bad_m || (use_handles && dw.n_manifolds)) {
                uint32_t zero[AVN_GRAPH_COLOR_COUNT + 1] = {0};
                dw.n_manifolds = 0; h_m_body1.clear(); h_m_body2.clear();
                set_color_offsets(zero);
                HIPCHK(hipMemcpyAsync(dw.color_offsets, zero, sizeof zero, hipMemcpyHostToDevice, stream));
                island_mode = false; islands_dirty = false;
            }
            if (bad_j) { dw.n_joints = 0; h_j_body1.clear(); h_j_body2.clear(); h_j_damped.clear(); h_j_collision_disabled.clear(); h_j_type.clear(); any_damped = false; }
            // a halo plan (level-2 sharding) names local body indices too: one that reaches past the new count is dropped with the rest
            bool bad_h = false;
            for (int32_t b : halo.send) bad_h = bad_h || (uint32_t)b >= n;
            for (int32_t b : halo.recv) bad_h = bad_h || (uint32_t)b >= n;
            if (bad_h) { halo = HaloPlan(); halo_on = false; }
            if (bad_c) { bp.n_colliders = 0; bp.n_intervals = 0; have_colliders = false; slot_entity.clear(); entity_slot.clear(); h_col_body.clear(); }
        }
        dw.n_bodies = n;
        size_t total = 0;
        total += al(sizeof(T) * 3 * n) * 7 + al(sizeof(T) * 4 * n) + al(sizeof(T) * 6 * n) + al(sizeof(T) * n) * 6 + al(n) * 4;
        avn_status st = stage_reserve(total + 64 * 32);
        if (st != AVN_OK) return st;
        BodyStage<T> s;
        std::memset(&s, 0, sizeof s);
#define SIN(field, src, cnt, U) do { st = stage_in<U>(src, cnt, &s.field); if (st != AVN_OK) return st; } while (0)
        SIN(position, b->position, 3 * (size_t)n, T); SIN(rotation, b->rotation, 4 * (size_t)n, T);
        SIN(linear_velocity, b->linear_velocity, 3 * (size_t)n, T); SIN(angular_velocity, b->angular_velocity, 3 * (size_t)n, T);
        SIN(inv_mass, b->inv_mass, n, T); SIN(inv_inertia_local, b->inv_inertia_local, 6 * (size_t)n, T);
        SIN(center_of_mass, b->center_of_mass, 3 * (size_t)n, T); SIN(linear_damping, b->linear_damping, n, T);
        SIN(angular_damping, b->angular_damping, n, T); SIN(gravity_scale, b->gravity_scale, n, T);
        SIN(accel_linear, b->accel_linear, 3 * (size_t)n, T); SIN(accel_angular, b->accel_angular, 3 * (size_t)n, T);
        SIN(max_linear_speed, b->max_linear_speed, n, T); SIN(max_angular_speed, b->max_angular_speed, n, T);
        SIN(rb_type, b->rb_type, n, uint8_t); SIN(locked_axes, b->locked_axes, n, uint8_t); SIN(body_flags, b->body_flags, n, uint8_t);
        SIN(dominance, b->dominance, n, int8_t);
        launch_pack_bodies<T>(dw, s, stream);
        HIPCHK(hipGetLastError());
        // host copy of "has SolverBody" for the joint schedules
        h_body_has_sb.resize(n); h_rb_type.resize(n); h_body_flags.resize(n);
        std::vector<uint32_t> asleep;
        for (uint32_t i = 0; i < n; ++i) {
            uint8_t fl = b->body_flags ? b->body_flags[i] : 0;
            h_rb_type[i] = b->rb_type[i];
            if (slp_on) {   // the Sleeping component is the island manager's (SleepIslands / WakeIslands), not the uploader's
                if (b->rb_type[i] != AVN_RB_STATIC && !(fl & AVN_BODY_DISABLED) && !isl.body_has_node(i)) { avn_status si = isl.body_add(i); if (si != AVN_OK) { error = isl.error; return si; } }
                if (isl.body_has_node(i) && isl.body_sleeps(i)) { fl |= AVN_BODY_SLEEPING; asleep.push_back(i); } else fl &= (uint8_t)~AVN_BODY_SLEEPING;
            }
            h_body_flags[i] = fl;
            h_body_has_sb[i] = b->rb_type[i] != AVN_RB_STATIC && !(fl & (AVN_BODY_SLEEPING | AVN_BODY_DISABLED));
        }
        if (slp_on) {
            // (the packed flags are the uploader's: put the manager's Sleeping flags back, clear the others)
            std::vector<uint32_t> awake_list;
            for (uint32_t i = 0; i < n; ++i) if (!(h_body_flags[i] & AVN_BODY_SLEEPING) && b->body_flags && (b->body_flags[i] & AVN_BODY_SLEEPING)) awake_list.push_back(i);
            HIPCHK(hipStreamSynchronize(stream));
            avn_status s2 = stage_reserve((asleep.size() + awake_list.size()) * 4 + 1024);
            if (s2 != AVN_OK) return s2;
            const uint32_t *d1 = nullptr, *d2 = nullptr;
            if ((s2 = stage_in<uint32_t>(asleep.data(), asleep.size(), &d1)) != AVN_OK || (s2 = stage_in<uint32_t>(awake_list.data(), awake_list.size(), &d2)) != AVN_OK) return s2;
            launch_bodies_set_sleeping<T>(dw, d1, (uint32_t)asleep.size(), 1u, nullptr, stream);
            launch_bodies_set_sleeping<T>(dw, d2, (uint32_t)awake_list.size(), 0u, nullptr, stream);
            if ((s2 = slp_grow_bodies(n)) != AVN_OK) return s2;   // spawned bodies: SleepTimer 0, the world's thresholds
        }
        joint_schedule_dirty = true;
        incidence_dirty = true;
        have_bodies = true;
        despawn_needs_bodies = false;
        HIPCHK(hipStreamSynchronize(stream));  // host arrays are only borrowed for the call
        return AVN_OK;
    }
    // AccumulatedLocalAcceleration (forces/mod.rs:661-673), consumed by apply_local_acceleration in front of integrate_velocities in every substep
    // (forces/plugin.rs:207-241): avn_body_ops.h integrate_velocities_one
    avn_status local_accelerations_upload(uint32_t count, const void* linear, const void* angular) override {
        if (count == 0 || (!linear && !angular)) {
            if (dw.lacc_l) { dw.lacc_l = dw.lacc_a = nullptr; graph_valid = false; }
            return AVN_OK;
        }
        if (!have_bodies || count != dw.n_bodies) { error = "local_accelerations_upload: count differs from the last bodies_upload"; return AVN_ERR_BAD_ARG; }
        if (despawn_needs_bodies) { error = "local_accelerations_upload: avn_despawn removed bodies: upload the remaining bodies (avn_bodies_upload) first"; return AVN_ERR_STATE; }
        slp_world_asleep = slp_world_idle = false;
        bool moved = false;
        Vec4<T>*dl = nullptr, *da = nullptr;
        HIPCHK(hipStreamSynchronize(stream));
        GROW(b_lacc_l, (size_t)cap_bodies, dl); GROW(b_lacc_a, (size_t)cap_bodies, da);
        avn_status st = stage_reserve(al(sizeof(T) * 3 * (size_t)count) * 2 + 1024);
        if (st != AVN_OK) return st;
        const T *sl = nullptr, *sa = nullptr;
        if ((st = stage_in<T>(linear, 3 * (size_t)count, &sl)) != AVN_OK || (st = stage_in<T>(angular, 3 * (size_t)count, &sa)) != AVN_OK) return st;
        launch_pack_local_accelerations<T>(dl, da, sl, sa, count, stream);
        HIPCHK(hipGetLastError());
        if (dw.lacc_l != dl || dw.lacc_a != da) { dw.lacc_l = dl; dw.lacc_a = da; graph_valid = false; }
        HIPCHK(hipStreamSynchronize(stream));  // host arrays are only borrowed for the call
        return AVN_OK;
    }
    avn_status bodies_download(const avn_bodies_out* o) override {
        if (!o) return AVN_ERR_BAD_ARG;
        size_t n = dw.n_bodies;
        avn_status st = stage_reserve(al(sizeof(T) * 3 * n) * 3 + al(sizeof(T) * 4 * n) + 1024);
        if (st != AVN_OK) return st;
        T* p = o->position ? stage_alloc<T>(3 * n) : nullptr;
        T* r = o->rotation ? stage_alloc<T>(4 * n) : nullptr;
        T* l = o->linear_velocity ? stage_alloc<T>(3 * n) : nullptr;
        T* a = o->angular_velocity ? stage_alloc<T>(3 * n) : nullptr;
        launch_unpack_bodies<T>(dw, p, r, l, a, stream);
        HIPCHK(hipGetLastError());
        if ((st = stage_out<T>(o->position, p, 3 * n)) != AVN_OK) return st;
        if ((st = stage_out<T>(o->rotation, r, 4 * n)) != AVN_OK) return st;
        if ((st = stage_out<T>(o->linear_velocity, l, 3 * n)) != AVN_OK) return st;
        if ((st = stage_out<T>(o->angular_velocity, a, 3 * n)) != AVN_OK) return st;
        HIPCHK(hipStreamSynchronize(stream));
        return AVN_OK;
    }
    avn_status solver_bodies_download(const avn_solver_bodies_out* o) override {
        if (!o) return AVN_ERR_BAD_ARG;
        size_t n = dw.n_bodies;
        avn_status st = stage_reserve(al(sizeof(T) * 3 * n) * 5 + al(sizeof(T) * 4 * n) + al(sizeof(T) * 6 * n) + al(sizeof(T) * n) * 3 + al(4 * n) + al(2 * n) + 4096);
        if (st != AVN_OK) return st;
        SolverBodiesStage<T> s;
        s.linear_velocity = o->linear_velocity ? stage_alloc<T>(3 * n) : nullptr;
        s.angular_velocity = o->angular_velocity ? stage_alloc<T>(3 * n) : nullptr;
        s.delta_position = o->delta_position ? stage_alloc<T>(3 * n) : nullptr;
        s.delta_rotation = o->delta_rotation ? stage_alloc<T>(4 * n) : nullptr;
        s.flags = o->flags ? stage_alloc<uint32_t>(n) : nullptr;
        s.inv_mass = o->inv_mass ? stage_alloc<T>(n) : nullptr;
        s.inv_inertia_world = o->inv_inertia_world ? stage_alloc<T>(6 * n) : nullptr;
        s.dominance = o->dominance ? stage_alloc<int16_t>(n) : nullptr;
        s.linear_increment = o->linear_increment ? stage_alloc<T>(3 * n) : nullptr;
        s.angular_increment = o->angular_increment ? stage_alloc<T>(3 * n) : nullptr;
        s.linear_damping_rhs = o->linear_damping_rhs ? stage_alloc<T>(n) : nullptr;
        s.angular_damping_rhs = o->angular_damping_rhs ? stage_alloc<T>(n) : nullptr;
        launch_unpack_solver_bodies<T>(dw, s, stream);
        HIPCHK(hipGetLastError());
#define SOUT(dst, src, cnt, U) do { if ((st = stage_out<U>(dst, src, cnt)) != AVN_OK) return st; } while (0)
        SOUT(o->linear_velocity, s.linear_velocity, 3 * n, T); SOUT(o->angular_velocity, s.angular_velocity, 3 * n, T);
        SOUT(o->delta_position, s.delta_position, 3 * n, T); SOUT(o->delta_rotation, s.delta_rotation, 4 * n, T);
        SOUT(o->flags, s.flags, n, uint32_t); SOUT(o->inv_mass, s.inv_mass, n, T); SOUT(o->inv_inertia_world, s.inv_inertia_world, 6 * n, T);
        SOUT(o->dominance, s.dominance, n, int16_t); SOUT(o->linear_increment, s.linear_increment, 3 * n, T);
        SOUT(o->angular_increment, s.angular_increment, 3 * n, T); SOUT(o->linear_damping_rhs, s.linear_damping_rhs, n, T);
        SOUT(o->angular_damping_rhs, s.angular_damping_rhs, n, T);
        HIPCHK(hipStreamSynchronize(stream));
        return AVN_OK;
    }

