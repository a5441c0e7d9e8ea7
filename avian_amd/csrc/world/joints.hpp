// world/joints.hpp -- fragment of the body of `template <class T> struct World` (avn_world.hip includes it inside the class):
// XpbdSolverPlugin's joint upload, read-back and the level schedules.

    // ---- joints ----------------------------------------------------------------------------------------
    avn_status distance_joints_upload(const avn_distance_joints* j) override {
        if (!j || (j->count && (!j->body1 || !j->body2 || !j->local_anchor1 || !j->local_anchor2 || !j->limit_min || !j->limit_max || !j->compliance))) {
            error = "distance_joints_upload: null array"; return AVN_ERR_BAD_ARG;
        }
        // the special case joint_type = DISTANCE of joints_upload
        std::vector<uint8_t> types(j->count, (uint8_t)AVN_JOINT_DISTANCE);
        std::vector<T> comp(3 * (size_t)j->count, T(0));
        for (size_t i = 0; i < j->count; ++i) comp[3 * i] = ((const T*)j->compliance)[i];
        avn_joints g;
        std::memset(&g, 0, sizeof g);
        g.count = j->count; g.joint_type = types.data(); g.body1 = j->body1; g.body2 = j->body2;
        g.local_anchor1 = j->local_anchor1; g.local_anchor2 = j->local_anchor2; g.limit_min = j->limit_min; g.limit_max = j->limit_max;
        g.compliance = comp.data(); g.damping_linear = j->damping_linear; g.damping_angular = j->damping_angular;
        g.collision_disabled = j->collision_disabled;
        return joints_upload(&g);
    }
    avn_status joints_upload(const avn_joints* j) override {
        slp_world_asleep = slp_world_idle = false;
        if (!have_bodies) { error = "joints_upload before bodies_upload"; return AVN_ERR_STATE; }
        if (!j || (j->count && (!j->joint_type || !j->body1 || !j->body2 || !j->local_anchor1 || !j->local_anchor2 || !j->compliance))) {
            error = "joints_upload: null array"; return AVN_ERR_BAD_ARG;
        }
        uint32_t J = j->count;
        for (uint32_t i = 0; i < J; ++i) {
            if (j->joint_type[i] >= AVN_JOINT_TYPE_COUNT) { error = "joints_upload: bad joint_type"; return AVN_ERR_BAD_ARG; }
            if (j->body1[i] < 0 || j->body2[i] < 0 || (uint32_t)j->body1[i] >= dw.n_bodies || (uint32_t)j->body2[i] >= dw.n_bodies || j->body1[i] == j->body2[i]) {
                error = "joints_upload: bad body index"; return AVN_ERR_BAD_ARG;
            }
        }
        if (despawn_needs_joints) {   // (avn_despawn took joints out: exactly the remaining set comes back, bodies in the new numbering)
            if (J != despawn_expected_joints) { error = "joints_upload: after avn_despawn exactly the remaining joints must be uploaded"; return AVN_ERR_STATE; }
            if (despawn_needs_bodies) { error = "joints_upload: after avn_despawn the remaining bodies are uploaded first"; return AVN_ERR_STATE; }
            // (the debt is cleared at the successful END of this call: a validation, allocation or copy that fails below must leave avn_step blocked -- ADVICE r5)
        }
        if (slp_on) {
            // the island manager links joints when they are added (PhysicsIslands::add_joint, islands/mod.rs:668-735) and has no way to take one
            // back without the bodies' JointGraph history: with sleeping on, an upload may only APPEND joints to the set it already knows
            if (J < h_j_body1.size()) { error = "joints_upload: with avn_sleeping_enable on, joints can only be appended (avn_sleeping_enable(NULL) first to change the set)"; return AVN_ERR_STATE; }
            for (size_t i = 0; i < h_j_body1.size(); ++i)
                if (h_j_body1[i] != j->body1[i] || h_j_body2[i] != j->body2[i]) { error = "joints_upload: with avn_sleeping_enable on, the bodies of an existing joint cannot change"; return AVN_ERR_STATE; }
            for (uint32_t i = (uint32_t)h_j_body1.size(); i < J; ++i) { const avn_status si = isl.joint_add(i, (uint32_t)j->body1[i], (uint32_t)j->body2[i]); if (si != AVN_OK) return slp_fail(si); }
        }
        bool moved = false;
        if (J > cap_joints) {
            HIPCHK(hipStreamSynchronize(stream));
            size_t c = std::max<size_t>(J, cap_joints + cap_joints / 2);
            GROW(b_j_bodies, c, dw.j_bodies); GROW(b_j_a1, c, dw.j_a1); GROW(b_j_a2, c, dw.j_a2); GROW(b_j_par, c, dw.j_par);
            GROW(b_j_b1, c, dw.j_b1); GROW(b_j_b2, c, dw.j_b2); GROW(b_j_ax, c, dw.j_ax); GROW(b_j_l2, c, dw.j_l2);
            GROW(b_j_r1, c, dw.j_r1); GROW(b_j_r2, c, dw.j_r2); GROW(b_j_cd, c, dw.j_cd); GROW(b_j_lag, c, dw.j_lag);
            GROW(b_j_s0, c, dw.j_s0); GROW(b_j_s1, c, dw.j_s1); GROW(b_j_s2, c, dw.j_s2); GROW(b_j_s3, c, dw.j_s3);
            GROW(b_j_rl0, c, dw.j_rl0); GROW(b_j_rl1, c, dw.j_rl1); GROW(b_j_force, c, dw.j_force); GROW(b_j_torque, c, dw.j_torque);
            cap_joints = (uint32_t)c;
        }
        if (moved || dw.n_joints != J) graph_valid = false;
        dw.n_joints = J;
        avn_status st = stage_reserve(al(4 * (size_t)J) * 2 + al(J) * 2 + al(sizeof(T) * 3 * J) * 4 + al(sizeof(T) * 4 * J) * 2 + al(sizeof(T) * J) * 6 + 8192);
        if (st != AVN_OK) return st;
        JointStage<T> s;
        std::memset(&s, 0, sizeof s);
        SIN(joint_type, j->joint_type, J, uint8_t); SIN(limit_flags, j->limit_flags, J, uint8_t);
        SIN(body1, j->body1, J, int32_t); SIN(body2, j->body2, J, int32_t);
        SIN(local_anchor1, j->local_anchor1, 3 * (size_t)J, T); SIN(local_anchor2, j->local_anchor2, 3 * (size_t)J, T);
        SIN(local_basis1, j->local_basis1, 4 * (size_t)J, T); SIN(local_basis2, j->local_basis2, 4 * (size_t)J, T);
        SIN(axis, j->axis, 3 * (size_t)J, T);
        SIN(limit_min, j->limit_min, J, T); SIN(limit_max, j->limit_max, J, T); SIN(limit2_min, j->limit2_min, J, T); SIN(limit2_max, j->limit2_max, J, T);
        SIN(compliance, j->compliance, 3 * (size_t)J, T);
        SIN(damping_linear, j->damping_linear, J, T); SIN(damping_angular, j->damping_angular, J, T);
        launch_pack_joints<T>(dw, s, stream);
        HIPCHK(hipGetLastError());
        h_j_body1.assign(j->body1, j->body1 + J);
        h_j_body2.assign(j->body2, j->body2 + J);
        h_j_type.assign(j->joint_type, j->joint_type + J);
        bool damp = j->damping_linear && j->damping_angular;
        h_j_damped.assign(J, damp ? 1 : 0);
        any_damped = damp && J > 0;
        // body pairs whose joints disable collision (reference broad_phase.rs:423-428)
        std::vector<uint64_t> disabled;
        h_j_collision_disabled.assign(J, 0);
        for (uint32_t i = 0; i < J; ++i)
            if (j->collision_disabled && j->collision_disabled[i]) {
                h_j_collision_disabled[i] = 1;
                uint32_t a = (uint32_t)j->body1[i], b = (uint32_t)j->body2[i];
                disabled.push_back(a < b ? ((uint64_t)a << 32) | b : ((uint64_t)b << 32) | a);
            }
        st = build_hash_set(b_disabled_set, bp.disabled_set, bp.disabled_cap, disabled.data(), (uint32_t)disabled.size());
        if (st != AVN_OK) return st;
        joint_schedule_dirty = true;
        HIPCHK(hipStreamSynchronize(stream));
        despawn_needs_joints = false;
        return AVN_OK;
    }
    avn_status joints_download(const avn_joints_out* o) override {
        if (!o) return AVN_ERR_BAD_ARG;
        if (despawn_needs_joints) { error = "joints_download: avn_despawn removed joints: upload the remaining joints (avn_joints_upload) first"; return AVN_ERR_STATE; }
        size_t J = dw.n_joints;
        avn_status st = stage_reserve(al(sizeof(T) * 3 * J) * 7 + 1024);
        if (st != AVN_OK) return st;
        T* a = o->world_r1 ? stage_alloc<T>(3 * J) : nullptr;
        T* b = o->world_r2 ? stage_alloc<T>(3 * J) : nullptr;
        T* c = o->center_difference ? stage_alloc<T>(3 * J) : nullptr;
        T* d = o->total_lagrange ? stage_alloc<T>(3 * J) : nullptr;
        T* e = o->force ? stage_alloc<T>(3 * J) : nullptr;
        T* f = o->total_rotation_lagrange ? stage_alloc<T>(3 * J) : nullptr;
        T* g = o->torque ? stage_alloc<T>(3 * J) : nullptr;
        launch_unpack_joints<T>(dw, a, b, c, d, e, f, g, stream);
        HIPCHK(hipGetLastError());
        SOUT(o->world_r1, a, 3 * J, T); SOUT(o->world_r2, b, 3 * J, T); SOUT(o->center_difference, c, 3 * J, T);
        SOUT(o->total_lagrange, d, 3 * J, T); SOUT(o->force, e, 3 * J, T);
        SOUT(o->total_rotation_lagrange, f, 3 * J, T); SOUT(o->torque, g, 3 * J, T);
        HIPCHK(hipStreamSynchronize(stream));
        return AVN_OK;
    }
    avn_status upload_u32(DevBuf& b, const std::vector<uint32_t>& v) {
        hipError_t err;
        b.ensure(std::max<size_t>(v.size(), 1) * 4, err);
        if (err != hipSuccess) { error = "hipMalloc failed"; return AVN_ERR_OOM; }
        if (!v.empty()) HIPCHK(hipMemcpyAsync(b.p, v.data(), v.size() * 4, hipMemcpyHostToDevice, stream));
        return AVN_OK;
    }
    avn_status rebuild_joint_schedules() {
        if (!joint_schedule_dirty) return AVN_OK;
        HIPCHK(hipStreamSynchronize(stream));
        uint32_t J = dw.n_joints, N = dw.n_bodies;
        // the reference's serial order: one system per joint type in the order of xpbd/plugin.rs:77-82 (= the AVN_JOINT_* ids),
        // each iterating its joints in array (= spawn) order
        std::vector<uint32_t> all(J);
        std::iota(all.begin(), all.end(), 0u);
        std::stable_sort(all.begin(), all.end(), [&](uint32_t a, uint32_t b) { return h_j_type[a] < h_j_type[b]; });
        std::vector<int32_t> k1(J), k2(J);
        for (uint32_t k = 0; k < J; ++k) {
            uint32_t i = all[k];
            // bodies without a SolverBody are DUMMY in solve_xpbd_joint: never modified => they do not serialise joints
            k1[k] = h_body_has_sb[h_j_body1[i]] ? h_j_body1[i] : -1;
            k2[k] = h_body_has_sb[h_j_body2[i]] ? h_j_body2[i] : -1;
        }
        sched_solve.build(all, k1, k2, N);
        std::vector<uint32_t> damped;
        std::vector<int32_t> d1, d2;
        sched_damp.touches_dummy = false;
        for (uint32_t k = 0; k < J; ++k) {
            uint32_t i = all[k];
            if (h_j_damped[i]) {
                damped.push_back(i);
                // joint_damping's DUMMY bodies are shared by the joints of ONE type and mutable: virtual bodies N + 2t, N + 2t + 1
                bool m1 = !h_body_has_sb[h_j_body1[i]], m2 = !h_body_has_sb[h_j_body2[i]];
                d1.push_back(m1 ? (int32_t)(N + 2u * h_j_type[i]) : h_j_body1[i]);
                d2.push_back(m2 ? (int32_t)(N + 2u * h_j_type[i] + 1u) : h_j_body2[i]);
                if (m1 || m2) sched_damp.touches_dummy = true;
            }
        }
        sched_damp.build(damped, d1, d2, N + DUMMY_SLOTS);
        avn_status st;
        for (JointSchedule* s : {&sched_solve, &sched_damp}) {
            if ((st = upload_u32(s->d_comp_level_begin, s->comp_level_begin)) != AVN_OK) return st;
            if ((st = upload_u32(s->d_level_offsets, s->level_offsets)) != AVN_OK) return st;
            if ((st = upload_u32(s->d_order, s->order)) != AVN_OK) return st;
            std::vector<uint32_t> rec;
            fill_joint_recs(*s, rec, s == &sched_solve);
            if ((st = upload_u32(s->d_rec, rec)) != AVN_OK) return st;
            if ((st = upload_u32(s->d_comp_bodies, s->comp_bodies)) != AVN_OK) return st;
            HIPCHK(hipStreamSynchronize(stream));   // (`rec` is a local: the copy must have left it)
        }
        HIPCHK(hipStreamSynchronize(stream));
        joint_schedule_dirty = false;
        groups_dirty = true;
        graph_valid = false;
        return AVN_OK;
    }
    // ---- island-level concurrency: joint-only islands on their own stream (DW::side_group) -------------------------------------------
    // Islands = connected components of the bodies that own a SolverBody under "share a joint or a manifold" (islands/mod.rs:1-10).  An
    // island with joints and without any manifold is a SIDE island: nothing in the contact passes touches its bodies, nothing in its joint
    // pass touches anybody else, so its substep loop (integrate_velocities, integrate_positions, the XPBD solve in the serial joint order of
    // xpbd/plugin.rs:77-82,145-189, velocity projection, joint damping) runs on `stream_side` next to the other islands' colour launches.
    // Per body and per joint the operation sequence is the single stream's: results are bit-identical.  Host-uploaded manifolds only (the
    // bodies of the manifolds are on the host then); joint damping against a body without a SolverBody couples the joints of one type
    // through the shared DUMMY (joint_damping::<T>): no split then.
    bool groups_dirty = true, groups_active = false, groups_enabled = !(avn_env("AVN_NO_ISLAND_STREAMS") && avn_env("AVN_NO_ISLAND_STREAMS")[0] && avn_env("AVN_NO_ISLAND_STREAMS")[0] != '0');
    JointSchedule sched_solve_main, sched_damp_main, sched_solve_side, sched_damp_side;
    DevBuf b_side_group;
    hipStream_t stream_side = nullptr;
    hipEvent_t ev_side_fork = nullptr, ev_side_done = nullptr;
    uint32_t side_bodies = 0, side_joints = 0;
    // (joint, body1, body2, local slots) per schedule slot + the per-component body counts of the LDS walk.  `lds`: the XPBD solve (damping touches virtual DUMMY bodies: global walk)
    bool joint_lds_enabled = !avn_env("AVN_NO_JOINT_LDS");
    std::vector<uint32_t> jl_loc, jl_gen;
    void fill_joint_recs(JointSchedule& sc, std::vector<uint32_t>& rec, bool lds) {
        const size_t J = sc.order.size();
        rec.assign(4 * J, 0u);
        for (size_t k = 0; k < J; ++k) { const uint32_t j = sc.order[k]; rec[4 * k] = j; rec[4 * k + 1] = (uint32_t)h_j_body1[j]; rec[4 * k + 2] = (uint32_t)h_j_body2[j]; }
        sc.comp_bodies.assign(sc.n_components, 0xFFFFFFFFu);
        sc.lds_bytes = 0;
        if (!lds || !joint_lds_enabled || !J) return;
        const uint32_t N = dw.n_bodies;
        if (jl_loc.size() < N) { jl_loc.resize(N); jl_gen.assign(N, 0u); }
        if (jl_gen.size() < N) jl_gen.resize(N, 0u);
        static uint32_t gen = 0;
        constexpr size_t LDS_LIMIT = 64 * 1024;   // (the default dynamic-LDS ceiling of a launch)
        for (uint32_t c = 0; c < sc.n_components; ++c) {
            const uint32_t s0 = sc.level_offsets[sc.comp_level_begin[c]], s1 = sc.level_offsets[sc.comp_level_begin[c + 1]];
            if (++gen == 0) { std::fill(jl_gen.begin(), jl_gen.end(), 0u); gen = 1; }
            uint32_t nb = 0;
            for (uint32_t k = s0; k < s1; ++k)
                for (int side = 0; side < 2; ++side) { const uint32_t b = rec[4 * k + 1 + side]; if (b < N && jl_gen[b] != gen) { jl_gen[b] = gen; jl_loc[b] = nb++; } }
            const size_t bytes = ((size_t)4 * nb + (size_t)15 * (s1 - s0)) * sizeof(V) + (size_t)4 * nb;
            if (nb > 0xFFFFu || bytes > LDS_LIMIT || s1 - s0 < 2) continue;   // (a single joint gains nothing from staging)
            for (uint32_t k = s0; k < s1; ++k) rec[4 * k + 3] = jl_loc[rec[4 * k + 1]] | (jl_loc[rec[4 * k + 2]] << 16);
            sc.comp_bodies[c] = nb;
            sc.lds_bytes = std::max(sc.lds_bytes, (uint32_t)bytes);
        }
    }
    avn_status upload_schedule(JointSchedule& sc) {
        avn_status st;
        if ((st = upload_u32(sc.d_comp_level_begin, sc.comp_level_begin)) != AVN_OK) return st;
        if ((st = upload_u32(sc.d_level_offsets, sc.level_offsets)) != AVN_OK) return st;
        std::vector<uint32_t> rec;
        fill_joint_recs(sc, rec, &sc == &sched_solve_main || &sc == &sched_solve_side);
        if ((st = upload_u32(sc.d_rec, rec)) != AVN_OK) return st;
        if ((st = upload_u32(sc.d_comp_bodies, sc.comp_bodies)) != AVN_OK) return st;
        HIPCHK(hipStreamSynchronize(stream));
        return AVN_OK;
    }
    avn_status rebuild_body_groups() {
        if (!groups_dirty) return AVN_OK;
        groups_dirty = false;
        const bool was = groups_active;
        groups_active = false;
        const uint32_t J = dw.n_joints, N = dw.n_bodies, M = dw.n_manifolds;
        dw.side_group = nullptr; dw.body_group = 0;
        if (was) graph_valid = false;
        if (!groups_enabled || pipe_on || use_handles || halo_on || !J || !M || sched_damp.touches_dummy || h_m_body1.size() != M || h_body_has_sb.size() != N) return AVN_OK;
        auto has_sb = [&](int32_t b) { return b >= 0 && (uint32_t)b < N && h_body_has_sb[(uint32_t)b]; };
        std::vector<uint32_t> parent(N);
        std::iota(parent.begin(), parent.end(), 0u);
        auto find = [&](uint32_t x) { while (parent[x] != x) { parent[x] = parent[parent[x]]; x = parent[x]; } return x; };
        auto unite = [&](uint32_t a, uint32_t b) { a = find(a); b = find(b); if (a != b) parent[std::max(a, b)] = std::min(a, b); };
        for (uint32_t j = 0; j < J; ++j) if (has_sb(h_j_body1[j]) && has_sb(h_j_body2[j])) unite((uint32_t)h_j_body1[j], (uint32_t)h_j_body2[j]);
        for (uint32_t m = 0; m < M; ++m) if (has_sb(h_m_body1[m]) && has_sb(h_m_body2[m])) unite((uint32_t)h_m_body1[m], (uint32_t)h_m_body2[m]);
        std::vector<uint8_t> has_manifold(N, 0), has_joint(N, 0);
        for (uint32_t m = 0; m < M; ++m) { if (has_sb(h_m_body1[m])) has_manifold[find((uint32_t)h_m_body1[m])] = 1; if (has_sb(h_m_body2[m])) has_manifold[find((uint32_t)h_m_body2[m])] = 1; }
        for (uint32_t j = 0; j < J; ++j) { if (has_sb(h_j_body1[j])) has_joint[find((uint32_t)h_j_body1[j])] = 1; if (has_sb(h_j_body2[j])) has_joint[find((uint32_t)h_j_body2[j])] = 1; }
        std::vector<uint8_t> side(N, 0);
        side_bodies = 0;
        for (uint32_t b = 0; b < N; ++b) if (h_body_has_sb[b]) { const uint32_t r = find(b); if (has_joint[r] && !has_manifold[r]) { side[b] = 1; ++side_bodies; } }
        if (!side_bodies) return AVN_OK;
        // the two halves of the joint schedules (a joint of a side island has both SolverBody-owning bodies in it; one without any goes with the main group)
        auto joint_side = [&](uint32_t j) { return (has_sb(h_j_body1[j]) && side[(uint32_t)h_j_body1[j]]) || (has_sb(h_j_body2[j]) && side[(uint32_t)h_j_body2[j]]); };
        std::vector<uint32_t> all(J);
        std::iota(all.begin(), all.end(), 0u);
        std::stable_sort(all.begin(), all.end(), [&](uint32_t a, uint32_t b) { return h_j_type[a] < h_j_type[b]; });
        side_joints = 0;
        for (int g = 0; g < 2; ++g) {
            std::vector<uint32_t> js, damped;
            std::vector<int32_t> k1, k2, d1, d2;
            for (uint32_t i : all) {
                if ((joint_side(i) ? 1 : 0) != g) continue;
                js.push_back(i);
                k1.push_back(has_sb(h_j_body1[i]) ? h_j_body1[i] : -1); k2.push_back(has_sb(h_j_body2[i]) ? h_j_body2[i] : -1);
                if (h_j_damped[i]) { damped.push_back(i); d1.push_back(h_j_body1[i]); d2.push_back(h_j_body2[i]); }   // (no DUMMY-touching damping here: checked above)
            }
            if (g) side_joints = (uint32_t)js.size();
            JointSchedule& so = g ? sched_solve_side : sched_solve_main; JointSchedule& da = g ? sched_damp_side : sched_damp_main;
            so.build(js, k1, k2, N);
            da.touches_dummy = false;
            da.build(damped, d1, d2, N + DUMMY_SLOTS);
            avn_status st;
            if ((st = upload_schedule(so)) != AVN_OK || (st = upload_schedule(da)) != AVN_OK) return st;
        }
        hipError_t err;
        b_side_group.ensure(std::max<size_t>(N, 1), err);
        if (err != hipSuccess) { error = "hipMalloc failed"; return AVN_ERR_OOM; }
        HIPCHK(hipMemcpyAsync(b_side_group.p, side.data(), N, hipMemcpyHostToDevice, stream));
        HIPCHK(hipStreamSynchronize(stream));
        if (!stream_side) {
            HIPCHK(hipStreamCreateWithFlags(&stream_side, hipStreamNonBlocking));
            HIPCHK(hipEventCreateWithFlags(&ev_side_fork, hipEventDisableTiming | EV_FLAGS));
            HIPCHK(hipEventCreateWithFlags(&ev_side_done, hipEventDisableTiming | EV_FLAGS));
        }
        dw.side_group = b_side_group.as<uint8_t>();
        groups_active = true;
        graph_valid = false;
        return AVN_OK;
    }

