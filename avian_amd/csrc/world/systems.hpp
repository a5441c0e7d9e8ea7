// world/systems.hpp -- fragment of the body of `template <class T> struct World` (avn_world.hip includes it inside the class):
// the systems of the schedule, the substep loop (hipGraph), avn_run_system / avn_step.

    // ---- systems -------------------------------------------------------------------------------------------
    avn_status need_bodies() { if (!have_bodies) { error = "no bodies uploaded"; return AVN_ERR_STATE; } return AVN_OK; }
    void prepare_solver_bodies() { launch_prepare_solver_bodies<T>(dw, stream); ++launches; }
    void prepare_joints() { if (dw.n_joints) { launch_prepare_joints<T>(dw, stream); ++launches; } }
    void prepare_contact_constraints() {
        // GraphColor::manifold_handles indirection (plugin.rs:389-398): the colours' manifolds are fetched from the contact table
        // handle mode: the constraints are generated straight from the table rows the handles name (the kernel also lays out the headers the solve
        // passes read); the overflow colour's CSR reads DW::m_bodies, so it follows
        // (Round 4, tried: m_bodies written by k_pg_build_handles so that the CSR chain and the slot table run on the broad-phase stream NEXT to
        //  the constraint generation: the chain does overlap (timeline), but its one-workgroup kernels wait for CUs behind the 818 workgroups of
        //  the generation (k_ovf_entries 4.8 -> 64 us) and the join costs another event: 2.841 / 2.852 ms per step against 2.823 / 2.846 without
        //  (same box).  Hoisting the point-record loads above the body gathers and 3 waves per SIMD for the generation: 0.1828 / 0.1816 ms
        //  against 0.1873 / 0.1772.  Neither kept.)
        if (!constraints_prepared_early) {   // (device closed loop: normally enqueued by pg_apply_ops, in front of its read-back)
            RowsView<T> rv{b_handles.as<uint32_t>(), ct.meta, bp.col_info, ct.rows};
            launch_prepare_contact_constraints<T>(dw, params, stream, constraint_count_clean, use_handles ? &rv : nullptr); ++launches;
            constraint_count_clean = false;
        }
        constraints_prepared_early = false;
        if (pipe_dev && ovf_csr_dirty && dw.n_manifolds) { overflow_csr_device(); ovf_csr_dirty = false; }
    }
    void store_contact_impulses() {
        launch_store_contact_impulses<T>(dw, stream); ++launches;
        if (use_handles && dw.n_manifolds) { launch_scatter_impulses<T>(dw, ct, b_handles.as<uint32_t>(), stream); ++launches; }
    }
    void pre_process_velocity_increments() { launch_pre_process_increments<T>(dw, params, stream); ++launches; }
    void integrate_velocities() { launch_integrate_velocities<T>(dw, params, stream); ++launches; }
    // warm start of ALL colours in one body-centric launch; `fused` also runs integrate_velocities for the body first
    void warm_start(bool fused) {
        if (dw.n_manifolds) {
            if (slots_dirty && dw.inc_slot) { launch_build_incidence_slots<T>(dw, stream); launches += 2; slots_dirty = false; }  // (normally done by prepare)
            launch_body_warm_start<T>(dw, params, fused, pipe_dev, stream); ++launches;
        }
        else if (fused) integrate_velocities();
    }
    void integrate_positions() { launch_integrate_positions<T>(dw, params, stream); ++launches; }
    void contact_pass(int pass) {
        if (!dw.n_manifolds) return;
        if (bias_skeleton && pass == PASS_SOLVE_BIAS) pass = PASS_MEMORY_SKELETON;   // AVN_BIAS_SKELETON=1: measurement aid, state unchanged
        if (pipe_dev) {   // overflow colour first (one dataflow launch), then colours 0..22
            // (launched whenever a grid is captured for it, whatever the colour's current population: the captured graph must not
            //  depend on the step's counts; an empty colour costs one launch of idle lanes)
            if (ovf_grid_blocks && ovf_epoch < PGC_OVF_TILES) {
                OverflowFlow of{b_ovf_rank.as<uint32_t>(), b_ovf_ticket.as<uint32_t>(), pg.ctr + PGC_OVF_TILE, pg.ctr + PGC_ERROR};
                if (const char* e = avn_env("AVN_OVF_POLL_SLEEP")) of.poll_sleep = (uint32_t)std::atoi(e);
                launch_overflow_flow<T>(dw, params, pass, of, ovf_epoch, ovf_grid_blocks, stream);
                ++ovf_epoch; ++launches;
            }
            uint32_t gb[AVN_GRAPH_COLOR_COUNT];
            std::memcpy(gb, grid_blocks, sizeof gb);
            gb[AVN_COLOR_OVERFLOW_INDEX] = 0;
            OverflowSchedule none{0, nullptr, nullptr, nullptr, nullptr, nullptr, 0};
            launches += launch_contact_pass<T>(dw, params, pass, gb, nullptr, none, stream, oct_mask);
            return;
        }
        OverflowSchedule ovf{sched_overflow.n_components, sched_overflow.d_comp_level_begin.as<uint32_t>(), sched_overflow.d_level_offsets.as<uint32_t>(),
                             sched_overflow.d_order.as<uint32_t>(), nullptr, nullptr, 0};
        if (sched_overflow.gorder.size() > overflow_level_threshold) {  // a big overflow colour: one device-wide launch per level instead of one workgroup per component
            ovf.gorder = sched_overflow.d_gorder.as<uint32_t>();
            ovf.glevel_offsets = sched_overflow.glevel_offsets.data();
            ovf.n_glevels = (uint32_t)sched_overflow.glevel_offsets.size() - 1;
        }
        launches += launch_contact_pass<T>(dw, params, pass, grid_blocks, use_handles ? nullptr : color_offsets, ovf, stream, oct_mask);
    }
    // The reference runs the snapshot and the velocity projection over ALL active bodies whenever XpbdSolverPlugin is
    // installed (xpbd/plugin.rs:61-76,192-240).  With no joints the projection adds 2 * (dq * conj(dq)).xyz / h:
    //  - f32 (glam's SIMD `Quat`, pairwise sums): every xyz component cancels exactly, e.g. y = (-wy + xz) + (yw - zx) is
    //    a + (-a) = 0, so the two systems are exact no-ops (up to the sign of a zero) and are skipped;
    //  - f64 (scalar `DQuat`, left-to-right sums): y = ((-wy + xz) + yw) - zx leaves a rounding residual of order
    //    ulp(wy), so the reference really perturbs omega every substep — replicate, don't "fix" (found by the cfg5 test).
    bool xpbd_body_passes_needed() const { return dw.n_joints != 0 || sizeof(T) == 8 || l2_global_joints; }   // (level 2: the UNSPLIT world holds joints, so it runs both passes over all bodies)
    void xpbd_solve(bool snapshot) {
        if (snapshot && xpbd_body_passes_needed()) { launch_xpbd_snapshot<T>(dw, stream); ++launches; }
        if (!dw.n_joints) return;
        JointSchedule& sc = dw.body_group == 1u ? sched_solve_main : sched_solve;   // (island streams: the side islands' joints run in substep_side)
        if (!sc.n_components) return;
        if (sc.lds_bytes) launch_joint_schedule_lds<T>(dw, params, (uint32_t)sc.n_components, sc.d_comp_level_begin.as<uint32_t>(), sc.d_level_offsets.as<uint32_t>(), sc.d_rec.as<int4>(), sc.d_comp_bodies.as<uint32_t>(), sc.lds_bytes, stream);
        else launch_joint_schedule<T>(dw, params, 0, (uint32_t)sc.n_components, sc.d_comp_level_begin.as<uint32_t>(), sc.d_level_offsets.as<uint32_t>(), sc.d_rec.as<int4>(), stream);
        ++launches;
    }
    void xpbd_velocity_projection() { if (xpbd_body_passes_needed()) { launch_xpbd_velocity_projection<T>(dw, params, stream); ++launches; } }
    void joint_damping() {
        if (dw.body_group == 1u) {
            if (!any_damped || !sched_damp_main.n_components) return;
            launch_joint_schedule<T>(dw, params, 1, (uint32_t)sched_damp_main.n_components, sched_damp_main.d_comp_level_begin.as<uint32_t>(),
                                     sched_damp_main.d_level_offsets.as<uint32_t>(), sched_damp_main.d_rec.as<int4>(), stream);
            ++launches;
            return;
        }
        if (!any_damped || !sched_damp.n_components) return;
        if (sched_damp.touches_dummy) {
            // reset the two virtual SolverBody::DUMMY slots (all-zero bit pattern = zero velocities)
            (void)hipMemsetAsync(&dw.sb_lin[dw.n_bodies], 0, 2 * DUMMY_SLOTS * sizeof(V), stream);  // DUMMY_SLOTS bodies x (lin | ang) slot
        }
        launch_joint_schedule<T>(dw, params, 1, (uint32_t)sched_damp.n_components, sched_damp.d_comp_level_begin.as<uint32_t>(),
                                 sched_damp.d_level_offsets.as<uint32_t>(), sched_damp.d_rec.as<int4>(), stream);
        ++launches;
    }
    void substep() {  // SubstepSchedule order (reference solver/schedule.rs:59-69, xpbd/plugin.rs:30-40)
        const bool dg = !cfg.use_graph && substep_index < DG_SUBSTEPS;   // (events captured into a hipGraph cannot be read back)
        hipEvent_t* de = ev_dgs + (size_t)substep_index * DG_PER;
        if (dg) (void)hipEventRecord(de[0], stream);
        warm_start(true);  // integrate_velocities + warm_start
        if (dg) (void)hipEventRecord(de[1], stream);
        // measurement hook: the dominant kernel's launches inside the step.  Direct launches only: events recorded as nodes of a
        // captured graph cannot be read back with hipEventElapsedTime on this runtime (hipErrorInvalidHandle).
        const bool timed = substep_index < BIAS_EV && dw.n_manifolds != 0 && !cfg.use_graph;
        if (timed) { (void)hipEventRecord(ev_bias[2 * substep_index], stream); bias_launches = launches; }
        for (uint32_t it = 0; it < cfg.solver_iterations; ++it) contact_pass(PASS_SOLVE_BIAS);
        if (timed) { (void)hipEventRecord(ev_bias[2 * substep_index + 1], stream); bias_launches = launches - bias_launches; bias_timed = substep_index + 1; }
        ++substep_index;
        if (dg) (void)hipEventRecord(de[2], stream);
        integrate_positions();
        if (dg) (void)hipEventRecord(de[3], stream);
        for (uint32_t it = 0; it < cfg.solver_iterations; ++it) contact_pass(PASS_SOLVE_RELAX);
        for (uint32_t it = 0; it < cfg.solver_iterations; ++it) xpbd_solve(it == 0);
        xpbd_velocity_projection();
        joint_damping();
        if (dg) { (void)hipEventRecord(de[4], stream); dg_substeps = substep_index; }
    }
    // the substep of the side islands (joints, no manifolds) on stream_side: the same systems in the same order, minus the contact passes
    void substep_side() {
        DW<T> ds = dw;
        ds.body_group = 2u;
        launch_integrate_velocities<T>(ds, params, stream_side); ++launches;
        launch_integrate_positions<T>(ds, params, stream_side); ++launches;
        for (uint32_t it = 0; it < cfg.solver_iterations; ++it) {
            if (it == 0) { launch_xpbd_snapshot<T>(ds, stream_side); ++launches; }
            if (sched_solve_side.n_components) {
                JointSchedule& ss = sched_solve_side;
                if (ss.lds_bytes) launch_joint_schedule_lds<T>(ds, params, (uint32_t)ss.n_components, ss.d_comp_level_begin.as<uint32_t>(), ss.d_level_offsets.as<uint32_t>(), ss.d_rec.as<int4>(), ss.d_comp_bodies.as<uint32_t>(), ss.lds_bytes, stream_side);
                else launch_joint_schedule<T>(ds, params, 0, (uint32_t)ss.n_components, ss.d_comp_level_begin.as<uint32_t>(), ss.d_level_offsets.as<uint32_t>(), ss.d_rec.as<int4>(), stream_side);
                ++launches;
            }
        }
        launch_xpbd_velocity_projection<T>(ds, params, stream_side); ++launches;
        if (any_damped && sched_damp_side.n_components) {
            launch_joint_schedule<T>(ds, params, 1, (uint32_t)sched_damp_side.n_components, sched_damp_side.d_comp_level_begin.as<uint32_t>(),
                                     sched_damp_side.d_level_offsets.as<uint32_t>(), sched_damp_side.d_rec.as<int4>(), stream_side);
            ++launches;
        }
    }
    // all substeps of a step: one stream, or the main islands on `stream` and the side islands on `stream_side` between a fork and a join
    avn_status substep_loop() {
        if (!groups_active) { for (uint32_t s = 0; s < cfg.substeps; ++s) substep(); return AVN_OK; }
        HIPCHK(hipEventRecord(ev_side_fork, stream));
        HIPCHK(hipStreamWaitEvent(stream_side, ev_side_fork, 0));
        dw.body_group = 1u;
        for (uint32_t s = 0; s < cfg.substeps; ++s) substep();
        dw.body_group = 0u;
        for (uint32_t s = 0; s < cfg.substeps; ++s) substep_side();
        HIPCHK(hipEventRecord(ev_side_done, stream_side));
        HIPCHK(hipStreamWaitEvent(stream, ev_side_done, 0));
        return AVN_OK;
    }
    bool islands_active() const { return island_mode && dw.n_joints == 0 && dw.n_manifolds != 0 && !halo_on; }
    avn_status run_substeps() {
        substep_index = 0;
        bias_timed = 0;
        dg_substeps = 0;
        if (halo_on) {   // level-2 sharding: direct launches, the exchanges are RCCL calls on the same stream
            if (!comm.handle) { error = "a halo plan is set but no communicator: call avn_comm_init, or drive the colours through avn_run_color_pass"; return AVN_ERR_STATE; }
            return level2_substeps();
        }
        if constexpr (sizeof(T) == 4) {
            if (islands_active()) {   // every substep of every island block in ONE launch (k_island_substeps)
                launch_island_substeps(dw, params, islands, cfg.substeps, cfg.solver_iterations, stream); ++launches;
                slot_clear_pending = false;   // (the island blocks do not use the slot table: a later colour-launch step clears it itself)
                // (device closed loop: the restitution pass after the loop still runs colour by colour; its overflow pass starts a fresh epoch count)
                if (pipe_dev && ovf_grid_blocks) { launch_overflow_reset(b_ovf_ticket.as<uint32_t>(), dw.n_bodies + 1, pg.ctr + PGC_OVF_TILE, PGC_OVF_TILES, stream); ++launches; }
                ovf_epoch = 0;
                return AVN_OK;
            }
        }
        // the body-centric warm start's slot table (not needed by the island blocks); outside the capture below
        if (slots_dirty && dw.n_manifolds && dw.inc_slot) {
            if (slot_clear_pending) { HIPCHK(hipStreamWaitEvent(stream, ev_slot_clear, 0)); }   // (the table was set to EMPTY on the broad-phase stream while the op pipeline ran)
            launch_build_incidence_slots<T>(dw, stream, slot_clear_pending); launches += 2; slots_dirty = false;
        }
        slot_clear_pending = false;
        const bool flow = pipe_dev && dw.n_manifolds && ovf_grid_blocks;
        if (flow && (uint64_t)cfg.substeps * 2 * cfg.solver_iterations + 2 > PGC_OVF_TILES) { error = "device closed loop: too many contact passes per step for the overflow tickets"; return AVN_ERR_CAPACITY; }
        if (!cfg.use_graph) {
            if (flow) { launch_overflow_reset(b_ovf_ticket.as<uint32_t>(), dw.n_bodies + 1, pg.ctr + PGC_OVF_TILE, PGC_OVF_TILES, stream); ++launches; }
            ovf_epoch = 0;
            avn_status sl = substep_loop();
            ovf_epoch_after_substeps = ovf_epoch;
            return sl;
        }
        // Round 6: the device closed loop launches its FIRST substep directly and replays the graph for the others.  A graph's first kernel starts ~43 us after
        // hipGraphLaunch returns (measured, above / profiles/r06_closed_loop_step110_timeline.txt) with the device idle behind the step's front; a direct launch
        // starts within a few microseconds, and the graph's start-up then hides under the ~190 us the first substep runs.  Same kernels, same order: bit-identical.
        const bool split = graph_split_enabled && pipe_dev && !groups_active && cfg.substeps >= 2;
        uint32_t epoch_after_first = 0;
        if (split) {
            if (flow) { launch_overflow_reset(b_ovf_ticket.as<uint32_t>(), dw.n_bodies + 1, pg.ctr + PGC_OVF_TILE, PGC_OVF_TILES, stream); ++launches; }
            ovf_epoch = 0;
            substep();
            epoch_after_first = ovf_epoch;
            ht("first substep launched directly");
        }
        if (graph_valid && (graph_is_split != split || (split && graph_first_epoch != epoch_after_first))) graph_valid = false;
        if (!graph_valid) {
            if (avn_env("AVN_DBG_CAPTURE")) std::fprintf(stderr, "[avn] substep graph re-captured (M %u, overflow grid %u)\n", dw.n_manifolds, ovf_grid_blocks);
            drop_graph();
            uint32_t before = launches;
            HIPCHK(hipStreamBeginCapture(stream, hipStreamCaptureModeThreadLocal));
            avn_status sl = AVN_OK;
            if (split) {   // substeps 1 .. S-1, continuing the first one's epochs
                ovf_epoch = epoch_after_first; substep_index = 1;
                for (uint32_t s = 1; s < cfg.substeps; ++s) substep();
            } else {
            // the overflow passes' tickets and tile counters restart with every step (a kernel node, replayed first)
            if (flow) { launch_overflow_reset(b_ovf_ticket.as<uint32_t>(), dw.n_bodies + 1, pg.ctr + PGC_OVF_TILE, PGC_OVF_TILES, stream); ++launches; }
            ovf_epoch = 0;
            sl = substep_loop();   // (the side islands' stream joins the capture through the fork event and leaves it at the join)
            }
            graph_is_split = split; graph_first_epoch = epoch_after_first;
            dw.body_group = 0u;
            ovf_epoch_after_substeps = ovf_epoch;
            // whatever went wrong inside the capture, the stream must leave capture mode and the partial graph must not survive
            hipError_t ce = hipStreamEndCapture(stream, &graph);
            if (ce == hipSuccess && sl != AVN_OK) ce = hipErrorUnknown;
            graph_launches = launches - before;
            launches = before;
            if (ce == hipSuccess) ce = hipGraphInstantiate(&graph_exec, graph, nullptr, nullptr, 0);
            if (ce != hipSuccess) {
                (void)hipGetLastError();
                drop_graph();
                error = std::string("substep graph capture failed: ") + hipGetErrorName(ce);
                return AVN_ERR_HIP;
            }
            graph_valid = true;
        }
        ht("substep graph: hipGraphLaunch called");
        // (Round 5, measured and removed: the replay on a stream of its own behind an event, so that its packets would already sit in a queue when the
        //  front finishes -- the device idles 43 us between k_build_incidence_slots and the graph's first kernel although hipGraphLaunch has returned
        //  40 us earlier (AVN_PIPE_HOST_TRACE).  Settled step 2.42 -> 2.62 / 2.64 ms on one box: the start gap grows to 52 us and the join costs more.)
        HIPCHK(hipGraphLaunch(graph_exec, stream));
        ht("hipGraphLaunch returned");
        launches += graph_launches;
        ovf_epoch = ovf_epoch_after_substeps;   // (the restitution pass after the loop continues the step's epochs)
        return AVN_OK;
    }
    uint32_t graph_launches = 0;
    bool graph_is_split = false;          // the captured graph holds substeps 1 .. S-1 (the first one is launched directly)
    uint32_t graph_first_epoch = 0;       // ... and continues the overflow epochs from here
    bool graph_split_enabled = !avn_env("AVN_NO_GRAPH_SPLIT");   // (A/B in `make measure` builds)
    bool bodies_prepared_early = false;    // prepare_solver_bodies + pre_process_velocity_increments of this step are already on the stream
    bool constraints_prepared_early = false;   // k_prepare_contact_constraints of this step is already on the stream (pg_apply_ops)
    bool slot_clear_pending = false;       // DW::inc_slot is being set to EMPTY on stream_bp (ev_slot_clear): build_incidence_slots skips its own memset
    hipEvent_t ev_slot_clear = nullptr;
    uint32_t island_backoff = 0;   // closed-loop steps for which the island blocks are not attempted again
    avn_status solver_front() {   // everything that only READS the rigid-body components
        avn_status st = need_bodies();
        if (st != AVN_OK) return st;
        if ((st = rebuild_joint_schedules()) != AVN_OK) return st;
        if ((st = rebuild_incidence()) != AVN_OK) return st;
        if ((st = rebuild_body_groups()) != AVN_OK) return st;
        if (!bodies_prepared_early) prepare_solver_bodies();   // (device closed loop: enqueued at the step's start, next to the broad phase)
        prepare_joints();
        prepare_contact_constraints();
        stamp(DG_PREP1);
        if (!bodies_prepared_early) pre_process_velocity_increments();
        bodies_prepared_early = false;
        stamp(DG_INC1);
        // host work that only the substep loop needs, done while the prepare kernels above run
        // (device closed loop: a scene whose islands did not fit a workgroup -- one big pile -- is not asked again for a while: the attempt
        //  costs a labelling pass, two read-backs and two synchronisations per step, 0.15 ms of a 1.1 ms Large Pyramid step)
        if (islands_dirty && pipe_dev && island_backoff) { --island_backoff; islands_dirty = false; }
        if (islands_dirty) {
            islands_dirty = false;
            if (pipe_dev) {   // the island builder is host code: fetch the (small) gathered body pairs
                const uint32_t M = dw.n_manifolds;
                std::vector<int2> mb(M);
                HIPCHK(hipMemcpyAsync(mb.data(), dw.m_bodies, (size_t)M * sizeof(int2), hipMemcpyDeviceToHost, stream));
                HIPCHK(hipStreamSynchronize(stream));
                h_m_body1.resize(M); h_m_body2.resize(M);
                for (uint32_t m = 0; m < M; ++m) { h_m_body1[m] = mb[m].x; h_m_body2[m] = mb[m].y; }
            }
            if ((st = rebuild_island_blocks()) != AVN_OK) return st;
            if (pipe_dev && !island_mode) island_backoff = 31;
        }
        ht("constraints / overflow CSR enqueued");
        HIPCHK(hipEventRecord(ev[2], stream));
        if ((st = run_substeps()) != AVN_OK) return st;
        HIPCHK(hipEventRecord(ev[3], stream));
        stamp(DG_SUB1);
        launch_clear_increments<T>(dw, stream); ++launches;
        // restitution == 0 everywhere: every manifold would early-out.  (Level 2: the exchanges are collective and `any_restitution` is a
        // per-rank fact, so the pass always runs there; a rank without restitution launches kernels whose lanes all early-out.)
        if (halo_on) { if ((st = level2_pass(PASS_RESTITUTION_)) != AVN_OK) return st; }
        else if (any_restitution) contact_pass(PASS_RESTITUTION_);
        stamp(DG_REST1);
        HIPCHK(hipGetLastError());
        return AVN_OK;
    }
    avn_status solver_back() {    // the write-back into Position / Rotation / velocities and the ContactGraph
        launch_writeback_solver_bodies<T>(dw, stream); ++launches;
        if (dw.n_joints) { launch_writeback_joint_forces<T>(dw, params, stream); ++launches; }
        stamp(DG_FIN1);
        store_contact_impulses();
        stamp(DG_STORE1);
        HIPCHK(hipGetLastError());
        return AVN_OK;
    }
    avn_status solver() {
        avn_status st = solver_front();
        if (st != AVN_OK) return st;
        return solver_back();
    }
    avn_status dispatch_system(avn_system sys) {
        avn_status st = AVN_OK;
        switch (sys) {
            case AVN_SYS_UPDATE_AABB: if ((st = update_aabb()) != AVN_OK) return st; break;
            case AVN_SYS_COLLECT_COLLISION_PAIRS: if ((st = collect_collision_pairs()) != AVN_OK) return st; break;
            case AVN_SYS_PREPARE_SOLVER_BODIES: prepare_solver_bodies(); break;
            case AVN_SYS_PREPARE_JOINTS: prepare_joints(); break;
            case AVN_SYS_PREPARE_CONTACT_CONSTRAINTS: prepare_contact_constraints(); break;
            case AVN_SYS_PRE_PROCESS_VELOCITY_INCREMENTS: pre_process_velocity_increments(); break;
            case AVN_SYS_INTEGRATE_VELOCITIES: integrate_velocities(); break;
            case AVN_SYS_WARM_START: warm_start(false); break;
            case AVN_SYS_SOLVE_CONTACTS_BIAS: contact_pass(PASS_SOLVE_BIAS); break;
            case AVN_SYS_INTEGRATE_POSITIONS: integrate_positions(); break;
            case AVN_SYS_SOLVE_CONTACTS_RELAX: contact_pass(PASS_SOLVE_RELAX); break;
            case AVN_SYS_XPBD_SOLVE: xpbd_solve(true); break;
            case AVN_SYS_XPBD_VELOCITY_PROJECTION: xpbd_velocity_projection(); break;
            case AVN_SYS_JOINT_DAMPING: joint_damping(); break;
            case AVN_SYS_CLEAR_VELOCITY_INCREMENTS: launch_clear_increments<T>(dw, stream); ++launches; break;
            case AVN_SYS_SOLVE_RESTITUTION: if (any_restitution) contact_pass(PASS_RESTITUTION_); break;
            case AVN_SYS_WRITEBACK_SOLVER_BODIES:
                launch_writeback_solver_bodies<T>(dw, stream); ++launches;
                if (dw.n_joints) { launch_writeback_joint_forces<T>(dw, params, stream); ++launches; }
                break;
            case AVN_SYS_STORE_CONTACT_IMPULSES: store_contact_impulses(); break;
            case AVN_SYS_NARROW_PHASE: if ((st = narrow_phase()) != AVN_OK) return st; break;
            case AVN_SYS_SUBSTEP: substep(); break;
            case AVN_SYS_SOLVER: {
                HIPCHK(hipEventRecord(ev[0], stream)); HIPCHK(hipEventRecord(ev[1], stream));
                if ((st = solver()) != AVN_OK) return st;
                HIPCHK(hipEventRecord(ev[4], stream));
                ev_valid = true;
                break;
            }
            default: error = "run_system: unknown system"; return AVN_ERR_BAD_ARG;
        }
        HIPCHK(hipGetLastError());
        return AVN_OK;
    }
    // single systems run outside avn_step: the dataflow passes' per-step state has to be fresh
    avn_status flow_begin_standalone() {
        if (!dw.n_manifolds || !(pipe_dev && ovf_grid_blocks)) return AVN_OK;
        launch_overflow_reset(b_ovf_ticket.as<uint32_t>(), dw.n_bodies + 1, pg.ctr + PGC_OVF_TILE, PGC_OVF_TILES, stream);
        ovf_epoch = 0;
        return AVN_OK;
    }
    avn_status run_system(avn_system sys) override {
        avn_status st = need_bodies();
        if (st != AVN_OK) return st;
        if (despawn_needs_joints) { error = "run_system: avn_despawn removed joints: upload the remaining joints (avn_joints_upload) first"; return AVN_ERR_STATE; }
        if (despawn_broken) { error = "run_system: an avn_despawn failed half-way; restart the closed loop"; return AVN_ERR_STATE; }
        if ((st = rebuild_joint_schedules()) != AVN_OK) return st;
        if ((st = rebuild_incidence()) != AVN_OK) return st;
        if (sys != AVN_SYS_SOLVER && (st = flow_begin_standalone()) != AVN_OK) return st;
        if ((st = dispatch_system(sys)) != AVN_OK) return st;
        HIPCHK(hipStreamSynchronize(stream));
        return AVN_OK;
    }
    avn_status profile_system(avn_system sys, uint32_t repeats, double* total_ms, uint32_t* n_launches) override {
        avn_status st = need_bodies();
        if (st != AVN_OK) return st;
        if ((st = rebuild_joint_schedules()) != AVN_OK) return st;
        if ((st = rebuild_incidence()) != AVN_OK) return st;
        if ((st = flow_begin_standalone()) != AVN_OK) return st;
        hipEvent_t a, b;
        HIPCHK(hipEventCreateWithFlags(&a, EV_FLAGS)); HIPCHK(hipEventCreateWithFlags(&b, EV_FLAGS));
        HIPCHK(hipStreamSynchronize(stream));
        uint32_t before = launches;
        HIPCHK(hipEventRecord(a, stream));  // events on the stream the kernels are launched on
        for (uint32_t r = 0; r < repeats; ++r)
            if ((st = dispatch_system(sys)) != AVN_OK) break;
        HIPCHK(hipEventRecord(b, stream));
        HIPCHK(hipStreamSynchronize(stream));
        float ms = 0;
        HIPCHK(hipEventElapsedTime(&ms, a, b));
        (void)hipEventDestroy(a); (void)hipEventDestroy(b);
        if (total_ms) *total_ms = ms;
        if (n_launches) *n_launches = launches - before;
        return st;
    }
    avn_status step() override {
        avn_status st = need_bodies();
        if (st != AVN_OK) return st;
        for (bool& b : dg_stamped) b = false;
        dg_np = false;
        if (pipe_on) return pipe_dev ? pipeline_step_device() : pipeline_step();
        launches = 0;
        static const bool host_trace = avn_env("AVN_HOST_TRACE") != nullptr;   // debugging aid: where the HOST spends a step (us since the call)
        const auto ht0 = std::chrono::steady_clock::now();
        auto hus = [&]() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - ht0).count(); };
        double h_bp = 0, h_front = 0, h_wait = 0;
        HIPCHK(hipEventRecord(ev[0], stream));
        if (host_trace) {   // (debugging aid) a ring of (start, end) events per step, read back by avn_synchronize: spans and the gaps BETWEEN steps
            if (trace_ev.empty()) { trace_ev.resize(2 * TRACE_STEPS); for (auto& e : trace_ev) HIPCHK(hipEventCreateWithFlags(&e, EV_FLAGS)); }
            HIPCHK(hipEventRecord(trace_ev[2 * (trace_n % TRACE_STEPS)], stream));
        }
        const bool overlap = overlap_bp && have_colliders;
        bp_timed = false;
        static const bool bp_first = avn_env("AVN_BP_ENQUEUE_FIRST") != nullptr;   // (A/B: round 2's order)
        const bool front_first = overlap && !bp_first;
        if (front_first) {
            // the solver's front only READS the rigid-body components, as the broad phase does: with the broad phase on its own stream the
            // order in which the HOST enqueues the two is free, and the world's stream is the one that must never wait for the host -- the
            // broad phase's ~20 launches take the host 70 us, and it has a millisecond of slack
            HIPCHK(hipEventRecord(ev[1], stream));
            if ((st = solver_front()) != AVN_OK) return st;
            h_front = hus();
        }
        if (have_colliders) {
            if (overlap) {
                HIPCHK(hipStreamWaitEvent(stream_bp, ev[0], 0));  // after the previous step's write-back
                bs = stream_bp;
                HIPCHK(hipEventRecord(ev_bp_t0, stream_bp));
            }
            st = update_aabb(1);
            if (st == AVN_OK) st = collect_launch();
            if (overlap) { (void)hipEventRecord(ev_bp_t1, stream_bp); bp_timed = true; }
            if (st != AVN_OK) { bs = stream; return st; }
        }
        h_bp = hus();
        if (!front_first) {
            HIPCHK(hipEventRecord(ev[1], stream));
            st = solver_front();                              // enqueued while the broad phase runs / its pair counters travel back
            h_front = hus();
        }
        if (st == AVN_OK) st = collect_finish();          // (emit pass only when the step found new pairs)
        h_wait = hus();
        if (overlap) {
            (void)hipEventRecord(ev_bp_done, stream_bp);
            (void)hipStreamWaitEvent(stream, ev_bp_done, 0);  // the write-back must not overtake k_update_aabb's reads
            bs = stream;
        }
        if (st != AVN_OK) return st;
        if ((st = solver_back()) != AVN_OK) return st;
        HIPCHK(hipEventRecord(ev[4], stream));
        if (host_trace) { HIPCHK(hipEventRecord(trace_ev[2 * (trace_n % TRACE_STEPS) + 1], stream)); ++trace_n; }
        ev_valid = true;
        last_timers.kernel_launches = launches;
        if (host_trace && avn_env("AVN_HOST_TRACE")[0] == '2') std::fprintf(stderr, "[avn host] broad phase enqueued %.0f us, solver front %.0f, counters back %.0f, step enqueued %.0f\n", h_bp, h_front, h_wait, hus());
        return AVN_OK;
    }
    static constexpr uint32_t TRACE_STEPS = 64;
    std::vector<hipEvent_t> trace_ev;
    uint32_t trace_n = 0;
    avn_status synchronize() override {
        HIPCHK(hipStreamSynchronize(stream)); HIPCHK(hipStreamSynchronize(stream_bp));
        if (trace_n >= 8 && trace_n <= TRACE_STEPS) {
            double span = 0, gap = 0;
            for (uint32_t i = 4; i + 1 < trace_n; ++i) {
                float a = 0, b = 0;
                HIPCHK(hipEventElapsedTime(&a, trace_ev[2 * i], trace_ev[2 * i + 1]));
                HIPCHK(hipEventElapsedTime(&b, trace_ev[2 * i + 1], trace_ev[2 * i + 2]));
                span += a; gap += b;
            }
            if (avn_env("AVN_HOST_TRACE")[0] == '3') { std::fprintf(stderr, "[avn trace] spans (ms):"); for (uint32_t i = 0; i < trace_n; ++i) { float a = 0; HIPCHK(hipEventElapsedTime(&a, trace_ev[2 * i], trace_ev[2 * i + 1])); std::fprintf(stderr, " %.3f", a); } std::fprintf(stderr, "\n"); }
            std::fprintf(stderr, "[avn trace] %u steps: mean span %.4f ms, mean gap to the next step's start %.4f ms\n", trace_n - 5, span / (trace_n - 5), gap / (trace_n - 5));
        }
        trace_n = 0;
        return pg_error_check();
    }

