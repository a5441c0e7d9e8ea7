// world/sleeping.hpp -- fragment of the body of `template <class T> struct World` (avn_world.hip includes it inside the class):
// persistent simulation islands and the ACTUATION of sleeping in the device closed loop (include/avian_mi355x.h: avn_sleeping_enable).
//
// Who does what (reference: dynamics/solver/islands/mod.rs, islands/sleeping.rs, collision/narrow_phase/system_param.rs:141-398):
//   * the island manager (avn_islands.hpp, host C++) keeps what the reference keeps in linked lists threaded through ECS components -- island
//     membership and body-list order, slab ids, the colliders' edge lists, which pairs touch / sleep -- and decides: merges inside the status
//     loop, the deferred split, which islands sleep and wake, and in WHICH ORDER their manifolds leave / re-enter the ConstraintGraph;
//   * the device does the arithmetic and the bulk state: SleepTimers from the SolverBody velocities (k_sleep_timers_flags), the narrow phase
//     that skips the pairs of ContactGraph::sleeping_pairs (AVN_CP_ROW_SLEEPING), the Sleeping flag of bodies (no SolverBody, inactive
//     interval), and the pops / pushes themselves: the manager's ordered list goes through the same op pipeline as the status loop's ops
//     (pg_apply_ops: dataflow greedy colouring, exact swap_remove replay), so colours and list orders are the reference's.
// Per step the host reads the loop's changes (8 bytes per change), the new pairs (28 bytes each) and the timers (5 bytes per body); with a
// sleeping pile all three are empty or tiny.  avn_step synchronises at its end in this mode (the Sleeping set needs the timers).
    bool slp_on = false;
    // Every body that could own a SolverBody sleeps (checked after the Sleeping set of a step; any upload, WakeBody, despawn or configuration
    // change clears it): nothing can move, no AABB changes, no pair starts or stops touching, no timer advances -- avn_step is the identity and
    // returns without a launch ("a sleeping pile costs nothing", islands/sleeping.rs:243-280).
    bool slp_world_asleep = false;
    // ... but only from the SECOND such step on: the step in which the last island fell asleep still moved its bodies, so the next step's AABBs and
    // contacts are new (a non-touching pair may start touching and wake the island again: DESIGN.md 4.8's flip-flop).  A step that STARTED with
    // everything asleep and changed nothing (no new pair, no status change, nothing woken) proves the state stationary.
    bool slp_world_idle = false, slp_step_started_asleep = false, slp_step_changed = false;
    void touched() override { slp_world_asleep = slp_world_idle = false; }   // (avn_abi.cpp: every state-changing entry point other than avn_step)
    IslandManager isl;
    SleepParams<T> slp_k;
    float slp_time_to_sleep = 0.5f;
    DevBuf b_slp_lin, b_slp_ang, b_slp_dis, b_slp_timer, b_slp_flags;
    Pinned pin_slp_ops, pin_slp_pairs, pin_slp_timers;
    std::vector<uint8_t> h_rb_type, h_body_flags;      // host copies of the uploaded RigidBody type / body flags (island nodes, Sleeping)
    uint32_t slp_n_awake = 0, slp_last_slept = 0, slp_last_woken = 0, slp_last_popped = 0, slp_last_pushed = 0;
    double slp_host_ms = 0;
    // measurement aid (`make measure` build, AVN_SLP_TRACE=1): host milliseconds of the island bookkeeping's phases, one line per step on stderr
    bool slp_trace = avn_env("AVN_SLP_TRACE") != nullptr;
    double slp_tr[10] = {0};
    static double slp_now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
    std::vector<uint32_t> slp_list;                    // scratch: cids | kinds of an op list
    // Round 6.  (a) A step whose status changes name no Sleeping body cannot queue a WakeIslands (k_pg_scan_classify counts them): the solver is enqueued FIRST and the
    // island manager digests the step's pairs and changes while the device solves (slp_fast_step).  (b) split_island's neighbour lists come from the device
    // (launch_isl_adjacency, k_graph.hip) when a split candidate exists at the step's start: the host's walk then reads a 4-byte CSR instead of its edge lists.
    bool slp_fast_step = false;
    bool slp_adj_enabled = !avn_env("AVN_SLP_HOST_SPLIT"), slp_fast_enabled = !avn_env("AVN_SLP_NO_FAST");   // (A/B switches of `make measure` builds)
    DevBuf b_adj_parent, b_adj_label, b_adj_ccctr;
    bool slp_async_enabled = !avn_env("AVN_SLP_SYNC_SPLIT");
    size_t slp_adj_min_bodies = avn_env("AVN_SLP_ADJ_MIN") ? (size_t)std::atol(avn_env("AVN_SLP_ADJ_MIN")) : 2048;   // candidate islands below this are split from the manager's own lists
    DevBuf b_adj_key2, b_adj_other, b_adj_body, b_adj_ka, b_adj_va, b_adj_kb, b_adj_vb, b_adj_off, b_adj_out, b_adj_rank, b_adj_hist, b_adj_sums;
    Pinned pin_adj2[2];   // double-buffered: a worker thread may still walk the previous build's lists
    uint32_t adj_flip = 0, adj_used = 0;
    hipEvent_t ev_adj = nullptr, ev_adj_go = nullptr;
    uint64_t adj_rank_epoch = 0;
    uint32_t adj_rank_slots = 0, adj_n = 0, adj_bodies = 0;
    bool adj_pending = false;
    std::vector<uint32_t> adj_rank_host;
    avn_status sleeping_adjacency_launch() {
        adj_pending = false;
        const uint32_t N = dw.n_bodies, n_slots = (uint32_t)slot_entity.size(), n = 2u * dw.n_manifolds;
        if (!slp_adj_enabled || !N || pipe_stats.pairs_added >= (1ull << 32) || n_slots >= (1u << 30) || dw.n_manifolds >= (1u << 30)) return AVN_OK;   // (the manager's own edge lists serve)
        hipError_t err = hipSuccess;
        if (adj_rank_epoch != isl.collider_epoch() || adj_rank_slots != n_slots) {
            adj_rank_host.resize(std::max<size_t>(n_slots, 1));
            isl.collider_ranks(slot_entity.data(), n_slots, adj_rank_host.data());
            HIPCHK(hipStreamSynchronize(stream_bp));
            b_adj_rank.ensure(adj_rank_host.size() * 4, err);
            if (err != hipSuccess) { error = "hipMalloc failed (collider ranks)"; return AVN_ERR_OOM; }
            HIPCHK(hipMemcpy(b_adj_rank.p, adj_rank_host.data(), (size_t)n_slots * 4, hipMemcpyHostToDevice));
            adj_rank_epoch = isl.collider_epoch(); adj_rank_slots = n_slots;
        }
        const size_t cap = std::max<size_t>(n, 64);
        if (b_adj_ka.cap < cap * 4) {
            HIPCHK(hipStreamSynchronize(stream_bp));
            const size_t c = cap + cap / 2;
            for (DevBuf* b : {&b_adj_key2, &b_adj_other, &b_adj_body, &b_adj_ka, &b_adj_va, &b_adj_kb, &b_adj_vb, &b_adj_out}) { b->ensure(c * 4, err); if (err != hipSuccess) { error = "hipMalloc failed (island adjacency)"; return AVN_ERR_OOM; } }
            b_adj_hist.ensure(((size_t)256 * radix_blocks((uint32_t)c) + 256) * 4, err);
            if (err != hipSuccess) { error = "hipMalloc failed (island adjacency)"; return AVN_ERR_OOM; }
            b_adj_sums.ensure((std::max<size_t>(scan_block_sums_needed(256 * radix_blocks((uint32_t)c)), scan_block_sums_needed((uint32_t)c)) + 16) * 4, err);
            if (err != hipSuccess) { error = "hipMalloc failed (island adjacency)"; return AVN_ERR_OOM; }
            HIPCHK(hipMemset(b_adj_sums.p, 0, b_adj_sums.cap));   // (the one-launch scan's state: zero once, self-cleaning afterwards)
        }
        b_adj_off.ensure(((size_t)N + 2) * 4, err); if (err != hipSuccess) { error = "hipMalloc failed (island adjacency)"; return AVN_ERR_OOM; }
        b_adj_parent.ensure((size_t)N * 4, err); if (err != hipSuccess) { error = "hipMalloc failed (island adjacency)"; return AVN_ERR_OOM; }
        b_adj_label.ensure((size_t)N * 4, err); if (err != hipSuccess) { error = "hipMalloc failed (island adjacency)"; return AVN_ERR_OOM; }
        b_adj_ccctr.ensure(64, err); if (err != hipSuccess) { error = "hipMalloc failed (island adjacency)"; return AVN_ERR_OOM; }
        if (pin_adj2[adj_flip].cap < ((size_t)2 * N + 2 + n) * 4 + 64) {
            avn_status js = isl.split_join();   // (growing frees the old block)
            if (js != AVN_OK) return slp_fail(js);
            if (pin_adj2[adj_flip].ensure(((size_t)2 * N + 2 + n) * 4 + 64) != hipSuccess) { error = "hipHostMalloc failed"; return AVN_ERR_OOM; }
        }
        adj_used = adj_flip;
        if (!ev_adj) { HIPCHK(hipEventCreateWithFlags(&ev_adj, hipEventDisableTiming)); HIPCHK(hipEventCreateWithFlags(&ev_adj_go, hipEventDisableTiming | EV_FLAGS)); }
        // behind every op batch of this step (the rows' colours are the constraint handles the walk follows), next to the solver
        HIPCHK(hipEventRecord(ev_adj_go, stream));
        HIPCHK(hipStreamWaitEvent(stream_bp, ev_adj_go, 0));
        IslAdj a{(uint32_t)(b_adj_ka.cap / 4), b_adj_key2.as<uint32_t>(), b_adj_other.as<uint32_t>(), b_adj_body.as<uint32_t>(),
                 b_adj_ka.as<uint32_t>(), b_adj_va.as<uint32_t>(), b_adj_kb.as<uint32_t>(), b_adj_vb.as<uint32_t>()};
        const uint32_t pad = 2u * n_slots;
        launch_isl_adjacency(pg, ct.meta, dw.bmeta, b_adj_rank.as<uint32_t>(), b_handles.as<uint32_t>(), dw.n_manifolds, N, std::max(1u, bits_for((uint32_t)pipe_stats.pairs_added)), bits_for(pad), pad, a,
                             b_adj_hist.as<uint32_t>(), b_adj_sums.as<uint32_t>(), b_adj_off.as<uint32_t>(), b_adj_out.as<uint32_t>(), stream_bp);
        HIPCHK(hipGetLastError());
        uint32_t* h = (uint32_t*)pin_adj2[adj_used].p;
        HIPCHK(hipMemcpyAsync(h, b_adj_off.p, ((size_t)N + 2) * 4, hipMemcpyDeviceToHost, stream_bp));
        if (n) HIPCHK(hipMemcpyAsync(h + N + 2, b_adj_out.p, (size_t)n * 4, hipMemcpyDeviceToHost, stream_bp));
        // the components of the same edges (+ joints), 4 bytes per body: with them the split's bookkeeping (pieces, keys, sizes) needs no walk, and the walk -- the order
        // inside the pieces' body lists -- leaves the step's critical path for a worker thread (IslandManager::split_candidate_labelled_async)
        HIPCHK(hipMemsetAsync(b_adj_ccctr.p, 0, 64, stream_bp));
        launch_islands_rows<T>(dw, pg.bodies, pg.color, b_handles.as<uint32_t>(), dw.n_manifolds, b_adj_parent.as<uint32_t>(), b_adj_label.as<uint32_t>(), b_adj_ccctr.as<uint32_t>(), stream_bp);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(h + N + 2 + n, b_adj_label.p, (size_t)N * 4, hipMemcpyDeviceToHost, stream_bp));
        HIPCHK(hipEventRecord(ev_adj, stream_bp));
        launches += 10;
        adj_pending = true; adj_n = n; adj_bodies = N;
        return AVN_OK;
    }
    // split_island(candidate), SolverSystems::Finalize
    avn_status sleeping_split() {
        if (!adj_pending) { const avn_status st = isl.split_candidate_now(); return st == AVN_OK ? st : slp_fail(st); }
        adj_pending = false;
        if (!isl.split_pending()) return AVN_OK;   // (the candidate merged away or has nothing to split: the CSR stays unread)
        HIPCHK(spin_event(ev_adj));
        const uint32_t* h = (const uint32_t*)pin_adj2[adj_used].p;
        if (h[adj_bodies + 1] > adj_n) { error = "island adjacency: more entries than 2 x constraint handles"; return AVN_ERR_STATE; }
        avn_status st = isl.split_join();   // (the previous walk, if it still runs, reads the other staging buffer)
        if (st != AVN_OK) return slp_fail(st);
        const uint32_t* lab = h + adj_bodies + 2 + adj_n;
        if (avn_env("AVN_SLP_CHECK_ADJ")) {   // (`make measure` builds: the device's lists against the manager's own)
            const std::string w = isl.check_adjacency(h, h + adj_bodies + 2, adj_bodies);
            if (!w.empty()) {
                error = "sleeping (AVN_SLP_CHECK_ADJ): the device-built adjacency differs from the island manager's edge lists: " + w;
                return AVN_ERR_STATE;
            }
        }
        if (slp_async_enabled) { st = isl.split_candidate_labelled_async(h, h + adj_bodies + 2, adj_bodies, lab); adj_flip ^= 1u; }   // (the walk reads this staging buffer: the next build fills the other)
        else st = isl.split_candidate_adjacency(h, h + adj_bodies + 2, adj_bodies);
        return st == AVN_OK ? st : slp_fail(st);
    }
    float slp_lin_default = 0.0f, slp_ang_default = 0.0f;   // the world-level SleepThreshold: what a body spawned after avn_sleeping_enable gets in the per-body arrays
    uint32_t slp_bodies = 0;                                 // bodies the per-body arrays (timer, flags, thresholds, SleepingDisabled) cover
    // bodies spawned inside the loop: SleepTimer 0 (the component's default, sleeping.rs:96-110), the world's thresholds, not SleepingDisabled
    avn_status slp_grow_bodies(uint32_t n) {
        if (n <= slp_bodies) return AVN_OK;
        HIPCHK(hipStreamSynchronize(stream));
        const uint32_t old = slp_bodies;
        hipError_t e;
        b_slp_timer.ensure((size_t)n * 4, e, true, stream);
        if (e != hipSuccess) { error = "hipMalloc failed (sleep timers)"; return AVN_ERR_OOM; }
        b_slp_flags.ensure((size_t)n, e);
        if (e != hipSuccess) { error = "hipMalloc failed (sleep flags)"; return AVN_ERR_OOM; }
        HIPCHK(hipMemsetAsync((float*)b_slp_timer.p + old, 0, (size_t)(n - old) * 4, stream));
        auto grow_f = [&](DevBuf& b, const float*& field, float fill) -> avn_status {
            if (!field) return AVN_OK;
            b.ensure((size_t)n * 4, e, true, stream);
            if (e != hipSuccess) { error = "hipMalloc failed (per-body sleep thresholds)"; return AVN_ERR_OOM; }
            std::vector<float> tail(n - old, fill);
            HIPCHK(hipMemcpy((float*)b.p + old, tail.data(), tail.size() * 4, hipMemcpyHostToDevice));
            field = (const float*)b.p;
            return AVN_OK;
        };
        avn_status st;
        if ((st = grow_f(b_slp_lin, slp_k.body_lin, slp_lin_default)) != AVN_OK || (st = grow_f(b_slp_ang, slp_k.body_ang, slp_ang_default)) != AVN_OK) return st;
        if (slp_k.body_disabled) {
            b_slp_dis.ensure((size_t)n, e, true, stream);
            if (e != hipSuccess) { error = "hipMalloc failed (SleepingDisabled flags)"; return AVN_ERR_OOM; }
            HIPCHK(hipMemsetAsync((uint8_t*)b_slp_dis.p + old, 0, n - old, stream));
            slp_k.body_disabled = (const uint8_t*)b_slp_dis.p;
        }
        HIPCHK(hipStreamSynchronize(stream));
        slp_bodies = n;
        return AVN_OK;
    }
    bool slp_node(uint32_t b) const { return h_rb_type[b] != AVN_RB_STATIC && !(h_body_flags[b] & AVN_BODY_DISABLED); }   // BodyIslandNode, islands/mod.rs:96-140
    avn_status slp_fail(avn_status st) { error = isl.error; return st; }

    avn_status sleeping_enable(const avn_sleep_params* p) override {
        slp_world_asleep = slp_world_idle = false;
        if (!p) {
            if (!slp_on) return AVN_OK;
            double ms = 0;
            for (uint32_t b = 0; b < dw.n_bodies; ++b)
                if (isl.body_has_node(b) && isl.body_sleeps(b)) { avn_status st = isl.wake_body(b); if (st != AVN_OK) return slp_fail(st); if ((st = sleeping_apply_result(false, ms)) != AVN_OK) return st; }
            slp_on = false;
            isl.reset();
            return AVN_OK;
        }
        if (p->struct_size != sizeof(avn_sleep_params)) { error = "sleeping_enable: bad params"; return AVN_ERR_BAD_ARG; }
        if (!pipe_on || !pipe_dev) { error = "sleeping_enable: needs the device closed loop (avn_pipeline_enable(1))"; return AVN_ERR_STATE; }
        if (pgm_next_id) { error = "sleeping_enable: enable it before the first step of the closed loop"; return AVN_ERR_STATE; }
        if (dsh_on) { error = "sleeping_enable: not combined with avn_dshard_enable"; return AVN_ERR_STATE; }
        const uint32_t n = dw.n_bodies;
        if (h_rb_type.size() != n) { error = "sleeping_enable: upload bodies first"; return AVN_ERR_STATE; }
        avn_status st;
        isl.reset();
        for (uint32_t b = 0; b < n; ++b) if (slp_node(b) && (st = isl.body_add(b)) != AVN_OK) return slp_fail(st);
        for (uint32_t s = 0; s < slot_entity.size(); ++s) {
            const uint32_t b = (uint32_t)h_col_body[s];
            if ((st = isl.collider_add(slot_entity[s], slp_node(b) ? b : IslandManager::NONE)) != AVN_OK) return slp_fail(st);
        }
        for (uint32_t j = 0; j < h_j_body1.size(); ++j) if ((st = isl.joint_add(j, (uint32_t)h_j_body1[j], (uint32_t)h_j_body2[j])) != AVN_OK) return slp_fail(st);
        for (uint32_t b = 0; b < n; ++b) if (slp_node(b) && (h_body_flags[b] & AVN_BODY_SLEEPING) && (st = isl.sleep_body(b)) != AVN_OK) return slp_fail(st);   // bodies uploaded asleep
        // the timer kernel's parameters and per-body components
        slp_k.length_unit_squared = (T)p->length_unit * (T)p->length_unit;
        slp_k.lin_threshold_squared = (T)(p->linear_threshold * std::fabs(p->linear_threshold));
        slp_k.ang_threshold_squared = (T)(p->angular_threshold * std::fabs(p->angular_threshold));
        slp_k.delta_secs = p->delta_secs; slp_k.time_to_sleep = p->time_to_sleep;
        slp_k.body_lin = nullptr; slp_k.body_ang = nullptr; slp_k.body_disabled = nullptr;
        slp_time_to_sleep = p->time_to_sleep;
        slp_lin_default = p->linear_threshold; slp_ang_default = p->angular_threshold; slp_bodies = n;
        HIPCHK(hipStreamSynchronize(stream));
        hipError_t err;
        auto up = [&](DevBuf& b, const void* src, size_t bytes) -> const void* {
            if (!src) return nullptr;
            b.ensure(std::max<size_t>(bytes, 1), err);
            if (err != hipSuccess) return nullptr;
            if (hipMemcpy(b.p, src, bytes, hipMemcpyHostToDevice) != hipSuccess) { err = hipErrorUnknown; return nullptr; }
            return b.p;
        };
        err = hipSuccess;
        slp_k.body_lin = (const float*)up(b_slp_lin, p->body_linear_threshold, (size_t)n * 4);
        slp_k.body_ang = (const float*)up(b_slp_ang, p->body_angular_threshold, (size_t)n * 4);
        slp_k.body_disabled = (const uint8_t*)up(b_slp_dis, p->body_sleeping_disabled, n);
        if (err != hipSuccess) { error = "sleeping_enable: device allocation / copy failed"; return AVN_ERR_OOM; }
        b_slp_timer.ensure(std::max<size_t>(n, 1) * 4, err); if (err != hipSuccess) { error = "hipMalloc failed"; return AVN_ERR_OOM; }
        b_slp_flags.ensure(std::max<size_t>(n, 1), err); if (err != hipSuccess) { error = "hipMalloc failed"; return AVN_ERR_OOM; }
        HIPCHK(hipMemset(b_slp_timer.p, 0, std::max<size_t>(n, 1) * 4));
        slp_n_awake = slp_last_slept = slp_last_woken = slp_last_popped = slp_last_pushed = 0;
        slp_on = true;
        return AVN_OK;
    }
    // The device side of a batch of the manager's commands (SleepIslands / WakeIslands): SLEEPING bits of the pairs, Sleeping flags of the
    // bodies (+ SleepTimer = 0 for woken ones), and the pops / pushes IN THE MANAGER'S ORDER through the op pipeline.
    avn_status sleeping_apply_result(bool count, double& host_ms) {
        const std::vector<uint32_t>&popped = isl.popped(), &pushed = isl.pushed(), &ps = isl.pairs_slept(), &pw = isl.pairs_woken(), &bsl = isl.bodies_slept(), &bw = isl.bodies_woken();
        if (count) { slp_last_popped = (uint32_t)popped.size(); slp_last_pushed = (uint32_t)pushed.size(); }
        const size_t n_ops = popped.size() + pushed.size();
        const size_t words = 2 * n_ops + ps.size() + pw.size() + bsl.size() + bw.size();
        if (!words) return AVN_OK;
        slp_list.clear();
        slp_list.insert(slp_list.end(), popped.begin(), popped.end()); slp_list.insert(slp_list.end(), pushed.begin(), pushed.end());
        slp_list.insert(slp_list.end(), popped.size(), 2u /* PG_KIND_POP */); slp_list.insert(slp_list.end(), pushed.size(), 1u /* PG_KIND_PUSH */);
        const size_t o_ps = slp_list.size(); slp_list.insert(slp_list.end(), ps.begin(), ps.end());
        const size_t o_pw = slp_list.size(); slp_list.insert(slp_list.end(), pw.begin(), pw.end());
        const size_t o_bs = slp_list.size(); slp_list.insert(slp_list.end(), bsl.begin(), bsl.end());
        const size_t o_bw = slp_list.size(); slp_list.insert(slp_list.end(), bw.begin(), bw.end());
        avn_status st = stage_reserve(slp_list.size() * 4 + 1024);
        if (st != AVN_OK) return st;
        const uint32_t* d;
        if ((st = stage_in<uint32_t>(slp_list.data(), slp_list.size(), &d)) != AVN_OK) return st;
        launch_rows_set_sleeping<T>(ct, d + o_ps, (uint32_t)ps.size(), 1u, stream);
        launch_rows_set_sleeping<T>(ct, d + o_pw, (uint32_t)pw.size(), 0u, stream);
        launch_bodies_set_sleeping<T>(dw, d + o_bs, (uint32_t)bsl.size(), 1u, nullptr, stream);
        launch_bodies_set_sleeping<T>(dw, d + o_bw, (uint32_t)bw.size(), 0u, b_slp_timer.as<float>(), stream);
        launches += 4;
        HIPCHK(hipGetLastError());
        for (uint32_t b : bsl) { h_body_flags[b] |= AVN_BODY_SLEEPING; h_body_has_sb[b] = 0; }
        for (uint32_t b : bw) { h_body_flags[b] &= (uint8_t)~AVN_BODY_SLEEPING; h_body_has_sb[b] = h_rb_type[b] != AVN_RB_STATIC && !(h_body_flags[b] & AVN_BODY_DISABLED); }
        if (!bsl.empty() || !bw.empty()) { joint_schedule_dirty = true; groups_dirty = true; incidence_dirty = true; bodies_prepared_early = false; }   // (which bodies own a SolverBody changed: prepare_solver_bodies again)
        if (n_ops) { if ((st = pg_apply_ops((uint32_t)n_ops, 0u, 0u, d, d + n_ops, host_ms)) != AVN_OK) return st; }
        else HIPCHK(hipStreamSynchronize(stream));   // (the staging arena is reused by the next call)
        return AVN_OK;
    }
    // after the status loop of the step (its ops are already in the colour lists): the manager sees the new pairs and the loop's link / unlink
    // events in the reference's order, then the deferred WakeIslands (system_param.rs:391-398) runs before the solver
    // the manager's half of the status loop: the step's new pairs (emission order) and status changes (ascending ContactId), then the deferred WakeIslands' decision
    avn_status sleeping_manager_digest(uint32_t n_new_pairs, uint32_t n_ops, double& host_ms) {
        auto t0 = std::chrono::steady_clock::now();
        avn_status st;
        const double m0 = slp_now();
        if (n_new_pairs) {
            const avn_pair* pr = (const avn_pair*)pin_slp_pairs.p;
            const uint32_t* ids = (const uint32_t*)((const char*)pin_slp_pairs.p + (size_t)n_new_pairs * sizeof(avn_pair));
            if ((st = isl.pairs_add(ids, pr, n_new_pairs)) != AVN_OK) return slp_fail(st);
        }
        const double m1 = slp_now();
        if (n_ops) {
            const uint32_t* cid = (const uint32_t*)pin_slp_ops.p;
            if ((st = isl.status_changes(cid, cid + n_ops, n_ops)) != AVN_OK) return slp_fail(st);
        }
        const double m2 = slp_now();
        if ((st = isl.flush_wake()) != AVN_OK) return slp_fail(st);
        host_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        slp_tr[0] = m1 - m0; slp_tr[1] = m2 - m1; slp_tr[2] = slp_now() - m2;
        return AVN_OK;
    }
    // after the status loop of the step (its ops are already in the colour lists): the manager sees the new pairs and the loop's link / unlink
    // events in the reference's order, then the deferred WakeIslands (system_param.rs:391-398) runs before the solver
    avn_status sleeping_after_status_loop(uint32_t n_new_pairs, uint32_t n_ops, double& host_ms) {
        avn_status st = sleeping_manager_digest(n_new_pairs, n_ops, host_ms);
        if (st != AVN_OK) return st;
        const double m3 = slp_now();
        st = sleeping_apply_result(false, host_ms);
        slp_tr[3] = slp_now() - m3;
        return st;
    }
    // the same digest AFTER the solver was enqueued (slp_fast_step: no change of the step names a Sleeping body, so nothing can wake): host work under the device's
    avn_status sleeping_digest_deferred(uint32_t n_new_pairs, uint32_t n_ops, double& host_ms) {
        avn_status st = sleeping_manager_digest(n_new_pairs, n_ops, host_ms);
        if (st != AVN_OK) return st;
        slp_tr[3] = 0;
        if (!isl.pushed().empty() || !isl.bodies_woken().empty() || !isl.pairs_woken().empty()) {
            error = "sleeping: the island manager woke an island in a step whose status changes named no Sleeping body (device flags and manager disagree)";
            return AVN_ERR_STATE;
        }
        return AVN_OK;
    }
    // split_island (SolverSystems::Finalize) and the Sleeping set (update_sleeping_states, wake_islands_with_sleeping_disabled, sleep_islands,
    // then SleepIslands / WakeIslands)
    avn_status sleeping_after_solver() {
        const uint32_t n = dw.n_bodies;
        double host_ms = 0;
        auto t0 = std::chrono::steady_clock::now();
        avn_status st;
        double m0 = slp_now();
        launch_sleep_timers_flags<T>(dw, slp_k, b_slp_timer.as<float>(), b_slp_flags.as<uint8_t>(), stream); ++launches;   // behind the write-back: SolverBody velocities of this step
        HIPCHK(hipGetLastError());
        if (pin_slp_timers.ensure((size_t)n * 5 + 64) != hipSuccess) { error = "hipHostMalloc failed"; return AVN_ERR_OOM; }
        float* h_timer = (float*)pin_slp_timers.p; uint8_t* h_flags = (uint8_t*)(h_timer + n);
        HIPCHK(hipMemcpyAsync(h_timer, b_slp_timer.p, (size_t)n * 4, hipMemcpyDeviceToHost, stream));
        HIPCHK(hipMemcpyAsync(h_flags, b_slp_flags.p, n, hipMemcpyDeviceToHost, stream));
        if ((st = sleeping_split()) != AVN_OK) return st;   // (host work while the solver's kernels run)
        host_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        double m1 = slp_now();
        HIPCHK(hipStreamSynchronize(stream));
        double m2 = slp_now();
        t0 = std::chrono::steady_clock::now();
        if ((st = isl.sleeping_systems(h_timer, h_flags, n, slp_time_to_sleep)) != AVN_OK) return slp_fail(st);
        slp_n_awake = isl.last_flag_awake();   // (counted inside the manager's one pass over the flags)
        slp_last_slept = isl.last_slept(); slp_last_woken = isl.last_woken();
        host_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        double m3 = slp_now();
        st = sleeping_apply_result(true, host_ms);
        slp_host_ms = host_ms;
        slp_world_asleep = st == AVN_OK && std::none_of(h_body_has_sb.begin(), h_body_has_sb.end(), [](uint8_t x) { return x != 0; });
        slp_world_idle = slp_world_asleep && slp_step_started_asleep && !slp_step_changed && isl.last_slept() == 0 && isl.last_woken() == 0;
        if (slp_trace) {
            const double m4 = slp_now();
            std::fprintf(stderr, "[avn slp] step %u (%s): pair_add %.3f status %.3f flush %.3f apply1 %.3f | split %.3f wait %.3f systems %.3f apply2 %.3f ms; awake %u slept %u woken %u popped %u pushed %u\n",
                         pipe_step_no, slp_fast_step ? "fast" : "slow", slp_tr[0], slp_tr[1], slp_tr[2], slp_tr[3], m1 - m0, m2 - m1, m3 - m2, m4 - m3, slp_n_awake, slp_last_slept, slp_last_woken, slp_last_popped, slp_last_pushed);
        }
        return st;
    }
    avn_status sleeping_stats_get(avn_sleeping_stats* o) override {
        if (!o) return AVN_ERR_BAD_ARG;
        std::memset(o, 0, sizeof *o);
        if (!slp_on) return AVN_OK;
        isl.stats(&o->islands);
        o->n_awake_bodies = slp_n_awake; o->last_islands_slept = slp_last_slept; o->last_islands_woken = slp_last_woken;
        o->last_manifolds_popped = slp_last_popped; o->last_manifolds_pushed = slp_last_pushed; o->last_host_ms = slp_host_ms;
        return AVN_OK;
    }
    avn_status sleeping_state_get(const avn_sleeping_out* o) override {
        if (!o) return AVN_ERR_BAD_ARG;
        if (!slp_on) { error = "sleeping_state_get: sleeping is not enabled"; return AVN_ERR_STATE; }
        const uint32_t n = dw.n_bodies;
        avn_status st = isl.state(n, o->island, o->next_in_island, nullptr, nullptr);
        if (st != AVN_OK) return slp_fail(st);
        if (o->sleeping) for (uint32_t b = 0; b < n; ++b) o->sleeping[b] = (h_body_flags[b] & AVN_BODY_SLEEPING) ? 1 : 0;
        if (o->sleep_timer) { HIPCHK(hipMemcpyAsync(o->sleep_timer, b_slp_timer.p, (size_t)n * 4, hipMemcpyDeviceToHost, stream)); HIPCHK(hipStreamSynchronize(stream)); }
        return AVN_OK;
    }
    avn_status wake_bodies(const uint32_t* ids, size_t n) override {   // WakeBody (sleeping.rs:438-452)
        slp_world_asleep = slp_world_idle = false;
        if (!slp_on) { error = "wake_bodies: sleeping is not enabled"; return AVN_ERR_STATE; }
        if (n && !ids) return AVN_ERR_BAD_ARG;
        double ms = 0;
        for (size_t i = 0; i < n; ++i) {
            avn_status st = isl.wake_body(ids[i]);
            if (st != AVN_OK) return slp_fail(st);
            if ((st = sleeping_apply_result(false, ms)) != AVN_OK) return st;
        }
        return AVN_OK;
    }
