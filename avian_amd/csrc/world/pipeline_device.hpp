// world/pipeline_device.hpp -- fragment of the body of `template <class T> struct World` (avn_world.hip includes it inside the class):
// closed loop with the bookkeeping on the device (k_graph.hip).

    // ---- closed loop, bookkeeping on the device -------------------------------------------------------------------------------
    template <class U> avn_status pg_buf(DevBuf& b, size_t count, U** field, bool keep = false) {
        hipError_t err;
        b.ensure(std::max<size_t>(count, 1) * sizeof(U), err, keep, stream);
        if (err != hipSuccess) { error = "hipMalloc failed"; return AVN_ERR_OOM; }
        *field = b.as<U>();
        return AVN_OK;
    }
    // per-row arrays follow CT::cap (contents kept: they are persistent state); per-op scratch is sized for one op per row
    avn_status pg_ensure_rows(uint32_t rows) {
        if (rows <= pg_rows) return AVN_OK;
        HIPCHK(hipStreamSynchronize(stream));
        const uint32_t old = pg_rows;
        avn_status st;
#define PGB(buf, cnt, field, keep) do { if ((st = pg_buf(buf, cnt, &(field), keep)) != AVN_OK) return st; } while (0)
        PGB(b_pg_bodies, rows, pg.bodies, true); PGB(b_pg_color, rows, pg.color, true); PGB(b_pg_lpos, rows, pg.lpos, true);
        PGB(b_pg_free_a, rows, pg.free_ids, true); PGB(b_pg_free_b, rows, pg.free_alt, true);
        PGB(b_pg_seq, rows, pg.seq, true);
        {   // colour lists: [24][stride] re-laid out for the new stride
            uint32_t* nl = nullptr;
            if (hipMalloc((void**)&nl, (size_t)AVN_GRAPH_COLOR_COUNT * rows * 4) != hipSuccess) { error = "hipMalloc failed"; return AVN_ERR_OOM; }
            for (int c = 0; c < AVN_GRAPH_COLOR_COUNT && old; ++c)
                if (pgm_len[c]) HIPCHK(hipMemcpy(nl + (size_t)c * rows, pg.lists + (size_t)c * old, (size_t)pgm_len[c] * 4, hipMemcpyDeviceToDevice));
            if (b_pg_lists.p) (void)hipFree(b_pg_lists.p);
            b_pg_lists.p = nl; b_pg_lists.cap = (size_t)AVN_GRAPH_COLOR_COUNT * rows * 4;
            pg.lists = nl; pg.list_stride = rows;
        }
        PGB(b_pg_chg, rows, pg.chg, true); PGB(b_pg_has, rows, pg.has, true);   // (kept: the overlapped narrow phase has written the old rows' changes when a step's new pairs grow the table)
        PGB(b_pg_off, rows + 1, pg.off, false);
        PGB(b_pg_op_cid, rows, pg.op_cid, false); PGB(b_pg_op_chg, rows, pg.op_chg, false); PGB(b_pg_op_info, rows, pg.op_info, false); PGB(b_pg_op_bodies, rows, pg.op_bodies, false);
        PGB(b_pg_ekey_a, 2 * (size_t)rows, pg.ekey_a, false); PGB(b_pg_eval_a, 2 * (size_t)rows, pg.eval_a, false);
        PGB(b_pg_ekey_b, 2 * (size_t)rows, pg.ekey_b, false); PGB(b_pg_eval_b, 2 * (size_t)rows, pg.eval_b, false);
        PGB(b_pg_epos, 2 * (size_t)rows, pg.epos, false); PGB(b_pg_popbefore, 2 * (size_t)rows, pg.popbefore, false);
        PGB(b_pg_prevpush, 2 * (size_t)rows, pg.prevpush, false); PGB(b_pg_est, 2 * (size_t)rows, pg.est, false);
        PGB(b_pg_tile_agg, 5 * (size_t)pg_scan_tiles(2 * rows) + 8, pg.tile_agg, false);
        PGB(b_pg_ckey_a, rows, pg.ckey_a, false); PGB(b_pg_cval_a, rows, pg.cval_a, false); PGB(b_pg_ckey_b, rows, pg.ckey_b, false); PGB(b_pg_cval_b, rows, pg.cval_b, false);
        PGB(b_pg_rem_flag, rows, pg.rem_flag, false); PGB(b_pg_rem_off, rows + 1, pg.rem_off, false); PGB(b_pg_rem_ids, rows, pg.rem_ids, false);
        uint32_t* dummy;
        PGB(b_pg_hist, (size_t)256 * radix_blocks(2 * rows) + 256, dummy, false);
        PGB(b_pg_sums, std::max<size_t>(scan_block_sums_needed(256 * radix_blocks(2 * rows)), scan_block_sums_needed(2 * rows)) + 16, dummy, false);
#undef PGB
        HIPCHK(hipMemset(b_pg_sums.p, 0, b_pg_sums.cap));   // the one-launch scan's state: zero once, self-cleaning afterwards (avn_scan.h)
        pg.rows = rows; pg_rows = rows;
        graph_valid = false;
        return AVN_OK;
    }
    // per-body colour masks (one 24-bit word per body): follow cap_bodies, contents kept, the new tail zeroed
    uint32_t pg_bcol_words = 0;
    avn_status pg_bcol_grow() {
        const uint32_t want = cap_bodies + 1;
        if (want <= pg_bcol_words) return AVN_OK;
        HIPCHK(hipStreamSynchronize(stream));
        avn_status st = pg_buf(b_pg_bcol, want, &pg.bcol, true);
        if (st != AVN_OK) return st;
        HIPCHK(hipMemsetAsync(pg.bcol + pg_bcol_words, 0, (size_t)(want - pg_bcol_words) * 4, stream));
        pg_bcol_words = want;
        graph_valid = false;
        return AVN_OK;
    }
    // collider entity -> slot (dense: Entity::index() values are small integers); rebuilt by every colliders_upload that changes the slots
    avn_status pg_upload_ent2slot() {
        uint32_t max_ent = 0;
        for (uint32_t e : slot_entity) max_ent = std::max(max_ent, e);
        if (max_ent > (1u << 27)) { error = "pipeline_enable: collider entity indices above 2^27 need the host bookkeeping (AVN_PIPELINE_HOST=1)"; return AVN_ERR_CAPACITY; }
        std::vector<uint32_t> e2s((size_t)max_ent + 1, 0u);
        for (uint32_t i = 0; i < slot_entity.size(); ++i) e2s[slot_entity[i]] = i;
        HIPCHK(hipStreamSynchronize(stream)); HIPCHK(hipStreamSynchronize(stream_bp));
        uint32_t* d;
        avn_status st = pg_buf(b_pg_ent2slot, e2s.size(), &d);
        if (st != AVN_OK) return st;
        HIPCHK(hipMemcpy(d, e2s.data(), e2s.size() * 4, hipMemcpyHostToDevice));
        pg.ent2slot = d;
        return AVN_OK;
    }
    avn_status pipeline_device_reset() {
        island_backoff = 0; isl_labels_step_valid = false;
        HIPCHK(hipStreamSynchronize(stream));
        HIPCHK(hipStreamSynchronize(stream_bp));
        avn_status st;
        hipError_t err;
        b_pg_ctr.ensure(PGC_WORDS * 4, err);
        if (err != hipSuccess) { error = "hipMalloc failed"; return AVN_ERR_OOM; }
        pg.ctr = b_pg_ctr.as<uint32_t>();
        HIPCHK(hipMemset(pg.ctr, 0, PGC_WORDS * 4));
        pg_bcol_words = 0;
        pg_batch_open = false;
        despawn_needs_bodies = despawn_needs_colliders = despawn_needs_joints = despawn_broken = false; despawn_expected_bodies = 0;   // a fresh loop owes no upload
        if (b_pg_sort_tab.p) HIPCHK(hipMemset(b_pg_sort_tab.p, 0xFF, b_pg_sort_tab.cap));
        if ((st = pg_bcol_grow()) != AVN_OK) return st;
        HIPCHK(hipMemset(pg.bcol, 0, (size_t)pg_bcol_words * 4));
        if ((st = pg_upload_ent2slot()) != AVN_OK) return st;
        if (ct.cap) HIPCHK(hipMemset(ct.meta, 0, (size_t)ct.cap * sizeof(uint4)));
        if ((st = ensure_contact_rows(std::max<uint32_t>(ct.cap, 1024u))) != AVN_OK) return st;
        pg_rows = 0;   // (re)allocate everything for the table's capacity
        std::memset(pgm_len, 0, sizeof pgm_len);
        if ((st = pg_ensure_rows(ct.cap)) != AVN_OK) return st;
        HIPCHK(hipMemset(pg.color, 0xFF, (size_t)pg_rows * 4));
        contact_keys_live = false; h_live_keys.clear();   // (the pair set keeps the keys the host announced: existing pairs stay existing)
        pgm_head = pgm_n_free = pgm_next_id = pgm_live = pgm_tomb = 0;
        pg_new_ids_count = 0; last_timers.pair_count = 0;
        slp_on = false; isl.reset();   // (avn_sleeping_enable follows avn_pipeline_enable)
        std::memset(&pipe_stats, 0, sizeof pipe_stats);
        std::memset(pipe_offsets, 0, sizeof pipe_offsets);
        uint32_t zero[AVN_GRAPH_COLOR_COUNT + 1] = {0};
        dw.n_manifolds = 0;
        set_color_offsets(zero);
        HIPCHK(hipMemcpy(dw.color_offsets, zero, sizeof zero, hipMemcpyHostToDevice));
        use_handles = true; any_restitution = materials_restitution || hk_restitution;
        incidence_dirty = true; graph_valid = false;
        if (pin_ctr.ensure(4096) != hipSuccess) { error = "hipHostMalloc failed"; return AVN_ERR_OOM; }
        return AVN_OK;
    }
    static uint32_t bits_for(uint32_t max_value) { uint32_t b = 1; while (b < 32 && (max_value >> b)) ++b; return b; }
    // ContactGraph::pair_set with room for `expect` more keys: rebuilt from the live rows when it would pass half full (tombstones count)
    avn_status pg_pair_set_reserve(uint32_t n_rows_now, uint32_t incoming) {
        const uint64_t need_keys = (uint64_t)n_pair_keys + pgm_live + pgm_tomb + incoming + 16;
        if (bp.pair_set_cap && 2 * need_keys <= bp.pair_set_cap) return AVN_OK;
        uint32_t need = 1024;
        while ((uint64_t)need < 4 * ((uint64_t)n_pair_keys + pgm_live + incoming + 16)) need <<= 1;
        HIPCHK(hipStreamSynchronize(bs));
        hipError_t err;
        b_pair_set.ensure((size_t)need * 8, err);
        if (err != hipSuccess) { error = "hipMalloc failed"; return AVN_ERR_OOM; }
        bp.pair_set = b_pair_set.as<uint64_t>();
        bp.pair_set_cap = need;
        HIPCHK(hipMemsetAsync(bp.pair_set, 0xFF, (size_t)need * 8, bs));
        launch_hs_insert(bp.pair_set, need, b_pair_keys.as<uint64_t>(), n_pair_keys, bs);   // keys announced by the host (avn_existing_pairs_upload, pairs collected outside the loop)
        launch_pg_rebuild_pair_set<T>(ct, bp, n_rows_now, bs);
        HIPCHK(hipGetLastError());
        pgm_tomb = 0;
        graph_valid = false;
        return AVN_OK;
    }
    // PGC_ERROR: bit 0 = k_pg_color's dataflow wait ran out (colouring), bit 1 = k_overflow_flow's ticket wait ran out (overflow colour's
    // solve), bit 2 = k_pack_contacts was asked to write a row that is not live.  The word is read back at the end of every step into pinned
    // memory (no extra synchronisation: the copy rides the stream); whoever synchronises next -- avn_synchronize, the next avn_step -- reports
    // it under the name of the kernel that raised it and clears it.
    hipEvent_t ev_np_fork = nullptr, ev_np_old = nullptr;
    // The closed loop reads three counter blocks per step, and the device idles while the host finds out that they arrived: a blocking
    // hipStreamSynchronize costs 20-35 us of that per read (interrupt + wake-up), polling the event 2-3.  AVN_NO_SPIN_SYNC=1: blocking waits.
    hipEvent_t ev_spin = nullptr;
    bool spin_enabled = !(avn_env("AVN_NO_SPIN_SYNC") && avn_env("AVN_NO_SPIN_SYNC")[0] && avn_env("AVN_NO_SPIN_SYNC")[0] != '0');
    hipError_t spin_event(hipEvent_t e) {
        if (!spin_enabled) return hipEventSynchronize(e);
        for (;;) {
            const hipError_t r = hipEventQuery(e);
            if (r != hipErrorNotReady) return r;
            __builtin_ia32_pause();
        }
    }
    hipError_t spin_sync(hipStream_t s) {
        if (!spin_enabled) return hipStreamSynchronize(s);
        if (!ev_spin) { const hipError_t r = hipEventCreateWithFlags(&ev_spin, hipEventDisableTiming); if (r != hipSuccess) return r; }
        const hipError_t r = hipEventRecord(ev_spin, s);
        return r != hipSuccess ? r : spin_event(ev_spin);
    }
    uint32_t pipe_step_no = 0;   // closed-loop steps taken by this world (measurement aids only)
    // measurement aid (`make measure` build, AVN_PIPE_HOST_TRACE=<step>): where the HOST is, in microseconds since the step's call, at the marked points of ONE closed-loop step
    int host_trace_step = avn_env("AVN_PIPE_HOST_TRACE") ? atoi(avn_env("AVN_PIPE_HOST_TRACE")) : -1;
    bool host_trace_on = false;
    std::chrono::steady_clock::time_point host_trace_t0;
    std::vector<std::pair<const char*, double>> host_trace_marks;
    void ht(const char* what) { if (host_trace_on) host_trace_marks.emplace_back(what, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - host_trace_t0).count()); }
    void ht_begin() { host_trace_on = host_trace_step >= 0 && (uint32_t)host_trace_step == pipe_step_no; host_trace_marks.clear(); host_trace_t0 = std::chrono::steady_clock::now(); }
    void ht_end() {
        if (!host_trace_on) return;
        ht("step call returns");
        std::fprintf(stderr, "[avn host trace] closed-loop step %d:", host_trace_step);
        for (auto& m : host_trace_marks) std::fprintf(stderr, "\n  %9.1f us  %s", m.second, m.first);
        std::fprintf(stderr, "\n");
        host_trace_on = false;
    }
#ifdef AVN_MEASURE
    int np_debug_step = avn_env("AVN_NP_DEBUG_STEP") ? atoi(avn_env("AVN_NP_DEBUG_STEP")) : -1;
#else
    static constexpr int np_debug_step = -1;
#endif
    bool np_overlap_enabled = !(avn_env("AVN_NO_NP_OVERLAP") && avn_env("AVN_NO_NP_OVERLAP")[0] && avn_env("AVN_NO_NP_OVERLAP")[0] != '0');
    uint32_t* h_pg_error = nullptr;   // pinned
    bool pg_error_pending = false;
    avn_status pg_error_fetch() {     // enqueue the read-back behind everything the step launched
        if (!pipe_dev || !pg.ctr) return AVN_OK;
        if (!h_pg_error) { HIPCHK(hipHostMalloc((void**)&h_pg_error, 64, hipHostMallocDefault)); *h_pg_error = 0; }
        HIPCHK(hipMemcpyAsync(h_pg_error, pg.ctr + PGC_ERROR, 4, hipMemcpyDeviceToHost, stream));
        pg_error_pending = true;
        return AVN_OK;
    }
    avn_status pg_error_report(uint32_t word) {   // the stream is idle here
        if (!word) return AVN_OK;
        error = "device closed loop:";
        if (word & 1u) error += " k_pg_color: the colouring's dataflow wait timed out (colour lists of this step are not the reference's);";
        if (word & 2u) error += " k_overflow_flow: a ticket wait of the overflow colour's solve timed out (the step's velocities are not the reference's);";
        if (word & 4u) error += " k_pack_contacts: avn_contacts_upload named a contact id whose row is not live (skipped);";
        if (word & 8u) error += " k_overflow_flow: a manifold's overflow rank and its constraint flags disagree about which body has a SolverBody;";
        if (word & 16u) error += " sharded closed loop: a manifold joins bodies of two ranks -- their islands have met (avn_bounds_exchange / re-partition before stepping on);";
        if (word & ~31u) error += " unknown bits in the error word;";
        error += " the error word has been cleared";
        HIPCHK(hipMemsetAsync(pg.ctr + PGC_ERROR, 0, 4, stream));
        HIPCHK(hipMemsetAsync(pg.ctr + PGC_BUCKET, 0, 32 * 4, stream));   // (an aborted batch never reached k_pg_build_handles, which cleans these)
        HIPCHK(hipMemsetAsync(pg.ctr + PGC_TILE, 0, 4, stream)); HIPCHK(hipMemsetAsync(pg.ctr + PGC_N_REM, 0, 4, stream)); HIPCHK(hipMemsetAsync(pg.ctr + PGC_N_SLEEP_OPS, 0, 4, stream));
        HIPCHK(hipMemsetAsync(pg.ctr + PGC_SORT_DUP, 0, AVN_GRAPH_COLOR_COUNT * 4, stream));
        if (b_pg_sort_tab.p) HIPCHK(hipMemsetAsync(b_pg_sort_tab.p, 0xFF, b_pg_sort_tab.cap, stream));
        pg_batch_open = false;
        HIPCHK(hipStreamSynchronize(stream));
        if (h_pg_error) *h_pg_error = 0;
        return AVN_ERR_STATE;
    }
    avn_status pg_error_check() {     // after a synchronisation of `stream`
        if (!pg_error_pending) return AVN_OK;
        pg_error_pending = false;
        return pg_error_report(*h_pg_error);
    }
    // One batch of ConstraintGraph ops through the device pipeline: ops from the rows' status changes (the status loop; list_cids == NULL) or
    // from a list (SleepIslands / WakeIslands, world/sleeping.hpp) -> body-sorted entries -> greedy colours as dataflow -> masks -> colour
    // buckets -> exact swap_remove replay -> (pair removals) -> the colours' lengths back to the host -> the concatenated handle list.
    // The scoped counters of an op batch (replay buckets, the colouring's tile ticket, the removal count, the sort table) are left clean by the batch's own
    // last kernels.  A batch that ended early -- an allocation failure, a device error word, a failed wait -- never reached them: the next one starts
    // with stale bucket counts and tile tickets, i.e. wrong colour lists with no error raised.  `pg_batch_open` is set when a batch's first kernel
    // (k_pg_scan_classify / k_pg_ops_from_list) is about to be launched and cleared behind k_pg_build_handles; still set at the next begin = clean up here.
    bool pg_batch_open = false;
    bool early_prepare_enabled = !avn_env("AVN_NO_EARLY_PREPARE");   // handle lists + constraint generation enqueued in front of the counters' read-back (A/B in `make measure` builds)
    bool handle_sort = true;          // the solver's body-sorted order inside colours 0..22 (AVN_NO_HANDLE_SORT=1 in `make measure` builds: A/B)
    DevBuf b_pg_sort_tab, b_pg_sort_cnt;
    avn_status pg_batch_begin() {
        if (pg_batch_open) {
            HIPCHK(hipMemsetAsync(pg.ctr + PGC_BUCKET, 0, 32 * 4, stream));
            HIPCHK(hipMemsetAsync(pg.ctr + PGC_TILE, 0, 4, stream)); HIPCHK(hipMemsetAsync(pg.ctr + PGC_N_REM, 0, 4, stream)); HIPCHK(hipMemsetAsync(pg.ctr + PGC_N_SLEEP_OPS, 0, 4, stream));
            HIPCHK(hipMemsetAsync(pg.ctr + PGC_SORT_DUP, 0, AVN_GRAPH_COLOR_COUNT * 4, stream));
            if (b_pg_sort_tab.p) HIPCHK(hipMemsetAsync(b_pg_sort_tab.p, 0xFF, b_pg_sort_tab.cap, stream));
        }
        pg_batch_open = true;
        return AVN_OK;
    }
    avn_status pg_sort_ensure() {   // [23][stride] words, all EMPTY between batches (k_pg_sort_emit cleans what k_pg_build_handles wrote)
        const size_t stride = pg_sort_stride(dw.n_bodies);
        hipError_t err;
        const bool fresh = b_pg_sort_tab.ensure((size_t)AVN_COLOR_OVERFLOW_INDEX * stride * 4, err);
        if (err != hipSuccess) { error = "hipMalloc failed"; return AVN_ERR_OOM; }
        if (fresh) HIPCHK(hipMemsetAsync(b_pg_sort_tab.p, 0xFF, b_pg_sort_tab.cap, stream));
        b_pg_sort_cnt.ensure((size_t)AVN_COLOR_OVERFLOW_INDEX * (stride / 2048u) * 4 + 64, err);
        if (err != hipSuccess) { error = "hipMalloc failed"; return AVN_ERR_OOM; }
        return AVN_OK;
    }
    avn_status pg_apply_ops(uint32_t n_ops, uint32_t n_rem, uint32_t n_rows, const uint32_t* list_cids, const uint32_t* list_kinds, double& host_ms) {
        avn_status st;
        // a list batch (SleepIslands / WakeIslands, avn_despawn) reuses the op arrays the status loop left its changes in: what avn_contact_changes_get
        // reports is fetched first (with sleeping on the pinned copy the island manager read is the source and stays valid)
        if (list_cids && !slp_on && (st = pipeline_device_changes_fetch()) != AVN_OK) return st;
        if (list_cids && (st = pg_batch_begin()) != AVN_OK) return st;   // (the status loop's batch was opened in front of k_pg_scan_classify)
        {
            // (ctr[PGC_BUCKET ..] and ctr[PGC_TILE] are zero here: k_pg_build_handles, the last kernel of every batch, leaves them so; the status
            //  loop's ops were classified by k_pg_scan_classify, the scan that counted them)
            if (list_cids) launch_pg_ops_from_list<T>(pg, ct, list_cids, list_kinds, n_ops, dw.n_bodies, stream);
            if (slp_on && !list_cids) {   // the island manager reads the loop's changes: (contact id, packed change) per op, in ascending id
                if (pin_slp_ops.ensure((size_t)n_ops * 8 + 64) != hipSuccess) { error = "hipHostMalloc failed"; return AVN_ERR_OOM; }
                HIPCHK(hipMemcpyAsync(pin_slp_ops.p, pg.op_cid, (size_t)n_ops * 4, hipMemcpyDeviceToHost, stream));
                HIPCHK(hipMemcpyAsync((uint32_t*)pin_slp_ops.p + n_ops, pg.op_chg, (size_t)n_ops * 4, hipMemcpyDeviceToHost, stream));
            }
            uint32_t *ek, *evv;
            launch_radix_sort_bits(pg.ekey_a, pg.eval_a, pg.ekey_b, pg.eval_b, 2 * n_ops, bits_for(dw.n_bodies), b_pg_hist.as<uint32_t>(), b_pg_sums.as<uint32_t>(), &ek, &evv, stream);
            launch_pg_entry_scan(pg, ek, evv, 2 * n_ops, dw.n_bodies, stream);
            launch_pg_color(pg, n_ops, stream);
            launch_pg_apply_masks(pg, ek, evv, 2 * n_ops, dw.n_bodies, stream);
            launch_pg_replay(pg, n_ops, stream);
            launches += 6 + ((bits_for(dw.n_bodies) + 7) / 8) * radix_pass_launches(2 * n_ops);
            if (n_rem) {   // ContactGraph::remove_edge_by_id + IdPool::free_id
                launch_exclusive_scan(pg.rem_flag, pg.rem_off, n_ops, b_pg_sums.as<uint32_t>(), nullptr, stream);
                launch_pg_remove<T>(pg, ct, bp, n_ops, stream);
                launch_pg_merge_free(pg, pgm_head, pgm_n_free, n_rem, stream);
                // the merged list IS the free list from here on: the two buffers change roles (no copy back)
                std::swap(b_pg_free_a.p, b_pg_free_b.p); std::swap(b_pg_free_a.cap, b_pg_free_b.cap);
                std::swap(pg.free_ids, pg.free_alt);
                pgm_head = 0; pgm_n_free += n_rem; pgm_live -= n_rem; pgm_tomb += n_rem;
                pipe_stats.pairs_removed += n_rem;
                launches += 3;
            }
            HIPCHK(hipGetLastError());
            if (dsh_on && (st = dsh_local_lists(n_ops)) != AVN_OK) return st;   // sharded closed loop: this rank's share of every colour list (ctr[PGC_LLEN ..])
            uint32_t* h = (uint32_t*)pin_ctr.p + 64;
            HIPCHK(hipMemcpyAsync(h, pg.ctr, 64 * 4, hipMemcpyDeviceToHost, stream));   // the counters block up to the colours' lengths, in one copy
            if (dsh_on) HIPCHK(hipMemcpyAsync(h + 64, pg.ctr + PGC_LLEN, AVN_GRAPH_COLOR_COUNT * 4, hipMemcpyDeviceToHost, stream));
            if (!ev_spin) HIPCHK(hipEventCreateWithFlags(&ev_spin, hipEventDisableTiming));
            HIPCHK(hipEventRecord(ev_spin, stream));   // (the host waits for THIS point, not for what it enqueues below)
            // Round 5: what follows the replay on the device -- handle lists, the body-sorted order, constraint generation: 90 us of a settled cfg2
            // step -- is enqueued BEFORE the host waits for the counters, bounded by M_ub = manifolds before the batch + ops (an op pushes at most
            // one) with the exact counts read on the device, so the device works while the host synchronises, decides about the substep graph and
            // launches it (the step-110 timeline had the device idle for 18 + 11 + 42 us there).  Not for list batches, and with sleeping on only in a step
            // that cannot wake anything (slp_fast_step): further batches may follow before the solver otherwise.
            bool early = !list_cids && (!slp_on || slp_fast_step) && early_prepare_enabled && use_handles;
            uint32_t M_ub = 0;
            if (early) {
                M_ub = dw.n_manifolds + n_ops;
                if ((st = ensure_manifold_capacity(M_ub)) != AVN_OK) return st;
                hipError_t e2;
                if (b_handles.ensure(std::max<size_t>(M_ub, 1) * 4, e2)) graph_valid = false;
                if (e2 != hipSuccess) { error = "hipMalloc failed"; return AVN_ERR_OOM; }
                const bool sorted_ub = handle_sort && M_ub >= 4096u && (uint64_t)AVN_COLOR_OVERFLOW_INDEX * dw.n_bodies <= 32ull * M_ub;
                if (sorted_ub && (st = pg_sort_ensure()) != AVN_OK) return st;
                launch_pg_build_handles(dsh_on ? dsh_pg() : pg, b_handles.as<uint32_t>(), dw.color_offsets, M_ub, ct.meta, sorted_ub ? b_pg_sort_tab.as<uint32_t>() : nullptr, b_pg_sort_cnt.as<uint32_t>(), dw.n_bodies, stream, dsh_on ? (uint32_t)PGC_LLEN : (uint32_t)PGC_LEN);
                launches += sorted_ub ? 3 : 1;
                DW<T> dwp = dw; dwp.n_manifolds = M_ub;
                RowsView<T> rv{b_handles.as<uint32_t>(), ct.meta, bp.col_info, ct.rows};
                launch_prepare_contact_constraints<T>(dwp, params, stream, constraint_count_clean, &rv); ++launches;
                constraint_count_clean = false;
                constraints_prepared_early = true;
                HIPCHK(hipGetLastError());
            }
            ht("op pipeline + early handles / constraints enqueued");
            HIPCHK(spin_event(ev_spin));
            ht("read 2 (counters) arrived");
            auto t0 = std::chrono::steady_clock::now();
            if (h[PGC_ERROR]) return pg_error_report(h[PGC_ERROR]);
            if (avn_env("AVN_PG_REPLAY_STATS")) {
                uint32_t d[96];
                HIPCHK(hipMemcpy(d, pg.ctr + PGC_DBG, sizeof d, hipMemcpyDeviceToHost));
                std::fprintf(stderr, "[avn replay] colour: ops/iterations/serial/reloads:");
                for (int c = 0; c < AVN_GRAPH_COLOR_COUNT; ++c) if (d[72 + c]) std::fprintf(stderr, " %d:%u/%u/%u/%u", c, d[72 + c], d[c], d[24 + c], d[48 + c]);
                std::fprintf(stderr, "\n");
            }
            if (const char* dir = avn_env("AVN_PG_DUMP")) {   // debugging aid (tools/debug_pg.py): this step's ops as the device saw them
                std::vector<uint32_t> a(n_ops), b(n_ops), o(n_ops), cnt(32);
                const uint32_t* order = pg.ekey_a;   // (the colour-partitioned op stream of the replay: contact id | push << 31)
                std::vector<int2> bd(n_ops);
                HIPCHK(hipMemcpy(a.data(), pg.op_cid, (size_t)n_ops * 4, hipMemcpyDeviceToHost));
                HIPCHK(hipMemcpy(b.data(), pg.op_info, (size_t)n_ops * 4, hipMemcpyDeviceToHost));
                HIPCHK(hipMemcpy(bd.data(), pg.op_bodies, (size_t)n_ops * 8, hipMemcpyDeviceToHost));
                HIPCHK(hipMemcpy(o.data(), order, (size_t)n_ops * 4, hipMemcpyDeviceToHost));
                HIPCHK(hipMemcpy(cnt.data(), pg.ctr + PGC_BUCKET, 32 * 4, hipMemcpyDeviceToHost));
                char path[512];
                std::snprintf(path, sizeof path, "%s/step_%04llu.bin", dir, (unsigned long long)pg_dump_step);
                if (FILE* f = std::fopen(path, "wb")) {
                    uint32_t hdr[4] = {n_ops, n_rem, 0, 0};
                    std::fwrite(hdr, 4, 4, f); std::fwrite(a.data(), 4, n_ops, f); std::fwrite(b.data(), 4, n_ops, f); std::fwrite(bd.data(), 8, n_ops, f);
                    std::fwrite(o.data(), 4, n_ops, f); std::fwrite(cnt.data(), 4, 32, f);
                    std::fclose(f);
                }
            }
            pipe_stats.manifolds_pushed = h[PGC_N_PUSH]; pipe_stats.manifolds_popped = h[PGC_N_POP];
            uint32_t offs[AVN_GRAPH_COLOR_COUNT + 1];
            uint32_t M = 0;
            // (sharded: the lists' lengths are the whole world's, the solver's offsets this rank's share)
            dsh_global_manifolds = 0;
            for (int c = 0; c < AVN_GRAPH_COLOR_COUNT; ++c) { pgm_len[c] = h[PGC_LEN + c]; dsh_global_manifolds += pgm_len[c]; const uint32_t len = dsh_on ? h[64 + c] : pgm_len[c]; offs[c] = M; M += len; }
            offs[AVN_GRAPH_COLOR_COUNT] = M;
            dsh_own_manifolds = M;
            if ((st = ensure_manifold_capacity(M)) != AVN_OK) return st;
            if ((dw.n_manifolds == 0) != (M == 0)) graph_valid = false;   // (no captured kernel reads DW::n_manifolds; only "any manifolds at all" shapes the substep)
            dw.n_manifolds = M;
            set_color_offsets(offs);
            hipError_t err;
            if (b_handles.ensure(std::max<size_t>(M, 1) * 4, err)) graph_valid = false;
            if (err != hipSuccess) { error = "hipMalloc failed"; return AVN_ERR_OOM; }
            // the solver's order inside colours 0..22: by key body (k_graph.hip, round 5) when the manifolds are dense enough in the bodies for the
            // 23 x n_bodies table walk to be cheaper than what the locality saves (a sparse scene's colour launches are tiny either way)
            if (!early) {
                const bool sorted = handle_sort && M >= 4096u && (uint64_t)AVN_COLOR_OVERFLOW_INDEX * dw.n_bodies <= 32ull * M;
                if (sorted && (st = pg_sort_ensure()) != AVN_OK) return st;
                launch_pg_build_handles(dsh_on ? dsh_pg() : pg, b_handles.as<uint32_t>(), dw.color_offsets, M, ct.meta, sorted ? b_pg_sort_tab.as<uint32_t>() : nullptr, b_pg_sort_cnt.as<uint32_t>(), dw.n_bodies, stream, dsh_on ? (uint32_t)PGC_LLEN : (uint32_t)PGC_LEN);
                launches += sorted ? 3 : 1;
                HIPCHK(hipGetLastError());
            } else if (M > M_ub) { error = "device closed loop: more manifolds after an op batch than manifolds before + ops"; return AVN_ERR_STATE; }
            pg_batch_open = false;
            incidence_dirty = true;
            host_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        }
        return AVN_OK;
    }
    avn_status pipeline_step_device() {
        avn_status st;
        if (despawn_needs_bodies || despawn_needs_colliders) { error = "avn_step: avn_despawn must be followed by avn_bodies_upload and avn_colliders_upload of what remains"; return AVN_ERR_STATE; }
        if (despawn_needs_joints) { error = "avn_step: avn_despawn removed joints: upload the remaining joints (avn_joints_upload) first"; return AVN_ERR_STATE; }
        if (despawn_broken) { error = "avn_step: an avn_despawn failed half-way and left the contact bookkeeping inconsistent; restart the closed loop (avn_pipeline_enable(0), uploads, avn_pipeline_enable(1))"; return AVN_ERR_STATE; }
        // whatever ends this step early must not leave the next one believing that prepare_solver_bodies already ran or that the slot table is being cleared
        struct StepGuard { World* w; bool ok = false; ~StepGuard() { if (!ok) { w->bodies_prepared_early = false; w->constraints_prepared_early = false; w->slot_clear_pending = false; w->bs = w->stream; } } } step_guard{this};
        launches = 0;
        if (slp_on && slp_world_idle) {   // nothing is awake and the last step proved the state stationary: the step is the identity (world/sleeping.hpp)
            for (hipEvent_t e : ev) HIPCHK(hipEventRecord(e, stream));
            ev_valid = true;
            pipe_stats.last_status_changes = 0; pipe_stats.last_host_ms = 0; last_timers.pair_count = 0; last_timers.kernel_launches = 0; pg_new_ids_count = 0;
            h_pairs.clear();
            slp_n_awake = slp_last_slept = slp_last_woken = slp_last_popped = slp_last_pushed = 0; slp_host_ms = 0;
            ++pipe_step_no; ++pg_dump_step;
            return AVN_OK;
        }
        slp_step_started_asleep = slp_on && slp_world_asleep; slp_step_changed = false;
        double host_ms = 0;
        auto t0 = std::chrono::steady_clock::now();
        auto lap = [&]() { auto t1 = std::chrono::steady_clock::now(); host_ms += std::chrono::duration<double, std::milli>(t1 - t0).count(); };
        ht_begin();
        HIPCHK(hipEventRecord(ev[0], stream));
        if ((st = update_aabb(2)) != AVN_OK) return st;
        // The narrow phase of the rows that exist at the START of the step needs the new AABBs and nothing else of the broad phase: it runs
        // on the world's stream while sort + sweep + emit run on the broad-phase stream (both are VALU-bound kernels that leave half the
        // chip idle on their own).  The rows a step ADDS are created only after that launch has finished (they may reuse freed ids: a row
        // must not come alive under a running launch) and get their own small launch.  Same per-row work, same outputs: chg / has per row,
        // so the scan still numbers the changes in ascending ContactId.  AVN_NO_NP_OVERLAP=1: the serial order (A/B runs).
        const uint32_t n_rows_old = pgm_next_id, head_old = pgm_head;
        StepParams<T> np_params = params;   // (AVN_NP_DEBUG_STEP: the measurement cut-offs of the narrow phase in ONE step of a run, so that the launch sees a real state)
        if (np_debug_step >= 0 && (uint32_t)np_debug_step != pipe_step_no) np_params.np_debug = 0u;
        ++pipe_step_no;
        const bool np_overlap = np_overlap_enabled && n_rows_old != 0 && bp.n_intervals != 0;
        const NpHostList hs_hl = hs_begin(stream);   // (host shapes: the list of this step's pairs whose manifold the host computes -- empty world/host_shapes.hpp when there are none)
        if (np_overlap) {
            if (!ev_np_fork) { HIPCHK(hipEventCreateWithFlags(&ev_np_fork, hipEventDisableTiming | EV_FLAGS)); HIPCHK(hipEventCreateWithFlags(&ev_np_old, hipEventDisableTiming | EV_FLAGS)); }
            HIPCHK(hipEventRecord(ev_np_fork, stream));
            HIPCHK(hipStreamWaitEvent(stream_bp, ev_np_fork, 0));
            // (issued BEFORE the broad phase's ~20 launches: the host needs ~150 us to enqueue those, and the narrow phase would start that late)
            launch_narrow_phase_dense<T>(dw, bp, ct, np_params, n_rows_old, pg.chg, pg.has, pg.ctr + PGC_N_REM, stream, false, hs_hl);
            hs_launches.push_back(HsLaunch{1, n_rows_old, 0, 0, 0, nullptr});
            ++launches;
            HIPCHK(hipEventRecord(ev_np_old, stream));
            bs = stream_bp;
        }
        // prepare_solver_bodies and pre_process_velocity_increments only read the rigid-body components: enqueued here, on the world's stream, they
        // run next to the broad phase instead of on the serial chain in front of the solver.
        // (Sleeping on: WakeIslands changes which bodies own a SolverBody between the status loop and the solver -- sleeping_apply_result then drops the flag and the
        //  solver's front runs both kernels again; they only read the components and rewrite the whole SolverBody.)
        prepare_solver_bodies(); pre_process_velocity_increments(); bodies_prepared_early = true;
        ht("narrow phase (old rows) + solver-body kernels enqueued");
        st = collect_launch();
        if (st != AVN_OK) { bs = stream; bodies_prepared_early = false; return st; }
        lap();
        ht("broad phase enqueued");
        // ---- new pairs (emission order) -> ids, rows, pair keys: all on the device; the host reads the pair COUNT ----
        uint32_t total = 0, used_ids = 0;
        auto fail = [&](avn_status e) { bs = stream; return e; };
        if (collect_pending) {
            collect_pending = false;
            HIPCHK(spin_event(ev_counters));
            ht("read 0 (pair count) arrived");
            t0 = std::chrono::steady_clock::now();
            if (h_counters[4]) {   // more long-interval chunks than slots: grow to the requested count and run the count pass again
                if ((st = grow_long_chunks(h_counters[3])) != AVN_OK) return fail(st);
                if ((st = collect_launch()) != AVN_OK) return fail(st);
                collect_pending = false;
                HIPCHK(hipEventSynchronize(ev_counters));
                if (h_counters[4]) { error = "collect_collision_pairs: long-interval chunk capacity exceeded"; return fail(AVN_ERR_CAPACITY); }
            }
            const uint32_t dropped = h_counters[0];
            total = h_counters[2];
            if (total) {
                hipError_t err;
                b_pairs.ensure((size_t)total * sizeof(avn_pair), err);
                if (err != hipSuccess) { error = "pair buffer allocation failed"; return fail(AVN_ERR_OOM); }
                launch_sweep<T>(bp, collect_n, true, sweep_scratch, b_counts.as<uint32_t>(), b_offsets.as<uint32_t>(), b_pairs.as<avn_pair>(), bs);
                launches += 2;
                if ((st = hk_filter_device(total, bs)) != AVN_OK) return fail(st);   // CollisionHooks::filter_pairs: the pairs the hook rejects never get an id (world/hooks.hpp)
            }
            if (total) {
                const uint32_t fresh = total > pgm_n_free ? total - pgm_n_free : 0u;
                if ((st = ensure_contact_rows(pgm_next_id + fresh)) != AVN_OK) return fail(st);   // (growing synchronises the world's stream first: the launch over the old rows is done)
                if (np_overlap) HIPCHK(hipStreamWaitEvent(stream_bp, ev_np_old, 0));               // nothing below may touch a row while that launch runs
                if ((st = pg_pair_set_reserve(pgm_next_id, total)) != AVN_OK) return fail(st);
                {   // the ids of the new pairs in emission order: the island manager's edge lists (sleeping on), avn_pipeline_new_pair_ids_get (a host's events)
                    hipError_t e2;
                    b_pg_new_ids.ensure((size_t)total * 4, e2);
                    if (e2 != hipSuccess) { error = "hipMalloc failed"; return fail(AVN_ERR_OOM); }
                    pg.new_ids = b_pg_new_ids.as<uint32_t>();
                }
                launch_pg_add_pairs<T>(pg, ct, b_pairs.as<avn_pair>(), total, bp.pair_set, bp.pair_set_cap, bs);   // ids, rows, PairKeys (add_edge_and_key_with), IdPool counters
                if (slp_on) {
                    if (pin_slp_pairs.ensure((size_t)total * (sizeof(avn_pair) + 4) + 64) != hipSuccess) { error = "hipHostMalloc failed"; return fail(AVN_ERR_OOM); }
                    HIPCHK(hipMemcpyAsync(pin_slp_pairs.p, b_pairs.p, (size_t)total * sizeof(avn_pair), hipMemcpyDeviceToHost, bs));
                    HIPCHK(hipMemcpyAsync((char*)pin_slp_pairs.p + (size_t)total * sizeof(avn_pair), pg.new_ids, (size_t)total * 4, hipMemcpyDeviceToHost, bs));
                }
                launches += 1;
                HIPCHK(hipGetLastError());
                const uint32_t used = std::min(total, pgm_n_free);
                used_ids = used;
                pgm_head += used; pgm_n_free -= used; pgm_next_id += total - used; pgm_live += total;
                pipe_stats.pairs_added += total;
            }
            bp.n_intervals = collect_n - dropped;
            last_timers.pair_count = total;
            pg_new_ids_count = total;
        }
        if (np_overlap) {
            HIPCHK(hipEventRecord(ev_bp_done, stream_bp));
            HIPCHK(hipStreamWaitEvent(stream, ev_bp_done, 0));
            bs = stream;
        }
        HIPCHK(hipEventRecord(ev[1], stream));
        // ---- narrow phase over every live row; changes numbered in ascending ContactId ----
        const uint32_t n_rows = pgm_next_id;
        uint32_t n_ops = 0, n_rem = 0, n_sleep_ops = 0;
        if (n_rows) {
            if (!np_overlap) { launch_narrow_phase_dense<T>(dw, bp, ct, np_params, n_rows, pg.chg, pg.has, pg.ctr + PGC_N_REM, stream, false, hs_hl); hs_launches.push_back(HsLaunch{1, n_rows, 0, 0, 0, nullptr}); ++launches; }
            else if (total) {   // the rows this step added: the lowest free ids first (k_pg_add_pairs), then the fresh ones
                launch_narrow_phase_rows<T>(dw, bp, ct, np_params, pg.free_ids + head_old, used_ids, n_rows_old, total - used_ids, pg.chg, pg.has, pg.ctr + PGC_N_REM, stream, hs_hl);
                hs_launches.push_back(HsLaunch{2, used_ids, n_rows_old, total - used_ids, 0, pg.free_ids + head_old});
                ++launches;
            }
            if (hs_hl.queries && (st = hs_manifolds(true, np_params, nullptr, pg.ctr + PGC_N_REM, pg.chg, pg.has, stream)) != AVN_OK) return fail(st);
            if (hs_hl.hook.count && (st = hk_modify(true, np_params, nullptr, pg.ctr + PGC_N_REM, pg.chg, pg.has, stream)) != AVN_OK) return fail(st);   // CollisionHooks::modify_contacts
            if ((st = pg_batch_begin()) != AVN_OK) return fail(st);
            launch_pg_scan_classify(pg, n_rows, dw.n_bodies, b_pg_sums.as<uint32_t>(), stream, slp_on ? dw.bmeta : nullptr);   // ops numbered in ascending ContactId AND classified
            ++launches;
            HIPCHK(hipGetLastError());
            uint32_t* h = (uint32_t*)pin_ctr.p;
            HIPCHK(hipMemcpyAsync(h, pg.ctr + PGC_N_OPS, (PGC_N_SLEEP_OPS - PGC_N_OPS + 1) * 4, hipMemcpyDeviceToHost, stream));   // N_OPS, N_REM, ERROR ... N_SLEEP_OPS
            lap();
            ht("new pairs + scan_classify enqueued");
            HIPCHK(spin_sync(stream));
            ht("read 1 (op count) arrived");
            t0 = std::chrono::steady_clock::now();
            n_ops = h[0]; n_rem = h[1];
            n_sleep_ops = h[PGC_N_SLEEP_OPS - PGC_N_OPS];
            pg_error_pending = false;
            if (h[2]) return pg_error_report(h[2]);   // raised by the previous step's solver passes (normally already reported by avn_synchronize)
        }
        pipe_stats.last_status_changes = n_ops;
        pg_changes_cached = false;   // (avn_contact_changes_get: this step's changes are in the op arrays now)
        if (!n_ops) pg_batch_open = false;   // (no change: k_pg_scan_classify touched none of the batch's counters)
        slp_fast_step = slp_on && slp_fast_enabled && n_sleep_ops == 0;   // no status change names a Sleeping body: nothing can wake before the solver (world/sleeping.hpp)
        if (n_ops || total) slp_step_changed = true;
        ++pg_dump_step;
        if (n_ops) {   // ---- the status-change loop: decisions, colours, handle lists ----
            lap();
            // the warm start's slot table will be rebuilt for the new lists: its 22 x cap_bodies words are set to EMPTY on the (idle) broad-phase
            // stream while the op pipeline runs, instead of by a memset in front of the solver
            if (dw.inc_slot && dw.inc_stride == cap_bodies && b_inc_slot.cap >= (size_t)AVN_COLOR_OVERFLOW_INDEX * cap_bodies * sizeof(uint32_t)) {
                if (!ev_slot_clear) HIPCHK(hipEventCreateWithFlags(&ev_slot_clear, hipEventDisableTiming | EV_FLAGS));
                HIPCHK(hipMemsetAsync(dw.inc_slot, 0xFF, (size_t)AVN_COLOR_OVERFLOW_INDEX * dw.inc_stride * sizeof(uint32_t), stream_bp));
                HIPCHK(hipEventRecord(ev_slot_clear, stream_bp));
                slot_clear_pending = true;
            }
            if ((st = pg_apply_ops(n_ops, n_rem, n_rows, nullptr, nullptr, host_ms)) != AVN_OK) return st;
            t0 = std::chrono::steady_clock::now();
        }
        // (the island-BLOCK builder takes its grouping from the manager: where the blocks are a candidate -- small scenes, where the digest is cheap -- the manager
        //  must have seen this step's links before the solver's front)
        const bool slp_defer = slp_on && slp_fast_step && !(island_candidate(dw.n_manifolds) && dw.n_joints == 0);
        if (slp_on && !slp_fast_step && (st = sleeping_after_status_loop(total, n_ops, host_ms)) != AVN_OK) return st;   // islands: new pairs, the loop's link / unlink, WakeIslands
        if (slp_on && slp_fast_step && !slp_defer && (st = sleeping_digest_deferred(total, n_ops, host_ms)) != AVN_OK) return st;
        // (a small candidate is walked over the manager's own edge lists in microseconds; the device lists cost ~20 launches and a hand-over)
        if (slp_on && isl.candidate_bodies() >= slp_adj_min_bodies && (st = sleeping_adjacency_launch()) != AVN_OK) return st;   // split_island's neighbour lists, next to the solver
        pipe_stats.last_overflow_manifolds = color_offsets[AVN_COLOR_OVERFLOW_INDEX + 1] - color_offsets[AVN_COLOR_OVERFLOW_INDEX];
        lap();
        pipe_stats.last_host_ms = host_ms;
        stamp(DG_NP1); dg_np = true;
        ht("bookkeeping done, solver() called");
        if ((st = solver()) != AVN_OK) return st;
        ht("solver() returned");
        if (slp_defer) {   // the manager's digest of this step, under the solver's kernels
            double slp_ms = 0;
            if ((st = sleeping_digest_deferred(total, n_ops, slp_ms)) != AVN_OK) return st;
            pipe_stats.last_host_ms += slp_ms;
        }
        if (slp_on && (st = sleeping_after_solver()) != AVN_OK) return st;   // split_island + the Sleeping set (synchronises: the host reads the timers)
        if (dsh_on && (st = dsh_exchange_in_step()) != AVN_OK) return st;   // sharded closed loop with a communicator: the ranks' own bodies to everybody (one all-gather)
        HIPCHK(hipEventRecord(ev[4], stream));
        ev_valid = true;
        last_timers.kernel_launches = launches;
        step_guard.ok = true;
        st = pg_error_fetch();
        ht_end();
        return st;
    }
    // the overflow colour's CSR + ranks, and the slot table of the other colours, from the gathered manifold arrays (all on the device)
    avn_status rebuild_incidence_device() {
        const uint32_t N = dw.n_bodies, M = dw.n_manifolds;
        incidence_dirty = false;
        island_mode = false; islands_dirty = false;
        if (M == 0) return AVN_OK;
        hipError_t err;
        bool moved = b_inc_slot.ensure((size_t)AVN_COLOR_OVERFLOW_INDEX * cap_bodies * sizeof(uint32_t), err);
        if (err != hipSuccess) { error = "hipMalloc failed"; return AVN_ERR_OOM; }
        if (moved || dw.inc_stride != cap_bodies) { graph_valid = false; slot_clear_pending = false; }   // (a new table: the early clear hit the old one)
        dw.inc_slot = b_inc_slot.as<uint32_t>(); dw.inc_stride = cap_bodies;
        slots_dirty = true;
        const uint32_t o0 = color_offsets[AVN_COLOR_OVERFLOW_INDEX], n23 = color_offsets[AVN_COLOR_OVERFLOW_INDEX + 1] - o0;
        moved = b_inc_off.ensure(((size_t)N + 2) * 4, err);
        if (err != hipSuccess) { error = "hipMalloc failed"; return AVN_ERR_OOM; }
        if (n23 > pg_ovf_cap) {
            HIPCHK(hipStreamSynchronize(stream));
            const size_t c = std::max<size_t>(2 * (size_t)n23 + 1024, (size_t)pg_ovf_cap * 3);
            for (DevBuf* b : {&b_inc_ent, &b_ovf_keys_a, &b_ovf_vals_a, &b_ovf_keys_b, &b_ovf_vals_b, &b_ovf_rank}) {
                b->ensure(c * 4, err);
                if (err != hipSuccess) { error = "hipMalloc failed"; return AVN_ERR_OOM; }
            }
            pg_ovf_cap = (uint32_t)(c / 2);
            moved = true;
        }
        if (b_inc_ent.cap == 0) { b_inc_ent.ensure(1024, err); b_ovf_rank.ensure(1024, err); moved = true; }
        if (b_ovf_ticket.ensure(((size_t)cap_bodies + 1) * 4, err)) moved = true;
        if (err != hipSuccess) { error = "hipMalloc failed"; return AVN_ERR_OOM; }
        if (moved || !dw.inc_off) graph_valid = false;
        dw.inc_off = b_inc_off.as<uint32_t>(); dw.inc_ent = b_inc_ent.as<uint32_t>();
        islands_dirty = island_candidate(M) && dw.n_joints == 0;
        return AVN_OK;
    }
    // after k_prepare_contact_constraints (the CSR reads DW::m_bodies of the overflow range, which that kernel lays out)
    void overflow_csr_device() {
        const uint32_t o0 = color_offsets[AVN_COLOR_OVERFLOW_INDEX], n23 = color_offsets[AVN_COLOR_OVERFLOW_INDEX + 1] - o0;
        uint32_t *k = b_ovf_keys_a.as<uint32_t>(), *v = b_ovf_vals_a.as<uint32_t>();
        if (n23) {
            launch_ovf_entries<T>(dw, o0, n23, k, v, stream);
            launch_radix_sort_bits(k, v, b_ovf_keys_b.as<uint32_t>(), b_ovf_vals_b.as<uint32_t>(), 2 * n23, bits_for(dw.n_bodies), b_pg_hist.as<uint32_t>(), b_pg_sums.as<uint32_t>(), &k, &v, stream);
            launches += 1 + ((bits_for(dw.n_bodies) + 7) / 8) * radix_pass_launches(2 * n23);
        }
        launch_ovf_csr<T>(dw, o0, n23, k, v, b_inc_off.as<uint32_t>(), b_inc_ent.as<uint32_t>(), b_ovf_rank.as<uint32_t>(), stream);
        launches += 2;
    }
