// world/broad_phase.hpp -- fragment of the body of `template <class T> struct World` (avn_world.hip includes it inside the class):
// BroadPhasePlugin's systems: update_aabb, collect_collision_pairs (launch / finish halves).

    // step_counters: inside a step the kernel also clears the counters of what follows it on the step's chain -- misc[32] the constraint
    // count, [33..34] dropped / unsorted, [35] pair total, [36..37] long-interval chunks / overflow -- instead of one memset launch each
    bool bp_counters_clean = false, constraint_count_clean = false;
    // (mode 1: the broad phase's counters only -- a frozen-manifold step prepares its constraints on the OTHER stream meanwhile; mode 2, the
    //  closed loop, where everything of the step is behind this kernel: the constraint count too)
    avn_status update_aabb(int step_counters = 0) {
        if (hs_any()) { avn_status sh = hs_aabbs(bs); if (sh != AVN_OK) return sh; }   // AnyCollider::aabb_with_context of the host's shapes (world/host_shapes.hpp)
        const bool ran = launch_update_aabb<T>(dw, bp, params, bs, step_counters ? b_misc.as<uint32_t>() + (step_counters == 2 ? 32 : 33) : nullptr, step_counters == 2 ? 6u : step_counters ? 5u : 0u);
        bp_counters_clean = step_counters && ran;
        constraint_count_clean = step_counters == 2 && ran;
        ++launches;
        HIPCHK(hipGetLastError());
        return AVN_OK;
    }
    // COLLECT_COLLISION_PAIRS is split in two so that avn_step can overlap the host round trip of the pair count with the
    // solver launches: collect_launch() enqueues sort + ranges + count pass + scan and an async read-back of the counters
    // into pinned memory (event-tracked); collect_finish() waits for THAT copy only, and runs the emit pass when new pairs exist.
    uint32_t* h_counters = nullptr;  // pinned: [dropped, unsorted, total, long chunks, long overflow]
    hipEvent_t ev_counters = nullptr;
    uint32_t collect_n = 0;
    bool collect_pending = false;
    avn_status collect_launch() {
        uint32_t n = bp.n_intervals;
        h_pairs.clear();
        last_timers.pair_count = 0;
        collect_n = n;
        collect_pending = false;
        if (n == 0) return AVN_OK;
        if (n > (1u << 26)) { error = "collect_collision_pairs: more than 2^26 intervals"; return AVN_ERR_CAPACITY; }
        if (!h_counters) {
            HIPCHK(hipHostMalloc((void**)&h_counters, 8 * sizeof(uint32_t), hipHostMallocDefault));
            HIPCHK(hipEventCreateWithFlags(&ev_counters, hipEventDisableTiming));
        }
        uint32_t* misc = b_misc.as<uint32_t>();
        uint32_t* d_dropped = misc + 33;   // [33] dropped, [34] unsorted
        uint32_t* d_total = misc + 35;
        sweep_scratch.n_long = misc + 36;  // [36] chunks, [37] overflow
        Key* keys_a = b_keys_a.as<Key>(); Key* keys_b = b_keys_b.as<Key>();
        uint32_t* vals_a = b_vals_a.as<uint32_t>(); uint32_t* vals_b = b_vals_b.as<uint32_t>();
        const bool clean = bp_counters_clean;   // (k_update_aabb of this step cleared them; a second count pass of one step clears them itself)
        bp_counters_clean = false;
        launch_interval_keys<T>(dw, bp, keys_a, vals_a, d_dropped, bs, clean);
        Key* keys_sorted; uint32_t* vals_sorted;
        launch_radix_sort<Key>(keys_a, vals_a, keys_b, vals_b, n, b_hist.as<uint32_t>(), b_block_sums.as<uint32_t>(), d_dropped + 1, &keys_sorted, &vals_sorted, bs);
        launch_gather_sorted<T>(dw, bp, vals_sorted, n, bs);
        launch_sweep_ranges<T>(bp, n, sweep_scratch, bs, clean);
        launch_sweep<T>(bp, n, false, sweep_scratch, b_counts.as<uint32_t>(), nullptr, nullptr, bs);
        launch_exclusive_scan(b_counts.as<uint32_t>(), b_offsets.as<uint32_t>(), n * sweep_count_slots(), b_block_sums.as<uint32_t>(), d_total, bs);
        launches += 3 + radix_sort_launches(n, (uint32_t)sizeof(Key)) + 3 + exclusive_scan_launches(n * sweep_count_slots());
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(h_counters, d_dropped, 5 * sizeof(uint32_t), hipMemcpyDeviceToHost, bs));
        HIPCHK(hipEventRecord(ev_counters, bs));
        collect_pending = true;
        return AVN_OK;
    }
    avn_status grow_long_chunks(uint32_t chunks_needed) {
        HIPCHK(hipStreamSynchronize(bs));
        const size_t lcap = (size_t)chunks_needed + chunks_needed / 4 + 65536;
        bool moved = false;
        uint8_t* dummy_b; uint32_t* dummy_u;
        GROW(b_long_items, lcap * sweep_long_item_bytes(), dummy_b);
        GROW(b_long_counts, lcap, dummy_u); GROW(b_long_off, lcap, dummy_u);
        sweep_scratch.long_items = b_long_items.p; sweep_scratch.long_counts = b_long_counts.as<uint32_t>();
        sweep_scratch.long_off = b_long_off.as<uint32_t>(); sweep_scratch.long_cap = (uint32_t)lcap;
        return AVN_OK;
    }
    avn_status collect_finish() {
        if (!collect_pending) return AVN_OK;
        collect_pending = false;
        uint32_t n = collect_n;
        HIPCHK(hipEventSynchronize(ev_counters));
        if (h_counters[4]) {   // more long-interval chunks than slots: grow to the requested count and run the count pass again
            avn_status st = grow_long_chunks(h_counters[3]);
            if (st != AVN_OK) return st;
            if ((st = collect_launch()) != AVN_OK) return st;
            collect_pending = false;
            HIPCHK(hipEventSynchronize(ev_counters));
            if (h_counters[4]) { error = "collect_collision_pairs: long-interval chunk capacity exceeded"; return AVN_ERR_CAPACITY; }
        }
        uint32_t dropped = h_counters[0], total = h_counters[2];
        if (total) {
            hipError_t err;
            b_pairs.ensure((size_t)total * sizeof(avn_pair), err);
            if (err != hipSuccess) { error = "pair buffer allocation failed"; return AVN_ERR_OOM; }
            launch_sweep<T>(bp, n, true, sweep_scratch, b_counts.as<uint32_t>(), b_offsets.as<uint32_t>(), b_pairs.as<avn_pair>(), bs);
            launches += 2;
            HIPCHK(hipGetLastError());
            h_pairs.resize(total);
            HIPCHK(hipMemcpyAsync(h_pairs.data(), b_pairs.p, (size_t)total * sizeof(avn_pair), hipMemcpyDeviceToHost, bs));
            // add_edge_and_key_with (reference contact_graph.rs:521-566): the new keys join the pair set
            HIPCHK(hipStreamSynchronize(bs));
            hk_filter_host(h_pairs);   // CollisionHooks::filter_pairs, when a callback is registered (world/hooks.hpp): rejected pairs never enter the pair set
            total = (uint32_t)h_pairs.size();
            std::vector<uint64_t> nk(total);
            for (uint32_t i = 0; i < total; ++i) { uint32_t a = h_pairs[i].collider1, b = h_pairs[i].collider2; nk[i] = a < b ? ((uint64_t)a << 32) | b : ((uint64_t)b << 32) | a; }
            if (contact_keys_live) h_live_keys.insert(nk.begin(), nk.end());
            b_pair_keys.ensure(((size_t)n_pair_keys + total) * 8, err, true, bs);
            if (err != hipSuccess) { error = "pair key list allocation failed"; return AVN_ERR_OOM; }
            HIPCHK(hipMemcpyAsync(b_pair_keys.as<uint64_t>() + n_pair_keys, nk.data(), (size_t)total * 8, hipMemcpyHostToDevice, bs));
            n_pair_keys += total;
            if (bp.pair_set_cap < 2 * (n_pair_keys + 16)) { avn_status st = rebuild_pair_set(n_pair_keys + n_pair_keys / 2); if (st != AVN_OK) return st; }
            else { launch_hs_insert(bp.pair_set, bp.pair_set_cap, b_pair_keys.as<uint64_t>() + (n_pair_keys - total), total, bs); HIPCHK(hipGetLastError()); HIPCHK(hipStreamSynchronize(bs)); }
        }
        bp.n_intervals = n - dropped;  // dropped intervals were sorted to the end
        last_timers.pair_count = total;
        return AVN_OK;
    }
    avn_status collect_collision_pairs() {
        avn_status st = collect_launch();
        if (st != AVN_OK) return st;
        return collect_finish();
    }

