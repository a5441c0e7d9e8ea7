// world/pipeline_host.hpp -- fragment of the body of `template <class T> struct World` (avn_world.hip includes it inside the class):
// closed loop with the ContactGraph / ConstraintGraph bookkeeping in host structures (round-1 path, A/B runs).

    // ---- standalone closed loop ------------------------------------------------------------------------------------------
    avn_status pipeline_enable(int on) override {
        slp_world_asleep = slp_world_idle = false;
        if (on && !have_colliders) { error = "pipeline_enable: upload bodies and colliders first"; return AVN_ERR_STATE; }
        if (on && pipe_on) return AVN_OK;
        if (pipe_on && pipe_dev) {   // leaving the device closed loop: its rows and keys go with it
            if (slp_on) { avn_status sw = sleeping_enable(nullptr); if (sw != AVN_OK) return sw; }   // the island manager goes too: no body may stay flagged Sleeping without it
            HIPCHK(hipStreamSynchronize(stream));
            if (ct.cap) HIPCHK(hipMemset(ct.meta, 0, (size_t)ct.cap * sizeof(uint4)));
            pipe_dev = false; pipe_on = false;
            despawn_needs_bodies = despawn_needs_colliders = despawn_needs_joints = despawn_broken = false; despawn_expected_bodies = 0;   // (the restart path after avn_despawn: any upload is welcome again)
            contact_keys_live = false; h_live_keys.clear();
            avn_status st = rebuild_pair_set(n_pair_keys);   // only the keys the host uploaded / collected outside the closed loop remain
            if (st != AVN_OK) return st;
            if (!on) return AVN_OK;
        }
        // on == 1: the bookkeeping runs on the device (k_graph.hip); on == 2 or AVN_PIPELINE_HOST=1: host structures (round-1 path, kept for A/B runs)
        const bool want_dev = on == 1 && !avn_env("AVN_PIPELINE_HOST");
        if (want_dev) {
            for (uint32_t id = 0; id < pipe_pairs.size(); ++id)
                if (pipe_pairs[id].used) { uint32_t cid = id; avn_status st = contact_pairs_remove(&cid, 1); if (st != AVN_OK) return st; }
            pipe_pairs.clear(); pipe_active.clear(); pipe_handles.clear();
            avn_status st = pipeline_device_reset();
            if (st != AVN_OK) return st;
            pipe_on = true; pipe_dev = true;
            return AVN_OK;
        }
        pipe_on = on != 0;
        // a fresh ContactGraph / ConstraintGraph: rows, ids, colour lists and the broad phase's pair set start empty
        for (uint32_t id = 0; id < pipe_pairs.size(); ++id)
            if (pipe_pairs[id].used) { uint32_t cid = id; avn_status st = contact_pairs_remove(&cid, 1); if (st != AVN_OK) return st; }
        pipe_pairs.clear(); pipe_active.clear(); pipe_handles.clear();
        for (auto& c : pipe_colors) { c.body_bits.clear(); c.handles.clear(); }
        pipe_free_ids = decltype(pipe_free_ids)();
        pipe_next_id = 0; pipe_handles_dirty = true; pipe_active_dirty = true;
        std::memset(&pipe_stats, 0, sizeof pipe_stats);
        std::memset(pipe_offsets, 0, sizeof pipe_offsets);
        return AVN_OK;
    }
    avn_status pipeline_stats_get(avn_pipeline_stats* o) override {
        if (!o) return AVN_ERR_BAD_ARG;
        if (pipe_dev) {
            avn_islands_stats is; std::memset(&is, 0, sizeof is);
            if (slp_on) isl.stats(&is);
            pipe_stats.active_pairs = pgm_live - is.sleeping_pairs;   // ContactGraph::active_pairs: the pairs of sleeping_pairs are not among them
            pipe_stats.manifolds = dw.n_manifolds; *o = pipe_stats; return AVN_OK;
        }
        pipe_stats.active_pairs = (uint32_t)pipe_active.size();
        pipe_stats.manifolds = (uint32_t)pipe_handles.size();
        *o = pipe_stats;
        return AVN_OK;
    }
    avn_status pipeline_handles_get(uint32_t* off, const uint32_t** ids, size_t* n) override {
        if (!off || !ids || !n) return AVN_ERR_BAD_ARG;
        if (pipe_dev) {   // the lists live on the device: fetched on request (tests, inspection)
            HIPCHK(hipStreamSynchronize(stream));
            // (offsets from the lists' own lengths: in the sharded closed loop the solver's colour offsets are this rank's share, the lists are the whole world's)
            pipe_offsets[0] = 0;
            for (int c = 0; c < AVN_GRAPH_COLOR_COUNT; ++c) pipe_offsets[c + 1] = pipe_offsets[c] + pgm_len[c];
            pipe_handles.resize(pipe_offsets[AVN_GRAPH_COLOR_COUNT]);
            // GraphColor::manifold_handles, i.e. the bookkeeping's lists in the reference's order (PG::lists) -- not the solver's arrays, whose order
            // inside colours 0..22 is by key body since round 5 (b_handles; the overflow colour is in list order there too)
            for (int c = 0; c < AVN_GRAPH_COLOR_COUNT; ++c) {
                const uint32_t len = pipe_offsets[c + 1] - pipe_offsets[c];
                if (len) HIPCHK(hipMemcpyAsync(pipe_handles.data() + pipe_offsets[c], pg.lists + (size_t)c * pg.list_stride, (size_t)len * 4, hipMemcpyDeviceToHost, stream));
            }
            HIPCHK(hipStreamSynchronize(stream));
        }
        std::memcpy(off, pipe_offsets, sizeof pipe_offsets);
        *ids = pipe_handles.data(); *n = pipe_handles.size();
        return AVN_OK;
    }
    // The ids of the last step's new pairs, in the order of avn_pairs_get (host pipeline: recorded as they are handed out; device: PG::new_ids)
    std::vector<uint32_t> h_new_pair_ids;
    // (device loop) how many entries of PG::new_ids the LAST COMPLETED closed-loop step wrote: 0 after avn_pipeline_enable, after avn_despawn and after a step that
    // found no pair -- last_timers.pair_count alone outlives those and named a stale or missing buffer
    uint32_t pg_new_ids_count = 0;
    avn_status pipeline_new_pair_ids_get(const uint32_t** ids, size_t* n) override {
        if (!ids || !n) return AVN_ERR_BAD_ARG;
        if (!pipe_on) { error = "pipeline_new_pair_ids_get: needs avn_pipeline_enable"; return AVN_ERR_STATE; }
        if (pipe_dev) {
            HIPCHK(hipStreamSynchronize(stream)); HIPCHK(hipStreamSynchronize(stream_bp));
            const uint32_t cnt = pg.new_ids ? pg_new_ids_count : 0u;
            h_new_pair_ids.resize(cnt);
            if (cnt) HIPCHK(hipMemcpy(h_new_pair_ids.data(), pg.new_ids, (size_t)cnt * 4, hipMemcpyDeviceToHost));
        }
        *ids = h_new_pair_ids.data(); *n = h_new_pair_ids.size();
        return AVN_OK;
    }
    // NarrowPhase::update's status changes of the last closed-loop step as avn_contact_change records (device loop: the op arrays the status scan
    // left, or their pinned copy when the island manager read them; decoded from the packed change word)
    bool pg_changes_cached = false;   // h_changes already holds the last step's changes (fetched before a later op batch reused the op arrays)
    avn_status pipeline_device_changes_fetch() {
        if (pg_changes_cached) return AVN_OK;
        const uint32_t n = pipe_stats.last_status_changes;
        h_changes.resize(n);
        pg_changes_cached = true;
        if (!n) return AVN_OK;
        std::vector<uint32_t> buf;
        const uint32_t *cid, *chg;
        if (slp_on && pin_slp_ops.p) { cid = (const uint32_t*)pin_slp_ops.p; chg = cid + n; HIPCHK(hipStreamSynchronize(stream)); }
        else {
            HIPCHK(hipStreamSynchronize(stream));
            buf.resize(2 * (size_t)n);
            HIPCHK(hipMemcpy(buf.data(), pg.op_cid, (size_t)n * 4, hipMemcpyDeviceToHost));
            HIPCHK(hipMemcpy(buf.data() + n, pg.op_chg, (size_t)n * 4, hipMemcpyDeviceToHost));
            cid = buf.data(); chg = cid + n;
        }
        for (uint32_t k = 0; k < n; ++k) {
            const uint32_t w = chg[k];
            h_changes[k] = avn_contact_change{cid[k], w & 0xFFFFu, (int32_t)((w >> 24) & 0xFFu) - 128, (w >> 16) & 0xFFu};
        }
        return AVN_OK;
    }
    static bool pbit_get(const std::vector<uint64_t>& s, uint32_t i) { return (i >> 6) < s.size() && ((s[i >> 6] >> (i & 63)) & 1ull); }
    static void pbit_set(std::vector<uint64_t>& s, uint32_t i) { if ((i >> 6) >= s.size()) s.resize((i >> 6) + 1, 0ull); s[i >> 6] |= 1ull << (i & 63); }
    static void pbit_unset(std::vector<uint64_t>& s, uint32_t i) { if ((i >> 6) < s.size()) s[i >> 6] &= ~(1ull << (i & 63)); }
    void pipe_push(uint32_t cid, uint32_t flags) {   // ConstraintGraph::push_manifold (constraint_graph.rs:163-236)
        PipePair& p = pipe_pairs[cid];
        if (p.n_handles) return;  // (one manifold per convex pair)
        const bool s1 = flags & AVN_CP_STATIC1, s2 = flags & AVN_CP_STATIC2;
        const uint32_t b1 = (uint32_t)p.b1, b2 = (uint32_t)p.b2;
        int color = AVN_COLOR_OVERFLOW_INDEX;
        if (!s1 && !s2) {
            for (int i = 0; i < AVN_DYNAMIC_COLOR_COUNT; ++i) {
                PipeColor& c = pipe_colors[i];
                if (pbit_get(c.body_bits, b1) || pbit_get(c.body_bits, b2)) continue;
                pbit_set(c.body_bits, b1); pbit_set(c.body_bits, b2);
                color = i;
                break;
            }
        } else if (!s1 || !s2) {
            const uint32_t body = !s1 ? b1 : b2;
            for (int i = AVN_COLOR_OVERFLOW_INDEX - 1; i >= 1; --i) {
                PipeColor& c = pipe_colors[i];
                if (pbit_get(c.body_bits, body)) continue;
                pbit_set(c.body_bits, body);
                color = i;
                break;
            }
        }
        p.color = (int8_t)color; p.color_pos = (uint32_t)pipe_colors[color].handles.size();
        pipe_colors[color].handles.push_back(cid);
        p.n_handles = 1; pipe_handles_dirty = true; ++pipe_stats.manifolds_pushed;
    }
    void pipe_pop(uint32_t cid) {                      // ConstraintGraph::pop_manifold (:245-296): swap-remove
        PipePair& p = pipe_pairs[cid];
        if (!p.n_handles) return;
        PipeColor& c = pipe_colors[p.color];
        if (p.color != AVN_COLOR_OVERFLOW_INDEX) { pbit_unset(c.body_bits, (uint32_t)p.b1); pbit_unset(c.body_bits, (uint32_t)p.b2); }
        uint32_t moved = c.handles.back();
        c.handles[p.color_pos] = moved; pipe_pairs[moved].color_pos = p.color_pos;
        c.handles.pop_back();
        p.n_handles = 0; p.color = -1; pipe_handles_dirty = true; ++pipe_stats.manifolds_popped;
    }
    avn_status pipeline_step() {
        avn_status st;
        launches = 0;
        HIPCHK(hipEventRecord(ev[0], stream));
        if ((st = update_aabb()) != AVN_OK) return st;
        if ((st = collect_collision_pairs()) != AVN_OK) return st;   // new pairs in h_pairs (emission order)
        HIPCHK(hipEventRecord(ev[1], stream));
        auto t0 = std::chrono::steady_clock::now();
        // ContactGraph::add_edge_and_key_with + IdPool::alloc_id for every new pair, in emission order
        h_new_pair_ids.clear();
        if (!h_pairs.empty()) {
            size_t n = h_pairs.size();
            std::vector<uint32_t> ids(n), c1(n), c2(n), fl(n);
            for (size_t i = 0; i < n; ++i) {
                uint32_t id;
                if (!pipe_free_ids.empty()) { id = pipe_free_ids.top(); pipe_free_ids.pop(); } else id = pipe_next_id++;
                if (id >= pipe_pairs.size()) pipe_pairs.resize(std::max<size_t>((size_t)id + 1, pipe_pairs.size() + pipe_pairs.size() / 2));
                const avn_pair& pr = h_pairs[i];
                PipePair& p = pipe_pairs[id];
                p.c1 = pr.collider1; p.c2 = pr.collider2; p.b1 = pr.body1; p.b2 = pr.body2; p.n_handles = 0; p.used = true;
                p.active_pos = (uint32_t)pipe_active.size();
                pipe_active.push_back(id);
                ids[i] = id; c1[i] = pr.collider1; c2[i] = pr.collider2; fl[i] = pr.flags;
            }
            avn_contact_pairs cp{(uint32_t)n, ids.data(), c1.data(), c2.data(), fl.data()};
            if ((st = contact_pairs_add(&cp)) != AVN_OK) return st;
            h_new_pair_ids = ids;
            pipe_stats.pairs_added += n;
            pipe_active_dirty = true;
        }
        if (pipe_active_dirty) {
            if ((st = active_pairs_set(pipe_active.data(), pipe_active.size())) != AVN_OK) return st;
            pipe_active_dirty = false;
        }
        double host_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        if ((st = narrow_phase()) != AVN_OK) return st;
        t0 = std::chrono::steady_clock::now();
        // the status-change loop of NarrowPhase::update (system_param.rs:141-389), ascending ContactId
        std::vector<uint32_t> removed;
        for (const avn_contact_change& c : h_changes) {
            const uint32_t cid = c.contact_id, flags = c.flags;
            const bool generates = flags & AVN_CP_GENERATE_CONSTRAINTS, touching = flags & AVN_CP_TOUCHING;
            PipePair& p = pipe_pairs[cid];
            if (flags & AVN_CP_DISJOINT_AABB) {
                if (generates) while (p.n_handles) pipe_pop(cid);
                removed.push_back(cid);
            } else if (flags & AVN_CP_STARTED_TOUCHING) {
                if (generates) for (uint32_t k = 0; k < c.manifold_count; ++k) pipe_push(cid, flags);
            } else if (flags & AVN_CP_STOPPED_TOUCHING) {
                if (generates) while (p.n_handles) pipe_pop(cid);
            } else if (touching && (flags & AVN_CP_STARTED_GENERATING_CONSTRAINTS)) {
                for (uint32_t k = 0; k < c.manifold_count; ++k) pipe_push(cid, flags);
            } else if (touching && generates && c.manifold_count_change > 0) {
                for (int32_t k = 0; k < c.manifold_count_change; ++k) pipe_push(cid, flags);
            } else if (touching && generates && c.manifold_count_change < 0) {
                for (int32_t k = 0; k < -c.manifold_count_change; ++k) pipe_pop(cid);
            }
        }
        pipe_stats.last_status_changes = (uint32_t)h_changes.size();
        if (!removed.empty()) {   // ContactGraph::remove_edge_by_id + IdPool::free_id
            if ((st = contact_pairs_remove(removed.data(), removed.size())) != AVN_OK) return st;
            for (uint32_t cid : removed) {
                PipePair& p = pipe_pairs[cid];
                uint32_t last = pipe_active.back();
                pipe_active[p.active_pos] = last; pipe_pairs[last].active_pos = p.active_pos; pipe_active.pop_back();
                p = PipePair();
                pipe_free_ids.push(cid);
            }
            pipe_stats.pairs_removed += removed.size();
            pipe_active_dirty = true;
        }
        if (pipe_handles_dirty) {
            size_t n = 0;
            for (int c = 0; c < AVN_GRAPH_COLOR_COUNT; ++c) { pipe_offsets[c] = (uint32_t)n; n += pipe_colors[c].handles.size(); }
            pipe_offsets[AVN_GRAPH_COLOR_COUNT] = (uint32_t)n;
            pipe_handles.resize(n);
            for (int c = 0; c < AVN_GRAPH_COLOR_COUNT; ++c)
                if (!pipe_colors[c].handles.empty()) std::memcpy(pipe_handles.data() + pipe_offsets[c], pipe_colors[c].handles.data(), pipe_colors[c].handles.size() * 4);
            if ((st = manifold_handles_upload(pipe_offsets, pipe_handles.data())) != AVN_OK) return st;
            pipe_handles_dirty = false;
        }
        pipe_stats.last_overflow_manifolds = pipe_offsets[AVN_COLOR_OVERFLOW_INDEX + 1] - pipe_offsets[AVN_COLOR_OVERFLOW_INDEX];
        pipe_stats.last_host_ms = host_ms + std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        stamp(DG_NP1); dg_np = true;
        if ((st = solver()) != AVN_OK) return st;
        HIPCHK(hipEventRecord(ev[4], stream));
        ev_valid = true;
        last_timers.kernel_launches = launches;
        return AVN_OK;
    }
