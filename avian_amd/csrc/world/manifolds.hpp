// world/manifolds.hpp -- fragment of the body of `template <class T> struct World` (avn_world.hip includes it inside the class):
// host-uploaded manifolds, the warm start's incidence, island blocks, impulse / constraint read-back.

    // ---- manifolds ---------------------------------------------------------------------------------------
    avn_status manifolds_upload(const avn_manifolds* m) override {
        if (!have_bodies) { error = "manifolds_upload before bodies_upload"; return AVN_ERR_STATE; }
        if (!m || !m->color_offsets || (m->count && (!m->body1 || !m->body2 || !m->normal || !m->friction || !m->restitution || !m->point_count ||
                                                    !m->anchor1 || !m->anchor2 || !m->penetration || !m->normal_speed))) {
            error = "manifolds_upload: null array"; return AVN_ERR_BAD_ARG;
        }
        uint32_t M = m->count;
        if (m->color_offsets[0] != 0 || m->color_offsets[AVN_GRAPH_COLOR_COUNT] != M) { error = "manifolds_upload: bad color_offsets"; return AVN_ERR_BAD_ARG; }
        for (int c = 0; c < AVN_GRAPH_COLOR_COUNT; ++c)
            if (m->color_offsets[c] > m->color_offsets[c + 1]) { error = "manifolds_upload: color_offsets not monotone"; return AVN_ERR_BAD_ARG; }
        if (use_handles) graph_valid = false;
        use_handles = false;  // the manifolds come from the host again
        avn_status st0 = ensure_manifold_capacity(M);
        if (st0 != AVN_OK) return st0;
        if (dw.n_manifolds != M) graph_valid = false;
        dw.n_manifolds = M;
        set_color_offsets(m->color_offsets);
        size_t total = al(4 * (size_t)M) * 2 + al(sizeof(T) * 3 * M) * 2 + al(sizeof(T) * M) * 2 + al(M) * 2 + al(sizeof(T) * 12 * M) * 2 + al(sizeof(T) * 4 * M) * 3 + al(sizeof(T) * 8 * M);
        avn_status st = stage_reserve(total + 64 * 32);
        if (st != AVN_OK) return st;
        HIPCHK(hipMemcpyAsync(dw.color_offsets, color_offsets, sizeof color_offsets, hipMemcpyHostToDevice, stream));
        ManifoldStage<T> s;
        std::memset(&s, 0, sizeof s);
        SIN(body1, m->body1, M, int32_t); SIN(body2, m->body2, M, int32_t); SIN(normal, m->normal, 3 * (size_t)M, T);
        SIN(friction, m->friction, M, T); SIN(restitution, m->restitution, M, T); SIN(tangent_velocity, m->tangent_velocity, 3 * (size_t)M, T);
        SIN(point_count, m->point_count, M, uint8_t); SIN(manifold_flags, m->manifold_flags, M, uint8_t);
        SIN(anchor1, m->anchor1, 12 * (size_t)M, T); SIN(anchor2, m->anchor2, 12 * (size_t)M, T);
        SIN(penetration, m->penetration, 4 * (size_t)M, T); SIN(normal_speed, m->normal_speed, 4 * (size_t)M, T);
        SIN(warm_n, m->warm_start_normal_impulse, 4 * (size_t)M, T); SIN(warm_t, m->warm_start_tangent_impulse, 8 * (size_t)M, T);
        launch_pack_manifolds<T>(dw, s, stream);
        HIPCHK(hipGetLastError());
        // (while the copies run) one pass over the host arrays: range checks, "any restitution", and the host copy of the ContactPair bodies (incidence CSR source)
        {
            const uint32_t nb = dw.n_bodies;
            const T* rest = (const T*)m->restitution;
            h_m_body1.resize(M); h_m_body2.resize(M);
            bool bad_body = false, bad_pc = false, any_r = false;
            for (uint32_t i = 0; i < M; ++i) {
                const int32_t a = m->body1[i], b = m->body2[i];
                bad_body |= (uint32_t)a >= nb || (uint32_t)b >= nb;   // (negative indices wrap above nb)
                bad_pc |= m->point_count[i] > AVN_MAX_MANIFOLD_POINTS;
                any_r |= !(rest[i] == T(0));
                h_m_body1[i] = a; h_m_body2[i] = b;
            }
            any_restitution = any_r;
            if (bad_body || bad_pc) {   // nothing of a rejected upload may be solved: the world is left without manifolds
                HIPCHK(hipStreamSynchronize(stream));
                uint32_t zero[AVN_GRAPH_COLOR_COUNT + 1] = {0};
                dw.n_manifolds = 0; h_m_body1.clear(); h_m_body2.clear();
                set_color_offsets(zero);
                HIPCHK(hipMemcpyAsync(dw.color_offsets, zero, sizeof zero, hipMemcpyHostToDevice, stream));
                HIPCHK(hipStreamSynchronize(stream));
                graph_valid = false; incidence_dirty = true;
                error = bad_body ? "manifolds_upload: body index out of range" : "manifolds_upload: point_count > 4";
                return AVN_ERR_BAD_ARG;
            }
        }
        incidence_dirty = true;
        HIPCHK(hipStreamSynchronize(stream));
        return AVN_OK;
    }
    // Incidence CSR of the body-centric warm start: per body that has a SolverBody, its (manifold, side) entries in SOLVE
    // order = overflow colour first, then colours 0..22 (reference plugin.rs:461-470), list order inside a colour.
    // Incidence of the body-centric warm start.  Colours 0..22: the slot table is (re)built ON THE DEVICE from the manifold arrays
    // (launch_build_incidence_slots, run after the manifolds are in place).  Host part: only the overflow colour -- its
    // per-body entry lists (CSR, list order) and its level schedule.
    bool slots_dirty = true;
    bool ovf_csr_dirty = false;
    avn_status rebuild_incidence() {
        if (!incidence_dirty) return AVN_OK;
        groups_dirty = true;   // bodies, manifolds or their mode changed: the island streams' grouping follows
        if (pipe_dev) { ovf_csr_dirty = true; return rebuild_incidence_device(); }
        uint32_t N = dw.n_bodies, M = dw.n_manifolds;
        if (M == 0) { incidence_dirty = false; island_mode = false; islands_dirty = false; return AVN_OK; }
        if (h_body_has_sb.size() != N || h_m_body1.size() != M) { error = "incidence: bodies / manifolds out of sync"; return AVN_ERR_STATE; }
        HIPCHK(hipStreamSynchronize(stream));
        hipError_t err;
        {   // slot table storage: 23 colour planes of cap_bodies entries
            bool moved = b_inc_slot.ensure((size_t)AVN_COLOR_OVERFLOW_INDEX * cap_bodies * sizeof(uint32_t), err);
            if (err != hipSuccess) { error = "hipMalloc failed"; return AVN_ERR_OOM; }
            if (moved || dw.inc_stride != cap_bodies) graph_valid = false;
            dw.inc_slot = b_inc_slot.as<uint32_t>();
            dw.inc_stride = cap_bodies;
            slots_dirty = true;
        }
        const uint32_t o0 = color_offsets[AVN_COLOR_OVERFLOW_INDEX], o1 = color_offsets[AVN_COLOR_OVERFLOW_INDEX + 1];
        std::vector<uint32_t>& off = inc_off_h; std::vector<uint32_t>& cursor = inc_cursor_h; std::vector<uint32_t>& ent = inc_ent_h;
        off.assign((size_t)N + 1, 0u);
        for (uint32_t m = o0; m < o1; ++m) {
            if (h_body_has_sb[h_m_body1[m]]) ++off[(size_t)h_m_body1[m] + 1];
            if (h_body_has_sb[h_m_body2[m]]) ++off[(size_t)h_m_body2[m] + 1];
        }
        for (uint32_t i = 0; i < N; ++i) off[i + 1] += off[i];
        cursor.assign(off.begin(), off.end() - 1);
        ent.resize(off[N]);
        for (uint32_t m = o0; m < o1; ++m) {
            uint32_t a = (uint32_t)h_m_body1[m], b = (uint32_t)h_m_body2[m];
            if (h_body_has_sb[a]) ent[cursor[a]++] = m;
            if (h_body_has_sb[b]) ent[cursor[b]++] = m | 0x80000000u;
        }
        bool moved = b_inc_off.ensure(((size_t)N + 1) * 4, err);
        if (err != hipSuccess) { error = "hipMalloc failed"; return AVN_ERR_OOM; }
        moved |= b_inc_ent.ensure(std::max<size_t>(ent.size(), 1) * sizeof(uint32_t), err);
        if (err != hipSuccess) { error = "hipMalloc failed"; return AVN_ERR_OOM; }
        if (moved || !dw.inc_off) graph_valid = false;
        dw.inc_off = b_inc_off.as<uint32_t>();
        dw.inc_ent = b_inc_ent.as<uint32_t>();
        if (o1 > o0 || overflow_csr_nonzero) {   // an all-zero offset array stays valid while the overflow colour is empty
            HIPCHK(hipMemcpyAsync(b_inc_off.p, off.data(), off.size() * 4, hipMemcpyHostToDevice, stream));
            if (!ent.empty()) HIPCHK(hipMemcpyAsync(b_inc_ent.p, ent.data(), ent.size() * sizeof(uint32_t), hipMemcpyHostToDevice, stream));
            HIPCHK(hipStreamSynchronize(stream));
            overflow_csr_nonzero = o1 > o0;
        } else if (moved || overflow_csr_bodies != N) {
            HIPCHK(hipMemsetAsync(b_inc_off.p, 0, ((size_t)N + 1) * 4, stream));
        }
        overflow_csr_bodies = N;
        {   // level schedule of the overflow colour (k_overflow_pass): keys = the bodies a manifold can modify
            std::vector<uint32_t> ms(o1 - o0);
            std::vector<int32_t> k1(o1 - o0), k2(o1 - o0);
            for (uint32_t m = o0; m < o1; ++m) {
                ms[m - o0] = m;
                k1[m - o0] = h_body_has_sb[h_m_body1[m]] ? h_m_body1[m] : -1;
                k2[m - o0] = h_body_has_sb[h_m_body2[m]] ? h_m_body2[m] : -1;
            }
            uint32_t before = sched_overflow.n_components;
            const bool levels_before = sched_overflow.gorder.size() > overflow_level_threshold;   // (the form contact_pass picks: one device-wide launch per LEVEL, sizes captured)
            const std::vector<uint32_t> glevels_before = levels_before ? sched_overflow.glevel_offsets : std::vector<uint32_t>();
            sched_overflow.build(ms, k1, k2, N, ms.size() > overflow_level_threshold);
            void* p0 = sched_overflow.d_order.p; void* p1 = sched_overflow.d_level_offsets.p; void* p2 = sched_overflow.d_comp_level_begin.p; void* p3 = sched_overflow.d_gorder.p;
            avn_status st;
            if ((st = upload_u32(sched_overflow.d_comp_level_begin, sched_overflow.comp_level_begin)) != AVN_OK) return st;
            if ((st = upload_u32(sched_overflow.d_level_offsets, sched_overflow.level_offsets)) != AVN_OK) return st;
            if ((st = upload_u32(sched_overflow.d_order, sched_overflow.order)) != AVN_OK) return st;
            if ((st = upload_u32(sched_overflow.d_gorder, sched_overflow.gorder)) != AVN_OK) return st;
            HIPCHK(hipStreamSynchronize(stream));
            // What the substep graph captured of the overflow colour: per-component form (k_overflow_pass) -- the grid (n_components) and the three pointers, checked
            // below; the arrays are read at replay.  Per-level form -- every level's (first, count) as launch parameters: re-capture when they differ.  (Until round 6
            // ANY non-empty overflow colour invalidated the graph with every upload: a host that re-sends a settled pile's manifolds every step -- HostNarrowPhase mode --
            // paid a capture + instantiate per step for a schedule that had not changed.)
            const bool levels_now = sched_overflow.gorder.size() > overflow_level_threshold;
            if (levels_now != levels_before || (levels_now && glevels_before != sched_overflow.glevel_offsets)) graph_valid = false;
            if (before != sched_overflow.n_components || p0 != sched_overflow.d_order.p || p1 != sched_overflow.d_level_offsets.p || p2 != sched_overflow.d_comp_level_begin.p || p3 != sched_overflow.d_gorder.p) graph_valid = false;
        }
        islands_dirty = true;   // rebuilt by solver_front AFTER the prepare kernels are enqueued (host work overlaps them)
        incidence_dirty = false;
        return AVN_OK;
    }
    // Island blocks (k_island_substeps).  Islands = connected components of the bodies that have a SolverBody under "share a
    // manifold" (a body without one -- static, sleeping, disabled -- is never written by the solver and joins nothing;
    // kinematic bodies DO have a SolverBody that the solver reads and re-writes, so they merge like dynamic ones).  Eligible
    // when f32, no joints, few enough manifolds for the colour launches to be latency-bound and every island fits a block.
    // (... and few enough BODIES: the block builder is host code that touches every body with a SolverBody, and the one launch stages all of
    //  them in LDS -- at cfg4's 10^6 free bodies around 2 k manifolds the attempt cost 5 ms of host time per step, found by the stepped cfg4 timing)
    size_t island_max_bodies_total = 65536;
    bool island_candidate(size_t M) const { return sizeof(T) == 4 && island_enabled && M != 0 && M <= island_max_manifolds && dw.n_bodies <= island_max_bodies_total; }
    avn_status rebuild_island_blocks() {
        island_mode = false;
        const uint32_t N = dw.n_bodies, M = dw.n_manifolds;
        if (!island_candidate(M) || dw.n_joints) return AVN_OK;
        auto has_sb = [&](int32_t b) { return b >= 0 && (uint32_t)b < N && h_body_has_sb[(uint32_t)b]; };
        std::vector<uint32_t>& parent = isl_parent;
        parent.resize(N);
        bool labelled = false;
        static const bool host_labels = avn_env("AVN_ISLAND_LABELS_HOST") != nullptr;   // A/B: the host union-find below
        if (pipe_dev && slp_on && !host_labels) {
            // sleeping on: the persistent islands ARE a grouping of the awake bodies no manifold crosses (an island pending its split is merely
            // coarser than necessary) -- taken straight from the manager: no labelling kernel, no read-back, no synchronisation
            std::vector<uint32_t>& rep = isl_roots;
            rep.assign((size_t)isl.island_key_bound(), 0xFFFFFFFFu);
            for (uint32_t i = 0; i < N; ++i) {
                parent[i] = i;
                if (!h_body_has_sb[i] || !isl.body_has_node(i)) continue;
                const uint32_t k = isl.island_of(i);
                if (rep[k] == 0xFFFFFFFFu) rep[k] = i;
                parent[i] = rep[k];
            }
            isl_labels_step_valid = false;
            labelled = true;
        } else if (pipe_dev && !host_labels) {
            // device closed loop: the manifolds' bodies are on the device already -- label the islands there (k_islands.hip: lock-free
            // union-find, root = lowest body index, only bodies with a SolverBody connect) and fetch 4 bytes per body; parent[] then holds
            // roots directly
            avn_status st = island_buffers();
            if (st != AVN_OK) return st;
            HIPCHK(hipMemsetAsync(b_isl_ctr.p, 0, 64, stream));
            bool reuse = false;
            if (isl_labels_step_valid && isl_roots.size() == N) {
                // last step's labels, checked against this step's manifolds on the device: one small kernel and a 4-byte read-back instead of
                // the union-find (100 us at 13 k manifolds: deep trees under "root = lowest index") and the read-back of every label
                launch_islands_validate<T>(dw, b_isl_blk_label.as<uint32_t>(), b_isl_ctr.as<uint32_t>() + 8, stream, 1u);
                ++launches;
                uint32_t invalid = 1;
                HIPCHK(hipMemcpyAsync(&invalid, b_isl_ctr.as<uint32_t>() + 8, 4, hipMemcpyDeviceToHost, stream));
                HIPCHK(hipStreamSynchronize(stream));
                reuse = invalid == 0;
            }
            if (reuse) parent = isl_roots;
            else {
                launch_islands<T>(dw, b_isl_parent.as<uint32_t>(), b_isl_blk_label.as<uint32_t>(), b_isl_ctr.as<uint32_t>(), stream, 1u);
                launches += 3;
                HIPCHK(hipMemcpyAsync(parent.data(), b_isl_blk_label.p, (size_t)N * 4, hipMemcpyDeviceToHost, stream));
                HIPCHK(hipStreamSynchronize(stream));
                isl_roots = parent;   // (as the device holds them: 0xFFFFFFFF for a body that was no island node)
                isl_labels_step_valid = true;
            }
            for (uint32_t i = 0; i < N; ++i) if (parent[i] == 0xFFFFFFFFu || !h_body_has_sb[i]) parent[i] = i;   // (bodies without a SolverBody: never asked)
            labelled = true;
        } else for (uint32_t i = 0; i < N; ++i) parent[i] = i;
        auto find = [&](uint32_t x) { while (parent[x] != x) { parent[x] = parent[parent[x]]; x = parent[x]; } return x; };
        for (uint32_t m = 0; m < M && !labelled; ++m) {
            int32_t a = h_m_body1[m], b = h_m_body2[m];
            if (has_sb(a) && has_sb(b)) { uint32_t ra = find((uint32_t)a), rb = find((uint32_t)b); if (ra != rb) parent[std::max(ra, rb)] = std::min(ra, rb); }
        }
        // islands numbered by their lowest body (the root: unions keep the smaller index on top), bodies inside in index order
        std::vector<uint32_t>& island_of = isl_island_of; std::vector<uint32_t>& count = isl_count;
        island_of.assign(N, 0xFFFFFFFFu);
        std::vector<uint32_t>& root_island = isl_root_island;
        root_island.assign(N, 0xFFFFFFFFu);
        count.clear();
        for (uint32_t i = 0; i < N; ++i) {
            if (!h_body_has_sb[i]) continue;
            uint32_t r = find(i);
            // (with REUSED labels the root -- the lowest index of the component when it was labelled -- may have lost its SolverBody since: it
            //  then only lends its index to the group, so the root's island is kept apart from island_of, which lists MEMBERS)
            if (root_island[r] == 0xFFFFFFFFu) { root_island[r] = (uint32_t)count.size(); count.push_back(0u); }
            island_of[i] = root_island[r];   // islands are numbered by their first member with a SolverBody, bodies inside in index order
            if (++count[island_of[i]] > ISLAND_MAX_BODIES) return AVN_OK;   // an island too big for one workgroup's LDS: device-wide path
        }
        const uint32_t n_islands = (uint32_t)count.size();
        if (!n_islands) return AVN_OK;
        // manifolds per island (a manifold belongs to the island of its body that has a SolverBody)
        std::vector<uint32_t>& mcount = isl_mcount;
        mcount.assign(n_islands, 0u);
        for (uint32_t m = 0; m < M; ++m) {
            int32_t a = h_m_body1[m], b = h_m_body2[m];
            if (has_sb(a)) ++mcount[island_of[(uint32_t)a]]; else if (has_sb(b)) ++mcount[island_of[(uint32_t)b]]; else ++mcount[0];
        }
        // blocks = runs of consecutive islands of at most island_pack_bodies bodies (one island may exceed that, up to the LDS cap);
        // a run is also closed when its bodies + constraint records would no longer fit the LDS staging of the kernel's CACHE variant
        std::vector<uint32_t>& block_of = isl_block_of_island; std::vector<uint32_t>& body_off = isl_body_off;
        block_of.resize(n_islands);
        body_off.assign(1, 0u);
        std::vector<uint32_t>& cursor = isl_cursor;   // per island: next LDS slot
        cursor.resize(n_islands);
        // pack target: enough blocks to cover the 256 CUs before blocks grow (a block's pass time is flat up to ~256 manifolds per colour)
        uint32_t n_sb = 0;
        for (uint32_t k = 0; k < n_islands; ++k) n_sb += count[k];
        const uint32_t pack = std::min<uint32_t>(island_pack_bodies, std::max<uint32_t>(64u, (n_sb + 255u) / 256u));
        uint32_t in_block = 0, m_in_block = 0, max_bodies = 0, max_manifolds = 0;
        auto close_block = [&]() { max_bodies = std::max(max_bodies, in_block); max_manifolds = std::max(max_manifolds, m_in_block); body_off.push_back(body_off.back() + in_block); in_block = 0; m_in_block = 0; };
        for (uint32_t k = 0; k < n_islands; ++k) {
            if (in_block && (in_block + count[k] > pack || 6u * (in_block + count[k]) + 20u * (m_in_block + mcount[k]) > ISLAND_LDS_VEC4)) close_block();
            block_of[k] = (uint32_t)body_off.size() - 1;
            cursor[k] = in_block;
            in_block += count[k];
            m_in_block += mcount[k];
        }
        close_block();
        // the LDS layout is the same for every block (sized by the largest body and manifold counts)
        const uint32_t lm_pad = (max_manifolds + 1u) & ~1u;   // (the entry list behind the records is uint2: keep it 16-byte aligned)
        const bool cache_records = island_cache_records && 6u * max_bodies + 20u * lm_pad <= ISLAND_LDS_VEC4;
        const uint32_t n_blocks = (uint32_t)body_off.size() - 1;
        std::vector<uint32_t>& slot = isl_slot; std::vector<uint32_t>& bodies = isl_bodies;
        slot.assign(N, 0u);
        bodies.resize(body_off.back());
        for (uint32_t i = 0; i < N; ++i) {
            uint32_t k = island_of[i];
            if (k == 0xFFFFFFFFu) continue;
            slot[i] = cursor[k]++;
            bodies[body_off[block_of[k]] + slot[i]] = i;
        }
        // entries: counting sort of the manifolds by (block, colour slot), ascending manifold index inside (= list order: the
        // overflow colour's serial order); colour slot 0 = overflow (solved first), 1 + c = colour c
        std::vector<uint32_t>& col_off = isl_col_off; std::vector<uint32_t>& ent = isl_ent;
        col_off.assign((size_t)n_blocks * AVN_GRAPH_COLOR_COUNT + 1, 0u);
        auto key_of = [&](uint32_t m, uint32_t c) -> size_t {
            int32_t a = h_m_body1[m], b = h_m_body2[m];
            uint32_t blk = has_sb(a) ? block_of[island_of[(uint32_t)a]] : has_sb(b) ? block_of[island_of[(uint32_t)b]] : 0u;  // (no SolverBody on either side: touches no body, any block)
            return (size_t)blk * AVN_GRAPH_COLOR_COUNT + (c == AVN_COLOR_OVERFLOW_INDEX ? 0u : c + 1u);
        };
        for (uint32_t c = 0; c < AVN_GRAPH_COLOR_COUNT; ++c)
            for (uint32_t m = color_offsets[c]; m < color_offsets[c + 1]; ++m) ++col_off[key_of(m, c) + 1];
        for (size_t i = 1; i < col_off.size(); ++i) col_off[i] += col_off[i - 1];
        std::vector<uint32_t> next(col_off.begin(), col_off.end() - 1);
        ent.resize((size_t)M * 2);
        for (uint32_t c = 0; c < AVN_GRAPH_COLOR_COUNT; ++c)
            for (uint32_t m = color_offsets[c]; m < color_offsets[c + 1]; ++m) {
                int32_t a = h_m_body1[m], b = h_m_body2[m];
                uint32_t e = next[key_of(m, c)]++;
                ent[2 * (size_t)e] = m;
                ent[2 * (size_t)e + 1] = (has_sb(a) ? slot[(uint32_t)a] : 0u) | (has_sb(b) ? slot[(uint32_t)b] : 0u) << 16;
            }
        // one pinned staging block -> one async copy on the solver's stream (the consumer); no synchronisation: the stream
        // was idle when the previous block went up (rebuild_incidence / the narrow-phase read-back synchronise it every step)
        const size_t w0 = body_off.size(), w1 = bodies.size(), w2 = col_off.size(), w3 = ent.size();
        const size_t o1 = (w0 + 63) & ~(size_t)63, o2 = o1 + ((w1 + 63) & ~(size_t)63), o3 = o2 + ((w2 + 63) & ~(size_t)63), words = o3 + w3;
        if (pin_islands.ensure(words * 4) != hipSuccess) { error = "hipHostMalloc failed"; return AVN_ERR_OOM; }
        uint32_t* h = (uint32_t*)pin_islands.p;
        std::memcpy(h, body_off.data(), w0 * 4); std::memcpy(h + o1, bodies.data(), w1 * 4);
        std::memcpy(h + o2, col_off.data(), w2 * 4); std::memcpy(h + o3, ent.data(), w3 * 4);
        if (words * 4 > b_isl_bodies.cap) HIPCHK(hipStreamSynchronize(stream));   // growing frees the old block: nothing may still read it
        hipError_t err;
        b_isl_bodies.ensure(words * 4, err);
        if (err != hipSuccess) { error = "hipMalloc failed"; return AVN_ERR_OOM; }
        HIPCHK(hipMemcpyAsync(b_isl_bodies.p, h, words * 4, hipMemcpyHostToDevice, stream));
        uint32_t* d = b_isl_bodies.as<uint32_t>();
        islands = IslandBlocks{d, d + o1, d + o2, (const uint2*)(d + o3), n_blocks, max_bodies, lm_pad, cache_records ? 1u : 0u};
        island_mode = true;
        return AVN_OK;
    }
    avn_status impulses_download(const avn_impulses_out* o) override {
        if (!o) return AVN_ERR_BAD_ARG;
        size_t M = dw.n_manifolds;
        avn_status st = stage_reserve(al(sizeof(T) * 4 * M) * 2 + al(sizeof(T) * 8 * M) + 1024);
        if (st != AVN_OK) return st;
        T* a = o->warm_start_normal_impulse ? stage_alloc<T>(4 * M) : nullptr;
        T* b = o->warm_start_tangent_impulse ? stage_alloc<T>(8 * M) : nullptr;
        T* c = o->normal_impulse ? stage_alloc<T>(4 * M) : nullptr;
        launch_unpack_impulses<T>(dw, a, b, c, stream);
        HIPCHK(hipGetLastError());
        SOUT(o->warm_start_normal_impulse, a, 4 * M, T); SOUT(o->warm_start_tangent_impulse, b, 8 * M, T); SOUT(o->normal_impulse, c, 4 * M, T);
        HIPCHK(hipStreamSynchronize(stream));
        return AVN_OK;
    }
    avn_status constraints_download(const avn_constraints_out* o) override {
        if (!o) return AVN_ERR_BAD_ARG;
        size_t M = dw.n_manifolds;
        avn_status st = stage_reserve(al(M) * 2 + al(2 * M) + al(sizeof(T) * 3 * M) + al(sizeof(T) * 12 * M) * 2 + al(sizeof(T) * 4 * M) * 4 + al(sizeof(T) * 8 * M) + 4096);
        if (st != AVN_OK) return st;
        ConstraintsStage<T> s;
        s.point_count = o->point_count ? stage_alloc<uint8_t>(M) : nullptr;
        s.softness_non_dynamic = o->softness_non_dynamic ? stage_alloc<uint8_t>(M) : nullptr;
        s.relative_dominance = o->relative_dominance ? stage_alloc<int16_t>(M) : nullptr;
        s.tangent1 = o->tangent1 ? stage_alloc<T>(3 * M) : nullptr;
        s.anchor1 = o->anchor1 ? stage_alloc<T>(12 * M) : nullptr;
        s.initial_separation = o->initial_separation ? stage_alloc<T>(4 * M) : nullptr;
        s.normal_impulse = o->normal_impulse ? stage_alloc<T>(4 * M) : nullptr;
        s.total_impulse = o->total_impulse ? stage_alloc<T>(4 * M) : nullptr;
        s.normal_effective_mass = o->normal_effective_mass ? stage_alloc<T>(4 * M) : nullptr;
        s.tangent_impulse = o->tangent_impulse ? stage_alloc<T>(8 * M) : nullptr;
        s.tangent_k = o->tangent_effective_inverse_mass ? stage_alloc<T>(12 * M) : nullptr;
        if (stage_off) HIPCHK(hipMemsetAsync(stage.p, 0, stage_off, stream));  // absent constraints read back as zeros
        launch_unpack_constraints<T>(dw, s, stream);
        HIPCHK(hipGetLastError());
        SOUT(o->point_count, s.point_count, M, uint8_t); SOUT(o->softness_non_dynamic, s.softness_non_dynamic, M, uint8_t);
        SOUT(o->relative_dominance, s.relative_dominance, M, int16_t); SOUT(o->tangent1, s.tangent1, 3 * M, T);
        SOUT(o->anchor1, s.anchor1, 12 * M, T); SOUT(o->initial_separation, s.initial_separation, 4 * M, T);
        SOUT(o->normal_impulse, s.normal_impulse, 4 * M, T); SOUT(o->total_impulse, s.total_impulse, 4 * M, T);
        SOUT(o->normal_effective_mass, s.normal_effective_mass, 4 * M, T); SOUT(o->tangent_impulse, s.tangent_impulse, 8 * M, T);
        SOUT(o->tangent_effective_inverse_mass, s.tangent_k, 12 * M, T);
        HIPCHK(hipStreamSynchronize(stream));
        return AVN_OK;
    }

