// world/contacts.hpp -- fragment of the body of `template <class T> struct World` (avn_world.hip includes it inside the class):
// ContactGraph side: materials, contact rows, narrow-phase driver for host bookkeeping, handle lists, row transfer, batch query.

    // ---- narrow phase, part 2 ------------------------------------------------------------------------------------------
    avn_status collider_materials_upload(const avn_collider_materials* m) override {
        slp_world_asleep = slp_world_idle = false;
        if (!m || m->count != bp.n_colliders) { error = "collider_materials_upload: count must equal the collider count"; return AVN_ERR_BAD_ARG; }
        uint32_t C = m->count;
        std::vector<V> mats(C);
        materials_restitution = false;
        for (uint32_t i = 0; i < C; ++i) {
            T fr = m->friction ? ((const T*)m->friction)[i] : T(0.5), re = m->restitution ? ((const T*)m->restitution)[i] : T(0);
            uint32_t fc = m->friction_combine ? m->friction_combine[i] : (uint32_t)AVN_COMBINE_AVERAGE, rc = m->restitution_combine ? m->restitution_combine[i] : (uint32_t)AVN_COMBINE_AVERAGE;
            if (fc < AVN_COMBINE_AVERAGE || fc > AVN_COMBINE_MAX || rc < AVN_COMBINE_AVERAGE || rc > AVN_COMBINE_MAX) { error = "collider_materials_upload: bad combine rule"; return AVN_ERR_BAD_ARG; }
            mats[i] = make4<T>(fr, re, bits_to_scalar(fc | (rc << 8), T(0)), T(0));
            if (!(re == T(0))) materials_restitution = true;
        }
        HIPCHK(hipStreamSynchronize(stream));
        if (C) HIPCHK(hipMemcpy(ct.col_mat, mats.data(), (size_t)C * sizeof(V), hipMemcpyHostToDevice));
        if (use_handles) any_restitution = materials_restitution || hk_restitution;
        return AVN_OK;
    }
    avn_status ensure_contact_rows(uint32_t rows) {
        if (rows <= ct.cap) return AVN_OK;
        HIPCHK(hipStreamSynchronize(stream));
        uint32_t old = ct.cap;
        size_t c = std::max<size_t>(rows, (size_t)old + old / 2);
        c = (c + 63) & ~(size_t)63;
        // grow with contents: the rows are persistent state
        auto grow_flat = [&](DevBuf& b, size_t elem, void** field) -> avn_status {
            hipError_t err;
            b.ensure(c * elem, err, true, stream);
            if (err != hipSuccess) { error = "hipMalloc failed"; return AVN_ERR_OOM; }
            *field = b.p;
            return AVN_OK;
        };
        avn_status st;
        if ((st = grow_flat(b_ct_meta, sizeof(uint4), (void**)&ct.meta)) != AVN_OK) return st;
        if ((st = grow_flat(b_ct_dcount, sizeof(int32_t), (void**)&ct.dcount)) != AVN_OK) return st;
        if ((st = grow_flat(b_ct_rows, AVN_CT_ROW_V4 * sizeof(V), (void**)&ct.rows)) != AVN_OK) return st;   // one block of 16 records per row: grows like a flat array
        HIPCHK(hipMemset((char*)ct.meta + (size_t)old * sizeof(uint4), 0, (c - old) * sizeof(uint4)));
        // the narrow phase's survivor list (scratch of one launch pair: nothing to keep) and its counters
        {
            hipError_t err;
            const size_t slots = c + np_survivor_list_slack();   // 64 lists, each rounded up to whole workgroups of the first kernel (k_narrow.hip)
            b_np_row.ensure(slots * sizeof(uint32_t), err);
            if (err == hipSuccess) b_np_axis.ensure(3 * slots * sizeof(T), err);
            if (err == hipSuccess && !b_np_ctr.p) { b_np_ctr.ensure(np_survivor_counter_bytes(), err); if (err == hipSuccess) HIPCHK(hipMemset(b_np_ctr.p, 0, np_survivor_counter_bytes())); }
            if (err != hipSuccess) { error = "hipMalloc failed"; return AVN_ERR_OOM; }
            ct.np_row = b_np_row.as<uint32_t>(); ct.np_axis = b_np_axis.as<T>(); ct.np_ctr = b_np_ctr.as<uint32_t>();
        }
        ct.cap = (uint32_t)c;
        h_ct_used.resize(c, 0); h_ct_c1.resize(c, 0); h_ct_c2.resize(c, 0); h_ct_b1.resize(c, -1); h_ct_b2.resize(c, -1);
        if (pipe_dev) return pg_ensure_rows(ct.cap);
        return AVN_OK;
    }
    avn_status contact_pairs_add(const avn_contact_pairs* p) override {
        if (!p || (p->count && (!p->contact_id || !p->collider1 || !p->collider2 || !p->pair_flags))) { error = "contact_pairs_add: null array"; return AVN_ERR_BAD_ARG; }
        uint32_t n = p->count;
        if (!n) return AVN_OK;
        uint32_t max_id = 0;
        for (uint32_t i = 0; i < n; ++i) max_id = std::max(max_id, p->contact_id[i]);
        avn_status st = ensure_contact_rows(max_id + 1);
        if (st != AVN_OK) return st;
        std::vector<uint32_t> s1(n), s2(n);
        for (uint32_t i = 0; i < n; ++i) {
            auto a = entity_slot.find(p->collider1[i]), b = entity_slot.find(p->collider2[i]);
            if (a == entity_slot.end() || b == entity_slot.end()) { error = "contact_pairs_add: unknown collider"; return AVN_ERR_BAD_ARG; }
            if (h_ct_used[p->contact_id[i]]) { error = "contact_pairs_add: contact id in use"; return AVN_ERR_STATE; }
            s1[i] = a->second; s2[i] = b->second;
        }
        for (uint32_t i = 0; i < n; ++i) {
            uint32_t id = p->contact_id[i];
            h_ct_used[id] = 1; h_ct_c1[id] = p->collider1[i]; h_ct_c2[id] = p->collider2[i];
            h_ct_b1[id] = h_col_body[s1[i]]; h_ct_b2[id] = h_col_body[s2[i]];
            uint32_t x = p->collider1[i], y = p->collider2[i];
            h_live_keys.insert(x < y ? ((uint64_t)x << 32) | y : ((uint64_t)y << 32) | x);
        }
        contact_keys_live = true;
        if ((st = stage_reserve(al(4 * (size_t)n) * 4 + 1024)) != AVN_OK) return st;
        const uint32_t *d_id, *d_s1, *d_s2, *d_pf;
        if ((st = stage_in<uint32_t>(p->contact_id, n, &d_id)) != AVN_OK) return st;
        if ((st = stage_in<uint32_t>(s1.data(), n, &d_s1)) != AVN_OK) return st;
        if ((st = stage_in<uint32_t>(s2.data(), n, &d_s2)) != AVN_OK) return st;
        if ((st = stage_in<uint32_t>(p->pair_flags, n, &d_pf)) != AVN_OK) return st;
        launch_init_contact_rows<T>(ct, d_id, d_s1, d_s2, d_pf, n, stream);
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(stream));
        return AVN_OK;
    }
    avn_status contact_pairs_remove(const uint32_t* ids, size_t n) override {
        if (n && !ids) return AVN_ERR_BAD_ARG;
        if (!n) return AVN_OK;
        std::vector<uint64_t> keys(n);
        for (size_t i = 0; i < n; ++i) {
            if (ids[i] >= ct.cap || !h_ct_used[ids[i]]) { error = "contact_pairs_remove: no such contact"; return AVN_ERR_STATE; }
            uint32_t x = h_ct_c1[ids[i]], y = h_ct_c2[ids[i]];
            keys[i] = x < y ? ((uint64_t)x << 32) | y : ((uint64_t)y << 32) | x;
        }
        for (size_t i = 0; i < n; ++i) { h_ct_used[ids[i]] = 0; h_live_keys.erase(keys[i]); }
        avn_status st = stage_reserve(al(4 * n) + al(8 * n) + 1024);
        if (st != AVN_OK) return st;
        const uint32_t* d_id; const uint64_t* d_keys;
        if ((st = stage_in<uint32_t>(ids, n, &d_id)) != AVN_OK) return st;
        if ((st = stage_in<uint64_t>(keys.data(), n, &d_keys)) != AVN_OK) return st;
        launch_clear_contact_rows<T>(ct, d_id, (uint32_t)n, stream);
        launch_hs_remove(bp.pair_set, bp.pair_set_cap, d_keys, (uint32_t)n, stream);
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(stream));
        return AVN_OK;
    }
    avn_status active_pairs_set(const uint32_t* ids, size_t n) override {
        if (n && !ids) return AVN_ERR_BAD_ARG;
        for (size_t i = 0; i < n; ++i)
            if (ids[i] >= ct.cap || !h_ct_used[ids[i]]) { error = "active_pairs_set: no such contact"; return AVN_ERR_STATE; }
        HIPCHK(hipStreamSynchronize(stream));
        hipError_t err;
        b_active.ensure(std::max<size_t>(n, 1) * 4, err);
        if (err != hipSuccess) { error = "hipMalloc failed"; return AVN_ERR_OOM; }
        b_changes.ensure(std::max<size_t>(n, 1) * sizeof(avn_contact_change) + 64, err);
        if (err != hipSuccess) { error = "hipMalloc failed"; return AVN_ERR_OOM; }
        if (n) HIPCHK(hipMemcpy(b_active.p, ids, n * 4, hipMemcpyHostToDevice));
        n_active = (uint32_t)n;
        return AVN_OK;
    }
    avn_status narrow_phase() {
        h_changes.clear();
        if (!n_active) return AVN_OK;
        uint32_t* d_count = b_misc.as<uint32_t>() + 40;
        const NpHostList hl = hs_begin(stream);
        launch_narrow_phase<T>(dw, bp, ct, params, b_active.as<uint32_t>(), n_active, b_changes.as<avn_contact_change>(), d_count, stream, hl);
        ++launches;
        HIPCHK(hipGetLastError());
        if (hl.any()) hs_launches.push_back(HsLaunch{0, n_active, 0, 0, 0, b_active.as<uint32_t>()});
        if (hl.queries) {   // pairs with a host-shaped collider: contact_manifolds_with_context on the host, the rest of update_contacts here (world/host_shapes.hpp)
            avn_status sh = hs_manifolds(false, params, b_changes.as<avn_contact_change>(), d_count, nullptr, nullptr, stream);
            if (sh != AVN_OK) return sh;
        }
        if (hl.hook.count) {   // pairs waiting for CollisionHooks::modify_contacts (world/hooks.hpp)
            avn_status sh = hk_modify(false, params, b_changes.as<avn_contact_change>(), d_count, nullptr, nullptr, stream);
            if (sh != AVN_OK) return sh;
        }
        // the count and the first CHANGES_PREFIX changes come back in one round trip (pinned memory, one synchronisation);
        // only a step with more changes than that pays a second copy
        const uint32_t prefix = std::min<uint32_t>(CHANGES_PREFIX, n_active);
        HIPCHK(pin_changes.ensure(64 + (size_t)CHANGES_PREFIX * sizeof(avn_contact_change)));
        uint32_t* h_cnt = (uint32_t*)pin_changes.p;
        avn_contact_change* h_pre = (avn_contact_change*)((char*)pin_changes.p + 64);
        HIPCHK(hipMemcpyAsync(h_cnt, d_count, 4, hipMemcpyDeviceToHost, stream));
        HIPCHK(hipMemcpyAsync(h_pre, b_changes.p, (size_t)prefix * sizeof(avn_contact_change), hipMemcpyDeviceToHost, stream));
        HIPCHK(hipStreamSynchronize(stream));
        const uint32_t cnt = *h_cnt;
        if (cnt) {
            h_changes.resize(cnt);
            std::memcpy(h_changes.data(), h_pre, (size_t)std::min(cnt, prefix) * sizeof(avn_contact_change));
            if (cnt > prefix)
                HIPCHK(hipMemcpy(h_changes.data() + prefix, b_changes.as<avn_contact_change>() + prefix, (size_t)(cnt - prefix) * sizeof(avn_contact_change), hipMemcpyDeviceToHost));
            // ContactStatusBits are walked in ascending contact id (system_param.rs:141-145)
            std::sort(h_changes.begin(), h_changes.end(), [](const avn_contact_change& a, const avn_contact_change& b) { return a.contact_id < b.contact_id; });
        }
        return AVN_OK;
    }
    avn_status contact_changes_get(const avn_contact_change** out, size_t* n) override {
        if (!out || !n) return AVN_ERR_BAD_ARG;
        if (pipe_on && pipe_dev) { const avn_status st = pipeline_device_changes_fetch(); if (st != AVN_OK) return st; }
        *out = h_changes.data(); *n = h_changes.size();
        return AVN_OK;
    }
    avn_status ensure_manifold_capacity(uint32_t M) {
        bool moved = false;
        if (M > cap_manifolds) {
            HIPCHK(hipStreamSynchronize(stream));
            size_t c = std::max<size_t>(M, cap_manifolds + cap_manifolds / 2);
            c = (c + 63) & ~(size_t)63;  // keep every point plane 1 KiB aligned
            GROW(b_m_bodies, c, dw.m_bodies); GROW(b_m_n, c, dw.m_n); GROW(b_m_tv, c, dw.m_tv); GROW(b_m_meta, c, dw.m_meta);
            GROW(b_mp_a1, 4 * c, dw.mp_a1); GROW(b_mp_a2, 4 * c, dw.mp_a2); GROW(b_mp_w, 4 * c, dw.mp_w);
            GROW(b_c_h1, c, dw.c_h1); GROW(b_c_pa, 4 * c, dw.c_pa); GROW(b_c_pb, 4 * c, dw.c_pb); GROW(b_c_pc, 4 * c, dw.c_pc); GROW(b_c_pd, 4 * c, dw.c_pd);
            GROW(b_c_reldom, c, dw.c_reldom);
            cap_manifolds = (uint32_t)c;
            dw.m_stride = cap_manifolds;
        }
        if (moved) graph_valid = false;
        return AVN_OK;
    }
    uint32_t ovf_grid_blocks = 0;   // device closed loop: captured grid of the overflow colour's dataflow pass (with slack, like the colours')
    // Colours small enough to leave most SIMDs idle run the eight-lanes-per-manifold solve (k_color_pass_oct, f32): chosen per colour with hysteresis, and only
    // ever changed together with a re-capture of the substep graph.  Bit-identical either way.  (AVN_NO_OCT=1 in `make measure` builds: A/B.)
    uint32_t oct_mask = 0;
    bool oct_enabled = sizeof(T) == 4;
    static constexpr uint32_t OCT_ON = 12288, OCT_OFF = 20480;
    void set_color_offsets(const uint32_t* offsets) {
        if (!use_handles && std::memcmp(color_offsets, offsets, sizeof color_offsets) != 0) graph_valid = false;  // (ranges captured as kernel arguments; handle mode reads them from the device)
        std::memcpy(color_offsets, offsets, sizeof color_offsets);
        {
            const uint32_t n23 = color_offsets[AVN_COLOR_OVERFLOW_INDEX + 1] - color_offsets[AVN_COLOR_OVERFLOW_INDEX];
            const uint32_t need = (n23 + 63u) / 64u;
            if (pipe_dev && (need > ovf_grid_blocks || ovf_grid_blocks > 4 * need + 64)) { ovf_grid_blocks = n23 ? (n23 + n23 / 4 + 64 + 63u) / 64u : 0u; graph_valid = false; }
        }
        // launch grids per colour: the kernels read the live colour ranges from device memory, so a captured grid stays
        // valid while it still covers the colour; grids are captured with 25 % slack and re-captured when outgrown
        for (int c = 0; c < AVN_GRAPH_COLOR_COUNT; ++c) {
            uint32_t cnt = color_offsets[c + 1] - color_offsets[c];
            if (c == AVN_COLOR_OVERFLOW_INDEX) {  // serial kernel: the "grid" is only an on/off flag
                if (cnt && !grid_blocks[c]) { grid_blocks[c] = 8; graph_valid = false; }
                continue;
            }
            uint32_t need = cnt ? color_grid_blocks(cnt) : 0u;
            if (need > grid_blocks[c] || grid_blocks[c] > 4 * need + 64) {
                grid_blocks[c] = cnt ? color_grid_blocks(cnt + cnt / 4 + 64) : 0u;
                graph_valid = false;
            }
            const bool was = (oct_mask >> c) & 1u;
            const bool now = oct_enabled && !halo_on && cnt != 0 && (was ? cnt <= OCT_OFF : cnt <= OCT_ON);
            if (now != was) { oct_mask ^= 1u << c; graph_valid = false; }
        }
    }
    avn_status manifold_handles_upload(const uint32_t* offsets, const uint32_t* ids) override {
        if (!have_bodies) { error = "manifold_handles_upload before bodies_upload"; return AVN_ERR_STATE; }
        if (!offsets || offsets[0] != 0) { error = "manifold_handles_upload: bad offsets"; return AVN_ERR_BAD_ARG; }
        for (int c = 0; c < AVN_GRAPH_COLOR_COUNT; ++c) if (offsets[c] > offsets[c + 1]) { error = "manifold_handles_upload: offsets not monotone"; return AVN_ERR_BAD_ARG; }
        uint32_t M = offsets[AVN_GRAPH_COLOR_COUNT];
        if (M && !ids) return AVN_ERR_BAD_ARG;
        h_m_body1.resize(M); h_m_body2.resize(M);
        for (uint32_t i = 0; i < M; ++i)
            if (ids[i] >= ct.cap || !h_ct_used[ids[i]]) { error = "manifold_handles_upload: no such contact"; return AVN_ERR_STATE; }
        // the host only needs the bodies of the OVERFLOW colour's manifolds (entry lists + level schedule); the incidence of
        // colours 0..22 is built on the device
        // (... and ALL of them when the set is small enough for the island blocks, whose entry lists are host-built)
        for (uint32_t i = island_candidate(M) ? 0u : offsets[AVN_COLOR_OVERFLOW_INDEX]; i < M; ++i) { h_m_body1[i] = h_ct_b1[ids[i]]; h_m_body2[i] = h_ct_b2[ids[i]]; }
        HIPCHK(hipStreamSynchronize(stream));
        avn_status st = ensure_manifold_capacity(M);
        if (st != AVN_OK) return st;
        if (dw.n_manifolds != M) graph_valid = false;
        dw.n_manifolds = M;
        set_color_offsets(offsets);
        hipError_t err;
        if (b_handles.ensure(std::max<size_t>(M, 1) * 4, err)) graph_valid = false;
        if (err != hipSuccess) { error = "hipMalloc failed"; return AVN_ERR_OOM; }
        HIPCHK(hipMemcpy(dw.color_offsets, color_offsets, sizeof color_offsets, hipMemcpyHostToDevice));
        if (M) HIPCHK(hipMemcpy(b_handles.p, ids, (size_t)M * 4, hipMemcpyHostToDevice));
        if (!use_handles) graph_valid = false;
        use_handles = true;
        any_restitution = materials_restitution || hk_restitution;
        incidence_dirty = true;
        return AVN_OK;
    }
    avn_status contacts_download(const uint32_t* ids, size_t n, const avn_contacts_out* o) override {
        if (!o || (n && !ids)) return AVN_ERR_BAD_ARG;
        for (size_t i = 0; i < n; ++i)
            if (ids[i] >= ct.cap || (!pipe_dev && !h_ct_used[ids[i]])) { error = "contacts_download: no such contact"; return AVN_ERR_STATE; }   // (device closed loop: liveness is a row flag)
        avn_status st = stage_reserve(al(4 * n) * 4 + al(n) + al(sizeof(T) * 3 * n) + al(sizeof(T) * n) * 2 + al(sizeof(T) * 12 * n) * 2 + al(sizeof(T) * 4 * n) * 4 + al(sizeof(T) * 8 * n) + al(16 * n) * 2 + 4096);
        if (st != AVN_OK) return st;
        const uint32_t* d_id;
        if ((st = stage_in<uint32_t>(ids, n, &d_id)) != AVN_OK) return st;
        ContactsStage<T> s;
        s.flags = o->flags ? stage_alloc<uint32_t>(n) : nullptr; s.point_count = o->point_count ? stage_alloc<uint8_t>(n) : nullptr;
        s.normal = o->normal ? stage_alloc<T>(3 * n) : nullptr; s.friction = o->friction ? stage_alloc<T>(n) : nullptr; s.restitution = o->restitution ? stage_alloc<T>(n) : nullptr;
        s.anchor1 = o->anchor1 ? stage_alloc<T>(12 * n) : nullptr; s.anchor2 = o->anchor2 ? stage_alloc<T>(12 * n) : nullptr;
        s.penetration = o->penetration ? stage_alloc<T>(4 * n) : nullptr; s.normal_speed = o->normal_speed ? stage_alloc<T>(4 * n) : nullptr;
        s.warm_n = o->warm_start_normal_impulse ? stage_alloc<T>(4 * n) : nullptr; s.warm_t = o->warm_start_tangent_impulse ? stage_alloc<T>(8 * n) : nullptr;
        s.normal_impulse = o->normal_impulse ? stage_alloc<T>(4 * n) : nullptr;
        s.feature_id1 = o->feature_id1 ? stage_alloc<uint32_t>(4 * n) : nullptr; s.feature_id2 = o->feature_id2 ? stage_alloc<uint32_t>(4 * n) : nullptr;
        launch_unpack_contacts<T>(ct, d_id, (uint32_t)n, s, stream);
        HIPCHK(hipGetLastError());
        SOUT(o->flags, s.flags, n, uint32_t); SOUT(o->point_count, s.point_count, n, uint8_t); SOUT(o->normal, s.normal, 3 * n, T);
        SOUT(o->friction, s.friction, n, T); SOUT(o->restitution, s.restitution, n, T); SOUT(o->anchor1, s.anchor1, 12 * n, T); SOUT(o->anchor2, s.anchor2, 12 * n, T);
        SOUT(o->penetration, s.penetration, 4 * n, T); SOUT(o->normal_speed, s.normal_speed, 4 * n, T); SOUT(o->warm_start_normal_impulse, s.warm_n, 4 * n, T);
        SOUT(o->warm_start_tangent_impulse, s.warm_t, 8 * n, T); SOUT(o->normal_impulse, s.normal_impulse, 4 * n, T);
        SOUT(o->feature_id1, s.feature_id1, 4 * n, uint32_t); SOUT(o->feature_id2, s.feature_id2, 4 * n, uint32_t);
        HIPCHK(hipStreamSynchronize(stream));
        return AVN_OK;
    }
    avn_status contacts_upload(const uint32_t* ids, size_t n, const avn_contacts_in* in) override {
        if (!in || (n && !ids)) return AVN_ERR_BAD_ARG;
        if (n && (!in->flags || !in->point_count || !in->normal || !in->friction || !in->restitution || !in->anchor1 || !in->anchor2 || !in->penetration || !in->normal_speed ||
                  !in->warm_start_normal_impulse || !in->warm_start_tangent_impulse || !in->normal_impulse || !in->feature_id1 || !in->feature_id2)) {
            error = "contacts_upload: every field of avn_contacts_in is required"; return AVN_ERR_BAD_ARG;
        }
        for (size_t i = 0; i < n; ++i) {
            if (ids[i] >= ct.cap || (!pipe_dev && !h_ct_used[ids[i]])) { error = "contacts_upload: no such contact (avn_contact_pairs_add first)"; return AVN_ERR_STATE; }
            if (in->point_count[i] > AVN_MAX_MANIFOLD_POINTS) { error = "contacts_upload: point_count > 4"; return AVN_ERR_BAD_ARG; }
        }
        if (!n) return AVN_OK;
        avn_status st = stage_reserve(al(4 * n) * 2 + al(n) + al(sizeof(T) * 3 * n) + al(sizeof(T) * n) * 2 + al(sizeof(T) * 12 * n) * 2 + al(sizeof(T) * 4 * n) * 4 + al(sizeof(T) * 8 * n) + al(16 * n) * 2 + 4096);
        if (st != AVN_OK) return st;
        const uint32_t *d_id, *d_flags, *d_f1, *d_f2; const uint8_t* d_pc;
        const T *d_n, *d_fr, *d_re, *d_a1, *d_a2, *d_pen, *d_ns, *d_wn, *d_wt, *d_ni;
        if ((st = stage_in<uint32_t>(ids, n, &d_id)) != AVN_OK || (st = stage_in<uint32_t>(in->flags, n, &d_flags)) != AVN_OK || (st = stage_in<uint8_t>(in->point_count, n, &d_pc)) != AVN_OK ||
            (st = stage_in<T>((const T*)in->normal, 3 * n, &d_n)) != AVN_OK || (st = stage_in<T>((const T*)in->friction, n, &d_fr)) != AVN_OK ||
            (st = stage_in<T>((const T*)in->restitution, n, &d_re)) != AVN_OK || (st = stage_in<T>((const T*)in->anchor1, 12 * n, &d_a1)) != AVN_OK ||
            (st = stage_in<T>((const T*)in->anchor2, 12 * n, &d_a2)) != AVN_OK || (st = stage_in<T>((const T*)in->penetration, 4 * n, &d_pen)) != AVN_OK ||
            (st = stage_in<T>((const T*)in->normal_speed, 4 * n, &d_ns)) != AVN_OK || (st = stage_in<T>((const T*)in->warm_start_normal_impulse, 4 * n, &d_wn)) != AVN_OK ||
            (st = stage_in<T>((const T*)in->warm_start_tangent_impulse, 8 * n, &d_wt)) != AVN_OK || (st = stage_in<T>((const T*)in->normal_impulse, 4 * n, &d_ni)) != AVN_OK ||
            (st = stage_in<uint32_t>(in->feature_id1, 4 * n, &d_f1)) != AVN_OK || (st = stage_in<uint32_t>(in->feature_id2, 4 * n, &d_f2)) != AVN_OK)
            return st;
        ContactsStage<T> s;   // read-only here; the struct is shared with the download direction
        s.flags = const_cast<uint32_t*>(d_flags); s.point_count = const_cast<uint8_t*>(d_pc); s.normal = const_cast<T*>(d_n); s.friction = const_cast<T*>(d_fr);
        s.restitution = const_cast<T*>(d_re); s.anchor1 = const_cast<T*>(d_a1); s.anchor2 = const_cast<T*>(d_a2); s.penetration = const_cast<T*>(d_pen);
        s.normal_speed = const_cast<T*>(d_ns); s.warm_n = const_cast<T*>(d_wn); s.warm_t = const_cast<T*>(d_wt); s.normal_impulse = const_cast<T*>(d_ni);
        s.feature_id1 = const_cast<uint32_t*>(d_f1); s.feature_id2 = const_cast<uint32_t*>(d_f2);
        // device closed loop: liveness is a row flag only the device knows -- the kernel skips ids without a live row and raises bit 2 of
        // the error word, read back here (the call synchronises anyway)
        uint32_t* d_err = pipe_dev ? pg.ctr + PGC_ERROR : nullptr;
        launch_pack_contacts<T>(ct, d_id, (uint32_t)n, s, d_err, stream);
        HIPCHK(hipGetLastError());
        if (d_err) { avn_status se = pg_error_fetch(); if (se != AVN_OK) return se; }
        HIPCHK(hipStreamSynchronize(stream));   // the staging buffer is reused by the next call
        return pg_error_check();
    }
    avn_status pairs_get(const avn_pair** out, size_t* n) override {
        if (!out || !n) return AVN_ERR_BAD_ARG;
        if (pipe_dev && h_pairs.empty() && last_timers.pair_count) {
            // device closed loop: the step's new pairs (emission order) never left the device -- fetched on request (tests, inspection)
            HIPCHK(hipStreamSynchronize(stream)); HIPCHK(hipStreamSynchronize(stream_bp));
            h_pairs.resize(last_timers.pair_count);
            HIPCHK(hipMemcpy(h_pairs.data(), b_pairs.p, (size_t)last_timers.pair_count * sizeof(avn_pair), hipMemcpyDeviceToHost));
        }
        *out = h_pairs.data();
        *n = h_pairs.size();
        return AVN_OK;
    }
    avn_status aabbs_download(void* mn, void* mx, uint32_t* ents, size_t* n_iv) override {
        size_t C = bp.n_colliders, I = bp.n_intervals;
        avn_status st = stage_reserve(al(sizeof(T) * 3 * C) * 2 + al(4 * I) + 1024);
        if (st != AVN_OK) return st;
        T* a = mn ? stage_alloc<T>(3 * C) : nullptr;
        T* b = mx ? stage_alloc<T>(3 * C) : nullptr;
        uint32_t* e = ents ? stage_alloc<uint32_t>(I) : nullptr;
        launch_unpack_aabbs<T>(bp, a, b, e, stream);
        HIPCHK(hipGetLastError());
        SOUT(mn, a, 3 * C, T); SOUT(mx, b, 3 * C, T); SOUT(ents, e, I, uint32_t);
        HIPCHK(hipStreamSynchronize(stream));
        if (n_iv) *n_iv = I;
        return AVN_OK;
    }
    avn_status dynamic_bounds(double* mn, double* mx) override {
        if (!mn || !mx) return AVN_ERR_BAD_ARG;
        const double inf = std::numeric_limits<double>::infinity();
        for (int k = 0; k < 3; ++k) { mn[k] = inf; mx[k] = -inf; }
        uint32_t nb = (bp.n_colliders + 255) / 256;
        if (!nb) return AVN_OK;
        avn_status st = stage_reserve((size_t)nb * 6 * sizeof(T) + 1024);
        if (st != AVN_OK) return st;
        T* part = stage_alloc<T>((size_t)nb * 6);
        launch_dynamic_bounds<T>(dw, bp, part, stream);
        HIPCHK(hipGetLastError());
        std::vector<T> h((size_t)nb * 6);
        HIPCHK(hipMemcpyAsync(h.data(), part, h.size() * sizeof(T), hipMemcpyDeviceToHost, stream));
        HIPCHK(hipStreamSynchronize(stream));
        for (uint32_t b = 0; b < nb; ++b)
            for (int k = 0; k < 3; ++k) { mn[k] = std::min(mn[k], (double)h[b * 6 + k]); mx[k] = std::max(mx[k], (double)h[b * 6 + 3 + k]); }
        return AVN_OK;
    }
    // batch contact_query::contact_manifolds (k_narrow.hip)
    avn_status contact_manifolds(const avn_shape_pairs* p, const avn_query_manifolds_out* o) override {
        if (!p || !o || (p->count && (!p->shape1 || !p->shape2 || !p->half_extents1 || !p->half_extents2 || !p->position1 || !p->position2 ||
                                      !p->rotation1 || !p->rotation2 || !p->prediction_distance))) { error = "contact_manifolds: null array"; return AVN_ERR_BAD_ARG; }
        size_t n = p->count;
        for (size_t i = 0; i < n; ++i)
            if (p->shape1[i] > AVN_SHAPE_BALL || p->shape2[i] > AVN_SHAPE_BALL) { error = "contact_manifolds: unknown shape"; return AVN_ERR_BAD_ARG; }
        const size_t Q = AVN_MAX_QUERY_POINTS;
        avn_status st = stage_reserve(al(n) * 3 + al(sizeof(T) * 3 * n) * 5 + al(sizeof(T) * 4 * n) * 2 + al(sizeof(T) * n) + al(sizeof(T) * 3 * Q * n) * 3 +
                                      al(sizeof(T) * Q * n) + al(4 * Q * n) * 2 + 64 * 32);
        if (st != AVN_OK) return st;
        QueryStage<T> s;
        std::memset(&s, 0, sizeof s);
        SIN(shape1, p->shape1, n, uint8_t); SIN(shape2, p->shape2, n, uint8_t);
        SIN(half_extents1, p->half_extents1, 3 * n, T); SIN(position1, p->position1, 3 * n, T); SIN(rotation1, p->rotation1, 4 * n, T);
        SIN(half_extents2, p->half_extents2, 3 * n, T); SIN(position2, p->position2, 3 * n, T); SIN(rotation2, p->rotation2, 4 * n, T);
        SIN(prediction, p->prediction_distance, n, T);
        s.point_count = o->point_count ? stage_alloc<uint8_t>(n) : nullptr;
        s.normal = o->normal ? stage_alloc<T>(3 * n) : nullptr;
        s.anchor1 = o->anchor1 ? stage_alloc<T>(3 * Q * n) : nullptr;
        s.anchor2 = o->anchor2 ? stage_alloc<T>(3 * Q * n) : nullptr;
        s.point = o->point ? stage_alloc<T>(3 * Q * n) : nullptr;
        s.penetration = o->penetration ? stage_alloc<T>(Q * n) : nullptr;
        s.feature_id1 = o->feature_id1 ? stage_alloc<uint32_t>(Q * n) : nullptr;
        s.feature_id2 = o->feature_id2 ? stage_alloc<uint32_t>(Q * n) : nullptr;
        launch_contact_manifolds_query<T>(s, (uint32_t)n, stream);
        HIPCHK(hipGetLastError());
        SOUT(o->point_count, s.point_count, n, uint8_t); SOUT(o->normal, s.normal, 3 * n, T);
        SOUT(o->anchor1, s.anchor1, 3 * Q * n, T); SOUT(o->anchor2, s.anchor2, 3 * Q * n, T); SOUT(o->point, s.point, 3 * Q * n, T);
        SOUT(o->penetration, s.penetration, Q * n, T); SOUT(o->feature_id1, s.feature_id1, Q * n, uint32_t); SOUT(o->feature_id2, s.feature_id2, Q * n, uint32_t);
        HIPCHK(hipStreamSynchronize(stream));
        return AVN_OK;
    }
