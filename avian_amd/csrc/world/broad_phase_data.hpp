// world/broad_phase_data.hpp -- fragment of the body of `template <class T> struct World` (avn_world.hip includes it inside the class):
// BroadPhasePlugin's state: pair set, joint-disabled set, colliders / AabbIntervals upkeep.

    // ---- broad phase -------------------------------------------------------------------------------------
    avn_status build_hash_set(DevBuf& buf, uint64_t*& tab, uint32_t& cap, const uint64_t* host_keys, uint32_t n) {
        if (n == 0) { if (cap) graph_valid = false; cap = 0; return AVN_OK; }
        uint32_t need = 64;
        while (need < 2 * n + 16) need <<= 1;
        hipError_t err;
        buf.ensure((size_t)need * 8, err);
        if (err != hipSuccess) { error = "hipMalloc failed"; return AVN_ERR_OOM; }
        tab = buf.as<uint64_t>();
        cap = need;
        avn_status st = stage_reserve((size_t)n * 8 + 1024);
        if (st != AVN_OK) return st;
        HIPCHK(hipMemsetAsync(tab, 0xFF, (size_t)cap * 8, stream));
        const uint64_t* d;
        if ((st = stage_in<uint64_t>(host_keys, n, &d)) != AVN_OK) return st;
        launch_hs_insert(tab, cap, d, n, stream);
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(stream));
        return AVN_OK;
    }
    avn_status existing_pairs_upload(const uint64_t* keys, size_t n) override {
        slp_world_asleep = slp_world_idle = false;
        if (n && !keys) return AVN_ERR_BAD_ARG;
        hipError_t err;
        b_pair_keys.ensure(std::max<size_t>(n, 1) * 8, err);
        if (err != hipSuccess) { error = "hipMalloc failed"; return AVN_ERR_OOM; }
        if (n) HIPCHK(hipMemcpyAsync(b_pair_keys.p, keys, n * 8, hipMemcpyHostToDevice, stream));
        n_pair_keys = (uint32_t)n;
        return rebuild_pair_set((uint32_t)n);
    }
    // (re)build the device pair set from the key list with room for `expect` keys
    avn_status rebuild_pair_set(uint32_t expect) {
        if (contact_keys_live) {
            // rows have been removed since the key list was built (contact_pairs_remove): rebuild from the live keys only
            std::vector<uint64_t> keys(h_live_keys.begin(), h_live_keys.end());
            hipError_t e2;
            b_pair_keys.ensure(std::max<size_t>(keys.size(), 1) * 8, e2);
            if (e2 != hipSuccess) { error = "hipMalloc failed"; return AVN_ERR_OOM; }
            HIPCHK(hipStreamSynchronize(bs));
            if (!keys.empty()) HIPCHK(hipMemcpy(b_pair_keys.p, keys.data(), keys.size() * 8, hipMemcpyHostToDevice));
            n_pair_keys = (uint32_t)keys.size();
            expect = std::max(expect, n_pair_keys + n_pair_keys / 2);
        }
        uint32_t need = 1024;
        while (need < 2 * (expect + 16)) need <<= 1;
        hipError_t err;
        b_pair_set.ensure((size_t)need * 8, err);
        if (err != hipSuccess) { error = "hipMalloc failed"; return AVN_ERR_OOM; }
        bp.pair_set = b_pair_set.as<uint64_t>();
        bp.pair_set_cap = need;
        HIPCHK(hipMemsetAsync(bp.pair_set, 0xFF, (size_t)need * 8, bs));
        launch_hs_insert(bp.pair_set, need, b_pair_keys.as<uint64_t>(), n_pair_keys, bs);
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(bs));
        return AVN_OK;
    }
    avn_status colliders_upload(const avn_colliders* c) override {
        slp_world_asleep = slp_world_idle = false;
        if (!have_bodies) { error = "colliders_upload before bodies_upload"; return AVN_ERR_STATE; }
        if (!c || (c->count && (!c->entity_index || !c->body || !c->shape || !c->half_extents))) { error = "colliders_upload: null array"; return AVN_ERR_BAD_ARG; }
        if (despawn_needs_bodies) { error = "colliders_upload: after avn_despawn the remaining bodies are uploaded first"; return AVN_ERR_STATE; }
        uint32_t C = c->count;
        for (uint32_t i = 0; i < C; ++i)
            if (c->body[i] < 0 || (uint32_t)c->body[i] >= dw.n_bodies) { error = "colliders_upload: body index out of range"; return AVN_ERR_BAD_ARG; }
        HIPCHK(hipStreamSynchronize(stream));
        HIPCHK(hipStreamSynchronize(stream_bp));
        bool same = slot_entity.size() == C && (C == 0 || std::memcmp(slot_entity.data(), c->entity_index, C * 4) == 0);
        std::vector<uint32_t> new_iv;
        std::vector<V> keep_min, keep_max;
        if (!same) {
            std::unordered_map<uint32_t, uint32_t> next_slot;
            next_slot.reserve(C * 2);
            for (uint32_t i = 0; i < C; ++i)
                if (!next_slot.emplace(c->entity_index[i], i).second) { error = "colliders_upload: duplicate entity_index"; return AVN_ERR_BAD_ARG; }
            // live contact rows name collider SLOTS: renumber them for the new upload order (a spawn appends, a reorder permutes); a row whose
            // collider is gone cannot be kept -- the reference pops such a pair's constraints when the collider is removed
            // (narrow_phase/mod.rs remove_collider_on), which the host must have done through avn_contact_pairs_remove / a loop restart
            if (ct.cap && !slot_entity.empty()) {
                std::vector<uint32_t> remap(slot_entity.size());
                bool identity = C >= slot_entity.size();
                for (size_t s = 0; s < slot_entity.size(); ++s) {
                    auto it = next_slot.find(slot_entity[s]);
                    remap[s] = it == next_slot.end() ? 0xFFFFFFFFu : it->second;
                    identity = identity && remap[s] == (uint32_t)s;
                }
                if (!identity) {
                    avn_status sr = stage_reserve(al(remap.size() * 4) + 1024);
                    if (sr != AVN_OK) return sr;
                    const uint32_t* d_map;
                    if ((sr = stage_in<uint32_t>(remap.data(), remap.size(), &d_map)) != AVN_OK) return sr;
                    uint32_t* d_orphans = b_misc.as<uint32_t>() + 41;
                    HIPCHK(hipMemsetAsync(d_orphans, 0, 4, stream));
                    launch_remap_row_slots<T>(ct, d_map, (uint32_t)remap.size(), 0u, d_orphans, stream);
                    uint32_t orphans = 0;
                    HIPCHK(hipMemcpyAsync(&orphans, d_orphans, 4, hipMemcpyDeviceToHost, stream));
                    HIPCHK(hipStreamSynchronize(stream));
                    if (orphans) {
                        error = "colliders_upload: " + std::to_string(orphans) + " live contact pair(s) name a collider that is no longer uploaded; remove the pairs first "
                                "(avn_contact_pairs_remove) or restart the closed loop (avn_pipeline_enable(0), upload, avn_pipeline_enable(1))";
                        return AVN_ERR_STATE;
                    }
                    launch_remap_row_slots<T>(ct, d_map, (uint32_t)remap.size(), 1u, d_orphans, stream);
                    HIPCHK(hipGetLastError());
                    HIPCHK(hipStreamSynchronize(stream));
                    graph_valid = false;
                }
            }
            // retain_mut (reference broad_phase.rs:230-279) on the current device order, then append the new ones
            std::vector<uint32_t> old_iv(bp.n_intervals);
            if (bp.n_intervals) HIPCHK(hipMemcpy(old_iv.data(), bp.iv_collider, (size_t)bp.n_intervals * 4, hipMemcpyDeviceToHost));
            std::vector<uint8_t> known(C, 0);
            // carry the ColliderAabb component of surviving colliders over to their new slot
            std::vector<V> omin(bp.n_colliders), omax(bp.n_colliders);
            if (bp.n_colliders) {
                HIPCHK(hipMemcpy(omin.data(), bp.aabb_min, (size_t)bp.n_colliders * sizeof(V), hipMemcpyDeviceToHost));
                HIPCHK(hipMemcpy(omax.data(), bp.aabb_max, (size_t)bp.n_colliders * sizeof(V), hipMemcpyDeviceToHost));
            }
            keep_min.assign(C, make4<T>(0, 0, 0, 0));
            keep_max.assign(C, make4<T>(0, 0, 0, 0));
            for (uint32_t s = 0; s < bp.n_colliders && s < slot_entity.size(); ++s) {
                auto it = next_slot.find(slot_entity[s]);
                if (it != next_slot.end()) { keep_min[it->second] = omin[s]; keep_max[it->second] = omax[s]; }
            }
            for (uint32_t iv : old_iv) {
                auto it = next_slot.find(slot_entity[iv]);
                if (it == next_slot.end()) continue;
                new_iv.push_back(it->second);
                known[it->second] = 1;
            }
            for (uint32_t i = 0; i < C; ++i)
                if (!known[i]) new_iv.push_back(i);  // add_new_aabb_intervals: appended at the END in upload order
            slot_entity.assign(c->entity_index, c->entity_index + C);
        }
        bool moved = false;
        if (C > cap_colliders) {
            size_t cc = std::max<size_t>(C, cap_colliders + cap_colliders / 2);
            GROW(b_col_info, cc, bp.col_info); GROW(b_col_he, cc, bp.col_he); GROW(b_col_spec, cc, bp.col_spec); GROW(b_col_layers, cc, bp.col_layers);
            GROW(b_aabb_min, cc, bp.aabb_min); GROW(b_aabb_max, cc, bp.aabb_max); GROW(b_iv, cc, bp.iv_collider);
            GROW(b_s_minx, cc, bp.s_minx); GROW(b_s_maxx, cc, bp.s_maxx); GROW(b_s_yz, cc + sweep_pad_records(), bp.s_yz); GROW(b_s_bb, sweep_bounds_words((uint32_t)cc), bp.s_bb); bp.s_bb2 = bp.s_bb + sweep_bounds_level2_offset((uint32_t)cc); GROW(b_s_end, cc, bp.s_end);
            GROW(b_s_info, cc, bp.s_info); GROW(b_s_flags, cc, bp.s_flags);
            Key* dummy_k; uint32_t* dummy_u;
            GROW(b_keys_a, cc, dummy_k); GROW(b_keys_b, cc, dummy_k); GROW(b_vals_a, cc, dummy_u); GROW(b_vals_b, cc, dummy_u);
            GROW(b_hist, (size_t)256 * radix_blocks((uint32_t)cc) + 256, dummy_u);
            GROW(b_block_sums, std::max<size_t>(scan_block_sums_needed(256 * radix_blocks((uint32_t)cc)), scan_block_sums_needed((uint32_t)cc * sweep_count_slots())) + 16, dummy_u);
            HIPCHK(hipMemset(b_block_sums.p, 0, b_block_sums.cap));   // the one-launch scan's state: zero once, self-cleaning afterwards (avn_scan.h)
            GROW(b_counts, cc * sweep_count_slots() + 1, dummy_u); GROW(b_offsets, cc * sweep_count_slots() + 1, dummy_u);
            GROW(b_sweep_hits, sweep_hit_words((uint32_t)cc), sweep_scratch.hits);
            {   // long-interval chunks: every interval may need one slot, plus room for the chunks of scene-spanning ones
                size_t lcap = cc + 65536;
                if (const char* e = avn_env("AVN_SWEEP_LONG_CAP")) lcap = std::max<size_t>(8, (size_t)strtoull(e, nullptr, 10));   // (tests: force the grow-and-retry path)
                uint8_t* dummy_b;
                GROW(b_long_items, lcap * sweep_long_item_bytes(), dummy_b);
                GROW(b_long_counts, lcap, dummy_u); GROW(b_long_off, lcap, dummy_u);
                sweep_scratch.long_items = b_long_items.p; sweep_scratch.long_counts = b_long_counts.as<uint32_t>();
                sweep_scratch.long_off = b_long_off.as<uint32_t>(); sweep_scratch.long_cap = (uint32_t)lcap;
            }
            cap_colliders = (uint32_t)cc;
        }
        bp.n_colliders = C;
        if (!same) {
            bp.n_intervals = (uint32_t)new_iv.size();
            if (!new_iv.empty()) HIPCHK(hipMemcpy(bp.iv_collider, new_iv.data(), new_iv.size() * 4, hipMemcpyHostToDevice));
            if (C) {
                HIPCHK(hipMemcpy(bp.aabb_min, keep_min.data(), (size_t)C * sizeof(V), hipMemcpyHostToDevice));
                HIPCHK(hipMemcpy(bp.aabb_max, keep_max.data(), (size_t)C * sizeof(V), hipMemcpyHostToDevice));
            }
        }
        avn_status st = stage_reserve(al(4 * (size_t)C) * 4 + al(C) * 2 + al(sizeof(T) * 3 * C) + al(sizeof(T) * C) * 2 + 4096);
        if (st != AVN_OK) return st;
        ColliderStage<T> s;
        std::memset(&s, 0, sizeof s);
        SIN(entity, c->entity_index, C, uint32_t); SIN(body, c->body, C, int32_t); SIN(shape, c->shape, C, uint8_t);
        SIN(half_extents, c->half_extents, 3 * (size_t)C, T); SIN(memberships, c->memberships, C, uint32_t); SIN(filters, c->filters, C, uint32_t);
        SIN(cflags, c->collider_flags, C, uint8_t); SIN(collision_margin, c->collision_margin, C, T); SIN(speculative_margin, c->speculative_margin, C, T);
        launch_pack_colliders<T>(bp, s, stream);
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(stream));
        entity_slot.clear();
        entity_slot.reserve((size_t)C * 2);
        h_col_body.assign(c->body, c->body + C);
        for (uint32_t i = 0; i < C; ++i) entity_slot.emplace(c->entity_index[i], i);
        {   // Friction / Restitution defaults until collider_materials_upload: DefaultFriction 0.5, DefaultRestitution 0, Average
            hipError_t err;
            b_col_mat.ensure(std::max<size_t>(C, 1) * sizeof(V), err);
            if (err != hipSuccess) { error = "hipMalloc failed"; return AVN_ERR_OOM; }
            ct.col_mat = b_col_mat.as<V>();
            std::vector<V> mats(C, make4<T>(T(0.5), T(0), bits_to_scalar((uint32_t)AVN_COMBINE_AVERAGE | ((uint32_t)AVN_COMBINE_AVERAGE << 8), T(0)), T(0)));
            if (C) HIPCHK(hipMemcpy(ct.col_mat, mats.data(), (size_t)C * sizeof(V), hipMemcpyHostToDevice));
            materials_restitution = false;
        }
        have_colliders = true;
        despawn_needs_colliders = false;
        { avn_status sh = hs_on_colliders_upload(c); if (sh != AVN_OK) return sh; }
        { avn_status sh = hk_on_colliders_upload(c); if (sh != AVN_OK) return sh; }
        bp.col_lpos = nullptr; bp.col_lrot = nullptr; tf_any = false;   // (every collider sits on its body again until avn_collider_transforms_upload says otherwise)
        if (slp_on) {   // colliders spawned inside the loop join the island manager's RigidBodyColliders lists (upload order = Add order)
            for (uint32_t i = 0; i < C; ++i) {
                if (isl.has_collider(c->entity_index[i])) continue;
                const uint32_t b = (uint32_t)c->body[i];
                const avn_status si = isl.collider_add(c->entity_index[i], (b < h_rb_type.size() && slp_node(b)) ? b : IslandManager::NONE);
                if (si != AVN_OK) return slp_fail(si);
            }
        }
        if (pipe_dev && !same) { avn_status se = pg_upload_ent2slot(); if (se != AVN_OK) return se; }   // k_pg_add_pairs turns the pairs' collider entities into slots
        return AVN_OK;
    }
