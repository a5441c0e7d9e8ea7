// world/despawn.hpp -- fragment of the body of `template <class T> struct World` (avn_world.hip includes it inside the class):
// avn_despawn -- bodies / colliders leave the device closed loop without restarting it (include/avian_mi355x.h).
//
// Reference: collision/narrow_phase/mod.rs:399-457 remove_collider, :459-560 remove_body_on / remove_collider_on; contact_types/contact_graph.rs:
// 641-700 remove_collider_with; data_structures/stable_graph.rs:251-315; dynamics/solver/islands/mod.rs:1336-1400 BodyIslandNode::on_remove.
//
// What decides bits here is ORDER: the pops of a removed collider's touching pairs happen in the ContactGraph's edge-list order (outgoing edges
// newest first, then incoming newest first), and swap_remove makes the colour lists remember it.  The device keeps an insertion stamp per row
// (PG::seq, written by k_pg_add_pairs): "newest first" is descending stamp.  One scan over the rows finds the removed colliders' edges, the host
// orders them (a despawn is a rare, host-initiated event: tens to thousands of edges), and the pops go through the same op pipeline as the
// status loop's (pg_apply_ops: exact swap_remove replay).  Then the rows are cleared, their PairKeys tombstoned, their ids merged into the
// sorted free list, and every body index the library holds is renumbered for the host's compacted arrays.
    bool despawn_needs_bodies = false, despawn_needs_colliders = false;
    bool despawn_needs_joints = false;   // joints left with the last avn_despawn: the device arrays still hold the old set until avn_joints_upload brings the remaining one
    uint32_t despawn_expected_joints = 0;
    bool despawn_broken = false;   // an avn_despawn failed after its first mutation: ContactGraph / islands are half-updated, only a restart of the closed loop clears it
    uint32_t despawn_expected_bodies = 0;
    DevBuf b_dsp_a, b_dsp_b;

    avn_status despawn(const avn_despawn_list* d) override {
        slp_world_asleep = slp_world_idle = false;
        if (!d || (d->struct_size != sizeof(avn_despawn_list) && d->struct_size != AVN_DESPAWN_LIST_SIZE_R4) || (d->n_colliders && !d->collider_entities) || (d->n_bodies && !d->bodies)) { error = "despawn: bad argument"; return AVN_ERR_BAD_ARG; }
        if (d->struct_size == sizeof(avn_despawn_list) && d->n_joints && !d->joints) { error = "despawn: bad argument"; return AVN_ERR_BAD_ARG; }
        if (!pipe_on || !pipe_dev) { error = "despawn: needs the device closed loop (avn_pipeline_enable(1))"; return AVN_ERR_STATE; }
        if (dsh_on) { error = "despawn: not inside a sharded closed loop (avn_dshard_enable(NULL), despawn on every rank, enable again)"; return AVN_ERR_STATE; }
        if (despawn_needs_bodies || despawn_needs_colliders) { error = "despawn: the previous avn_despawn is still waiting for avn_bodies_upload / avn_colliders_upload"; return AVN_ERR_STATE; }
        if (despawn_broken) { error = "despawn: an earlier avn_despawn failed half-way; restart the closed loop (avn_pipeline_enable(0), uploads, avn_pipeline_enable(1))"; return AVN_ERR_STATE; }
        pg_new_ids_count = 0; last_timers.pair_count = 0;   // (the last step's new-pair ids may name rows that leave now: avn_pipeline_new_pair_ids_get reports an empty list until the next step)
        const avn_status ds = despawn_body(d);
        if (ds != AVN_OK && despawn_mutating) despawn_broken = true;
        despawn_mutating = false;
        return ds;
    }
    bool despawn_mutating = false;
    avn_status despawn_body(const avn_despawn_list* d) {
        const uint32_t n_old = dw.n_bodies, C = bp.n_colliders;
        std::vector<uint8_t> gone_body(n_old, 0);
        for (uint32_t i = 0; i < d->n_bodies; ++i) {
            const uint32_t b = d->bodies[i];
            if (b >= n_old || gone_body[b]) { error = "despawn: body index out of range or listed twice"; return AVN_ERR_BAD_ARG; }
            gone_body[b] = 1;
        }
        for (uint32_t i = 0; i < d->n_colliders; ++i) if (!entity_slot.count(d->collider_entities[i])) { error = "despawn: unknown collider"; return AVN_ERR_BAD_ARG; }
        const uint32_t n_gone_joints = d->struct_size == sizeof(avn_despawn_list) ? d->n_joints : 0u;
        std::vector<uint8_t> gone_joint(h_j_body1.size(), 0);
        for (uint32_t i = 0; i < n_gone_joints; ++i) {
            const uint32_t j = d->joints[i];
            if (j >= gone_joint.size() || gone_joint[j]) { error = "despawn: joint index out of range or listed twice"; return AVN_ERR_BAD_ARG; }
            gone_joint[j] = 1;
        }
        for (size_t j = 0; j < h_j_body1.size(); ++j)
            if (!gone_joint[j] && (gone_body[(uint32_t)h_j_body1[j]] || gone_body[(uint32_t)h_j_body2[j]])) { error = "despawn: a joint names a despawned body: list it in avn_despawn_list::joints"; return AVN_ERR_STATE; }
        HIPCHK(hipStreamSynchronize(stream)); HIPCHK(hipStreamSynchronize(stream_bp));
        avn_status st = pg_error_check();
        if (st != AVN_OK) return st;
        // ---- the removal order: single colliders first, then every body's colliders in slot (= upload = RigidBodyColliders) order ----
        struct Unit { uint32_t body; std::vector<uint32_t> slots; };   // body = NONE: a collider despawned on its own
        std::vector<Unit> units;
        std::vector<uint32_t> rm_rank(C, PG_NONE);
        uint32_t n_rm = 0;
        for (uint32_t i = 0; i < d->n_colliders; ++i) {
            const uint32_t s_ = entity_slot[d->collider_entities[i]];
            if (rm_rank[s_] != PG_NONE) continue;   // (listed twice: the second remove_collider finds no node)
            rm_rank[s_] = n_rm++;
            units.push_back(Unit{IslandManager::NONE, {s_}});
        }
        {
            std::vector<std::vector<uint32_t>> of_body(d->n_bodies);
            std::unordered_map<uint32_t, uint32_t> unit_of;
            for (uint32_t i = 0; i < d->n_bodies; ++i) unit_of.emplace(d->bodies[i], i);
            for (uint32_t s_ = 0; s_ < C; ++s_) { auto it = unit_of.find((uint32_t)h_col_body[s_]); if (it != unit_of.end()) of_body[it->second].push_back(s_); }
            for (uint32_t i = 0; i < d->n_bodies; ++i) {
                Unit u{d->bodies[i], {}};
                for (uint32_t s_ : of_body[i]) if (rm_rank[s_] == PG_NONE) { rm_rank[s_] = n_rm++; u.slots.push_back(s_); }
                units.push_back(std::move(u));
            }
        }
        // ---- the edges of the removed colliders, from the device ----
        const uint32_t n_rows = pgm_next_id;
        std::vector<PGEdgeRec> recs;
        if (n_rows && n_rm) {
            hipError_t e;
            b_dsp_a.ensure((size_t)C * 4 + 64, e);
            if (e != hipSuccess) { error = "hipMalloc failed"; return AVN_ERR_OOM; }
            HIPCHK(hipMemcpy(b_dsp_a.p, rm_rank.data(), (size_t)C * 4, hipMemcpyHostToDevice));
            uint32_t cap = 4096 + 64 * n_rm;
            for (int attempt = 0; attempt < 2; ++attempt) {
                b_dsp_b.ensure((size_t)cap * sizeof(PGEdgeRec), e);
                if (e != hipSuccess) { error = "hipMalloc failed"; return AVN_ERR_OOM; }
                launch_pg_collect_edges<T>(pg, ct, n_rows, b_dsp_a.as<uint32_t>(), b_dsp_b.as<PGEdgeRec>(), cap, stream);
                uint32_t found = 0;
                HIPCHK(hipMemcpyAsync(&found, pg.ctr + PGC_COLLECT, 4, hipMemcpyDeviceToHost, stream));
                HIPCHK(hipStreamSynchronize(stream));
                if (found <= cap) { recs.resize(found); if (found) HIPCHK(hipMemcpy(recs.data(), b_dsp_b.p, (size_t)found * sizeof(PGEdgeRec), hipMemcpyDeviceToHost)); break; }
                cap = found + 64;
                if (attempt == 1) { error = "despawn: edge collection overflowed twice"; return AVN_ERR_CAPACITY; }
            }
        }
        // per removed collider: outgoing edges (it is collider1) by descending stamp, then incoming edges by descending stamp
        std::vector<std::vector<uint32_t>> out_of(n_rm), in_of(n_rm);   // indices into recs
        for (uint32_t k = 0; k < recs.size(); ++k) {
            const PGEdgeRec& r = recs[k];
            if (rm_rank[r.slot1] != PG_NONE) out_of[rm_rank[r.slot1]].push_back(k);
            if (rm_rank[r.slot2] != PG_NONE) in_of[rm_rank[r.slot2]].push_back(k);
        }
        auto stamp_of = [&](uint32_t k) { return ((uint64_t)recs[k].seq_hi << 32) | recs[k].seq_lo; };
        auto newest_first = [&](uint32_t a, uint32_t b) { return stamp_of(a) > stamp_of(b); };
        for (auto& v : out_of) std::sort(v.begin(), v.end(), newest_first);
        for (auto& v : in_of) std::sort(v.begin(), v.end(), newest_first);
        // (everything above only reads; from here on a failure leaves the bookkeeping half-updated: despawn() marks the loop as broken)
        despawn_mutating = true;
        // ---- walk the units in order; pops accumulate into one op batch until a WakeIslands with an effect has to run in between ----
        std::vector<uint8_t> edge_done(recs.size(), 0);
        std::unordered_map<uint32_t, uint32_t> rec_of;   // contact id -> record (built when a wake makes it necessary)
        std::vector<uint32_t> pops, removed;
        double host_ms = 0;
        auto flush_pops = [&]() -> avn_status {
            if (pops.empty()) return AVN_OK;
            avn_status s2 = stage_reserve(pops.size() * 8 + 1024);
            if (s2 != AVN_OK) return s2;
            std::vector<uint32_t> kinds(pops.size(), 2u /* PG_KIND_POP */);
            const uint32_t *d_c, *d_k;
            if ((s2 = stage_in<uint32_t>(pops.data(), pops.size(), &d_c)) != AVN_OK || (s2 = stage_in<uint32_t>(kinds.data(), kinds.size(), &d_k)) != AVN_OK) return s2;
            s2 = pg_apply_ops((uint32_t)pops.size(), 0u, 0u, d_c, d_k, host_ms);
            pops.clear();
            return s2;
        };
        // ---- joints leave first: remove_joint_from_graph per joint (joint_graph/plugin.rs:163-194), a sleeping island is woken (its manifolds are pushed back) ----
        if (n_gone_joints) {
            if (slp_on)
                for (uint32_t i = 0; i < n_gone_joints; ++i) {
                    if ((st = isl.joint_remove(d->joints[i])) != AVN_OK) return slp_fail(st);
                    if (!isl.pushed().empty() || !isl.bodies_woken().empty() || !isl.pairs_woken().empty()) {
                        if ((st = sleeping_apply_result(false, host_ms)) != AVN_OK) return st;
                    }
                }
            // the joint set shrinks; the device arrays are rewritten by the upload the host owes (avn_joints_upload packs every array, prepare_joints the rest)
            std::vector<uint32_t> jmap(gone_joint.size(), IslandManager::NONE);
            uint32_t nj = 0;
            for (size_t j = 0; j < gone_joint.size(); ++j) if (!gone_joint[j]) jmap[j] = nj++;
            auto keep = [&](auto& v) { if (v.size() != gone_joint.size()) return; size_t k = 0; for (size_t j = 0; j < gone_joint.size(); ++j) if (!gone_joint[j]) v[k++] = v[j]; v.resize(k); };
            keep(h_j_body1); keep(h_j_body2); keep(h_j_type); keep(h_j_damped); keep(h_j_collision_disabled);
            if (slp_on && (st = isl.renumber_joints(jmap.data(), (uint32_t)jmap.size())) != AVN_OK) return slp_fail(st);
            dw.n_joints = nj;
            joint_schedule_dirty = true; groups_dirty = true; graph_valid = false;
            despawn_needs_joints = true; despawn_expected_joints = nj;
        }
        for (const Unit& u : units) {
            uint32_t island = IslandManager::NONE;
            const uint32_t owner = u.body != IslandManager::NONE ? u.body : (u.slots.empty() ? IslandManager::NONE : (uint32_t)h_col_body[u.slots[0]]);
            if (slp_on && owner != IslandManager::NONE && isl.body_has_node(owner)) island = isl.island_of(owner);
            for (uint32_t s_ : u.slots) {
                const uint32_t r = rm_rank[s_];
                for (int dir = 0; dir < 2; ++dir)
                    for (uint32_t k : (dir == 0 ? out_of[r] : in_of[r])) {
                        if (edge_done[k]) continue;
                        edge_done[k] = 1;
                        const PGEdgeRec& e = recs[k];
                        if ((e.flags & AVN_CP_TOUCHING) && e.color != PG_NONE) pops.push_back(e.cid);   // constraint_handles.len() = 1 (convex pairs)
                        if (slp_on) { if ((st = isl.remove_collider_edge(e.cid)) != AVN_OK) return slp_fail(st); }
                        removed.push_back(e.cid);
                    }
                if (slp_on) isl.collider_forget(slot_entity[s_]);
            }
            if (slp_on) {
                if (u.body != IslandManager::NONE) { if ((st = isl.body_remove(u.body, false)) != AVN_OK) return slp_fail(st); }
                if ((st = isl.wake_island(island)) != AVN_OK) return slp_fail(st);   // the queued WakeIslands([island]): a no-op unless it sleeps
                if (!isl.pushed().empty() || !isl.bodies_woken().empty() || !isl.pairs_woken().empty()) {
                    if ((st = flush_pops()) != AVN_OK) return st;
                    // (the woken island's pairs are back in the ConstraintGraph: an edge record collected before the wake that belongs to a LATER
                    //  unit of this despawn has a handle now -- its pop reads the colour on the device, the record only says "pop me")
                    if (rec_of.empty()) for (uint32_t k = 0; k < recs.size(); ++k) rec_of.emplace(recs[k].cid, k);
                    for (uint32_t cid : isl.pushed()) { auto it = rec_of.find(cid); if (it != rec_of.end()) recs[it->second].color = 0u; }
                    if ((st = sleeping_apply_result(false, host_ms)) != AVN_OK) return st;
                }
            }
        }
        if ((st = flush_pops()) != AVN_OK) return st;
        // ---- the rows leave: cleared, keys tombstoned, ids back into the sorted free list ----
        if (!removed.empty()) {
            std::vector<uint32_t> ids(removed);
            std::sort(ids.begin(), ids.end());
            const uint32_t n_rem = (uint32_t)ids.size();
            HIPCHK(hipStreamSynchronize(stream));
            if ((st = stage_reserve((size_t)n_rem * 4 + 1024)) != AVN_OK) return st;
            const uint32_t* d_ids;
            if ((st = stage_in<uint32_t>(ids.data(), n_rem, &d_ids)) != AVN_OK) return st;
            launch_pg_remove_list<T>(pg, ct, bp, d_ids, n_rem, stream);
            launch_pg_merge_free(pg, pgm_head, pgm_n_free, n_rem, stream);
            std::swap(b_pg_free_a.p, b_pg_free_b.p); std::swap(b_pg_free_a.cap, b_pg_free_b.cap);
            std::swap(pg.free_ids, pg.free_alt);
            pgm_head = 0; pgm_n_free += n_rem; pgm_live -= n_rem; pgm_tomb += n_rem;
            pipe_stats.pairs_removed += n_rem;
            HIPCHK(hipGetLastError());
            HIPCHK(hipStreamSynchronize(stream));
        }
        // ---- renumbering: body i -> i - #removed below i, in everything that names a body ----
        if (d->n_bodies) {
            std::vector<uint32_t> new_index(n_old + 1u, PG_NONE);
            uint32_t n_new = 0;
            for (uint32_t b = 0; b < n_old; ++b) if (!gone_body[b]) new_index[b] = n_new++;
            new_index[n_old] = n_new;   // ("no body" keys of per-step scratch map to the new count)
            hipError_t e;
            b_dsp_a.ensure(((size_t)n_old + 1) * 4 + 64, e);
            if (e != hipSuccess) { error = "hipMalloc failed"; return AVN_ERR_OOM; }
            HIPCHK(hipMemcpy(b_dsp_a.p, new_index.data(), ((size_t)n_old + 1) * 4, hipMemcpyHostToDevice));
            const uint32_t* d_map = b_dsp_a.as<uint32_t>();
            launch_pg_renumber_rows<T>(pg, ct, pgm_next_id, d_map, stream);
            if (dw.n_joints) launch_renumber_int2(dw.j_bodies, dw.n_joints, d_map, stream);
            // per-body arrays that are STATE: the colour masks, and with sleeping on the SleepTimers and the per-body thresholds
            auto compact32 = [&](DevBuf& buf, size_t words_needed) -> avn_status {
                if (!buf.p) return AVN_OK;
                b_dsp_b.ensure(std::max(buf.cap, words_needed * 4), e);
                if (e != hipSuccess) { error = "hipMalloc failed"; return AVN_ERR_OOM; }
                HIPCHK(hipMemsetAsync(b_dsp_b.p, 0, buf.cap, stream));
                launch_compact_u32(buf.as<uint32_t>(), b_dsp_b.as<uint32_t>(), d_map, n_old, stream);
                HIPCHK(hipMemcpyAsync(buf.p, b_dsp_b.p, buf.cap, hipMemcpyDeviceToDevice, stream));
                return AVN_OK;
            };
            if ((st = compact32(b_pg_bcol, n_old + 1u)) != AVN_OK) return st;
            if (slp_on) {
                if ((st = compact32(b_slp_timer, n_old)) != AVN_OK) return st;
                if (slp_k.body_lin && (st = compact32(b_slp_lin, n_old)) != AVN_OK) return st;
                if (slp_k.body_ang && (st = compact32(b_slp_ang, n_old)) != AVN_OK) return st;
                if (slp_k.body_disabled) {
                    b_dsp_b.ensure(std::max<size_t>(b_slp_dis.cap, n_old), e);
                    if (e != hipSuccess) { error = "hipMalloc failed"; return AVN_ERR_OOM; }
                    HIPCHK(hipMemsetAsync(b_dsp_b.p, 0, b_slp_dis.cap, stream));
                    launch_compact_u8(b_slp_dis.as<uint8_t>(), b_dsp_b.as<uint8_t>(), d_map, n_old, stream);
                    HIPCHK(hipMemcpyAsync(b_slp_dis.p, b_dsp_b.p, b_slp_dis.cap, hipMemcpyDeviceToDevice, stream));
                }
                slp_bodies = n_new;
                if ((st = isl.renumber_bodies(new_index.data(), n_old)) != AVN_OK) return slp_fail(st);
            }
            HIPCHK(hipGetLastError());
            HIPCHK(hipStreamSynchronize(stream));
            // host mirrors
            auto compact_host = [&](std::vector<uint8_t>& v) { if (v.size() < n_old) return; std::vector<uint8_t> o; o.reserve(n_new); for (uint32_t b = 0; b < n_old; ++b) if (!gone_body[b]) o.push_back(v[b]); v.swap(o); };
            compact_host(h_body_has_sb); compact_host(h_rb_type); compact_host(h_body_flags);
            for (size_t j = 0; j < h_j_body1.size(); ++j) { h_j_body1[j] = (int32_t)new_index[(uint32_t)h_j_body1[j]]; h_j_body2[j] = (int32_t)new_index[(uint32_t)h_j_body2[j]]; }
            if (!h_j_body1.empty()) {   // the joint-disabled body pairs are keyed by body index
                std::vector<uint64_t> disabled;
                for (size_t j = 0; j < h_j_body1.size(); ++j)
                    if (j < h_j_collision_disabled.size() && h_j_collision_disabled[j]) { const uint32_t a = (uint32_t)h_j_body1[j], b = (uint32_t)h_j_body2[j]; disabled.push_back(a < b ? ((uint64_t)a << 32) | b : ((uint64_t)b << 32) | a); }
                if ((st = build_hash_set(b_disabled_set, bp.disabled_set, bp.disabled_cap, disabled.data(), (uint32_t)disabled.size())) != AVN_OK) return st;
            }
            for (uint32_t s_ = 0; s_ < h_col_body.size(); ++s_) h_col_body[s_] = (h_col_body[s_] >= 0 && !gone_body[(uint32_t)h_col_body[s_]]) ? (int32_t)new_index[(uint32_t)h_col_body[s_]] : -1;
            dw.lacc_l = dw.lacc_a = nullptr;   // (header: a despawn of bodies drops the local accelerations; the host uploads them again for what remains)
            joint_schedule_dirty = true; groups_dirty = true; incidence_dirty = true; graph_valid = false;
            isl_labels_step_valid = false; island_backoff = 0;
            despawn_needs_bodies = true; despawn_expected_bodies = n_new;
        }
        for (uint32_t s_ = 0; s_ < C; ++s_) if (rm_rank[s_] != PG_NONE) h_col_body[s_] = -1;
        despawn_needs_colliders = true;
        return AVN_OK;
    }
