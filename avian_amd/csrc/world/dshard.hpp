// world/dshard.hpp -- fragment of the body of `template <class T> struct World` (avn_world.hip includes it inside the class):
// the DEVICE closed loop sharded by islands (include/avian_mi355x.h: avn_dshard_*).
//
// Every rank runs the whole integer / geometry front of the step on ALL bodies (replicated: equal inputs, equal code, equal ids / colours / list positions), simulates
// only the bodies it owns (the others carry AVN_BODY_FOREIGN: no SolverBody), solves only its share of the colour lists (k_pg_local_lists: a stable restriction), and
// receives the other ranks' bodies after every step (one all-gather of 4 records per body).
    bool dsh_on = false;
    uint32_t dsh_ranks = 1, dsh_rank = 0, dsh_exchanges = 0, dsh_own_manifolds = 0, dsh_global_manifolds = 0;
    std::vector<int32_t> dsh_owner;                 // per body
    std::vector<uint32_t> dsh_list, dsh_off;        // bodies by owner (ascending inside a rank), dsh_off[r] .. dsh_off[r + 1]
    uint32_t dsh_max_own = 0;
    bool dsh_island_enabled_before = true;
    DevBuf b_dsh_owner, b_dsh_list, b_dsh_send, b_dsh_recv, b_dsh_local, b_dsh_cnt;
    avn_status dsh_apply_flags() {   // FOREIGN flags on the device + the host's "has a SolverBody" mirror
        launch_dsh_set_foreign<T>(dw, dsh_on ? b_dsh_owner.as<int32_t>() : nullptr, dsh_rank, stream); ++launches;
        HIPCHK(hipGetLastError());
        for (uint32_t b = 0; b < dw.n_bodies && b < h_body_has_sb.size(); ++b) {
            const bool foreign = dsh_on && dsh_owner[b] >= 0 && (uint32_t)dsh_owner[b] != dsh_rank;
            h_body_has_sb[b] = h_rb_type[b] != AVN_RB_STATIC && !(h_body_flags[b] & (AVN_BODY_SLEEPING | AVN_BODY_DISABLED)) && !foreign;
        }
        joint_schedule_dirty = true; groups_dirty = true; incidence_dirty = true; graph_valid = false;
        bodies_prepared_early = false;
        return AVN_OK;
    }
    avn_status dshard_enable(const avn_dshard_config* c) override {
        HIPCHK(hipStreamSynchronize(stream));
        if (!c) {
            if (!dsh_on) return AVN_OK;
            dsh_on = false; island_enabled = dsh_island_enabled_before;
            return dsh_apply_flags();
        }
        if (c->struct_size != sizeof(avn_dshard_config) || !c->n_ranks || c->rank >= c->n_ranks || !c->body_owner) { error = "dshard_enable: bad argument"; return AVN_ERR_BAD_ARG; }
        if (!pipe_on || !pipe_dev) { error = "dshard_enable: needs the device closed loop (avn_pipeline_enable(1))"; return AVN_ERR_STATE; }
        if (slp_on) { error = "dshard_enable: not combined with avn_sleeping_enable (the island manager is per world)"; return AVN_ERR_STATE; }
        if (pgm_next_id) { error = "dshard_enable: enable it before the first step of the closed loop (the solver's handle lists are cut when they change)"; return AVN_ERR_STATE; }
        const uint32_t n = dw.n_bodies;
        if (h_rb_type.size() != n) { error = "dshard_enable: upload bodies first"; return AVN_ERR_STATE; }
        for (uint32_t b = 0; b < n; ++b) {
            const int32_t o = c->body_owner[b];
            if (o >= (int32_t)c->n_ranks) { error = "dshard_enable: body_owner names a rank that does not exist"; return AVN_ERR_BAD_ARG; }
            if (o < 0 && h_rb_type[b] != AVN_RB_STATIC) { error = "dshard_enable: every non-static body needs an owner"; return AVN_ERR_BAD_ARG; }
        }
        dsh_owner.assign(c->body_owner, c->body_owner + n);
        dsh_ranks = c->n_ranks; dsh_rank = c->rank;
        dsh_off.assign(dsh_ranks + 1, 0u);
        for (uint32_t b = 0; b < n; ++b) if (dsh_owner[b] >= 0) ++dsh_off[dsh_owner[b] + 1];
        dsh_max_own = 0;
        for (uint32_t r = 0; r < dsh_ranks; ++r) { dsh_max_own = std::max(dsh_max_own, dsh_off[r + 1]); dsh_off[r + 1] += dsh_off[r]; }
        dsh_list.resize(dsh_off[dsh_ranks]);
        { std::vector<uint32_t> cur(dsh_off.begin(), dsh_off.end() - 1); for (uint32_t b = 0; b < n; ++b) if (dsh_owner[b] >= 0) dsh_list[cur[dsh_owner[b]]++] = b; }
        hipError_t err;
        b_dsh_owner.ensure(std::max<size_t>(n, 1) * 4, err); if (err != hipSuccess) { error = "hipMalloc failed"; return AVN_ERR_OOM; }
        b_dsh_list.ensure(std::max<size_t>(dsh_list.size(), 1) * 4, err); if (err != hipSuccess) { error = "hipMalloc failed"; return AVN_ERR_OOM; }
        b_dsh_send.ensure(std::max<size_t>(dsh_max_own, 1) * 4 * sizeof(V), err); if (err != hipSuccess) { error = "hipMalloc failed"; return AVN_ERR_OOM; }
        b_dsh_recv.ensure((size_t)dsh_ranks * std::max<size_t>(dsh_max_own, 1) * 4 * sizeof(V), err); if (err != hipSuccess) { error = "hipMalloc failed"; return AVN_ERR_OOM; }
        HIPCHK(hipMemcpy(b_dsh_owner.p, dsh_owner.data(), (size_t)n * 4, hipMemcpyHostToDevice));
        if (!dsh_list.empty()) HIPCHK(hipMemcpy(b_dsh_list.p, dsh_list.data(), dsh_list.size() * 4, hipMemcpyHostToDevice));
        dsh_on = true; dsh_exchanges = 0;
        dsh_island_enabled_before = island_enabled; island_enabled = false;   // (the island-block builder groups by labels of the whole world: the sharded solver runs the colour launches, which give the same bits)
        return dsh_apply_flags();
    }
    // the local lists live next to PG::lists, same stride
    avn_status dsh_local_lists(uint32_t n_ops) {
        hipError_t err;
        if (b_dsh_local.ensure((size_t)AVN_GRAPH_COLOR_COUNT * pg.list_stride * 4, err)) graph_valid = false;
        if (err != hipSuccess) { error = "hipMalloc failed"; return AVN_ERR_OOM; }
        // chunks of 2 048 entries: a list is at most its length before the batch + the batch's ops long (the exact lengths are read on the device)
        uint32_t longest = 0;
        for (int c = 0; c < AVN_GRAPH_COLOR_COUNT; ++c) longest = std::max(longest, pgm_len[c]);
        const uint32_t n_chunks = std::max(1u, (longest + n_ops + 2047u) / 2048u);
        b_dsh_cnt.ensure((size_t)AVN_GRAPH_COLOR_COUNT * n_chunks * 4, err);
        if (err != hipSuccess) { error = "hipMalloc failed"; return AVN_ERR_OOM; }
        launch_pg_local_lists(pg, ct.meta, b_dsh_owner.as<int32_t>(), dsh_rank, b_dsh_local.as<uint32_t>(), b_dsh_cnt.as<uint32_t>(), n_chunks, stream); launches += 2;
        HIPCHK(hipGetLastError());
        return AVN_OK;
    }
    PG dsh_pg() const { PG l = pg; l.lists = b_dsh_local.as<uint32_t>(); return l; }
    uint32_t dsh_own_count() const { return dsh_off[dsh_rank + 1] - dsh_off[dsh_rank]; }
    avn_status dshard_bodies_pack(void* out, size_t cap, size_t* bytes) override {
        if (!dsh_on) { error = "dshard_bodies_pack: avn_dshard_enable first"; return AVN_ERR_STATE; }
        const uint32_t n = dsh_own_count();
        const size_t need = (size_t)n * 4 * sizeof(V);
        if (bytes) *bytes = need;
        if (!out || cap < need) { error = "dshard_bodies_pack: the buffer is too small"; return AVN_ERR_BAD_ARG; }
        launch_dsh_pack<T>(dw, b_dsh_list.as<uint32_t>() + dsh_off[dsh_rank], n, b_dsh_send.p, stream); ++launches;
        HIPCHK(hipGetLastError());
        if (need) HIPCHK(hipMemcpyAsync(out, b_dsh_send.p, need, hipMemcpyDeviceToHost, stream));
        HIPCHK(hipStreamSynchronize(stream));
        return AVN_OK;
    }
    avn_status dshard_bodies_unpack(uint32_t from, const void* in, size_t bytes) override {
        if (!dsh_on) { error = "dshard_bodies_unpack: avn_dshard_enable first"; return AVN_ERR_STATE; }
        if (from >= dsh_ranks || from == dsh_rank) { error = "dshard_bodies_unpack: another rank of the shard"; return AVN_ERR_BAD_ARG; }
        const uint32_t n = dsh_off[from + 1] - dsh_off[from];
        if (bytes != (size_t)n * 4 * sizeof(V) || (n && !in)) { error = "dshard_bodies_unpack: the byte count is not that rank's bodies x 16 scalars"; return AVN_ERR_BAD_ARG; }
        if (!n) return AVN_OK;
        HIPCHK(hipMemcpyAsync(b_dsh_recv.as<V>() + (size_t)from * dsh_max_own * 4, in, bytes, hipMemcpyHostToDevice, stream));
        launch_dsh_unpack<T>(dw, b_dsh_list.as<uint32_t>() + dsh_off[from], n, b_dsh_recv.as<V>() + (size_t)from * dsh_max_own * 4, stream); ++launches;
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(stream));
        return AVN_OK;
    }
    // inside avn_step, behind the write-back, when the library holds a communicator: pack -> ncclAllGather (equal blocks of the largest rank's size) -> unpack, all on the world's stream
    avn_status dsh_exchange_in_step() {
        if (!dsh_on || !comm.handle) return AVN_OK;
        if ((uint32_t)comm.n_ranks != dsh_ranks || (uint32_t)comm.rank != dsh_rank) { error = "avn_step: the communicator's ranks are not the shard's (avn_comm_init / avn_dshard_enable)"; return AVN_ERR_STATE; }
        const size_t block = (size_t)std::max(dsh_max_own, 1u) * 4 * sizeof(V);
        launch_dsh_pack<T>(dw, b_dsh_list.as<uint32_t>() + dsh_off[dsh_rank], dsh_own_count(), b_dsh_send.p, stream); ++launches;
        avn_status st = comm.all_gather(b_dsh_send.p, b_dsh_recv.p, block, stream, error);
        if (st != AVN_OK) return st;
        for (uint32_t r = 0; r < dsh_ranks; ++r) {
            if (r == dsh_rank) continue;
            launch_dsh_unpack<T>(dw, b_dsh_list.as<uint32_t>() + dsh_off[r], dsh_off[r + 1] - dsh_off[r], b_dsh_recv.as<V>() + (size_t)r * std::max(dsh_max_own, 1u) * 4, stream); ++launches;
        }
        HIPCHK(hipGetLastError());
        ++dsh_exchanges;
        return AVN_OK;
    }
    avn_status dshard_stats_get(avn_dshard_stats* o) override {
        if (!o) return AVN_ERR_BAD_ARG;
        std::memset(o, 0, sizeof *o);
        if (!dsh_on) return AVN_OK;
        o->n_ranks = dsh_ranks; o->rank = dsh_rank; o->own_bodies = dsh_own_count(); o->own_manifolds = dsh_own_manifolds; o->global_manifolds = dsh_global_manifolds;
        o->exchanges = dsh_exchanges; o->bytes_sent_per_step = (uint64_t)dsh_own_count() * 4 * sizeof(V);
        return AVN_OK;
    }
