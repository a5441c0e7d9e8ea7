// world/level2.hpp -- fragment of the body of `template <class T> struct World` (avn_world.hip includes it inside the class):
// level-2 sharding: halo plan, per-colour passes, exchange.

    // ---- level-2 sharding (include/avian_mi355x.h: avn_halo_plan) ------------------------------------------------------------------------
    // One contact island over several worlds: global colouring, and after every colour launch the (linear, angular) velocity records of the
    // shared bodies this world's manifolds moved go to the other holders.  Exchange records of one colour are contiguous over the peers
    // (list k = colour * n_peers + peer), so a colour costs one pack launch, one grouped RCCL send/recv and one unpack launch.
    struct HaloPlan {
        std::vector<int32_t> peers, send, recv;
        std::vector<uint32_t> send_off, recv_off;   // [n_slots * n_peers + 1]
    } halo;
    // exchange slots: colours 0..22, then the overflow colour -- ONE slot (the world's whole overflow colour), or one per level of the GLOBAL overflow list when a
    // manifold of it touches a shared body (avn_halo_overflow_levels_upload; round 6).  l2_order = this world's overflow manifolds grouped by level, l2_off[l] the groups.
    uint32_t l2_levels = 1;
    std::vector<uint32_t> l2_level_of, l2_order, l2_off;
    DevBuf b_l2_order;
    // round 6: the joint slot (header: avn_halo_joint_slot_set) -- behind the colours and levels; its records are the whole SolverBody (4 Vec4 per body instead of 2)
    bool l2_joint_slot = false, l2_global_joints = false;
    uint32_t halo_slots() const { return (uint32_t)AVN_COLOR_OVERFLOW_INDEX + l2_levels + (l2_joint_slot ? 1u : 0u); }
    uint32_t joint_slot_index() const { return (uint32_t)AVN_COLOR_OVERFLOW_INDEX + l2_levels; }
    bool is_joint_slot(uint32_t slot) const { return l2_joint_slot && slot == joint_slot_index(); }
    avn_status halo_joint_slot_set(uint32_t joint_slot, uint32_t global_joints) override {
        HIPCHK(hipStreamSynchronize(stream));
        l2_joint_slot = joint_slot != 0; l2_global_joints = global_joints != 0;
        halo = HaloPlan(); halo_on = false;   // (a plan uploaded before counted other slots)
        drop_graph();
        return AVN_OK;
    }
    avn_status halo_overflow_levels_upload(uint32_t n_levels, const uint32_t* level_of, size_t count) override {
        if (count && !level_of) { error = "halo_overflow_levels_upload: null array"; return AVN_ERR_BAD_ARG; }
        for (size_t i = 0; i < count; ++i) if (level_of[i] >= std::max(n_levels, 1u)) { error = "halo_overflow_levels_upload: level out of range"; return AVN_ERR_BAD_ARG; }
        HIPCHK(hipStreamSynchronize(stream));
        l2_levels = std::max(n_levels, 1u);
        l2_level_of.assign(level_of, level_of + count);
        halo = HaloPlan(); halo_on = false;   // (a plan uploaded before counted other slots)
        drop_graph();
        return AVN_OK;
    }
    // this world's overflow manifolds grouped by level (a counting sort: list order inside a level, though manifolds of a level share no body)
    avn_status l2_build_levels() {
        const uint32_t o0 = color_offsets[AVN_COLOR_OVERFLOW_INDEX], n23 = color_offsets[AVN_COLOR_OVERFLOW_INDEX + 1] - o0;
        if (l2_level_of.size() != n23) { error = "level-2: avn_halo_overflow_levels_upload named " + std::to_string(l2_level_of.size()) + " overflow manifolds, the world holds " + std::to_string(n23); return AVN_ERR_STATE; }
        l2_off.assign(l2_levels + 1, 0u);
        for (uint32_t l : l2_level_of) ++l2_off[l + 1];
        for (uint32_t l = 0; l < l2_levels; ++l) l2_off[l + 1] += l2_off[l];
        l2_order.resize(n23);
        std::vector<uint32_t> cur(l2_off.begin(), l2_off.end() - 1);
        for (uint32_t i = 0; i < n23; ++i) l2_order[cur[l2_level_of[i]]++] = o0 + i;
        hipError_t err;
        b_l2_order.ensure(std::max<size_t>(n23, 1) * 4, err);
        if (err != hipSuccess) { error = "hipMalloc failed"; return AVN_ERR_OOM; }
        if (n23) HIPCHK(hipMemcpy(b_l2_order.p, l2_order.data(), (size_t)n23 * 4, hipMemcpyHostToDevice));
        l2_built_for = dw.n_manifolds;
        return AVN_OK;
    }
    uint32_t l2_built_for = 0xFFFFFFFFu;
    DevBuf b_halo_send, b_halo_recv, b_halo_out, b_halo_in;
    bool halo_on = false;
#ifdef AVN_MEASURE   // measurement build only (make measure): the default library has no switch that changes results
    bool bias_skeleton = avn_env("AVN_BIAS_SKELETON") != nullptr && avn_env("AVN_BIAS_SKELETON")[0] == '1';
#else
    static constexpr bool bias_skeleton = false;
#endif
    Comm comm;
    std::vector<CommXfer> xf_send, xf_recv;
    avn_status halo_plan_upload(const avn_halo_plan* p) override {
        if (!p) { error = "halo_plan_upload: null plan"; return AVN_ERR_BAD_ARG; }
        const size_t n = (size_t)halo_slots() * p->n_peers;
        if (p->n_peers && (!p->peer_rank || !p->send_offsets || !p->recv_offsets)) { error = "halo_plan_upload: null array"; return AVN_ERR_BAD_ARG; }
        HaloPlan h;
        if (p->n_peers) {
            h.peers.assign(p->peer_rank, p->peer_rank + p->n_peers);
            h.send_off.assign(p->send_offsets, p->send_offsets + n + 1); h.recv_off.assign(p->recv_offsets, p->recv_offsets + n + 1);
            for (size_t k = 0; k < n; ++k)
                if (h.send_off[k] > h.send_off[k + 1] || h.recv_off[k] > h.recv_off[k + 1]) { error = "halo_plan_upload: offsets must ascend"; return AVN_ERR_BAD_ARG; }
            if (h.send_off[0] || h.recv_off[0]) { error = "halo_plan_upload: offsets must start at 0"; return AVN_ERR_BAD_ARG; }
            if ((h.send_off[n] && !p->send_bodies) || (h.recv_off[n] && !p->recv_bodies)) { error = "halo_plan_upload: null body list"; return AVN_ERR_BAD_ARG; }
            h.send.assign(p->send_bodies, p->send_bodies + h.send_off[n]); h.recv.assign(p->recv_bodies, p->recv_bodies + h.recv_off[n]);
            const int64_t nb = have_bodies ? (int64_t)dw.n_bodies : INT32_MAX;
            for (int32_t b : h.send) if (b < 0 || b >= nb) { error = "halo_plan_upload: body index out of range"; return AVN_ERR_BAD_ARG; }
            for (int32_t b : h.recv) if (b < 0 || b >= nb) { error = "halo_plan_upload: body index out of range"; return AVN_ERR_BAD_ARG; }
        }
        HIPCHK(hipStreamSynchronize(stream));
        halo = std::move(h);
        halo_on = !halo.peers.empty();
        groups_dirty = true;
        drop_graph();
        if (!halo_on) return AVN_OK;
        hipError_t err;
        b_halo_send.ensure(std::max<size_t>(halo.send.size(), 1) * 4, err); if (err != hipSuccess) { error = "hipMalloc failed"; return AVN_ERR_OOM; }
        b_halo_recv.ensure(std::max<size_t>(halo.recv.size(), 1) * 4, err); if (err != hipSuccess) { error = "hipMalloc failed"; return AVN_ERR_OOM; }
        b_halo_out.ensure(std::max<size_t>(halo.send.size(), 1) * 4 * sizeof(V), err); if (err != hipSuccess) { error = "hipMalloc failed"; return AVN_ERR_OOM; }
        b_halo_in.ensure(std::max<size_t>(halo.recv.size(), 1) * 4 * sizeof(V), err);   // (record k of list entry i at 2 i: the joint slot, the LAST slot, spills into the doubled tail) if (err != hipSuccess) { error = "hipMalloc failed"; return AVN_ERR_OOM; }
        if (!halo.send.empty()) HIPCHK(hipMemcpyAsync(b_halo_send.p, halo.send.data(), halo.send.size() * 4, hipMemcpyHostToDevice, stream));
        if (!halo.recv.empty()) HIPCHK(hipMemcpyAsync(b_halo_recv.p, halo.recv.data(), halo.recv.size() * 4, hipMemcpyHostToDevice, stream));
        HIPCHK(hipStreamSynchronize(stream));
        return AVN_OK;
    }
    // one colour of one contact pass, in the single-world launch shape (overflow colour: the host schedule's launches); slot >= 23 with levels: ONE level of the overflow colour
    avn_status color_pass_enqueue(int pass, uint32_t color) {
        if (pipe_dev) { error = "level-2 colour passes need host-uploaded manifolds (not the device closed loop)"; return AVN_ERR_STATE; }
        if (is_joint_slot(color)) return AVN_OK;   // (not a contact slot)
        if (l2_levels > 1 && color >= (uint32_t)AVN_COLOR_OVERFLOW_INDEX) {
            if (!dw.n_manifolds) return AVN_OK;
            if (l2_built_for != dw.n_manifolds || l2_off.size() != l2_levels + 1) { avn_status sb = l2_build_levels(); if (sb != AVN_OK) return sb; }
            const uint32_t l = color - (uint32_t)AVN_COLOR_OVERFLOW_INDEX;
            if (l2_off[l + 1] == l2_off[l]) return AVN_OK;   // none of this world's manifolds in the level
            uint32_t gb[AVN_GRAPH_COLOR_COUNT] = {0};
            gb[AVN_COLOR_OVERFLOW_INDEX] = 1;
            OverflowSchedule ovf{0, nullptr, nullptr, nullptr, b_l2_order.as<uint32_t>(), l2_off.data() + l, 1};   // one device-wide launch over the level's manifolds
            launches += launch_contact_pass<T>(dw, params, pass, gb, use_handles ? nullptr : color_offsets, ovf, stream);
            return AVN_OK;
        }
        if (color >= AVN_GRAPH_COLOR_COUNT) return AVN_OK;
        if (!dw.n_manifolds || !grid_blocks[color]) return AVN_OK;
        uint32_t gb[AVN_GRAPH_COLOR_COUNT] = {0};
        gb[color] = grid_blocks[color];
        OverflowSchedule ovf{0, nullptr, nullptr, nullptr, nullptr, nullptr, 0};
        if (color == AVN_COLOR_OVERFLOW_INDEX) {
            ovf = OverflowSchedule{sched_overflow.n_components, sched_overflow.d_comp_level_begin.as<uint32_t>(), sched_overflow.d_level_offsets.as<uint32_t>(),
                                   sched_overflow.d_order.as<uint32_t>(), nullptr, nullptr, 0};
            if (sched_overflow.gorder.size() > overflow_level_threshold) {
                ovf.gorder = sched_overflow.d_gorder.as<uint32_t>();
                ovf.glevel_offsets = sched_overflow.glevel_offsets.data();
                ovf.n_glevels = (uint32_t)sched_overflow.glevel_offsets.size() - 1;
            }
        }
        launches += launch_contact_pass<T>(dw, params, pass, gb, use_handles ? nullptr : color_offsets, ovf, stream);
        return AVN_OK;
    }
    static int color_pass_of(avn_system sys) {
        switch (sys) {
            case AVN_SYS_WARM_START: return PASS_WARM_START_COLORS;
            case AVN_SYS_SOLVE_CONTACTS_BIAS: return PASS_SOLVE_BIAS;
            case AVN_SYS_SOLVE_CONTACTS_RELAX: return PASS_SOLVE_RELAX;
            case AVN_SYS_SOLVE_RESTITUTION: return PASS_RESTITUTION_;
            default: return -1;
        }
    }
    avn_status run_color_pass(avn_system sys, uint32_t color) override {
        if (color >= halo_slots() || is_joint_slot(color)) { error = "run_color_pass: colour / slot out of range"; return AVN_ERR_BAD_ARG; }
        const int pass = color_pass_of(sys);
        if (pass < 0) { error = "run_color_pass: not a contact pass"; return AVN_ERR_BAD_ARG; }
        avn_status st = need_bodies();
        if (st != AVN_OK) return st;
        if ((st = rebuild_incidence()) != AVN_OK) return st;
        if (pass == PASS_RESTITUTION_ && !any_restitution) return AVN_OK;
        if ((st = color_pass_enqueue(pass, color)) != AVN_OK) return st;
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(stream));
        return AVN_OK;
    }
    // where list entry `a` of slot `slot` lives in b_halo_out / b_halo_in (in Vec4 records): 2 per body; the joint slot -- the last one -- 4 per body from its own start
    size_t halo_rec(uint32_t slot, const std::vector<uint32_t>& off, size_t a) const {
        if (!is_joint_slot(slot)) return 2 * a;
        const size_t s0 = off[(size_t)slot * halo.peers.size()];
        return 2 * s0 + 4 * (a - s0);
    }
    avn_status halo_list(uint32_t color, uint32_t peer, const std::vector<uint32_t>& off, size_t* b0, size_t* b1) {
        if (color >= halo_slots() || peer >= halo.peers.size()) { error = "halo: colour / slot or peer out of range"; return AVN_ERR_BAD_ARG; }
        const size_t k = (size_t)color * halo.peers.size() + peer;
        *b0 = off[k]; *b1 = off[k + 1];
        return AVN_OK;
    }
    avn_status halo_pack(uint32_t color, uint32_t peer, void* out, size_t* count) override {
        if (!count) { error = "halo_pack: null count"; return AVN_ERR_BAD_ARG; }
        size_t b0, b1;
        avn_status st = halo_list(color, peer, halo.send_off, &b0, &b1);
        if (st != AVN_OK) return st;
        *count = b1 - b0;
        if (b1 == b0) return AVN_OK;
        if (!out) { error = "halo_pack: null output"; return AVN_ERR_BAD_ARG; }
        if ((st = need_bodies()) != AVN_OK) return st;
        const size_t rv = is_joint_slot(color) ? 4 : 2;
        V* rec = b_halo_out.as<V>() + halo_rec(color, halo.send_off, b0);
        if (rv == 4) launch_halo_pack_joint<T>(dw, b_halo_send.as<int32_t>() + b0, (uint32_t)(b1 - b0), rec, stream);
        else launch_halo_pack<T>(dw, b_halo_send.as<int32_t>() + b0, (uint32_t)(b1 - b0), rec, stream);
        ++launches;
        HIPCHK(hipMemcpyAsync(out, rec, (b1 - b0) * rv * sizeof(V), hipMemcpyDeviceToHost, stream));
        HIPCHK(hipStreamSynchronize(stream));
        return AVN_OK;
    }
    avn_status halo_unpack(uint32_t color, uint32_t peer, const void* in, size_t count) override {
        size_t b0, b1;
        avn_status st = halo_list(color, peer, halo.recv_off, &b0, &b1);
        if (st != AVN_OK) return st;
        if (count != b1 - b0 || (count && !in)) { error = "halo_unpack: count does not match the plan"; return AVN_ERR_BAD_ARG; }
        if (!count) return AVN_OK;
        if ((st = need_bodies()) != AVN_OK) return st;
        const size_t rv = is_joint_slot(color) ? 4 : 2;
        V* rec = b_halo_in.as<V>() + halo_rec(color, halo.recv_off, b0);
        HIPCHK(hipMemcpyAsync(rec, in, count * rv * sizeof(V), hipMemcpyHostToDevice, stream));
        if (rv == 4) launch_halo_unpack_joint<T>(dw, b_halo_recv.as<int32_t>() + b0, (uint32_t)count, rec, stream);
        else launch_halo_unpack<T>(dw, b_halo_recv.as<int32_t>() + b0, (uint32_t)count, rec, stream);
        ++launches;
        HIPCHK(hipStreamSynchronize(stream));
        return AVN_OK;
    }
    avn_status comm_init(const uint8_t* unique_id, int n_ranks, int rank) override {
        HIPCHK(hipStreamSynchronize(stream));
        return comm.init(unique_id, n_ranks, rank, error);
    }
    // ---- level-1 sharding: the bounds all-gather behind the ABI (avn_bounds_exchange) ---------------------------------------------------------
    DevBuf b_bounds_send, b_bounds_recv;
    avn_status bounds_exchange(double* bounds, uint32_t cap_ranks, uint32_t* n_ranks_out, uint32_t* overlaps, uint32_t cap_overlaps, uint32_t* n_overlaps) override {
        const uint32_t nr = comm.handle ? (uint32_t)comm.n_ranks : 1u;
        if (n_ranks_out) *n_ranks_out = nr;
        if (!bounds || cap_ranks < nr) { error = "bounds_exchange: the bounds array must hold n_ranks x 6 doubles"; return AVN_ERR_BAD_ARG; }
        avn_status st = need_bodies();
        if (st != AVN_OK) return st;
        hipError_t err;
        b_bounds_send.ensure(6 * sizeof(double), err); if (err != hipSuccess) { error = "hipMalloc failed"; return AVN_ERR_OOM; }
        b_bounds_recv.ensure((size_t)nr * 6 * sizeof(double), err); if (err != hipSuccess) { error = "hipMalloc failed"; return AVN_ERR_OOM; }
        const uint32_t nb = (bp.n_colliders + 255) / 256;
        if ((st = stage_reserve((size_t)std::max(nb, 1u) * 6 * sizeof(T) + 1024)) != AVN_OK) return st;
        T* part = stage_alloc<T>((size_t)std::max(nb, 1u) * 6);
        launch_dynamic_bounds<T>(dw, bp, part, stream);
        launch_bounds_reduce<T>(part, nb, b_bounds_send.as<double>(), stream);
        launches += 2;
        HIPCHK(hipGetLastError());
        if (comm.handle) { if ((st = comm.all_gather(b_bounds_send.p, b_bounds_recv.p, 6 * sizeof(double), stream, error)) != AVN_OK) return st; }
        else HIPCHK(hipMemcpyAsync(b_bounds_recv.p, b_bounds_send.p, 6 * sizeof(double), hipMemcpyDeviceToDevice, stream));
        HIPCHK(hipMemcpyAsync(bounds, b_bounds_recv.p, (size_t)nr * 6 * sizeof(double), hipMemcpyDeviceToHost, stream));
        HIPCHK(hipStreamSynchronize(stream));
        uint32_t found = 0;
        for (uint32_t i = 0; i < nr; ++i)
            for (uint32_t j = i + 1; j < nr; ++j) {
                const double *a = bounds + 6 * i, *b = bounds + 6 * j;
                bool hit = true;
                for (int k = 0; k < 3; ++k) hit = hit && a[k] <= b[3 + k] && a[3 + k] >= b[k];   // ColliderAabb::intersects (closed intervals); empty bounds (+inf, -inf) never hit
                if (hit) { if (overlaps && found < cap_overlaps) { overlaps[2 * found] = i; overlaps[2 * found + 1] = j; } ++found; }
            }
        if (n_overlaps) *n_overlaps = found;
        return AVN_OK;
    }
    // the exchange after colour c inside avn_step: everything is enqueued on the world's stream, no host code waits
    avn_status halo_exchange(uint32_t c) {
        const size_t np = halo.peers.size(), k0 = (size_t)c * np;
        const size_t s0 = halo.send_off[k0], s1 = halo.send_off[k0 + np], r0 = halo.recv_off[k0], r1 = halo.recv_off[k0 + np];
        if (s1 == s0 && r1 == r0) return AVN_OK;
        const bool jt = is_joint_slot(c);
        const size_t rv = jt ? 4 : 2;
        if (s1 > s0) {
            if (jt) launch_halo_pack_joint<T>(dw, b_halo_send.as<int32_t>() + s0, (uint32_t)(s1 - s0), b_halo_out.as<V>() + halo_rec(c, halo.send_off, s0), stream);
            else launch_halo_pack<T>(dw, b_halo_send.as<int32_t>() + s0, (uint32_t)(s1 - s0), b_halo_out.as<V>() + 2 * s0, stream);
            ++launches;
        }
        xf_send.clear(); xf_recv.clear();
        for (size_t p = 0; p < np; ++p) {
            const size_t a = halo.send_off[k0 + p], b = halo.send_off[k0 + p + 1], ra = halo.recv_off[k0 + p], rb = halo.recv_off[k0 + p + 1];
            if (b > a) xf_send.push_back(CommXfer{b_halo_out.as<V>() + halo_rec(c, halo.send_off, a), (b - a) * rv * sizeof(V), halo.peers[p]});
            if (rb > ra) xf_recv.push_back(CommXfer{b_halo_in.as<V>() + halo_rec(c, halo.recv_off, ra), (rb - ra) * rv * sizeof(V), halo.peers[p]});
        }
        avn_status st = comm.exchange(xf_send.data(), xf_send.size(), xf_recv.data(), xf_recv.size(), stream, error);
        if (st != AVN_OK) return st;
        ++halo_exchanges;
        if (r1 > r0) {
            if (jt) launch_halo_unpack_joint<T>(dw, b_halo_recv.as<int32_t>() + r0, (uint32_t)(r1 - r0), b_halo_in.as<V>() + halo_rec(c, halo.recv_off, r0), stream);
            else launch_halo_unpack<T>(dw, b_halo_recv.as<int32_t>() + r0, (uint32_t)(r1 - r0), b_halo_in.as<V>() + 2 * r0, stream);
            ++launches;
        }
        return AVN_OK;
    }
    uint32_t halo_exchanges = 0;
    // one contact pass in level-2 form: colours in solve order (overflow first), exchange after each
    avn_status level2_pass(int pass) {
        // solve order: the overflow colour first (its levels in order), then colours 0..22
        for (uint32_t k = 0; k < (uint32_t)AVN_COLOR_OVERFLOW_INDEX + l2_levels; ++k) {   // (the contact slots; the joint slot, when there is one, follows the joint systems)
            const uint32_t c = k < l2_levels ? (uint32_t)AVN_COLOR_OVERFLOW_INDEX + k : k - l2_levels;
            avn_status st = color_pass_enqueue(pass, c);
            if (st == AVN_OK) st = halo_exchange(c);
            if (st != AVN_OK) return st;
        }
        return AVN_OK;
    }
    avn_status level2_substeps() {   // SubstepSchedule with the contact passes split by colour (avian_amd/shard.py: level2_solver)
        for (uint32_t s = 0; s < cfg.substeps; ++s) {
            integrate_velocities();
            avn_status st = level2_pass(PASS_WARM_START_COLORS);
            for (uint32_t it = 0; it < cfg.solver_iterations && st == AVN_OK; ++it) st = level2_pass(PASS_SOLVE_BIAS);
            if (st != AVN_OK) return st;
            integrate_positions();
            for (uint32_t it = 0; it < cfg.solver_iterations && st == AVN_OK; ++it) st = level2_pass(PASS_SOLVE_RELAX);
            if (st != AVN_OK) return st;
            for (uint32_t it = 0; it < cfg.solver_iterations; ++it) xpbd_solve(it == 0);
            xpbd_velocity_projection();
            joint_damping();
            if (l2_joint_slot && (st = halo_exchange(joint_slot_index())) != AVN_OK) return st;   // the joint components' shared bodies: owner -> the other holders
        }
        return AVN_OK;
    }
