// world/host_shapes.hpp -- fragment of the body of `template <class T> struct World` (avn_world.hip includes it inside the class):
// host shapes (include/avian_mi355x.h "host shapes", round 6) -- colliders whose AnyCollider::aabb_with_context / contact_manifolds_with_context stay on the host
// (collision/collider/mod.rs), everything else of update_aabb / update_contacts on the device.  Two round trips per step, only when such colliders exist, carrying only
// their queries and answers.

    avn_host_aabb_fn hs_aabb_fn = nullptr;
    avn_host_manifolds_fn hs_manifolds_fn = nullptr;
    void* hs_user = nullptr;
    std::vector<uint32_t> hs_slots;          // collider slots uploaded with AVN_SHAPE_HOST (upload order)
    DevBuf b_hs_slots, b_hs_aq, b_hs_ab, b_hs_mq, b_hs_mm, b_hs_cnt;
    Pinned pin_hs, pin_hs_aabb;
    uint32_t hs_mq_cap = 0;
    avn_host_shape_stats hs_stats{};
    struct HsLaunch { int form; uint32_t a, b, c, d; const uint32_t* list; };   // the light launches of the step's narrow phase (a retry repeats them host-only)
    std::vector<HsLaunch> hs_launches;
    static constexpr size_t HS_AQ = sizeof(T) == 4 ? sizeof(avn_host_aabb_query_f32) : sizeof(avn_host_aabb_query_f64);
    static constexpr size_t HS_AB = sizeof(T) == 4 ? sizeof(avn_host_aabb_f32) : sizeof(avn_host_aabb_f64);
    static constexpr size_t HS_MQ = sizeof(T) == 4 ? sizeof(avn_host_manifold_query_f32) : sizeof(avn_host_manifold_query_f64);
    static constexpr size_t HS_MM = sizeof(T) == 4 ? sizeof(avn_host_manifold_f32) : sizeof(avn_host_manifold_f64);

    avn_status host_shapes_set(avn_host_aabb_fn a, avn_host_manifolds_fn m, void* user) override {
        if ((a == nullptr) != (m == nullptr)) { error = "host_shapes_set: both callbacks or none"; return AVN_ERR_BAD_ARG; }
        hs_aabb_fn = a; hs_manifolds_fn = m; hs_user = user;
        return AVN_OK;
    }
    avn_status host_shape_stats_get(avn_host_shape_stats* o) override {
        if (!o) return AVN_ERR_BAD_ARG;
        hs_stats.host_colliders = (uint32_t)hs_slots.size();
        *o = hs_stats;
        return AVN_OK;
    }
    // colliders_upload: which slots are the host's
    avn_status hs_on_colliders_upload(const avn_colliders* c) {
        hs_slots.clear();
        for (uint32_t i = 0; i < c->count; ++i) if (c->shape[i] == AVN_SHAPE_HOST) hs_slots.push_back(i);
        if (hs_slots.empty()) return AVN_OK;
        hipError_t err;
        const size_t n = hs_slots.size();
        b_hs_slots.ensure(n * 4, err); if (err != hipSuccess) { error = "hipMalloc failed"; return AVN_ERR_OOM; }
        b_hs_aq.ensure(n * HS_AQ, err); if (err != hipSuccess) { error = "hipMalloc failed"; return AVN_ERR_OOM; }
        b_hs_ab.ensure(n * HS_AB, err); if (err != hipSuccess) { error = "hipMalloc failed"; return AVN_ERR_OOM; }
        b_hs_cnt.ensure(64, err); if (err != hipSuccess) { error = "hipMalloc failed"; return AVN_ERR_OOM; }
        HIPCHK(hipMemcpy(b_hs_slots.p, hs_slots.data(), n * 4, hipMemcpyHostToDevice));
        HIPCHK(hipMemset(b_hs_cnt.p, 0, 64));
        if (const char* e = avn_env("AVN_HS_QUERY_CAP")) return hs_reserve_queries((uint32_t)std::max<long>(1, atol(e)));   // (measure build, tests: force the grow-and-retry path)
        return hs_reserve_queries((uint32_t)std::max<size_t>(1024, 8 * n));
    }
    avn_status hs_reserve_queries(uint32_t cap) {
        if (cap <= hs_mq_cap) return AVN_OK;
        hipError_t err;
        b_hs_mq.ensure((size_t)cap * HS_MQ, err); if (err != hipSuccess) { error = "hipMalloc failed"; return AVN_ERR_OOM; }
        b_hs_mm.ensure((size_t)cap * HS_MM, err); if (err != hipSuccess) { error = "hipMalloc failed"; return AVN_ERR_OOM; }
        hs_mq_cap = cap;
        return AVN_OK;
    }
    bool hs_any() const { return !hs_slots.empty(); }
    avn_status hs_need_callbacks() { if (!hs_aabb_fn) { error = "the world holds AVN_SHAPE_HOST colliders and no callbacks (avn_host_shapes_set)"; return AVN_ERR_STATE; } return AVN_OK; }
    // ---- update_aabb's half: one box per host collider, written before k_update_aabb (which skips those colliders) ----
    avn_status hs_aabbs(hipStream_t s) {
        avn_status st = hs_need_callbacks();
        if (st != AVN_OK) return st;
        const uint32_t n = (uint32_t)hs_slots.size();
        if (pin_hs_aabb.ensure((size_t)n * (HS_AQ + HS_AB) + 64) != hipSuccess) { error = "hipHostMalloc failed"; return AVN_ERR_OOM; }
        launch_host_aabb_queries<T>(dw, bp, params, b_hs_slots.as<uint32_t>(), n, b_hs_aq.p, s); ++launches;
        HIPCHK(hipGetLastError());
        char* hq = (char*)pin_hs_aabb.p; char* ha = hq + (size_t)n * HS_AQ;
        HIPCHK(hipMemcpyAsync(hq, b_hs_aq.p, (size_t)n * HS_AQ, hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        const auto t0 = std::chrono::steady_clock::now();
        hs_aabb_fn(hs_user, (uint32_t)(8 * sizeof(T)), n, hq, ha);
        hs_stats.last_callback_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        HIPCHK(hipMemcpyAsync(b_hs_ab.p, ha, (size_t)n * HS_AB, hipMemcpyHostToDevice, s));
        launch_host_aabb_apply<T>(bp, params, b_hs_slots.as<uint32_t>(), n, b_hs_ab.p, s); ++launches;
        HIPCHK(hipGetLastError());
        hs_stats.last_aabb_queries = n; hs_stats.bytes_to_host += (uint64_t)n * HS_AQ; hs_stats.bytes_from_host += (uint64_t)n * HS_AB;
        return AVN_OK;
    }
    // ---- update_contacts' half ----
    // before the step's first narrow-phase launch: the list is empty
    NpHostList hs_begin(hipStream_t s) {
        hs_launches.clear();
        NpHostList l;
        if (hs_any()) { (void)hipMemsetAsync(b_hs_cnt.p, 0, 4, s); l = hs_list(); }
        if (hk_begin(l, s) != AVN_OK) l.hook = NpHookList();   // (collision hooks, world/hooks.hpp: phase 1 rides the same launches)
        l.locals = tf_any;                                      // (child colliders: the HS instantiations compute their poses)
        return l;
    }
    NpHookList hk_phase1() const { NpHookList h; if (hk_modify_active() && b_hk_cnt.p) { h.count = b_hk_cnt.as<uint32_t>(); h.phase = 1; } return h; }
    NpHostList hs_list(uint32_t host_only = 0) const {
        NpHostList l;
        if (!hs_any()) return l;
        l.queries = b_hs_mq.p; l.count = b_hs_cnt.as<uint32_t>(); l.cap = hs_mq_cap; l.host_only = host_only; l.locals = tf_any;
        return l;
    }
    // after the step's light + heavy launches (all on `s`): the pairs with a host collider have left their queries; ask the host, then run update_contacts' remainder
    // for them.  dense: the closed loop's outputs (chg / has per row, *n_changes = the removal counter); else the change list.
    avn_status hs_manifolds(bool dense, const StepParams<T>& np, avn_contact_change* changes, uint32_t* n_changes, uint32_t* chg, uint32_t* has, hipStream_t s) {
        if (!hs_any()) return AVN_OK;
        avn_status st = hs_need_callbacks();
        if (st != AVN_OK) return st;
        uint32_t* h_cnt = (uint32_t*)pin_hs.p;
        if (!h_cnt) { if (pin_hs.ensure((size_t)hs_mq_cap * (HS_MQ + HS_MM) + 64) != hipSuccess) { error = "hipHostMalloc failed"; return AVN_ERR_OOM; } h_cnt = (uint32_t*)pin_hs.p; }
        HIPCHK(hipMemcpyAsync(h_cnt, b_hs_cnt.p, 4, hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        uint32_t n = *h_cnt;
        hs_stats.last_manifold_queries = n; hs_stats.last_manifolds_with_points = 0;
        if (!n) return AVN_OK;
        uint32_t listed = n;   // entries in the device list (a retry may hold a pair twice, below)
        if (n > hs_mq_cap) {   // more pairs than the list held: grow, then only those pairs again (they wrote nothing; the list is refilled from its start)
            // A row whose id was recycled in this step is in the range of the launch over the OLD rows (it was a free id when that launch ran) and in the list of the
            // NEW ones: the retry's relaunch of both meets it twice.  The list is sized for that (every pair twice at worst) and the duplicates leave below.
            if ((st = hs_reserve_queries(2 * n + 16)) != AVN_OK) return st;
            HIPCHK(hipMemsetAsync(b_hs_cnt.p, 0, 4, s));
            hs_rerun(hs_list(1), dense, np, changes, n_changes, chg, has, s);
            HIPCHK(hipGetLastError());
            HIPCHK(hipMemcpyAsync(h_cnt, b_hs_cnt.p, 4, hipMemcpyDeviceToHost, s));
            HIPCHK(hipStreamSynchronize(s));
            listed = *h_cnt;
            if (listed < n || listed > hs_mq_cap) { error = "host shapes: the retry found " + std::to_string(listed) + " pairs, the first pass " + std::to_string(n); return AVN_ERR_STATE; }
        }
        if (pin_hs.cap < (size_t)listed * (HS_MQ + HS_MM) + 64) {
            if (pin_hs.ensure((size_t)std::max(hs_mq_cap, listed) * (HS_MQ + HS_MM) + 64) != hipSuccess) { error = "hipHostMalloc failed"; return AVN_ERR_OOM; }
        }
        char* hq = (char*)pin_hs.p + 64;
        HIPCHK(hipMemcpyAsync(hq, b_hs_mq.p, (size_t)listed * HS_MQ, hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        {   // ascending contact id: the order update_contacts would meet them in, and a deterministic callback; a pair listed twice keeps one entry
            std::vector<uint32_t> idx(listed);
            for (uint32_t i = 0; i < listed; ++i) idx[i] = i;
            auto cid = [&](uint32_t i) { return *(const uint32_t*)(hq + (size_t)i * HS_MQ); };
            std::sort(idx.begin(), idx.end(), [&](uint32_t a, uint32_t b) { return cid(a) < cid(b); });
            idx.erase(std::unique(idx.begin(), idx.end(), [&](uint32_t a, uint32_t b) { return cid(a) == cid(b); }), idx.end());
            if (idx.size() != n) { error = "host shapes: the retry found " + std::to_string(idx.size()) + " distinct pairs, the first pass " + std::to_string(n); return AVN_ERR_STATE; }
            std::vector<char> tmp((size_t)n * HS_MQ);
            for (uint32_t i = 0; i < n; ++i) std::memcpy(tmp.data() + (size_t)i * HS_MQ, hq + (size_t)idx[i] * HS_MQ, HS_MQ);
            std::memcpy(hq, tmp.data(), tmp.size());
        }
        char* hm = hq + (size_t)n * HS_MQ;
        std::memset(hm, 0, (size_t)n * HS_MM);
        const auto t0 = std::chrono::steady_clock::now();
        hs_manifolds_fn(hs_user, (uint32_t)(8 * sizeof(T)), n, hq, hm);
        hs_stats.last_callback_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        for (uint32_t i = 0; i < n; ++i) {
            const uint32_t pc = *(const uint32_t*)(hm + (size_t)i * HS_MM);
            if (pc > AVN_MAX_QUERY_POINTS) { error = "host shapes: a manifold with more than AVN_MAX_QUERY_POINTS points"; return AVN_ERR_BAD_ARG; }
            hs_stats.last_manifolds_with_points += pc != 0;
        }
        HIPCHK(hipMemcpyAsync(b_hs_mq.p, hq, (size_t)n * HS_MQ, hipMemcpyHostToDevice, s));
        HIPCHK(hipMemcpyAsync(b_hs_mm.p, hm, (size_t)n * HS_MM, hipMemcpyHostToDevice, s));
        launch_narrow_phase_host<T>(dw, bp, ct, np, dense, changes, n_changes, chg, has, b_hs_mq.p, b_hs_mm.p, n, s, hk_phase1()); ++launches;
        HIPCHK(hipGetLastError());
        hs_stats.bytes_to_host += (uint64_t)n * HS_MQ; hs_stats.bytes_from_host += (uint64_t)n * (HS_MQ + HS_MM);
        return AVN_OK;
    }
