// world/hooks.hpp -- fragment of the body of `template <class T> struct World` (avn_world.hip includes it inside the class):
// collision hooks (include/avian_mi355x.h "collision hooks", round 6) -- CollisionHooks::filter_pairs (broad_phase.rs:431-439) and CollisionHooks::modify_contacts
// (narrow_phase/system_param.rs:770-778) as callbacks, so that a world with ActiveCollisionHooks colliders stays in the closed loop.  Only the pairs the hooks are
// asked about cross the bus; everything on either side of the hook stays on the device.

    avn_filter_pairs_fn hk_filter_fn = nullptr;
    avn_modify_contacts_fn hk_modify_fn = nullptr;
    void* hk_user = nullptr;
    bool hk_restitution = false;   // a hook has left a restitution != 0 in a record: the restitution pass runs from then on whatever the materials say (sticky)
    bool hk_any_filter = false, hk_any_modify = false;   // a collider has carried the flag since the world was created (rows keep MODIFY_CONTACTS from their creation: sticky)
    DevBuf b_hk_cnt, b_hk_rec, b_hk_fq, b_hk_rej, b_pairs_alt;
    Pinned pin_hk;
    uint32_t hk_rec_cap = 0;
    avn_collision_hook_stats hk_stats{};
    static constexpr size_t HK_REC = sizeof(T) == 4 ? sizeof(avn_hook_contact_f32) : sizeof(avn_hook_contact_f64);

    avn_status collision_hooks_set(avn_filter_pairs_fn f, avn_modify_contacts_fn m, void* user) override { hk_filter_fn = f; hk_modify_fn = m; hk_user = user; return AVN_OK; }
    avn_status collision_hook_stats_get(avn_collision_hook_stats* o) override { if (!o) return AVN_ERR_BAD_ARG; *o = hk_stats; return AVN_OK; }
    avn_status hk_on_colliders_upload(const avn_colliders* c) {
        for (uint32_t i = 0; i < c->count && c->collider_flags; ++i) {
            hk_any_filter |= (c->collider_flags[i] & AVN_COLLIDER_FILTER_PAIRS) != 0;
            hk_any_modify |= (c->collider_flags[i] & AVN_COLLIDER_MODIFY_CONTACTS) != 0;
        }
        return AVN_OK;
    }
    bool hk_filter_active() const { return hk_filter_fn && hk_any_filter; }
    bool hk_modify_active() const { return hk_modify_fn && hk_any_modify; }
    avn_status hk_counters() {
        if (b_hk_cnt.p) return AVN_OK;
        hipError_t err;
        b_hk_cnt.ensure(64, err); if (err != hipSuccess) { error = "hipMalloc failed"; return AVN_ERR_OOM; }
        HIPCHK(hipMemset(b_hk_cnt.p, 0, 64));
        return AVN_OK;
    }
    // ---- filter_pairs ----
    // the callback over `n` candidate pairs (sorted by emission index here); rej <- the emission indices it rejected, ascending
    void hk_ask_filter(avn_hook_pair* q, uint32_t n, std::vector<uint32_t>& rej) {
        std::sort(q, q + n, [](const avn_hook_pair& a, const avn_hook_pair& b) { return a.index < b.index; });
        std::vector<uint8_t> keep(n, 1);
        const auto t0 = std::chrono::steady_clock::now();
        hk_filter_fn(hk_user, n, q, keep.data());
        hk_stats.last_callback_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        rej.clear();
        for (uint32_t i = 0; i < n; ++i) if (!keep[i]) rej.push_back(q[i].index);
        hk_stats.last_filter_queries = n; hk_stats.last_filter_rejected = (uint32_t)rej.size();
        hk_stats.bytes_to_host += (uint64_t)n * sizeof(avn_hook_pair); hk_stats.bytes_from_host += n;
    }
    // host forms of collect_collision_pairs (AVN_SYS_BROAD_PHASE, the host-bookkeeping loop): the emitted pairs are on the host already
    void hk_filter_host(std::vector<avn_pair>& pairs) {
        hk_stats.last_filter_queries = hk_stats.last_filter_rejected = 0; hk_stats.last_callback_ms = 0;
        if (!hk_filter_fn) return;
        std::vector<avn_hook_pair> q;
        for (uint32_t i = 0; i < pairs.size(); ++i) if (pairs[i].flags & AVN_PAIR_NEEDS_CUSTOM_FILTER) q.push_back(avn_hook_pair{i, pairs[i].collider1, pairs[i].collider2});
        if (q.empty()) return;
        std::vector<uint32_t> rej;
        hk_ask_filter(q.data(), (uint32_t)q.size(), rej);
        if (rej.empty()) return;
        size_t r = 0, o = 0;
        for (uint32_t i = 0; i < pairs.size(); ++i) {
            if (r < rej.size() && rej[r] == i) { ++r; continue; }
            pairs[o++] = pairs[i];
        }
        pairs.resize(o);
    }
    // device closed loop: b_pairs[0 .. total) in emission order, on `s`; total <- the pairs that passed (b_pairs compacted, order kept)
    avn_status hk_filter_device(uint32_t& total, hipStream_t s) {
        hk_stats.last_filter_queries = hk_stats.last_filter_rejected = 0; hk_stats.last_callback_ms = 0;
        if (!hk_filter_active() || !total) return AVN_OK;
        avn_status st = hk_counters();
        if (st != AVN_OK) return st;
        hipError_t err;
        b_hk_fq.ensure((size_t)total * sizeof(avn_hook_pair), err); if (err != hipSuccess) { error = "hipMalloc failed"; return AVN_ERR_OOM; }
        if (pin_hk.ensure(std::max<size_t>((size_t)total * sizeof(avn_hook_pair) + 64, 4096)) != hipSuccess) { error = "hipHostMalloc failed"; return AVN_ERR_OOM; }
        uint32_t* d_cnt = b_hk_cnt.as<uint32_t>() + 1;
        HIPCHK(hipMemsetAsync(d_cnt, 0, 4, s));
        launch_hook_filter_collect(b_pairs.as<avn_pair>(), total, b_hk_fq.as<avn_hook_pair>(), d_cnt, s); ++launches;
        HIPCHK(hipGetLastError());
        uint32_t* h_cnt = (uint32_t*)pin_hk.p;
        HIPCHK(hipMemcpyAsync(h_cnt, d_cnt, 4, hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        const uint32_t n = *h_cnt;
        if (!n) return AVN_OK;
        avn_hook_pair* q = (avn_hook_pair*)((char*)pin_hk.p + 64);
        HIPCHK(hipMemcpyAsync(q, b_hk_fq.p, (size_t)n * sizeof(avn_hook_pair), hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        std::vector<uint32_t> rej;
        hk_ask_filter(q, n, rej);
        if (rej.empty()) return AVN_OK;
        b_hk_rej.ensure(rej.size() * 4, err); if (err != hipSuccess) { error = "hipMalloc failed"; return AVN_ERR_OOM; }
        b_pairs_alt.ensure((size_t)total * sizeof(avn_pair), err); if (err != hipSuccess) { error = "hipMalloc failed"; return AVN_ERR_OOM; }
        HIPCHK(hipMemcpyAsync(b_hk_rej.p, rej.data(), rej.size() * 4, hipMemcpyHostToDevice, s));
        launch_hook_filter_compact(b_pairs.as<avn_pair>(), b_pairs_alt.as<avn_pair>(), total, b_hk_rej.as<uint32_t>(), (uint32_t)rej.size(), s); ++launches;
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(s));   // (`rej` is pageable host memory)
        std::swap(b_pairs.p, b_pairs_alt.p); std::swap(b_pairs.cap, b_pairs_alt.cap);
        total -= (uint32_t)rej.size();
        return AVN_OK;
    }
    // ---- modify_contacts ----
    // the step's narrow-phase launches carry this list: phase 1 (count + pending bit) when a hook is registered and a collider asks for it
    avn_status hk_begin(NpHostList& l, hipStream_t s) {
        hk_stats.last_modify_queries = hk_stats.last_modify_rejected = 0;
        if (!hk_filter_active()) hk_stats.last_callback_ms = 0;
        if (!hk_modify_active()) return AVN_OK;
        avn_status st = hk_counters();
        if (st != AVN_OK) return st;
        HIPCHK(hipMemsetAsync(b_hk_cnt.p, 0, 4, s));
        l.hook.records = nullptr; l.hook.count = b_hk_cnt.as<uint32_t>(); l.hook.cap = 0; l.hook.phase = 1;
        return AVN_OK;
    }
    // the launches of one narrow phase again, for a second pass that visits only some rows (host shapes: the retry after the query list grew; hooks: phase 2)
    void hs_rerun(const NpHostList& l, bool dense, const StepParams<T>& np, avn_contact_change* changes, uint32_t* n_changes, uint32_t* chg, uint32_t* has, hipStream_t s) {
        (void)dense;
        for (const HsLaunch& k : hs_launches) {
            if (k.form == 0) launch_narrow_phase<T>(dw, bp, ct, np, k.list, k.a, changes, n_changes, s, l);
            else if (k.form == 1) launch_narrow_phase_dense<T>(dw, bp, ct, np, k.a, chg, has, n_changes, s, false, l);
            else launch_narrow_phase_rows<T>(dw, bp, ct, np, k.list, k.a, k.b, k.c, chg, has, n_changes, s, l);
            ++launches;
        }
    }
    // after the step's narrow-phase launches and the host shapes' answers (all on `s`): phase 2 collects the pending pairs' records, the hook sees them, phase 3 finishes the pairs
    avn_status hk_modify(bool dense, const StepParams<T>& np, avn_contact_change* changes, uint32_t* n_changes, uint32_t* chg, uint32_t* has, hipStream_t s) {
        if (!hk_modify_active() || !b_hk_cnt.p) return AVN_OK;
        if (pin_hk.ensure(4096) != hipSuccess) { error = "hipHostMalloc failed"; return AVN_ERR_OOM; }
        uint32_t* d_cnt = b_hk_cnt.as<uint32_t>();
        HIPCHK(hipMemcpyAsync(pin_hk.p, d_cnt, 4, hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        const uint32_t n = *(uint32_t*)pin_hk.p;
        hk_stats.last_modify_queries = n;
        if (!n) return AVN_OK;
        hipError_t err;
        if (n > hk_rec_cap) {
            const uint32_t cap = std::max<uint32_t>(n + n / 2, 256u);
            b_hk_rec.ensure((size_t)cap * HK_REC, err); if (err != hipSuccess) { error = "hipMalloc failed"; return AVN_ERR_OOM; }
            hk_rec_cap = cap;
        }
        if (pin_hk.ensure((size_t)n * HK_REC + 64) != hipSuccess) { error = "hipHostMalloc failed"; return AVN_ERR_OOM; }
        HIPCHK(hipMemsetAsync(d_cnt, 0, 4, s));
        NpHostList l;
        l.hook.records = b_hk_rec.p; l.hook.count = d_cnt; l.hook.cap = hk_rec_cap; l.hook.phase = 2; l.locals = tf_any;
        hs_rerun(l, dense, np, changes, n_changes, chg, has, s);
        if (hs_any() && hs_stats.last_manifold_queries)   // (pending pairs with a host-shaped collider: their manifolds are still in the answer list)
            { launch_narrow_phase_host<T>(dw, bp, ct, np, dense, changes, n_changes, chg, has, b_hs_mq.p, b_hs_mm.p, hs_stats.last_manifold_queries, s, l.hook); ++launches; }
        HIPCHK(hipGetLastError());
        uint32_t* h_cnt = (uint32_t*)pin_hk.p;
        char* rec = (char*)pin_hk.p + 64;
        HIPCHK(hipMemcpyAsync(h_cnt, d_cnt, 4, hipMemcpyDeviceToHost, s));
        HIPCHK(hipMemcpyAsync(rec, b_hk_rec.p, (size_t)n * HK_REC, hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        if (*h_cnt != n) { error = "collision hooks: the second pass found " + std::to_string(*h_cnt) + " pending pairs, the first " + std::to_string(n); return AVN_ERR_STATE; }
        {   // ascending contact id: a deterministic callback (the reference's par_iter has no order)
            std::vector<uint32_t> idx(n);
            for (uint32_t i = 0; i < n; ++i) idx[i] = i;
            std::sort(idx.begin(), idx.end(), [&](uint32_t a, uint32_t b) { return *(const uint32_t*)(rec + (size_t)a * HK_REC) < *(const uint32_t*)(rec + (size_t)b * HK_REC); });
            std::vector<char> tmp((size_t)n * HK_REC);
            for (uint32_t i = 0; i < n; ++i) std::memcpy(tmp.data() + (size_t)i * HK_REC, rec + (size_t)idx[i] * HK_REC, HK_REC);
            std::memcpy(rec, tmp.data(), tmp.size());
        }
        const auto t0 = std::chrono::steady_clock::now();
        hk_modify_fn(hk_user, (uint32_t)(8 * sizeof(T)), n, rec);
        hk_stats.last_callback_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        for (uint32_t i = 0; i < n; ++i) {
            const uint32_t* r = (const uint32_t*)(rec + (size_t)i * HK_REC);   // (contact_id, collider1, collider2, body1, body2, flags, touching, manifold_count, point_count)
            if (r[8] > (uint32_t)AVN_MAX_MANIFOLD_POINTS) { error = "collision hooks: modify_contacts left more than 4 points in a manifold"; return AVN_ERR_BAD_ARG; }
            hk_stats.last_modify_rejected += r[6] == 0u;
            const T* rs = (const T*)(rec + (size_t)i * HK_REC + 10 * sizeof(uint32_t));   // (normal[3], friction, restitution, ...)
            if (r[6] && !(rs[4] == T(0))) hk_restitution = true;
        }
        if (hk_restitution) any_restitution = true;
        HIPCHK(hipMemcpyAsync(b_hk_rec.p, rec, (size_t)n * HK_REC, hipMemcpyHostToDevice, s));
        launch_narrow_phase_hooked<T>(dw, bp, ct, np, dense, changes, n_changes, chg, has, b_hk_rec.p, n, s); ++launches;
        HIPCHK(hipGetLastError());
        hk_stats.bytes_to_host += (uint64_t)n * HK_REC; hk_stats.bytes_from_host += (uint64_t)n * HK_REC;
        return AVN_OK;
    }

    // ---- child colliders (include/avian_mi355x.h "child colliders"): ColliderTransform per collider slot; the kernels compute the child's pose from its body's ----
    DevBuf b_col_lpos, b_col_lrot;
    bool tf_any = false;
    avn_status collider_transforms_upload(const avn_collider_transforms* t) override {
        HIPCHK(hipStreamSynchronize(stream)); HIPCHK(hipStreamSynchronize(stream_bp));
        if (!t || !t->count) { bp.col_lpos = nullptr; bp.col_lrot = nullptr; tf_any = false; return AVN_OK; }
        if (t->count != bp.n_colliders) { error = "collider_transforms_upload: count differs from the last colliders_upload"; return AVN_ERR_BAD_ARG; }
        if (!t->is_child || !t->translation || !t->rotation) { error = "collider_transforms_upload: null array"; return AVN_ERR_BAD_ARG; }
        const uint32_t C = t->count;
        std::vector<V> lp(C), lr(C);
        bool any = false;
        const T* tr = (const T*)t->translation; const T* ro = (const T*)t->rotation;
        for (uint32_t i = 0; i < C; ++i) {
            const bool child = t->is_child[i] != 0;
            any |= child;
            lp[i] = make4<T>(tr[3 * i], tr[3 * i + 1], tr[3 * i + 2], child ? T(1) : T(0));
            lr[i] = make4<T>(ro[4 * i], ro[4 * i + 1], ro[4 * i + 2], ro[4 * i + 3]);
        }
        tf_any = any;
        if (!any) { bp.col_lpos = nullptr; bp.col_lrot = nullptr; return AVN_OK; }
        hipError_t err;
        b_col_lpos.ensure((size_t)C * sizeof(V), err); if (err != hipSuccess) { error = "hipMalloc failed"; return AVN_ERR_OOM; }
        b_col_lrot.ensure((size_t)C * sizeof(V), err); if (err != hipSuccess) { error = "hipMalloc failed"; return AVN_ERR_OOM; }
        HIPCHK(hipMemcpy(b_col_lpos.p, lp.data(), (size_t)C * sizeof(V), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(b_col_lrot.p, lr.data(), (size_t)C * sizeof(V), hipMemcpyHostToDevice));
        bp.col_lpos = b_col_lpos.as<V>(); bp.col_lrot = b_col_lrot.as<V>();
        return AVN_OK;
    }
