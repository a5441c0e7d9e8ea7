// avn_level2.cpp — the host planners of the multi-GPU modes behind the C ABI (include/avian_mi355x.h: avn_level2_plan_*, avn_slab_select,
// avn_interval_orders_merge).  Host integer work only.
#include <algorithm>
#include <map>
#include <new>
#include <string>
#include <utility>
#include <vector>

#include "../../include/avian_mi355x.h"

struct avn_level2_plan {
    struct Rank {
        std::vector<int32_t> bodies;
        std::vector<uint32_t> manifolds, color_offsets;
        std::vector<int32_t> peers, send_bodies, recv_bodies;
        std::vector<uint32_t> send_offsets, recv_offsets;
        std::vector<uint32_t> overflow_level;   // per local overflow manifold (local order): its level in the GLOBAL overflow list, 0-based
        std::vector<uint32_t> joints;           // global joint indices owned (ascending)
    };
    uint32_t n_overflow_levels = 1;
    bool joint_slot = false, global_joints = false;
    std::vector<Rank> ranks;
};

static avn_status plan_create(const avn_level2_in* in, const avn_level2_joints* jn, avn_level2_plan** out) {
    if (jn && jn->n_joints && (!jn->body1 || !jn->body2 || !jn->joint_type)) return AVN_ERR_BAD_ARG;
    if (!in || !out || in->n_ranks == 0 || (in->n_bodies && (!in->rb_type || !in->center_x)) || (in->n_manifolds && (!in->body1 || !in->body2)) || !in->color_offsets)
        return AVN_ERR_BAD_ARG;
    *out = nullptr;
    try {
        const uint32_t N = in->n_bodies, M = in->n_manifolds, R = in->n_ranks, C = AVN_GRAPH_COLOR_COUNT;
        if (in->color_offsets[0] != 0 || in->color_offsets[C] != M) return AVN_ERR_BAD_ARG;
        for (uint32_t c = 0; c < C; ++c) if (in->color_offsets[c] > in->color_offsets[c + 1]) return AVN_ERR_BAD_ARG;
        for (uint32_t m = 0; m < M; ++m)
            if (in->body1[m] < 0 || in->body2[m] < 0 || (uint32_t)in->body1[m] >= N || (uint32_t)in->body2[m] >= N) return AVN_ERR_BAD_ARG;
        auto is_static = [&](uint32_t b) { return in->rb_type[b] == AVN_RB_STATIC; };
        // slab of every non-static body: cuts at the quantiles of their x
        std::vector<double> xs;
        for (uint32_t b = 0; b < N; ++b) if (!is_static(b)) xs.push_back(in->center_x[b]);
        std::stable_sort(xs.begin(), xs.end());
        std::vector<double> cuts;
        for (uint32_t r = 1; r < R && !xs.empty(); ++r) cuts.push_back(xs[std::min<size_t>(xs.size() - 1, xs.size() * (size_t)r / R)]);
        std::vector<int32_t> owner(N, -1);
        for (uint32_t b = 0; b < N; ++b)
            if (!is_static(b)) owner[b] = (int32_t)(std::upper_bound(cuts.begin(), cuts.end(), in->center_x[b]) - cuts.begin());   // cuts <= x
        std::vector<int32_t> m_owner(M);
        for (uint32_t m = 0; m < M; ++m) {
            const uint32_t a = (uint32_t)in->body1[m], b = (uint32_t)in->body2[m];
            m_owner[m] = is_static(a) ? owner[b] : owner[a];
            if (m_owner[m] < 0) return AVN_ERR_BAD_ARG;   // a manifold between two static bodies has no owner
        }
        // held[r][b]
        std::vector<std::vector<uint8_t>> held(R, std::vector<uint8_t>(N, 0));
        for (uint32_t r = 0; r < R; ++r)
            for (uint32_t b = 0; b < N; ++b) held[r][b] = is_static(b) || owner[b] == (int32_t)r;
        for (uint32_t m = 0; m < M; ++m) { held[m_owner[m]][in->body1[m]] = 1; held[m_owner[m]][in->body2[m]] = 1; }
        // joints (header: avn_halo_joint_slot_set): components over non-static bodies and -- with JointDamping -- the per-type DUMMY pair (virtual nodes N + 2 t, + 1);
        // owner of a component = the slab of its lowest non-static body; the owner holds every body of the component
        const uint32_t J = jn ? jn->n_joints : 0u;
        std::vector<uint8_t> jointed(N, 0);
        std::vector<int32_t> body_comp_owner(N, -1), j_owner(J, 0);
        if (J) {
            std::vector<uint32_t> parent(N + 2u * AVN_JOINT_TYPE_COUNT);
            for (uint32_t i = 0; i < parent.size(); ++i) parent[i] = i;
            auto find = [&](uint32_t x) { while (parent[x] != x) { parent[x] = parent[parent[x]]; x = parent[x]; } return x; };
            std::vector<int64_t> any_node(J, -1);
            for (uint32_t j = 0; j < J; ++j) {
                if (jn->body1[j] < 0 || jn->body2[j] < 0 || (uint32_t)jn->body1[j] >= N || (uint32_t)jn->body2[j] >= N || jn->joint_type[j] >= AVN_JOINT_TYPE_COUNT) return AVN_ERR_BAD_ARG;
                const uint32_t a = (uint32_t)jn->body1[j], b = (uint32_t)jn->body2[j], t = jn->joint_type[j];
                const int64_t na = is_static(a) ? (jn->damped ? (int64_t)(N + 2u * t) : -1) : (int64_t)a, nb = is_static(b) ? (jn->damped ? (int64_t)(N + 2u * t + 1u) : -1) : (int64_t)b;
                if (na >= 0 && nb >= 0) { const uint32_t ra = find((uint32_t)na), rb = find((uint32_t)nb); if (ra != rb) parent[std::max(ra, rb)] = std::min(ra, rb); }
                any_node[j] = std::max(na, nb);
                if (!is_static(a)) jointed[a] = 1;
                if (!is_static(b)) jointed[b] = 1;
            }
            std::vector<int32_t> comp_owner(parent.size(), -1);
            for (uint32_t b = 0; b < N; ++b) if (jointed[b]) { const uint32_t r = find(b); if (comp_owner[r] < 0) comp_owner[r] = owner[b]; }   // ascending: the lowest body decides
            for (uint32_t b = 0; b < N; ++b) if (jointed[b]) { body_comp_owner[b] = comp_owner[find(b)]; held[(uint32_t)body_comp_owner[b]][b] = 1; }
            for (uint32_t j = 0; j < J; ++j) { const int32_t o = any_node[j] >= 0 ? comp_owner[find((uint32_t)any_node[j])] : -1; j_owner[j] = o < 0 ? 0 : o; }
        }
        std::vector<uint8_t> shared(N, 0);
        for (uint32_t b = 0; b < N; ++b) {
            if (is_static(b)) continue;
            uint32_t k = 0;
            for (uint32_t r = 0; r < R; ++r) k += held[r][b];
            shared[b] = k > 1;
        }
        // Exchange SLOTS: slot c < 23 = colour c.  The overflow colour is solved serially in list order (solver/plugin.rs:461-467); only the relative order of manifolds
        // that share a body matters, so it is cut into LEVELS over the GLOBAL list -- level(m) = 1 + the highest level of an earlier overflow manifold on one of m's
        // non-static bodies; manifolds of one level share no body -- and slot 23 + l = level l: every world runs its own manifolds of the level, then the shared bodies they
        // moved travel, exactly as after a colour.  (Round 5 refused an overflow manifold on a shared body.)  Without one, the overflow colour stays ONE slot (23).
        const uint32_t o0 = in->color_offsets[AVN_COLOR_OVERFLOW_INDEX], o1 = in->color_offsets[AVN_COLOR_OVERFLOW_INDEX + 1];
        bool ovf_shared = false;
        for (uint32_t m = o0; m < o1 && !ovf_shared; ++m) ovf_shared = shared[in->body1[m]] || shared[in->body2[m]];
        std::vector<uint32_t> level(o1 - o0, 0u);
        uint32_t n_levels = 1;
        if (ovf_shared) {
            std::vector<uint32_t> last(N, 0u);
            n_levels = 0;
            for (uint32_t m = o0; m < o1; ++m) {
                const uint32_t a = (uint32_t)in->body1[m], b = (uint32_t)in->body2[m];
                const uint32_t lv = 1u + std::max(is_static(a) ? 0u : last[a], is_static(b) ? 0u : last[b]);
                if (!is_static(a)) last[a] = lv;
                if (!is_static(b)) last[b] = lv;
                level[m - o0] = lv - 1u;
                n_levels = std::max(n_levels, lv);
            }
            n_levels = std::max(n_levels, 1u);
        }
        bool joint_slot = false;
        for (uint32_t b = 0; b < N && !joint_slot; ++b) joint_slot = jointed[b] && shared[b];
        const uint32_t S = (uint32_t)AVN_COLOR_OVERFLOW_INDEX + n_levels + (joint_slot ? 1u : 0u);
        auto slot_of = [&](uint32_t c, uint32_t m) { return c == (uint32_t)AVN_COLOR_OVERFLOW_INDEX ? (uint32_t)AVN_COLOR_OVERFLOW_INDEX + level[m - o0] : c; };
        // sends[s][slot][r] = global bodies rank s hands to rank r after the slot
        std::vector<std::vector<std::map<uint32_t, std::vector<int32_t>>>> sends(R, std::vector<std::map<uint32_t, std::vector<int32_t>>>(S));
        for (uint32_t c = 0; c < C; ++c)
            for (uint32_t m = in->color_offsets[c]; m < in->color_offsets[c + 1]; ++m) {
                const uint32_t s = (uint32_t)m_owner[m], slot = slot_of(c, m);
                const int32_t bb[2] = {in->body1[m], in->body2[m]};
                for (int32_t b : bb) {
                    if (!shared[b]) continue;
                    for (uint32_t r = 0; r < R; ++r) if (r != s && held[r][b]) sends[s][slot][r].push_back(b);
                }
            }
        if (joint_slot)   // the joint slot: a component's shared bodies, from its owner to the other holders (ascending body index)
            for (uint32_t b = 0; b < N; ++b)
                if (jointed[b] && shared[b])
                    for (uint32_t r = 0; r < R; ++r) if ((int32_t)r != body_comp_owner[b] && held[r][b]) sends[(uint32_t)body_comp_owner[b]][S - 1][r].push_back((int32_t)b);
        avn_level2_plan* pl = new avn_level2_plan;
        pl->ranks.resize(R);
        pl->n_overflow_levels = n_levels;
        pl->joint_slot = joint_slot; pl->global_joints = J != 0;
        for (uint32_t j = 0; j < J; ++j) pl->ranks[(uint32_t)j_owner[j]].joints.push_back(j);
        for (uint32_t r = 0; r < R; ++r) {
            avn_level2_plan::Rank& k = pl->ranks[r];
            std::vector<int32_t> g2l(N, -1);
            for (uint32_t b = 0; b < N; ++b) if (held[r][b]) { g2l[b] = (int32_t)k.bodies.size(); k.bodies.push_back((int32_t)b); }
            std::vector<uint8_t> is_peer(R, 0);
            for (uint32_t c = 0; c < S; ++c) {
                for (auto& kv : sends[r][c]) is_peer[kv.first] = 1;
                for (uint32_t s = 0; s < R; ++s) if (s != r && sends[s][c].count(r)) is_peer[s] = 1;
            }
            for (uint32_t p = 0; p < R; ++p) if (is_peer[p]) k.peers.push_back((int32_t)p);
            k.send_offsets.push_back(0); k.recv_offsets.push_back(0);
            if (!k.peers.empty())
                for (uint32_t c = 0; c < S; ++c)
                    for (int32_t p : k.peers) {
                        std::vector<int32_t> a, b;
                        auto it = sends[r][c].find((uint32_t)p); if (it != sends[r][c].end()) a = it->second;
                        auto jt = sends[p][c].find(r); if (jt != sends[p][c].end()) b = jt->second;
                        std::sort(a.begin(), a.end()); std::sort(b.begin(), b.end());
                        for (int32_t g : a) k.send_bodies.push_back(g2l[g]);
                        for (int32_t g : b) k.recv_bodies.push_back(g2l[g]);
                        k.send_offsets.push_back((uint32_t)k.send_bodies.size()); k.recv_offsets.push_back((uint32_t)k.recv_bodies.size());
                    }
            k.color_offsets.assign(C + 1, 0);
            for (uint32_t c = 0; c < C; ++c) {
                for (uint32_t m = in->color_offsets[c]; m < in->color_offsets[c + 1]; ++m)
                    if ((uint32_t)m_owner[m] == r) { k.manifolds.push_back(m); if (c == (uint32_t)AVN_COLOR_OVERFLOW_INDEX) k.overflow_level.push_back(level[m - o0]); }
                k.color_offsets[c + 1] = (uint32_t)k.manifolds.size();
            }
        }
        *out = pl;
        return AVN_OK;
    } catch (const std::bad_alloc&) { return AVN_ERR_OOM; } catch (...) { return AVN_ERR_STATE; }
}

extern "C" {

AVN_API avn_status avn_level2_plan_create(const avn_level2_in* in, avn_level2_plan** out) { return plan_create(in, nullptr, out); }
AVN_API avn_status avn_level2_plan_create_joints(const avn_level2_in* in, const avn_level2_joints* joints, avn_level2_plan** out) {
    if (!joints) return AVN_ERR_BAD_ARG;
    return plan_create(in, joints, out);
}
AVN_API avn_status avn_level2_plan_rank_joints(const avn_level2_plan* plan, uint32_t rank, uint32_t* n_joints, const uint32_t** joints, uint32_t* joint_slot, uint32_t* global_joints) {
    if (!plan || rank >= plan->ranks.size() || !n_joints || !joints || !joint_slot || !global_joints) return AVN_ERR_BAD_ARG;
    *n_joints = (uint32_t)plan->ranks[rank].joints.size(); *joints = plan->ranks[rank].joints.data();
    *joint_slot = plan->joint_slot ? 1u : 0u; *global_joints = plan->global_joints ? 1u : 0u;
    return AVN_OK;
}

AVN_API void avn_level2_plan_destroy(avn_level2_plan* plan) { delete plan; }

AVN_API avn_status avn_slab_select(const avn_slab_in* in, uint32_t* local, uint8_t* owned, uint32_t* n_local, uint32_t* next_order, uint32_t* n_next) {
    if (!in || !local || !owned || !n_local || in->n_ranks == 0 || in->rank >= in->n_ranks || (in->n_colliders && (!in->aabb_min_x || !in->aabb_max_x)) ||
        (in->n_prev && !in->prev_order) || (next_order && !n_next))
        return AVN_ERR_BAD_ARG;
    try {
        const uint32_t n = in->n_colliders, R = in->n_ranks;
        for (uint32_t i = 0; i < in->n_prev; ++i) if (in->prev_order[i] >= n) return AVN_ERR_BAD_ARG;
        // the persistent order before this frame's sort: prev_order, then the colliders it does not know yet
        std::vector<uint32_t> po;
        po.reserve(n);
        std::vector<uint8_t> seen(n, 0);
        if (in->prev_order) for (uint32_t i = 0; i < in->n_prev; ++i) { if (seen[in->prev_order[i]]) return AVN_ERR_BAD_ARG; seen[in->prev_order[i]] = 1; po.push_back(in->prev_order[i]); }
        for (uint32_t c = 0; c < n; ++c) if (!seen[c]) po.push_back(c);
        // slab boundaries on key values
        std::vector<double> xs(in->aabb_min_x, in->aabb_min_x + n);
        std::stable_sort(xs.begin(), xs.end(), [](double a, double b) { return a < b || (b != b && a == a); });   // NaN last, like a numeric sort
        const double inf = 1.0 / 0.0;
        std::vector<double> splits(R + 1, inf);
        splits[0] = -inf;
        for (uint32_t r = 1; r < R; ++r) splits[r] = n ? xs[std::min<size_t>(n - 1, (size_t)n * r / R)] : inf;
        for (uint32_t r = 1; r <= R; ++r) splits[r] = std::max(splits[r], splits[r - 1]);
        auto slab_of = [&](double x) {   // splits[s] <= x < splits[s + 1], clipped
            int64_t s = (int64_t)(std::upper_bound(splits.begin(), splits.end(), x) - splits.begin()) - 1;
            return (uint32_t)std::min<int64_t>(std::max<int64_t>(s, 0), (int64_t)R - 1);
        };
        double reach = -inf;
        bool any = false;
        for (uint32_t c = 0; c < n; ++c)
            if (slab_of(in->aabb_min_x[c]) == in->rank) { any = true; const double m = in->aabb_max_x[c]; if (m != m || reach != reach) reach = 0.0 / 0.0; else reach = std::max(reach, m); }
        uint32_t k = 0;
        if (any)
            for (uint32_t c : po) {
                const uint32_t s = slab_of(in->aabb_min_x[c]);
                const bool own = s == in->rank, halo = s > in->rank && in->aabb_min_x[c] <= reach;
                if (own || halo) { local[k] = c; owned[k] = own ? 1 : 0; ++k; }
            }
        *n_local = k;
        if (next_order) {
            std::vector<std::pair<double, uint32_t>> keyed;   // (key, position in po): a stable sort on the key alone
            std::vector<uint32_t> fin;
            for (uint32_t c : po) { const double x = in->aabb_min_x[c] + 0.0; if (x - x == 0.0) fin.push_back(c); }   // finite keys only
            std::stable_sort(fin.begin(), fin.end(), [&](uint32_t a, uint32_t b) { return in->aabb_min_x[a] + 0.0 < in->aabb_min_x[b] + 0.0; });
            for (size_t i = 0; i < fin.size(); ++i) next_order[i] = fin[i];
            *n_next = (uint32_t)fin.size();
        }
        return AVN_OK;
    } catch (const std::bad_alloc&) { return AVN_ERR_OOM; } catch (...) { return AVN_ERR_STATE; }
}

AVN_API avn_status avn_interval_orders_merge(uint32_t n_lists, const uint32_t* const* entities, const double* const* keys, const uint32_t* lengths, uint32_t* out, uint32_t* n_out) {
    if (!n_out || (n_lists && (!entities || !keys || !lengths))) return AVN_ERR_BAD_ARG;
    try {
        std::vector<uint32_t> head(n_lists, 0);
        std::vector<uint32_t> merged;
        uint32_t max_e = 0;
        for (uint32_t l = 0; l < n_lists; ++l) for (uint32_t i = 0; i < lengths[l]; ++i) max_e = std::max(max_e, entities[l][i]);
        std::vector<uint8_t> seen((size_t)max_e + 1, 0);
        const double ninf = -1.0 / 0.0;
        auto key_of = [&](uint32_t l, uint32_t i) { const double k = keys[l][i]; return k != k ? ninf : k; };
        for (;;) {
            int best = -1;
            for (uint32_t l = 0; l < n_lists; ++l) {
                if (head[l] >= lengths[l]) continue;
                if (best < 0) { best = (int)l; continue; }
                const double ka = key_of(l, head[l]), kb = key_of((uint32_t)best, head[best]);
                const uint32_t ea = entities[l][head[l]], eb = entities[best][head[best]];
                if (ka < kb || (ka == kb && ea < eb)) best = (int)l;
            }
            if (best < 0) break;
            const uint32_t e = entities[best][head[best]++];
            if (!seen[e]) { seen[e] = 1; merged.push_back(e); }
        }
        if (out) for (size_t i = 0; i < merged.size(); ++i) out[i] = merged[i];
        *n_out = (uint32_t)merged.size();
        return AVN_OK;
    } catch (const std::bad_alloc&) { return AVN_ERR_OOM; } catch (...) { return AVN_ERR_STATE; }
}

AVN_API avn_status avn_level2_plan_rank(const avn_level2_plan* plan, uint32_t rank, avn_level2_rank* out) {
    if (!plan || !out || rank >= plan->ranks.size()) return AVN_ERR_BAD_ARG;
    const avn_level2_plan::Rank& k = plan->ranks[rank];
    out->n_bodies = (uint32_t)k.bodies.size(); out->bodies = k.bodies.data();
    out->n_manifolds = (uint32_t)k.manifolds.size(); out->manifolds = k.manifolds.data();
    out->color_offsets = k.color_offsets.data();
    out->halo.n_peers = (uint32_t)k.peers.size(); out->halo.peer_rank = k.peers.data();
    out->halo.send_offsets = k.send_offsets.data(); out->halo.send_bodies = k.send_bodies.data();
    out->halo.recv_offsets = k.recv_offsets.data(); out->halo.recv_bodies = k.recv_bodies.data();
    return AVN_OK;
}

AVN_API avn_status avn_level2_plan_rank_overflow(const avn_level2_plan* plan, uint32_t rank, uint32_t* n_levels, const uint32_t** level_of_local_overflow_manifold) {
    if (!plan || rank >= plan->ranks.size() || !n_levels || !level_of_local_overflow_manifold) return AVN_ERR_BAD_ARG;
    *n_levels = plan->n_overflow_levels;
    *level_of_local_overflow_manifold = plan->ranks[rank].overflow_level.data();
    return AVN_OK;
}

}  // extern "C"
