// avn_level2.cpp — the level-2 sharding planner behind the C ABI (include/avian_mi355x.h: avn_level2_plan_*).  Host integer work only.
#include <algorithm>
#include <map>
#include <new>
#include <string>
#include <vector>

#include "../../include/avian_mi355x.h"

struct avn_level2_plan {
    struct Rank {
        std::vector<int32_t> bodies;
        std::vector<uint32_t> manifolds, color_offsets;
        std::vector<int32_t> peers, send_bodies, recv_bodies;
        std::vector<uint32_t> send_offsets, recv_offsets;
    };
    std::vector<Rank> ranks;
};

extern "C" {

AVN_API avn_status avn_level2_plan_create(const avn_level2_in* in, avn_level2_plan** out) {
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
        std::vector<uint8_t> shared(N, 0);
        for (uint32_t b = 0; b < N; ++b) {
            if (is_static(b)) continue;
            uint32_t k = 0;
            for (uint32_t r = 0; r < R; ++r) k += held[r][b];
            shared[b] = k > 1;
        }
        // sends[s][c][r] = global bodies rank s hands to rank r after colour c
        std::vector<std::vector<std::map<uint32_t, std::vector<int32_t>>>> sends(R, std::vector<std::map<uint32_t, std::vector<int32_t>>>(C));
        for (uint32_t c = 0; c < C; ++c)
            for (uint32_t m = in->color_offsets[c]; m < in->color_offsets[c + 1]; ++m) {
                const uint32_t s = (uint32_t)m_owner[m];
                const int32_t bb[2] = {in->body1[m], in->body2[m]};
                for (int32_t b : bb) {
                    if (!shared[b]) continue;
                    if (c == AVN_COLOR_OVERFLOW_INDEX) return AVN_ERR_BAD_ARG;   // solved serially across worlds: not supported
                    for (uint32_t r = 0; r < R; ++r) if (r != s && held[r][b]) sends[s][c][r].push_back(b);
                }
            }
        avn_level2_plan* pl = new avn_level2_plan;
        pl->ranks.resize(R);
        for (uint32_t r = 0; r < R; ++r) {
            avn_level2_plan::Rank& k = pl->ranks[r];
            std::vector<int32_t> g2l(N, -1);
            for (uint32_t b = 0; b < N; ++b) if (held[r][b]) { g2l[b] = (int32_t)k.bodies.size(); k.bodies.push_back((int32_t)b); }
            std::vector<uint8_t> is_peer(R, 0);
            for (uint32_t c = 0; c < C; ++c) {
                for (auto& kv : sends[r][c]) is_peer[kv.first] = 1;
                for (uint32_t s = 0; s < R; ++s) if (s != r && sends[s][c].count(r)) is_peer[s] = 1;
            }
            for (uint32_t p = 0; p < R; ++p) if (is_peer[p]) k.peers.push_back((int32_t)p);
            k.send_offsets.push_back(0); k.recv_offsets.push_back(0);
            if (!k.peers.empty())
                for (uint32_t c = 0; c < C; ++c)
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
                for (uint32_t m = in->color_offsets[c]; m < in->color_offsets[c + 1]; ++m) if ((uint32_t)m_owner[m] == r) k.manifolds.push_back(m);
                k.color_offsets[c + 1] = (uint32_t)k.manifolds.size();
            }
        }
        *out = pl;
        return AVN_OK;
    } catch (const std::bad_alloc&) { return AVN_ERR_OOM; } catch (...) { return AVN_ERR_STATE; }
}

AVN_API void avn_level2_plan_destroy(avn_level2_plan* plan) { delete plan; }

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

}  // extern "C"
