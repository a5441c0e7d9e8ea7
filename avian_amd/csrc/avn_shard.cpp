// avn_shard.cpp -- the closed loop sharded by ISLANDS: the replicated integer bookkeeping, host C++ behind the ABI (include/avian_mi355x.h, "sharded closed
// loop"; round 4 had it as a Python prototype: per-pair loops, heapq, dicts, pickled payloads).
//
// Why it is replicated: the reference's IdPool hands out the LOWEST free ContactId in the broad phase's GLOBAL emission order
// (data_structures/id_pool.rs:31-40, collision/broad_phase.rs:387-388, 443-468), NarrowPhase::update walks the status changes in ascending id
// (collision/narrow_phase/system_param.rs:141-145), and pop_manifold's swap_remove moves the LAST handle of a colour's list into the hole
// (dynamics/solver/constraint_graph.rs:245-296): an island's colours and the order of its overflow manifolds depend on what OTHER islands freed and
// popped.  So every rank replays everything that is integer -- the global AabbIntervals order (a stable sort of last frame's order by this frame's
// min.x keys), the ContactIds, the ConstraintGraph with all colour lists -- from flat arrays the ranks all-gather (12 bytes per collider, 24 bytes per
// new pair, 16 bytes per status change), and runs only its own islands' physics through the low-level ABI.  A rank's handle lists are the global
// colour lists restricted to its own pairs: a restriction keeps the relative order, which is all the overflow colour's serial solve needs.
#include <algorithm>
#include <cstring>
#include <new>
#include <numeric>
#include <queue>
#include <string>
#include <unordered_map>
#include <vector>

#include "avn_world.hpp"

struct avn_shard {
    uint32_t rank = 0, n_colliders = 0;
    std::unordered_map<uint32_t, uint32_t> slot_of_entity;   // collider Entity::index() -> global slot (upload order)
    std::vector<uint32_t> order, gpos;                       // the replicated AabbIntervals order and its inverse
    std::vector<double> minx;
    avn::ConstraintGraphHost graph;                          // the replicated ConstraintGraph
    std::priority_queue<uint32_t, std::vector<uint32_t>, std::greater<uint32_t>> free_ids;   // IdPool: lowest free id first
    uint32_t next_id = 0;
    struct Pair { uint32_t c1 = 0, c2 = 0; int32_t b1 = -1, b2 = -1; uint32_t owner = 0; uint32_t n_handles = 0; bool live = false; };
    std::vector<Pair> pairs;                                 // by ContactId
    std::vector<uint32_t> active;                            // this rank's active pairs, in the order they were added
    // results of the last phase
    std::vector<uint32_t> new_ids, new_c1, new_c2, new_flags, removed_local, loc_handles, glob_handles;
    uint32_t loc_off[AVN_GRAPH_COLOR_COUNT + 1] = {0}, glob_off[AVN_GRAPH_COLOR_COUNT + 1] = {0};
    avn_shard_stats stats{};
    std::string error;
    struct Rec { uint32_t p1, p2; avn_shard_pair q; };
    std::vector<Rec> recs;
    std::vector<avn_contact_change> ch;
};

extern "C" {

AVN_API avn_status avn_shard_create(uint32_t n_colliders, const uint32_t* collider_entities, uint32_t rank, avn_shard** out) {
    if (!out || (n_colliders && !collider_entities)) return AVN_ERR_BAD_ARG;
    avn_shard* s = new (std::nothrow) avn_shard();
    if (!s) return AVN_ERR_OOM;
    try {
        s->rank = rank; s->n_colliders = n_colliders;
        s->order.resize(n_colliders); std::iota(s->order.begin(), s->order.end(), 0u);
        s->gpos.resize(n_colliders); s->minx.assign(n_colliders, 0.0);
        for (uint32_t i = 0; i < n_colliders; ++i) s->slot_of_entity.emplace(collider_entities[i], i);
        if (s->slot_of_entity.size() != n_colliders) { delete s; return AVN_ERR_BAD_ARG; }
    } catch (...) { delete s; return AVN_ERR_OOM; }
    *out = s;
    return AVN_OK;
}
AVN_API void avn_shard_destroy(avn_shard* s) { delete s; }
AVN_API const char* avn_shard_last_error(const avn_shard* s) { return s ? s->error.c_str() : ""; }

#define SH_TRY if (!s) return AVN_ERR_BAD_ARG; try
#define SH_CATCH catch (const std::bad_alloc&) { s->error = "out of host memory"; return AVN_ERR_OOM; } catch (...) { s->error = "unexpected C++ exception"; return AVN_ERR_STATE; }

// phase 2: the global interval order from every collider's min.x key, ContactIds for the ranks' new pairs in the GLOBAL emission order
AVN_API avn_status avn_shard_phase2(avn_shard* s, const uint32_t* key_collider, const double* key_min_x, size_t n_keys, const avn_shard_pair* pairs, size_t n_pairs) {
    SH_TRY {
        if ((n_keys && (!key_collider || !key_min_x)) || (n_pairs && !pairs)) { s->error = "shard_phase2: null array"; return AVN_ERR_BAD_ARG; }
        // Everything is validated on COPIES first; the replicated state (interval order, IdPool, pair table, active list) is only touched once the whole call is
        // known to go through -- a rank that fails here must not be left half a step ahead of the others.
        for (size_t i = 0; i < n_keys; ++i)
            if (key_collider[i] >= s->n_colliders) { s->error = "shard_phase2: collider slot out of range"; return AVN_ERR_BAD_ARG; }
        for (size_t i = 0; i < n_pairs; ++i)
            if (!s->slot_of_entity.count(pairs[i].collider1) || !s->slot_of_entity.count(pairs[i].collider2)) { s->error = "shard_phase2: a pair names an unknown collider"; return AVN_ERR_BAD_ARG; }
        std::vector<double> minx = s->minx;
        for (size_t i = 0; i < n_keys; ++i) minx[key_collider[i]] = key_min_x[i];
        // sweep_and_prune's insertion sort (broad_phase.rs:373-387, 479-487) is a STABLE sort of last frame's order by this frame's min.x
        std::vector<uint32_t> order = s->order, gpos(s->n_colliders);
        std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return minx[a] < minx[b]; });
        for (uint32_t i = 0; i < s->n_colliders; ++i) gpos[order[i]] = i;
        s->recs.clear(); s->recs.reserve(n_pairs);
        for (size_t i = 0; i < n_pairs; ++i) {
            const uint32_t p1 = gpos[s->slot_of_entity.find(pairs[i].collider1)->second], p2 = gpos[s->slot_of_entity.find(pairs[i].collider2)->second];
            if (p1 >= p2) { s->error = "shard_phase2: collider1 must be the earlier interval of a pair"; return AVN_ERR_STATE; }
            s->recs.push_back({p1, p2, pairs[i]});
        }
        // pairs are emitted i-major over the sorted intervals, j ascending (broad_phase.rs:387-388)
        std::sort(s->recs.begin(), s->recs.end(), [](const avn_shard::Rec& x, const avn_shard::Rec& y) { return x.p1 != y.p1 ? x.p1 < y.p1 : x.p2 < y.p2; });
        // ---- commit ----
        s->minx.swap(minx); s->order.swap(order); s->gpos.swap(gpos);
        s->new_ids.clear(); s->new_c1.clear(); s->new_c2.clear(); s->new_flags.clear();
        for (const avn_shard::Rec& r : s->recs) {
            uint32_t cid;
            if (!s->free_ids.empty()) { cid = s->free_ids.top(); s->free_ids.pop(); } else cid = s->next_id++;
            if (s->pairs.size() <= cid) s->pairs.resize(std::max<size_t>((size_t)cid + 1, s->pairs.size() + s->pairs.size() / 2));
            avn_shard::Pair& p = s->pairs[cid];
            p.c1 = r.q.collider1; p.c2 = r.q.collider2; p.b1 = r.q.body1; p.b2 = r.q.body2; p.owner = r.q.owner; p.n_handles = 0; p.live = true;
            if (r.q.owner == s->rank) {
                s->new_ids.push_back(cid); s->new_c1.push_back(r.q.collider1); s->new_c2.push_back(r.q.collider2); s->new_flags.push_back(r.q.flags);
                s->active.push_back(cid);
            }
        }
        s->stats.pairs_added += (uint32_t)std::min<size_t>(n_pairs, 0xFFFFFFFFu);
        s->stats.next_id = s->next_id; s->stats.n_free = (uint32_t)s->free_ids.size();
        return AVN_OK;
    } SH_CATCH
}
AVN_API avn_status avn_shard_new_local_pairs(avn_shard* s, const uint32_t** ids, const uint32_t** c1, const uint32_t** c2, const uint32_t** flags, size_t* n) {
    if (!s || !ids || !c1 || !c2 || !flags || !n) return AVN_ERR_BAD_ARG;
    *ids = s->new_ids.data(); *c1 = s->new_c1.data(); *c2 = s->new_c2.data(); *flags = s->new_flags.data(); *n = s->new_ids.size();
    return AVN_OK;
}
AVN_API avn_status avn_shard_active(avn_shard* s, const uint32_t** ids, size_t* n) {
    if (!s || !ids || !n) return AVN_ERR_BAD_ARG;
    *ids = s->active.data(); *n = s->active.size();
    return AVN_OK;
}
// phase 3: every rank's status changes on the replicated graph in ascending ContactId (system_param.rs:141-389), the removals, the handle lists
AVN_API avn_status avn_shard_phase3(avn_shard* s, const avn_contact_change* changes, size_t n) {
    SH_TRY {
        if (n && !changes) { s->error = "shard_phase3: null array"; return AVN_ERR_BAD_ARG; }
        s->ch.assign(changes, changes + n);
        std::stable_sort(s->ch.begin(), s->ch.end(), [](const avn_contact_change& a, const avn_contact_change& b) { return a.contact_id < b.contact_id; });
        s->removed_local.clear();
        std::vector<uint32_t> removed;
        for (const avn_contact_change& c : s->ch) {
            const uint32_t cid = c.contact_id, flags = c.flags;
            if (cid >= s->pairs.size() || !s->pairs[cid].live) { s->error = "shard_phase3: a status change names a contact id that is not live"; return AVN_ERR_STATE; }
            avn_shard::Pair& p = s->pairs[cid];
            const bool generates = flags & AVN_CP_GENERATE_CONSTRAINTS, touching = flags & AVN_CP_TOUCHING;
            auto push = [&](uint32_t k) {
                for (uint32_t i = 0; i < k; ++i) {
                    s->graph.push_manifold(((uint64_t)cid << 8) | p.n_handles, (uint32_t)p.b1, (uint32_t)p.b2, flags & AVN_CP_STATIC1, flags & AVN_CP_STATIC2);
                    ++p.n_handles; ++s->stats.pushes;
                }
            };
            auto pop = [&](uint32_t k) {
                for (uint32_t i = 0; i < k && p.n_handles; ++i) { --p.n_handles; s->graph.pop_manifold(((uint64_t)cid << 8) | p.n_handles); ++s->stats.pops; }
            };
            if (flags & AVN_CP_DISJOINT_AABB) {
                if (generates) pop(p.n_handles);
                removed.push_back(cid);
                if (p.owner == s->rank) s->removed_local.push_back(cid);
            } else if (flags & AVN_CP_STARTED_TOUCHING) { if (generates) push(c.manifold_count); }
            else if (flags & AVN_CP_STOPPED_TOUCHING) { if (generates) pop(p.n_handles); }
            else if (touching && (flags & AVN_CP_STARTED_GENERATING_CONSTRAINTS)) push(c.manifold_count);
            else if (touching && generates && c.manifold_count_change > 0) push((uint32_t)c.manifold_count_change);
            else if (touching && generates && c.manifold_count_change < 0) pop((uint32_t)(-c.manifold_count_change));
        }
        if (!s->removed_local.empty()) {
            std::vector<uint8_t> gone(s->pairs.size(), 0);
            for (uint32_t cid : s->removed_local) gone[cid] = 1;
            s->active.erase(std::remove_if(s->active.begin(), s->active.end(), [&](uint32_t a) { return gone[a] != 0; }), s->active.end());
        }
        for (uint32_t cid : removed) { s->pairs[cid] = avn_shard::Pair(); s->free_ids.push(cid); }
        s->stats.pairs_removed += (uint32_t)removed.size();
        s->stats.next_id = s->next_id; s->stats.n_free = (uint32_t)s->free_ids.size();
        // GraphColor::manifold_handles of all colours, and their restriction to this rank's pairs (order kept)
        s->glob_handles.clear(); s->loc_handles.clear();
        for (int c = 0; c < AVN_GRAPH_COLOR_COUNT; ++c) {
            s->glob_off[c] = (uint32_t)s->glob_handles.size(); s->loc_off[c] = (uint32_t)s->loc_handles.size();
            for (const avn::ConstraintGraphHost::Handle& h : s->graph.colors[c].manifold_handles) {
                const uint32_t cid = (uint32_t)(h.handle >> 8);
                s->glob_handles.push_back(cid);
                if (s->pairs[cid].owner == s->rank) s->loc_handles.push_back(cid);
            }
        }
        s->glob_off[AVN_GRAPH_COLOR_COUNT] = (uint32_t)s->glob_handles.size(); s->loc_off[AVN_GRAPH_COLOR_COUNT] = (uint32_t)s->loc_handles.size();
        s->stats.last_status_changes = (uint32_t)n;
        return AVN_OK;
    } SH_CATCH
}
AVN_API avn_status avn_shard_removed_local(avn_shard* s, const uint32_t** ids, size_t* n) {
    if (!s || !ids || !n) return AVN_ERR_BAD_ARG;
    *ids = s->removed_local.data(); *n = s->removed_local.size();
    return AVN_OK;
}
AVN_API avn_status avn_shard_handles(avn_shard* s, int global, uint32_t* offsets, const uint32_t** ids, size_t* n) {
    if (!s || !offsets || !ids || !n) return AVN_ERR_BAD_ARG;
    std::memcpy(offsets, global ? s->glob_off : s->loc_off, sizeof s->loc_off);
    const std::vector<uint32_t>& v = global ? s->glob_handles : s->loc_handles;
    *ids = v.data(); *n = v.size();
    return AVN_OK;
}
AVN_API avn_status avn_shard_stats_get(avn_shard* s, avn_shard_stats* o) {
    if (!s || !o) return AVN_ERR_BAD_ARG;
    *o = s->stats;
    return AVN_OK;
}

}  // extern "C"
