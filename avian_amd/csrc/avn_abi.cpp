// avn_abi.cpp — the extern "C" boundary (include/avian_mi355x.h) over the C++ host world.
// No exception or HIP error crosses it: every entry point returns avn_status.
#include <algorithm>
#include <cmath>
#include <new>
#include <numeric>
#include <string>
#include <vector>

#include "avn_world.hpp"

struct avn_world { avn::WorldBase* impl; };
struct avn_constraint_graph { avn::ConstraintGraphHost g; };
static thread_local std::string g_create_error;

#define GUARD(expr)                                                     \
    do {                                                                \
        if (!w || !w->impl) return AVN_ERR_BAD_ARG;                     \
        try { w->impl->bind(); return w->impl->expr; }                                   \
        catch (const std::bad_alloc&) { w->impl->error = "out of host memory"; return AVN_ERR_OOM; } \
        catch (...) { w->impl->error = "unexpected C++ exception"; return AVN_ERR_STATE; }           \
    } while (0)

// entry points that can change the state a step starts from (everything but avn_step, avn_synchronize and the *_get / *_download calls)
#define GUARD_MUT(expr)                                                 \
    do {                                                                \
        if (!w || !w->impl) return AVN_ERR_BAD_ARG;                     \
        try { w->impl->bind(); w->impl->touched(); return w->impl->expr; }               \
        catch (const std::bad_alloc&) { w->impl->error = "out of host memory"; return AVN_ERR_OOM; } \
        catch (...) { w->impl->error = "unexpected C++ exception"; return AVN_ERR_STATE; }           \
    } while (0)

extern "C" {

AVN_API avn_status avn_world_create(const avn_config* cfg, avn_world** out) {
    if (!cfg || !out) { g_create_error = "world_create: null argument"; return AVN_ERR_BAD_ARG; }
    if (cfg->struct_size != sizeof(avn_config)) { g_create_error = "world_create: struct_size mismatch"; return AVN_ERR_BAD_ARG; }
    *out = nullptr;
    avn_status st = AVN_OK;
    avn::WorldBase* w = nullptr;
    try {
        if (cfg->scalar_bits == 32) w = avn::make_world_f32(cfg, &st, &g_create_error);
        else if (cfg->scalar_bits == 64) w = avn::make_world_f64(cfg, &st, &g_create_error);
        else { g_create_error = "world_create: scalar_bits must be 32 or 64"; return AVN_ERR_BAD_ARG; }
    } catch (...) { g_create_error = "world_create: unexpected C++ exception"; return AVN_ERR_OOM; }
    if (!w) return st;
    *out = new (std::nothrow) avn_world{w};
    if (!*out) { delete w; return AVN_ERR_OOM; }
    return AVN_OK;
}
AVN_API void avn_world_destroy(avn_world* w) { if (w) { if (w->impl) w->impl->bind(); delete w->impl; delete w; } }
AVN_API const char* avn_last_error(const avn_world* w) { return (w && w->impl) ? w->impl->error.c_str() : g_create_error.c_str(); }
AVN_API avn_status avn_config_set(avn_world* w, const avn_config* c) { GUARD_MUT(config_set(c)); }
AVN_API avn_status avn_bodies_upload(avn_world* w, const avn_bodies* b) { GUARD_MUT(bodies_upload(b)); }
AVN_API avn_status avn_bodies_download(avn_world* w, const avn_bodies_out* o) { GUARD(bodies_download(o)); }
AVN_API avn_status avn_solver_bodies_download(avn_world* w, const avn_solver_bodies_out* o) { GUARD(solver_bodies_download(o)); }
AVN_API avn_status avn_manifolds_upload(avn_world* w, const avn_manifolds* m) { GUARD_MUT(manifolds_upload(m)); }
AVN_API avn_status avn_impulses_download(avn_world* w, const avn_impulses_out* o) { GUARD(impulses_download(o)); }
AVN_API avn_status avn_constraints_download(avn_world* w, const avn_constraints_out* o) { GUARD(constraints_download(o)); }
AVN_API avn_status avn_distance_joints_upload(avn_world* w, const avn_distance_joints* j) { GUARD_MUT(distance_joints_upload(j)); }
AVN_API avn_status avn_joints_upload(avn_world* w, const avn_joints* j) { GUARD_MUT(joints_upload(j)); }
AVN_API avn_status avn_joints_download(avn_world* w, const avn_joints_out* o) { GUARD(joints_download(o)); }
AVN_API avn_status avn_colliders_upload(avn_world* w, const avn_colliders* c) { GUARD_MUT(colliders_upload(c)); }
AVN_API avn_status avn_collider_transforms_upload(avn_world* w, const avn_collider_transforms* t) { GUARD_MUT(collider_transforms_upload(t)); }
AVN_API avn_status avn_local_accelerations_upload(avn_world* w, uint32_t count, const void* linear, const void* angular) { GUARD_MUT(local_accelerations_upload(count, linear, angular)); }
AVN_API avn_status avn_existing_pairs_upload(avn_world* w, const uint64_t* k, size_t n) { GUARD_MUT(existing_pairs_upload(k, n)); }
AVN_API avn_status avn_pairs_get(avn_world* w, const avn_pair** o, size_t* n) { GUARD(pairs_get(o, n)); }
AVN_API avn_status avn_aabbs_download(avn_world* w, void* mn, void* mx, uint32_t* e, size_t* n) { GUARD(aabbs_download(mn, mx, e, n)); }
AVN_API avn_status avn_run_system(avn_world* w, avn_system s) { GUARD_MUT(run_system(s)); }
AVN_API avn_status avn_step(avn_world* w) { GUARD(step()); }
AVN_API avn_status avn_synchronize(avn_world* w) { GUARD(synchronize()); }
AVN_API avn_status avn_timers_get(avn_world* w, avn_timers* t) { GUARD(timers(t)); }
AVN_API avn_status avn_diagnostics_get(avn_world* w, avn_diagnostics* d) { GUARD(diagnostics(d)); }
AVN_API avn_status avn_profile_system(avn_world* w, avn_system s, uint32_t r, double* ms, uint32_t* l) { GUARD_MUT(profile_system(s, r, ms, l)); }
AVN_API avn_status avn_dynamic_bounds(avn_world* w, double* mn, double* mx) { GUARD(dynamic_bounds(mn, mx)); }
AVN_API avn_status avn_contact_manifolds(avn_world* w, const avn_shape_pairs* p, const avn_query_manifolds_out* o) { GUARD(contact_manifolds(p, o)); }
AVN_API avn_status avn_collider_materials_upload(avn_world* w, const avn_collider_materials* m) { GUARD_MUT(collider_materials_upload(m)); }
AVN_API avn_status avn_contact_pairs_add(avn_world* w, const avn_contact_pairs* p) { GUARD_MUT(contact_pairs_add(p)); }
AVN_API avn_status avn_contact_pairs_remove(avn_world* w, const uint32_t* ids, size_t n) { GUARD_MUT(contact_pairs_remove(ids, n)); }
AVN_API avn_status avn_active_pairs_set(avn_world* w, const uint32_t* ids, size_t n) { GUARD_MUT(active_pairs_set(ids, n)); }
AVN_API avn_status avn_contact_changes_get(avn_world* w, const avn_contact_change** o, size_t* n) { GUARD(contact_changes_get(o, n)); }
AVN_API avn_status avn_manifold_handles_upload(avn_world* w, const uint32_t* off, const uint32_t* ids) { GUARD_MUT(manifold_handles_upload(off, ids)); }
AVN_API avn_status avn_contacts_download(avn_world* w, const uint32_t* ids, size_t n, const avn_contacts_out* o) { GUARD(contacts_download(ids, n, o)); }
AVN_API avn_status avn_contacts_upload(avn_world* w, const uint32_t* ids, size_t n, const avn_contacts_in* in) { GUARD_MUT(contacts_upload(ids, n, in)); }
AVN_API avn_status avn_pipeline_enable(avn_world* w, int on) { GUARD_MUT(pipeline_enable(on)); }
AVN_API avn_status avn_pipeline_stats_get(avn_world* w, avn_pipeline_stats* o) { GUARD(pipeline_stats_get(o)); }
AVN_API avn_status avn_pipeline_handles_get(avn_world* w, uint32_t* off, const uint32_t** ids, size_t* n) { GUARD(pipeline_handles_get(off, ids, n)); }
AVN_API avn_status avn_pipeline_new_pair_ids_get(avn_world* w, const uint32_t** ids, size_t* n) { GUARD(pipeline_new_pair_ids_get(ids, n)); }

// Interaction islands + x-slab assignment (host integer work; see the header).  Union-find with path halving; islands
// are numbered by their smallest body index; slabs cut the islands, ordered by mean x then id, at equal cumulative weight.
AVN_API avn_status avn_islands_partition(const avn_islands_in* in, int32_t* island_of_body, int32_t* rank_of_body, uint32_t* n_islands) {
    if (!in || !island_of_body || !rank_of_body || !n_islands || in->n_ranks == 0 || (in->n_bodies && (!in->rb_type || !in->center_x)) ||
        (in->n_edges && (!in->edge_body1 || !in->edge_body2)))
        return AVN_ERR_BAD_ARG;
    try {
        const uint32_t n = in->n_bodies;
        std::vector<int32_t> parent(n);
        std::iota(parent.begin(), parent.end(), 0);
        auto find = [&](int32_t x) { while (parent[x] != x) { parent[x] = parent[parent[x]]; x = parent[x]; } return x; };
        auto is_static = [&](int32_t b) { return in->rb_type[b] == AVN_RB_STATIC; };
        std::vector<uint32_t> edge_count(n, 0);
        for (uint32_t e = 0; e < in->n_edges; ++e) {
            int32_t a = in->edge_body1[e], b = in->edge_body2[e];
            if (a < 0 || b < 0 || (uint32_t)a >= n || (uint32_t)b >= n) return AVN_ERR_BAD_ARG;
            bool sa = is_static(a), sb = is_static(b);
            if (sa && sb) continue;
            if (!sa && !sb) { int32_t ra = find(a), rb = find(b); if (ra != rb) { if (ra < rb) parent[rb] = ra; else parent[ra] = rb; } }
            edge_count[sa ? b : a] += 1;  // the edge is carried by (one of) its non-static bodies
        }
        // number the islands by smallest member (roots are the smallest index because unions keep the smaller root)
        std::vector<int32_t> id_of_root(n, -1);
        uint32_t k = 0;
        for (uint32_t b = 0; b < n; ++b) {
            if (is_static((int32_t)b)) { island_of_body[b] = -1; continue; }
            int32_t r = find((int32_t)b);
            if (id_of_root[r] < 0) id_of_root[r] = (int32_t)k++;
            island_of_body[b] = id_of_root[r];
        }
        *n_islands = k;
        std::vector<double> sum_x(k, 0.0);
        std::vector<uint64_t> weight(k, 0), members(k, 0);
        for (uint32_t b = 0; b < n; ++b) {
            int32_t i = island_of_body[b];
            if (i < 0) continue;
            sum_x[i] += in->center_x[b]; members[i] += 1; weight[i] += 1 + edge_count[b];
        }
        std::vector<uint32_t> order(k);
        std::iota(order.begin(), order.end(), 0u);
        std::vector<double> mean_x(k);
        for (uint32_t i = 0; i < k; ++i) mean_x[i] = sum_x[i] / (double)members[i];
        std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return mean_x[a] < mean_x[b]; });
        uint64_t total = 0;
        for (uint32_t i = 0; i < k; ++i) total += weight[i];
        std::vector<int32_t> rank_of_island(k, 0);
        uint64_t cum = 0;
        for (uint32_t o = 0; o < k; ++o) {
            uint32_t i = order[o];
            // the island goes to the slab that contains its weight midpoint: rank = floor((cum + w/2) * R / total)
            unsigned __int128 mid2 = (unsigned __int128)(2 * cum + weight[i]) * in->n_ranks;
            uint64_t r = total ? (uint64_t)(mid2 / ((unsigned __int128)2 * total)) : 0;
            rank_of_island[i] = (int32_t)std::min<uint64_t>(r, in->n_ranks - 1);
            cum += weight[i];
        }
        for (uint32_t b = 0; b < n; ++b) rank_of_body[b] = island_of_body[b] < 0 ? -1 : rank_of_island[island_of_body[b]];
        return AVN_OK;
    } catch (...) { return AVN_ERR_OOM; }
}

AVN_API avn_status avn_host_shapes_set(avn_world* w, avn_host_aabb_fn aabb, avn_host_manifolds_fn manifolds, void* user) { GUARD_MUT(host_shapes_set(aabb, manifolds, user)); }
AVN_API avn_status avn_host_shape_stats_get(avn_world* w, avn_host_shape_stats* out) { GUARD(host_shape_stats_get(out)); }
AVN_API avn_status avn_collision_hooks_set(avn_world* w, avn_filter_pairs_fn filter, avn_modify_contacts_fn modify, void* user) { GUARD_MUT(collision_hooks_set(filter, modify, user)); }
AVN_API avn_status avn_collision_hook_stats_get(avn_world* w, avn_collision_hook_stats* out) { GUARD(collision_hook_stats_get(out)); }
AVN_API avn_status avn_halo_plan_upload(avn_world* w, const avn_halo_plan* p) { GUARD_MUT(halo_plan_upload(p)); }
AVN_API avn_status avn_halo_overflow_levels_upload(avn_world* w, uint32_t n_levels, const uint32_t* level_of, size_t count) { GUARD_MUT(halo_overflow_levels_upload(n_levels, level_of, count)); }
AVN_API avn_status avn_halo_joint_slot_set(avn_world* w, uint32_t joint_slot, uint32_t global_joints) { GUARD_MUT(halo_joint_slot_set(joint_slot, global_joints)); }
AVN_API avn_status avn_run_color_pass(avn_world* w, avn_system pass, uint32_t color) { GUARD_MUT(run_color_pass(pass, color)); }
AVN_API avn_status avn_halo_pack(avn_world* w, uint32_t color, uint32_t peer, void* out, size_t* count) { GUARD(halo_pack(color, peer, out, count)); }
AVN_API avn_status avn_halo_unpack(avn_world* w, uint32_t color, uint32_t peer, const void* in, size_t count) { GUARD_MUT(halo_unpack(color, peer, in, count)); }
AVN_API avn_status avn_islands_get(avn_world* w, uint32_t* island_of_body, uint32_t* n_islands) { GUARD(islands_get(island_of_body, n_islands)); }
AVN_API avn_status avn_sleep_update(avn_world* w, const avn_sleep_params* p, avn_sleep_stats* st) { GUARD(sleep_update(p, st)); }
AVN_API avn_status avn_sleep_get(avn_world* w, const avn_sleep_out* o) { GUARD(sleep_get(o)); }
AVN_API avn_status avn_sleep_reset(avn_world* w, const uint32_t* bodies, size_t n) { GUARD_MUT(sleep_reset(bodies, n)); }
AVN_API avn_status avn_bounds_exchange(avn_world* w, double* bounds, uint32_t cap_ranks, uint32_t* n_ranks, uint32_t* overlaps, uint32_t cap_overlaps, uint32_t* n_overlaps) { GUARD(bounds_exchange(bounds, cap_ranks, n_ranks, overlaps, cap_overlaps, n_overlaps)); }
AVN_API avn_status avn_sleeping_enable(avn_world* w, const avn_sleep_params* p) { GUARD_MUT(sleeping_enable(p)); }
AVN_API avn_status avn_sleeping_stats_get(avn_world* w, avn_sleeping_stats* o) { GUARD(sleeping_stats_get(o)); }
AVN_API avn_status avn_sleeping_state_get(avn_world* w, const avn_sleeping_out* o) { GUARD(sleeping_state_get(o)); }
AVN_API avn_status avn_wake_bodies(avn_world* w, const uint32_t* bodies, size_t n) { GUARD_MUT(wake_bodies(bodies, n)); }
AVN_API avn_status avn_despawn(avn_world* w, const avn_despawn_list* d) { GUARD_MUT(despawn(d)); }
AVN_API avn_status avn_dshard_enable(avn_world* w, const avn_dshard_config* c) { GUARD_MUT(dshard_enable(c)); }
AVN_API avn_status avn_dshard_bodies_pack(avn_world* w, void* out, size_t cap, size_t* bytes) { GUARD(dshard_bodies_pack(out, cap, bytes)); }
AVN_API avn_status avn_dshard_bodies_unpack(avn_world* w, uint32_t from_rank, const void* in, size_t bytes) { GUARD_MUT(dshard_bodies_unpack(from_rank, in, bytes)); }
AVN_API avn_status avn_dshard_stats_get(avn_world* w, avn_dshard_stats* out) { GUARD(dshard_stats_get(out)); }
AVN_API avn_status avn_comm_unique_id(uint8_t* out) {
    try { return avn::comm_unique_id(out, g_create_error); }
    catch (...) { g_create_error = "unexpected C++ exception"; return AVN_ERR_STATE; }
}
AVN_API avn_status avn_comm_init(avn_world* w, const uint8_t* unique_id, int n_ranks, int rank) { GUARD(comm_init(unique_id, n_ranks, rank)); }

AVN_API uint64_t avn_pair_key(uint32_t a, uint32_t b) { return a < b ? ((uint64_t)a << 32) | b : ((uint64_t)b << 32) | a; }

AVN_API avn_status avn_constraint_graph_create(uint32_t, avn_constraint_graph** out) {
    if (!out) return AVN_ERR_BAD_ARG;
    *out = new (std::nothrow) avn_constraint_graph();
    return *out ? AVN_OK : AVN_ERR_OOM;
}
AVN_API void avn_constraint_graph_destroy(avn_constraint_graph* g) { delete g; }
AVN_API int32_t avn_constraint_graph_push(avn_constraint_graph* g, uint64_t h, uint32_t b1, uint32_t b2, int s1, int s2) {
    if (!g) return -1;
    try { return g->g.push_manifold(h, b1, b2, s1 != 0, s2 != 0); } catch (...) { return -1; }
}
AVN_API avn_status avn_constraint_graph_push_batch(avn_constraint_graph* g, size_t n, const uint64_t* h, const uint32_t* b1, const uint32_t* b2,
                                                 const uint8_t* s1, const uint8_t* s2, int8_t* colors) {
    if (!g || (n && (!h || !b1 || !b2 || !s1 || !s2))) return AVN_ERR_BAD_ARG;
    try {
        for (size_t i = 0; i < n; ++i) {
            int c = g->g.push_manifold(h[i], b1[i], b2[i], s1[i] != 0, s2[i] != 0);
            if (colors) colors[i] = (int8_t)c;
            if (c < 0) return AVN_ERR_STATE;  // duplicate handle
        }
    } catch (...) { return AVN_ERR_OOM; }
    return AVN_OK;
}
AVN_API avn_status avn_constraint_graph_pop(avn_constraint_graph* g, uint64_t h) {
    if (!g) return AVN_ERR_BAD_ARG;
    try { return g->g.pop_manifold(h) ? AVN_OK : AVN_ERR_STATE; } catch (...) { return AVN_ERR_STATE; }
}
AVN_API avn_status avn_constraint_graph_lists(const avn_constraint_graph* g, uint32_t* offsets, uint64_t* handles, size_t cap, size_t* count) {
    if (!g || !offsets || !count) return AVN_ERR_BAD_ARG;
    size_t n = 0;
    for (int c = 0; c < AVN_GRAPH_COLOR_COUNT; ++c) {
        offsets[c] = (uint32_t)n;
        for (const auto& h : g->g.colors[c].manifold_handles) { if (handles && n < cap) handles[n] = h.handle; ++n; }
    }
    offsets[AVN_GRAPH_COLOR_COUNT] = (uint32_t)n;
    *count = n;
    return (handles && n > cap) ? AVN_ERR_CAPACITY : AVN_OK;
}
}
