// ORACLE — TEST INFRASTRUCTURE ONLY (see avo_world.hpp).  Exports the SAME C ABI as the product
// (include/avian_mi355x.h) under the `avo_` prefix so tests drive both with identical calls.
#define AVN_PREFIX_ORACLE 1
#include "avo_world.hpp"

#include <cmath>
#include <new>
#include <map>
#include <set>

struct avn_world { avo::WorldBase* impl; };
struct avn_constraint_graph { avo::ConstraintGraph g; };
struct avn_island_manager { avo::IslandManager m; };
static thread_local std::string g_create_error;

extern "C" {

avn_status avo_world_create(const avn_config* cfg, avn_world** out) {
    if (!cfg || !out) { g_create_error = "world_create: null argument"; return AVN_ERR_BAD_ARG; }
    if (cfg->struct_size != sizeof(avn_config)) { g_create_error = "world_create: struct_size mismatch"; return AVN_ERR_BAD_ARG; }
    avo::WorldBase* w = nullptr;
    if (cfg->scalar_bits == 32) w = new (std::nothrow) avo::World<float>();
    else if (cfg->scalar_bits == 64) w = new (std::nothrow) avo::World<double>();
    else { g_create_error = "world_create: scalar_bits must be 32 or 64"; return AVN_ERR_BAD_ARG; }
    if (!w) return AVN_ERR_OOM;
    avn_status st = w->config_set(cfg);
    if (st != AVN_OK) { g_create_error = w->error; delete w; return st; }
    *out = new avn_world{w};
    return AVN_OK;
}
void avo_world_destroy(avn_world* w) { if (w) { delete w->impl; delete w; } }
const char* avo_last_error(const avn_world* w) { return w ? w->impl->error.c_str() : g_create_error.c_str(); }
#define FWD(call) do { if (!w) return AVN_ERR_BAD_ARG; return w->impl->call; } while (0)
avn_status avo_config_set(avn_world* w, const avn_config* c) { FWD(config_set(c)); }
avn_status avo_bodies_upload(avn_world* w, const avn_bodies* b) { FWD(bodies_upload(b)); }
avn_status avo_bodies_download(avn_world* w, const avn_bodies_out* o) { FWD(bodies_download(o)); }
avn_status avo_solver_bodies_download(avn_world* w, const avn_solver_bodies_out* o) { FWD(solver_bodies_download(o)); }
avn_status avo_manifolds_upload(avn_world* w, const avn_manifolds* m) { FWD(manifolds_upload(m)); }
avn_status avo_impulses_download(avn_world* w, const avn_impulses_out* o) { FWD(impulses_download(o)); }
avn_status avo_constraints_download(avn_world* w, const avn_constraints_out* o) { FWD(constraints_download(o)); }
avn_status avo_distance_joints_upload(avn_world* w, const avn_distance_joints* j) { FWD(distance_joints_upload(j)); }
avn_status avo_joints_upload(avn_world* w, const avn_joints* j) { FWD(joints_upload(j)); }
avn_status avo_joints_download(avn_world* w, const avn_joints_out* o) { FWD(joints_download(o)); }
avn_status avo_colliders_upload(avn_world* w, const avn_colliders* c) { FWD(colliders_upload(c)); }
avn_status avo_existing_pairs_upload(avn_world* w, const uint64_t* k, size_t n) { FWD(existing_pairs_upload(k, n)); }
avn_status avo_pairs_get(avn_world* w, const avn_pair** o, size_t* n) { FWD(pairs_get(o, n)); }
avn_status avo_aabbs_download(avn_world* w, void* mn, void* mx, uint32_t* e, size_t* n) { FWD(aabbs_download(mn, mx, e, n)); }
avn_status avo_run_system(avn_world* w, avn_system s) { FWD(run_system(s)); }
avn_status avo_step(avn_world* w) { FWD(step()); }
avn_status avo_synchronize(avn_world* w) { return w ? AVN_OK : AVN_ERR_BAD_ARG; }
avn_status avo_timers_get(avn_world* w, avn_timers* t) { FWD(timers(t)); }
avn_status avo_diagnostics_get(avn_world* w, avn_diagnostics* d) { FWD(diagnostics(d)); }
avn_status avo_islands_get(avn_world* w, uint32_t* island_of_body, uint32_t* n_islands) { FWD(islands_get(island_of_body, n_islands)); }
avn_status avo_sleep_update(avn_world* w, const avn_sleep_params* p, avn_sleep_stats* st) { FWD(sleep_update(p, st)); }
avn_status avo_sleep_get(avn_world* w, const avn_sleep_out* o) { FWD(sleep_get(o)); }
avn_status avo_sleep_reset(avn_world* w, const uint32_t* bodies, size_t n) { FWD(sleep_reset(bodies, n)); }
avn_status avo_halo_plan_upload(avn_world* w, const avn_halo_plan* p) { FWD(halo_plan_upload(p)); }
avn_status avo_run_color_pass(avn_world* w, avn_system pass, uint32_t color) { FWD(run_color_pass(pass, color)); }
avn_status avo_halo_pack(avn_world* w, uint32_t color, uint32_t peer, void* out, size_t* count) { FWD(halo_pack(color, peer, out, count)); }
avn_status avo_halo_unpack(avn_world* w, uint32_t color, uint32_t peer, const void* in, size_t count) { FWD(halo_unpack(color, peer, in, count)); }
// the CPU checker has no device transport: the host moves the records (avn_halo_pack / avn_halo_unpack)
avn_status avo_comm_unique_id(uint8_t* out) { if (!out) return AVN_ERR_BAD_ARG; std::memset(out, 0, AVN_COMM_ID_BYTES); return AVN_OK; }
avn_status avo_comm_init(avn_world*, const uint8_t*, int, int) { return AVN_ERR_STATE; }
// (no transport in the checker: a world is its own only rank)
avn_status avo_bounds_exchange(avn_world* w, double* bounds, uint32_t cap_ranks, uint32_t* n_ranks, uint32_t*, uint32_t, uint32_t* n_overlaps) {
    if (!w || !bounds || cap_ranks < 1) return AVN_ERR_BAD_ARG;
    if (n_ranks) *n_ranks = 1;
    if (n_overlaps) *n_overlaps = 0;
    return w->impl->dynamic_bounds(bounds, bounds + 3);
}
avn_status avo_profile_system(avn_world* w, avn_system s, uint32_t r, double* ms, uint32_t* l) { FWD(profile_system(s, r, ms, l)); }
avn_status avo_dynamic_bounds(avn_world* w, double* mn, double* mx) { FWD(dynamic_bounds(mn, mx)); }
avn_status avo_contact_manifolds(avn_world* w, const avn_shape_pairs* p, const avn_query_manifolds_out* o) { FWD(contact_manifolds(p, o)); }
avn_status avo_collider_materials_upload(avn_world* w, const avn_collider_materials* m) { FWD(collider_materials_upload(m)); }
avn_status avo_contact_pairs_add(avn_world* w, const avn_contact_pairs* p) { FWD(contact_pairs_add(p)); }
avn_status avo_contact_pairs_remove(avn_world* w, const uint32_t* ids, size_t n) { FWD(contact_pairs_remove(ids, n)); }
avn_status avo_active_pairs_set(avn_world* w, const uint32_t* ids, size_t n) { FWD(active_pairs_set(ids, n)); }
avn_status avo_contact_changes_get(avn_world* w, const avn_contact_change** o, size_t* n) { FWD(contact_changes_get(o, n)); }
avn_status avo_manifold_handles_upload(avn_world* w, const uint32_t* off, const uint32_t* ids) { FWD(manifold_handles_upload(off, ids)); }
avn_status avo_contacts_download(avn_world* w, const uint32_t* ids, size_t n, const avn_contacts_out* o) { FWD(contacts_download(ids, n, o)); }
avn_status avo_contacts_upload(avn_world* w, const uint32_t* ids, size_t n, const avn_contacts_in* in) { FWD(contacts_upload(ids, n, in)); }
avn_status avo_pipeline_enable(avn_world* w, int on) { FWD(pipeline_enable(on)); }
avn_status avo_pipeline_stats_get(avn_world* w, avn_pipeline_stats* o) { FWD(pipeline_stats_get(o)); }
avn_status avo_pipeline_handles_get(avn_world* w, uint32_t* off, const uint32_t** ids, size_t* n) { FWD(pipeline_handles_get(off, ids, n)); }
avn_status avo_pipeline_new_pair_ids_get(avn_world* w, const uint32_t** ids, size_t* n) { FWD(pipeline_new_pair_ids_get(ids, n)); }
avn_status avo_sleeping_enable(avn_world* w, const avn_sleep_params* p) { FWD(sleeping_enable(p)); }
avn_status avo_sleeping_stats_get(avn_world* w, avn_sleeping_stats* o) { FWD(sleeping_stats_get(o)); }
avn_status avo_sleeping_state_get(avn_world* w, const avn_sleeping_out* o) { FWD(sleeping_state_get(o)); }
avn_status avo_wake_bodies(avn_world* w, const uint32_t* ids, size_t n) { FWD(wake_bodies(ids, n)); }
avn_status avo_despawn(avn_world* w, const avn_despawn_list* d) { FWD(despawn(d)); }
avn_status avo_dshard_enable(avn_world* w, const avn_dshard_config* c) { FWD(dshard_enable(c)); }
avn_status avo_dshard_bodies_pack(avn_world* w, void* out, size_t cap, size_t* bytes) { FWD(dshard_bodies_pack(out, cap, bytes)); }
avn_status avo_dshard_bodies_unpack(avn_world* w, uint32_t from, const void* in, size_t bytes) { FWD(dshard_bodies_unpack(from, in, bytes)); }
avn_status avo_dshard_stats_get(avn_world* w, avn_dshard_stats* o) { FWD(dshard_stats_get(o)); }
// Checker for avn_level2_plan_* (header).  Deliberately organised the other way round from the product's planner: per colour a
// body -> owner-of-its-manifold table (a non-static body is in at most one manifold per colour), then the send lists are read off BODY by
// body in ascending index, so they come out sorted without sorting.
struct avn_level2_plan {
    struct Rank { std::vector<int32_t> bodies, peers, send_bodies, recv_bodies; std::vector<uint32_t> manifolds, color_offsets, send_offsets, recv_offsets, overflow_level, joints; };
    std::vector<Rank> ranks;
    uint32_t n_overflow_levels = 1;
    bool joint_slot = false, global_joints = false;
};
static avn_status level2_plan_create(const avn_level2_in* in, const avn_level2_joints* jn, avn_level2_plan** out);
avn_status avo_level2_plan_create(const avn_level2_in* in, avn_level2_plan** out) { return level2_plan_create(in, nullptr, out); }
avn_status avo_level2_plan_create_joints(const avn_level2_in* in, const avn_level2_joints* jn, avn_level2_plan** out) { return jn ? level2_plan_create(in, jn, out) : AVN_ERR_BAD_ARG; }
static avn_status level2_plan_create(const avn_level2_in* in, const avn_level2_joints* jn, avn_level2_plan** out) {
    if (jn && jn->n_joints && (!jn->body1 || !jn->body2 || !jn->joint_type)) return AVN_ERR_BAD_ARG;
    if (!in || !out || in->n_ranks == 0 || !in->color_offsets || (in->n_bodies && (!in->rb_type || !in->center_x)) || (in->n_manifolds && (!in->body1 || !in->body2))) return AVN_ERR_BAD_ARG;
    *out = nullptr;
    const uint32_t N = in->n_bodies, M = in->n_manifolds, R = in->n_ranks, C = AVN_GRAPH_COLOR_COUNT;
    if (in->color_offsets[0] != 0 || in->color_offsets[C] != M) return AVN_ERR_BAD_ARG;
    std::vector<uint32_t> dyn;
    for (uint32_t b = 0; b < N; ++b) if (in->rb_type[b] != AVN_RB_STATIC) dyn.push_back(b);
    std::vector<double> xs(dyn.size());
    for (size_t i = 0; i < dyn.size(); ++i) xs[i] = in->center_x[dyn[i]];
    std::sort(xs.begin(), xs.end());
    std::vector<int32_t> slab(N, -1);
    for (uint32_t b : dyn) {   // slab = number of cuts <= x
        int32_t k = 0;
        for (uint32_t r = 1; r < R; ++r) if (xs[std::min<size_t>(xs.size() - 1, xs.size() * (size_t)r / R)] <= in->center_x[b]) ++k;
        slab[b] = k;
    }
    std::vector<int32_t> owner_of(M);
    std::vector<std::set<uint32_t>> holders(N);
    for (uint32_t m = 0; m < M; ++m) {
        const int32_t a = in->body1[m], b = in->body2[m];
        if (a < 0 || b < 0 || (uint32_t)a >= N || (uint32_t)b >= N) return AVN_ERR_BAD_ARG;
        owner_of[m] = in->rb_type[a] == AVN_RB_STATIC ? slab[b] : slab[a];
        if (owner_of[m] < 0) return AVN_ERR_BAD_ARG;
        holders[a].insert((uint32_t)owner_of[m]); holders[b].insert((uint32_t)owner_of[m]);
    }
    for (uint32_t b : dyn) holders[b].insert((uint32_t)slab[b]);
    // joints (header: avn_halo_joint_slot_set): a flood over the joint graph -- nodes = non-static bodies + (with JointDamping) the two DUMMY stand-ins of every joint type --
    // started from the bodies in ascending order, so a component's first body is its lowest and decides the owner; the owner holds every body of the component
    const uint32_t J = jn ? jn->n_joints : 0u, NV = N + 2u * AVN_JOINT_TYPE_COUNT;
    std::vector<int32_t> comp_owner_of_node(NV, -1), joint_owner(J, 0);
    std::vector<uint8_t> jointed(N, 0);
    if (J) {
        std::vector<std::vector<uint32_t>> adj(NV);
        std::vector<int64_t> a_node(J, -1);
        for (uint32_t j = 0; j < J; ++j) {
            const int32_t a = jn->body1[j], b = jn->body2[j];
            if (a < 0 || b < 0 || (uint32_t)a >= N || (uint32_t)b >= N || jn->joint_type[j] >= AVN_JOINT_TYPE_COUNT) return AVN_ERR_BAD_ARG;
            const bool sa = in->rb_type[a] == AVN_RB_STATIC, sb = in->rb_type[b] == AVN_RB_STATIC;
            const int64_t na = sa ? (jn->damped ? (int64_t)N + 2 * jn->joint_type[j] : -1) : a, nb = sb ? (jn->damped ? (int64_t)N + 2 * jn->joint_type[j] + 1 : -1) : b;
            if (na >= 0 && nb >= 0) { adj[(size_t)na].push_back((uint32_t)nb); adj[(size_t)nb].push_back((uint32_t)na); }
            a_node[j] = na >= 0 ? na : nb;
            if (!sa) jointed[a] = 1;
            if (!sb) jointed[b] = 1;
        }
        for (uint32_t b0 : dyn) {
            if (!jointed[b0] || comp_owner_of_node[b0] >= 0) continue;
            std::vector<uint32_t> stack{b0};
            comp_owner_of_node[b0] = slab[b0];
            while (!stack.empty()) {
                const uint32_t x = stack.back(); stack.pop_back();
                for (uint32_t y : adj[x]) if (comp_owner_of_node[y] < 0) { comp_owner_of_node[y] = slab[b0]; stack.push_back(y); }
            }
        }
        for (uint32_t b : dyn) if (jointed[b]) holders[b].insert((uint32_t)comp_owner_of_node[b]);
        for (uint32_t j = 0; j < J; ++j) joint_owner[j] = a_node[j] >= 0 && comp_owner_of_node[(size_t)a_node[j]] >= 0 ? comp_owner_of_node[(size_t)a_node[j]] : 0;
    }
    auto shared = [&](uint32_t b) { return in->rb_type[b] != AVN_RB_STATIC && holders[b].size() > 1; };
    // Exchange slots: colours 0..22, then the LEVELS of the overflow colour when one of its manifolds touches a shared body (header: avn_level2_plan_rank_overflow).
    // A manifold's level = how many overflow manifolds lie in front of it on the deepest chain through its bodies: per body the depth reached so far, walked in list order.
    const uint32_t o0 = in->color_offsets[AVN_COLOR_OVERFLOW_INDEX], o1 = in->color_offsets[AVN_COLOR_OVERFLOW_INDEX + 1];
    bool levelled = false;
    for (uint32_t m = o0; m < o1; ++m) levelled = levelled || shared((uint32_t)in->body1[m]) || shared((uint32_t)in->body2[m]);
    std::vector<uint32_t> lev(o1 - o0, 0u), depth(N, 0u);
    uint32_t n_levels = 1;
    if (levelled)
        for (uint32_t m = o0; m < o1; ++m) {
            uint32_t d = 0;
            for (int32_t b : {in->body1[m], in->body2[m]}) if (in->rb_type[b] != AVN_RB_STATIC) d = std::max(d, depth[b]);
            lev[m - o0] = d;
            for (int32_t b : {in->body1[m], in->body2[m]}) if (in->rb_type[b] != AVN_RB_STATIC) depth[b] = d + 1;
            n_levels = std::max(n_levels, d + 1);
        }
    bool joint_slot = false;
    for (uint32_t b : dyn) joint_slot = joint_slot || (jointed[b] && shared(b));
    const uint32_t S = (uint32_t)AVN_COLOR_OVERFLOW_INDEX + n_levels + (joint_slot ? 1u : 0u);
    // mover[slot][b] = the rank whose manifold of that slot touches shared body b (-1: none)
    std::vector<std::vector<int32_t>> mover(S, std::vector<int32_t>(N, -1));
    for (uint32_t c = 0; c < C; ++c)
        for (uint32_t m = in->color_offsets[c]; m < in->color_offsets[c + 1]; ++m)
            for (int32_t b : {in->body1[m], in->body2[m]})
                if (shared((uint32_t)b)) mover[c == (uint32_t)AVN_COLOR_OVERFLOW_INDEX ? c + lev[m - o0] : c][b] = owner_of[m];
    if (joint_slot) for (uint32_t b : dyn) if (jointed[b] && shared(b)) mover[S - 1][b] = comp_owner_of_node[b];   // the joint slot: the component's owner moves its shared bodies
    avn_level2_plan* pl = new avn_level2_plan;
    pl->ranks.resize(R);
    pl->n_overflow_levels = n_levels;
    pl->joint_slot = joint_slot; pl->global_joints = J != 0;
    for (uint32_t j = 0; j < J; ++j) pl->ranks[(uint32_t)joint_owner[j]].joints.push_back(j);
    for (uint32_t r = 0; r < R; ++r) {
        auto& k = pl->ranks[r];
        std::vector<int32_t> local(N, -1);
        for (uint32_t b = 0; b < N; ++b)
            if (in->rb_type[b] == AVN_RB_STATIC || holders[b].count(r)) { local[b] = (int32_t)k.bodies.size(); k.bodies.push_back((int32_t)b); }
        std::set<int32_t> peers;
        for (uint32_t c = 0; c < S; ++c)
            for (uint32_t b : dyn) {
                if (mover[c][b] < 0 || !holders[b].count(r)) continue;
                if ((uint32_t)mover[c][b] == r) { for (uint32_t h : holders[b]) if (h != r) peers.insert((int32_t)h); }
                else peers.insert(mover[c][b]);
            }
        k.peers.assign(peers.begin(), peers.end());
        k.send_offsets.push_back(0); k.recv_offsets.push_back(0);
        if (!k.peers.empty())
            for (uint32_t c = 0; c < S; ++c)
                for (int32_t p : k.peers) {
                    for (uint32_t b : dyn) {
                        if (mover[c][b] == (int32_t)r && holders[b].count((uint32_t)p)) k.send_bodies.push_back(local[b]);
                        if (mover[c][b] == p && holders[b].count(r)) k.recv_bodies.push_back(local[b]);
                    }
                    k.send_offsets.push_back((uint32_t)k.send_bodies.size()); k.recv_offsets.push_back((uint32_t)k.recv_bodies.size());
                }
        k.color_offsets.assign(C + 1, 0);
        for (uint32_t c = 0; c < C; ++c) {
            for (uint32_t m = in->color_offsets[c]; m < in->color_offsets[c + 1]; ++m)
                if ((uint32_t)owner_of[m] == r) { k.manifolds.push_back(m); if (c == (uint32_t)AVN_COLOR_OVERFLOW_INDEX) k.overflow_level.push_back(lev[m - o0]); }
            k.color_offsets[c + 1] = (uint32_t)k.manifolds.size();
        }
    }
    *out = pl;
    return AVN_OK;
}
void avo_level2_plan_destroy(avn_level2_plan* plan) { delete plan; }
avn_status avo_level2_plan_rank_overflow(const avn_level2_plan* plan, uint32_t rank, uint32_t* n_levels, const uint32_t** level_of) {
    if (!plan || rank >= plan->ranks.size() || !n_levels || !level_of) return AVN_ERR_BAD_ARG;
    *n_levels = plan->n_overflow_levels; *level_of = plan->ranks[rank].overflow_level.data();
    return AVN_OK;
}
avn_status avo_host_shapes_set(avn_world* w, avn_host_aabb_fn a, avn_host_manifolds_fn m, void* user) { FWD(host_shapes_set(a, m, user)); }
avn_status avo_host_shape_stats_get(avn_world* w, avn_host_shape_stats* o) { FWD(host_shape_stats_get(o)); }
avn_status avo_collider_transforms_upload(avn_world* w, const avn_collider_transforms* t) { FWD(collider_transforms_upload(t)); }
avn_status avo_local_accelerations_upload(avn_world* w, uint32_t count, const void* linear, const void* angular) { FWD(local_accelerations_upload(count, linear, angular)); }
avn_status avo_collision_hooks_set(avn_world* w, avn_filter_pairs_fn f, avn_modify_contacts_fn m, void* user) { FWD(collision_hooks_set(f, m, user)); }
avn_status avo_collision_hook_stats_get(avn_world* w, avn_collision_hook_stats* o) { FWD(collision_hook_stats_get(o)); }
avn_status avo_level2_plan_rank_joints(const avn_level2_plan* plan, uint32_t rank, uint32_t* n_joints, const uint32_t** joints, uint32_t* joint_slot, uint32_t* global_joints) {
    if (!plan || rank >= plan->ranks.size() || !n_joints || !joints || !joint_slot || !global_joints) return AVN_ERR_BAD_ARG;
    *n_joints = (uint32_t)plan->ranks[rank].joints.size(); *joints = plan->ranks[rank].joints.data();
    *joint_slot = plan->joint_slot; *global_joints = plan->global_joints;
    return AVN_OK;
}
avn_status avo_halo_joint_slot_set(avn_world* w, uint32_t joint_slot, uint32_t global_joints) { FWD(halo_joint_slot_set(joint_slot, global_joints)); }
avn_status avo_halo_overflow_levels_upload(avn_world* w, uint32_t n_levels, const uint32_t* level_of, size_t count) { FWD(halo_overflow_levels_upload(n_levels, level_of, count)); }
avn_status avo_level2_plan_rank(const avn_level2_plan* plan, uint32_t rank, avn_level2_rank* out) {
    if (!plan || !out || rank >= plan->ranks.size()) return AVN_ERR_BAD_ARG;
    const auto& k = plan->ranks[rank];
    out->n_bodies = (uint32_t)k.bodies.size(); out->bodies = k.bodies.data();
    out->n_manifolds = (uint32_t)k.manifolds.size(); out->manifolds = k.manifolds.data(); out->color_offsets = k.color_offsets.data();
    out->halo.n_peers = (uint32_t)k.peers.size(); out->halo.peer_rank = k.peers.data();
    out->halo.send_offsets = k.send_offsets.data(); out->halo.send_bodies = k.send_bodies.data();
    out->halo.recv_offsets = k.recv_offsets.data(); out->halo.recv_bodies = k.recv_bodies.data();
    return AVN_OK;
}

// Checkers for avn_slab_select / avn_interval_orders_merge (header), written the slow obvious way: slab of a collider = number of
// boundaries at or below its key, orders by repeated selection.
avn_status avo_slab_select(const avn_slab_in* in, uint32_t* local, uint8_t* owned, uint32_t* n_local, uint32_t* next_order, uint32_t* n_next) {
    if (!in || !local || !owned || !n_local || in->n_ranks == 0 || in->rank >= in->n_ranks || (next_order && !n_next)) return AVN_ERR_BAD_ARG;
    const uint32_t n = in->n_colliders, R = in->n_ranks;
    std::vector<uint32_t> order;
    std::vector<bool> have(n, false);
    for (uint32_t i = 0; i < in->n_prev; ++i) { if (in->prev_order[i] >= n || have[in->prev_order[i]]) return AVN_ERR_BAD_ARG; have[in->prev_order[i]] = true; order.push_back(in->prev_order[i]); }
    for (uint32_t c = 0; c < n; ++c) if (!have[c]) order.push_back(c);
    std::vector<double> xs;
    for (uint32_t c = 0; c < n; ++c) if (in->aabb_min_x[c] == in->aabb_min_x[c]) xs.push_back(in->aabb_min_x[c]);
    std::sort(xs.begin(), xs.end());
    for (uint32_t c = 0; c < n; ++c) if (in->aabb_min_x[c] != in->aabb_min_x[c]) xs.push_back(in->aabb_min_x[c]);   // NaNs last
    std::vector<double> bound;   // boundaries 1 .. R - 1, made non-decreasing
    for (uint32_t r = 1; r < R; ++r) {
        double b = n ? xs[std::min<size_t>(n - 1, (size_t)n * r / R)] : HUGE_VAL;
        if (!bound.empty() && !(b >= bound.back())) b = bound.back();
        bound.push_back(b);
    }
    auto slab = [&](double x) { uint32_t s = 0; for (double b : bound) if (b <= x) ++s; if (x != x) s = R - 1; return std::min(s, R - 1); };
    bool any = false, nan_reach = false;
    double reach = -HUGE_VAL;
    for (uint32_t c = 0; c < n; ++c) if (slab(in->aabb_min_x[c]) == in->rank) { any = true; if (in->aabb_max_x[c] != in->aabb_max_x[c]) nan_reach = true; else if (in->aabb_max_x[c] > reach) reach = in->aabb_max_x[c]; }
    uint32_t k = 0;
    if (any)
        for (uint32_t c : order) {
            const uint32_t s = slab(in->aabb_min_x[c]);
            if (s == in->rank) { local[k] = c; owned[k++] = 1; }
            else if (s > in->rank && !nan_reach && in->aabb_min_x[c] <= reach) { local[k] = c; owned[k++] = 0; }
        }
    *n_local = k;
    if (next_order) {
        std::vector<uint32_t> fin;
        for (uint32_t c : order) if (std::isfinite(in->aabb_min_x[c])) fin.push_back(c);
        // insertion sort (stable), like the reference's own (broad_phase.rs:479-487)
        for (size_t i = 1; i < fin.size(); ++i) {
            const uint32_t v = fin[i]; size_t j = i;
            while (j > 0 && in->aabb_min_x[fin[j - 1]] + 0.0 > in->aabb_min_x[v] + 0.0) { fin[j] = fin[j - 1]; --j; }
            fin[j] = v;
        }
        for (size_t i = 0; i < fin.size(); ++i) next_order[i] = fin[i];
        *n_next = (uint32_t)fin.size();
    }
    return AVN_OK;
}
avn_status avo_interval_orders_merge(uint32_t n_lists, const uint32_t* const* entities, const double* const* keys, const uint32_t* lengths, uint32_t* out, uint32_t* n_out) {
    if (!n_out || (n_lists && (!entities || !keys || !lengths))) return AVN_ERR_BAD_ARG;
    std::vector<uint32_t> at(n_lists, 0);
    std::set<uint32_t> done;
    uint32_t k = 0;
    while (true) {
        int pick = -1;
        for (uint32_t l = 0; l < n_lists; ++l) {
            if (at[l] == lengths[l]) continue;
            if (pick < 0) { pick = (int)l; continue; }
            double a = keys[l][at[l]], b = keys[pick][at[pick]];
            if (a != a) a = -HUGE_VAL;
            if (b != b) b = -HUGE_VAL;
            if (a < b || (a == b && entities[l][at[l]] < entities[pick][at[pick]])) pick = (int)l;
        }
        if (pick < 0) break;
        const uint32_t e = entities[pick][at[pick]++];
        if (done.insert(e).second) { if (out) out[k] = e; ++k; }
    }
    *n_out = k;
    return AVN_OK;
}

// Checker for avn_islands_partition (header).  Deliberately a DIFFERENT algorithm from the product's union-find:
// breadth-first flood fill over an adjacency list, islands discovered in ascending body index (= numbered by their
// smallest member), then the same slab rule (reference island statistics: islands/mod.rs:213-232).
avn_status avo_islands_partition(const avn_islands_in* in, int32_t* island_of_body, int32_t* rank_of_body, uint32_t* n_islands) {
    if (!in || !island_of_body || !rank_of_body || !n_islands || in->n_ranks == 0) return AVN_ERR_BAD_ARG;
    const uint32_t n = in->n_bodies;
    std::vector<std::vector<int32_t>> adj(n);
    std::vector<uint64_t> carried(n, 0);
    for (uint32_t e = 0; e < in->n_edges; ++e) {
        int32_t a = in->edge_body1[e], b = in->edge_body2[e];
        if (a < 0 || b < 0 || (uint32_t)a >= n || (uint32_t)b >= n) return AVN_ERR_BAD_ARG;
        bool sa = in->rb_type[a] == AVN_RB_STATIC, sb = in->rb_type[b] == AVN_RB_STATIC;
        if (sa && sb) continue;
        if (!sa && !sb) { adj[a].push_back(b); adj[b].push_back(a); }
        carried[sa ? b : a] += 1;
    }
    for (uint32_t b = 0; b < n; ++b) island_of_body[b] = -1;
    std::vector<double> sum_x; std::vector<uint64_t> weight, members;
    uint32_t k = 0;
    std::vector<int32_t> queue;
    for (uint32_t s = 0; s < n; ++s) {
        if (in->rb_type[s] == AVN_RB_STATIC || island_of_body[s] >= 0) continue;
        sum_x.push_back(0); weight.push_back(0); members.push_back(0);
        queue.assign(1, (int32_t)s); island_of_body[s] = (int32_t)k;
        for (size_t q = 0; q < queue.size(); ++q) {
            int32_t b = queue[q];
            sum_x[k] += in->center_x[b]; members[k] += 1; weight[k] += 1 + carried[b];
            for (int32_t o : adj[b]) if (island_of_body[o] < 0) { island_of_body[o] = (int32_t)k; queue.push_back(o); }
        }
        ++k;
    }
    // NOTE: sum_x accumulates in BFS order here and in index order in the product; the slab ORDER only needs the means to
    // compare the same way, which tests guarantee by using scenes whose island means are well separated or equal by construction
    // (ties are broken by island id in both).
    *n_islands = k;
    std::vector<uint32_t> order(k);
    for (uint32_t i = 0; i < k; ++i) order[i] = i;
    std::vector<double> mean(k);
    for (uint32_t i = 0; i < k; ++i) mean[i] = sum_x[i] / (double)members[i];
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return mean[a] < mean[b]; });
    unsigned __int128 total = 0;
    for (uint32_t i = 0; i < k; ++i) total += weight[i];
    std::vector<int32_t> rank_of_island(k, 0);
    unsigned __int128 cum = 0;
    for (uint32_t o = 0; o < k; ++o) {
        uint32_t i = order[o];
        unsigned __int128 r = total ? ((2 * cum + weight[i]) * in->n_ranks) / (2 * total) : 0;
        if (r > in->n_ranks - 1) r = in->n_ranks - 1;
        rank_of_island[i] = (int32_t)r;
        cum += weight[i];
    }
    for (uint32_t b = 0; b < n; ++b) rank_of_body[b] = island_of_body[b] < 0 ? -1 : rank_of_island[island_of_body[b]];
    return AVN_OK;
}
uint64_t avo_pair_key(uint32_t a, uint32_t b) { return avo::pair_key(a, b); }

// ---- avn_islands_* (header): the linked-list restatement of avo_islands.hpp ----
avn_island_manager* avo_islands_create(void) { return new (std::nothrow) avn_island_manager(); }
void avo_islands_destroy(avn_island_manager* m) { delete m; }
avn_status avo_islands_body_add(avn_island_manager* m, uint32_t body) { return m ? m->m.body_add(body) : AVN_ERR_BAD_ARG; }
avn_status avo_islands_collider_add(avn_island_manager* m, uint32_t collider, uint32_t body) { return m ? m->m.collider_add(collider, body) : AVN_ERR_BAD_ARG; }
avn_status avo_islands_joint_add(avn_island_manager* m, uint32_t joint, uint32_t b1, uint32_t b2) { return m ? m->m.joint_add(joint, b1, b2) : AVN_ERR_BAD_ARG; }
avn_status avo_islands_pair_add(avn_island_manager* m, uint32_t id, uint32_t c1, uint32_t c2) { return m ? m->m.pair_add(id, c1, c2) : AVN_ERR_BAD_ARG; }
avn_status avo_islands_status_change(avn_island_manager* m, uint32_t id, uint32_t flags, uint32_t mc) { return m ? m->m.status_change(id, flags, mc) : AVN_ERR_BAD_ARG; }
avn_status avo_islands_flush_wake(avn_island_manager* m) { return m ? m->m.flush_wake() : AVN_ERR_BAD_ARG; }
avn_status avo_islands_split_candidate(avn_island_manager* m) { return m ? m->m.split_candidate_now() : AVN_ERR_BAD_ARG; }
// (the oracle has no worker thread: `labels` only ever allow a shortcut, the checked serial walk is always right)
avn_status avo_islands_split_candidate_adjacency(avn_island_manager* m, const uint32_t* off, const uint32_t* adj, uint32_t n, const uint32_t*) { return m ? m->m.split_candidate_adjacency(off, adj, n) : AVN_ERR_BAD_ARG; }
avn_status avo_islands_split_join(avn_island_manager* m) { return m ? AVN_OK : AVN_ERR_BAD_ARG; }
avn_status avo_islands_sleeping_systems(avn_island_manager* m, const float* t, const uint8_t* f, uint32_t n, float tts) { return m ? m->m.sleeping_systems(t, f, n, tts) : AVN_ERR_BAD_ARG; }
avn_status avo_islands_wake_body(avn_island_manager* m, uint32_t body) { return m ? m->m.wake_body(body) : AVN_ERR_BAD_ARG; }
avn_status avo_islands_sleep_body(avn_island_manager* m, uint32_t body) { return m ? m->m.sleep_body(body) : AVN_ERR_BAD_ARG; }
avn_status avo_islands_last_result(avn_island_manager* m, avn_islands_result* o) {
    if (!m || !o) return AVN_ERR_BAD_ARG;
    avo::IslandManager& g = m->m;
    avn_islands_result& r = *o;
    r.popped = g.popped.data(); r.n_popped = g.popped.size(); r.pushed = g.pushed.data(); r.n_pushed = g.pushed.size();
    r.pairs_slept = g.pairs_slept.data(); r.n_pairs_slept = g.pairs_slept.size(); r.pairs_woken = g.pairs_woken.data(); r.n_pairs_woken = g.pairs_woken.size();
    r.bodies_slept = g.bodies_slept.data(); r.n_bodies_slept = g.bodies_slept.size(); r.bodies_woken = g.bodies_woken.data(); r.n_bodies_woken = g.bodies_woken.size();
    r.pairs_removed = g.pairs_removed.data(); r.n_pairs_removed = g.pairs_removed.size();
    return AVN_OK;
}
avn_status avo_islands_collider_remove(avn_island_manager* m, uint32_t collider) { return m ? m->m.collider_remove_full(collider) : AVN_ERR_BAD_ARG; }
avn_status avo_islands_body_remove(avn_island_manager* m, uint32_t body) { return m ? m->m.body_remove_and_wake(body) : AVN_ERR_BAD_ARG; }
avn_status avo_islands_joint_remove(avn_island_manager* m, uint32_t joint) { return m ? m->m.joint_remove_and_wake(joint) : AVN_ERR_BAD_ARG; }
avn_status avo_islands_renumber_joints(avn_island_manager* m, const uint32_t* new_index, uint32_t n_old) {
    if (!m || (n_old && !new_index)) return AVN_ERR_BAD_ARG;
    m->m.renumber_joints(std::vector<uint32_t>(new_index, new_index + n_old));
    return AVN_OK;
}
avn_status avo_islands_renumber_bodies(avn_island_manager* m, const uint32_t* new_index, uint32_t n_old) {
    if (!m || (n_old && !new_index)) return AVN_ERR_BAD_ARG;
    std::vector<uint32_t> map(new_index, new_index + n_old);
    uint32_t n_new = 0;
    for (uint32_t v : map) if (v != avo::IslandManager::NONE) n_new = std::max(n_new, v + 1u);
    m->m.renumber_bodies(map, n_new);
    return AVN_OK;
}
avn_status avo_islands_stats_get(avn_island_manager* m, avn_islands_stats* o) { return m ? m->m.stats(o) : AVN_ERR_BAD_ARG; }
avn_status avo_islands_state(avn_island_manager* m, uint32_t n, uint32_t* a, uint32_t* b, uint8_t* c, uint32_t* d) { return m ? m->m.state(n, a, b, c, d) : AVN_ERR_BAD_ARG; }
// debug aid of the oracle only (not in the header): PhysicsIsland::validate over every island; 1 = consistent
int avo_islands_validate(avn_island_manager* m) { std::string why; return m && m->m.validate(why) ? 1 : 0; }

avn_status avo_constraint_graph_create(uint32_t, avn_constraint_graph** out) {
    if (!out) return AVN_ERR_BAD_ARG;
    *out = new (std::nothrow) avn_constraint_graph();
    return *out ? AVN_OK : AVN_ERR_OOM;
}
void avo_constraint_graph_destroy(avn_constraint_graph* g) { delete g; }
int32_t avo_constraint_graph_push(avn_constraint_graph* g, uint64_t h, uint32_t b1, uint32_t b2, int s1, int s2) {
    return g ? g->g.push_manifold(h, b1, b2, s1 != 0, s2 != 0) : -1;
}
avn_status avo_constraint_graph_push_batch(avn_constraint_graph* g, size_t n, const uint64_t* h, const uint32_t* b1, const uint32_t* b2,
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
avn_status avo_constraint_graph_pop(avn_constraint_graph* g, uint64_t h) {
    if (!g) return AVN_ERR_BAD_ARG;
    return g->g.pop_manifold(h) ? AVN_OK : AVN_ERR_STATE;
}
avn_status avo_constraint_graph_lists(const avn_constraint_graph* g, uint32_t* offsets, uint64_t* handles, size_t cap, size_t* count) {
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
// test hook: switch the oracle's sin/cos to the host libm (to measure the deterministic kernel's deviation)
void avo_use_libm_trig(int on) { avo::use_libm_trig() = on != 0; }
// test hook: glam's scalar (left-to-right) f32 quaternion product instead of the SSE2 association
void avo_use_scalar_quat(int on) { avo::use_scalar_quat_mul() = on != 0; }
void avo_sin_cos_f32(float a, float* s, float* c) { avo::sin_cos_det(a, *s, *c); }
void avo_sin_cos_f64(double a, double* s, double* c) { avo::sin_cos_det(a, *s, *c); }
float avo_asin_f32(float x) { return avo::asin_det(x); }
double avo_asin_f64(double x) { return avo::asin_det(x); }

// ---- the closed loop sharded by islands: the replicated bookkeeping as the SINGLE world's own structures (PipelineState's: an ordered set of free
// ids, a map of pairs, the ConstraintGraph restatement) fed with every rank's events -- the checker of avn_shard_* (avian_amd/csrc/avn_shard.cpp) ----
struct avn_shard {
    uint32_t rank = 0, n_colliders = 0;
    std::map<uint32_t, uint32_t> slot_of_entity;
    std::vector<uint32_t> order;
    std::vector<double> minx;
    avo::ConstraintGraph graph;
    std::set<uint32_t> free_ids;
    uint32_t next_id = 0;
    struct Pair { uint32_t c1, c2; int32_t b1, b2; uint32_t owner, n_handles; };
    std::map<uint32_t, Pair> pairs;
    std::vector<uint32_t> active, new_ids, new_c1, new_c2, new_flags, removed_local, loc_handles, glob_handles;
    uint32_t loc_off[AVN_GRAPH_COLOR_COUNT + 1] = {0}, glob_off[AVN_GRAPH_COLOR_COUNT + 1] = {0};
    avn_shard_stats stats{};
    std::string error;
};
avn_status avo_shard_create(uint32_t n_colliders, const uint32_t* ents, uint32_t rank, avn_shard** out) {
    if (!out || (n_colliders && !ents)) return AVN_ERR_BAD_ARG;
    avn_shard* s = new (std::nothrow) avn_shard();
    if (!s) return AVN_ERR_OOM;
    s->rank = rank; s->n_colliders = n_colliders;
    for (uint32_t i = 0; i < n_colliders; ++i) { s->slot_of_entity[ents[i]] = i; s->order.push_back(i); }
    s->minx.assign(n_colliders, 0.0);
    *out = s;
    return AVN_OK;
}
void avo_shard_destroy(avn_shard* s) { delete s; }
const char* avo_shard_last_error(const avn_shard* s) { return s ? s->error.c_str() : ""; }
avn_status avo_shard_phase2(avn_shard* s, const uint32_t* kc, const double* kx, size_t n_keys, const avn_shard_pair* pairs, size_t n_pairs) {
    if (!s || (n_keys && (!kc || !kx)) || (n_pairs && !pairs)) return AVN_ERR_BAD_ARG;
    for (size_t i = 0; i < n_keys; ++i) { if (kc[i] >= s->n_colliders) return AVN_ERR_BAD_ARG; s->minx[kc[i]] = kx[i]; }
    // broad_phase.rs:479-487, literally: insertion sort, swap while prev.min.x > cur.min.x
    for (size_t i = 1; i < s->order.size(); ++i) {
        size_t j = i;
        while (j > 0 && s->minx[s->order[j - 1]] > s->minx[s->order[j]]) { std::swap(s->order[j - 1], s->order[j]); --j; }
    }
    std::vector<uint32_t> pos(s->n_colliders);
    for (uint32_t i = 0; i < s->n_colliders; ++i) pos[s->order[i]] = i;
    // the single world's emission order: for i over the sorted intervals, for j > i (broad_phase.rs:387-388) -- an ordered map keyed by (i, j)
    std::map<std::pair<uint32_t, uint32_t>, avn_shard_pair> emitted;
    for (size_t i = 0; i < n_pairs; ++i) {
        auto a = s->slot_of_entity.find(pairs[i].collider1), b = s->slot_of_entity.find(pairs[i].collider2);
        if (a == s->slot_of_entity.end() || b == s->slot_of_entity.end()) { s->error = "shard_phase2: unknown collider"; return AVN_ERR_BAD_ARG; }
        emitted[{pos[a->second], pos[b->second]}] = pairs[i];
    }
    s->new_ids.clear(); s->new_c1.clear(); s->new_c2.clear(); s->new_flags.clear();
    for (const auto& kv : emitted) {
        if (kv.first.first >= kv.first.second) { s->error = "shard_phase2: collider1 must be the earlier interval"; return AVN_ERR_STATE; }
        uint32_t id;
        if (!s->free_ids.empty()) { id = *s->free_ids.begin(); s->free_ids.erase(s->free_ids.begin()); } else id = s->next_id++;   // IdPool::alloc_id
        const avn_shard_pair& q = kv.second;
        s->pairs[id] = {q.collider1, q.collider2, q.body1, q.body2, q.owner, 0u};
        if (q.owner == s->rank) { s->new_ids.push_back(id); s->new_c1.push_back(q.collider1); s->new_c2.push_back(q.collider2); s->new_flags.push_back(q.flags); s->active.push_back(id); }
    }
    s->stats.pairs_added += (uint32_t)n_pairs; s->stats.next_id = s->next_id; s->stats.n_free = (uint32_t)s->free_ids.size();
    return AVN_OK;
}
avn_status avo_shard_new_local_pairs(avn_shard* s, const uint32_t** ids, const uint32_t** c1, const uint32_t** c2, const uint32_t** fl, size_t* n) {
    if (!s || !ids || !c1 || !c2 || !fl || !n) return AVN_ERR_BAD_ARG;
    *ids = s->new_ids.data(); *c1 = s->new_c1.data(); *c2 = s->new_c2.data(); *fl = s->new_flags.data(); *n = s->new_ids.size();
    return AVN_OK;
}
avn_status avo_shard_active(avn_shard* s, const uint32_t** ids, size_t* n) { if (!s || !ids || !n) return AVN_ERR_BAD_ARG; *ids = s->active.data(); *n = s->active.size(); return AVN_OK; }
avn_status avo_shard_phase3(avn_shard* s, const avn_contact_change* changes, size_t n) {
    if (!s || (n && !changes)) return AVN_ERR_BAD_ARG;
    std::map<uint32_t, avn_contact_change> by_id;   // ContactStatusBits: walked in ascending id (system_param.rs:141-145)
    for (size_t i = 0; i < n; ++i) by_id[changes[i].contact_id] = changes[i];
    s->removed_local.clear();
    std::vector<uint32_t> removed;
    for (const auto& kv : by_id) {
        const uint32_t cid = kv.first, flags = kv.second.flags;
        auto it = s->pairs.find(cid);
        if (it == s->pairs.end()) { s->error = "shard_phase3: no such contact"; return AVN_ERR_STATE; }
        avn_shard::Pair& p = it->second;
        const bool generates = flags & AVN_CP_GENERATE_CONSTRAINTS, touching = flags & AVN_CP_TOUCHING;
        auto push = [&](uint32_t k) { for (uint32_t i = 0; i < k; ++i) { s->graph.push_manifold(((uint64_t)cid << 8) | p.n_handles, (uint32_t)p.b1, (uint32_t)p.b2, flags & AVN_CP_STATIC1, flags & AVN_CP_STATIC2); ++p.n_handles; ++s->stats.pushes; } };
        auto pop = [&](uint32_t k) { for (uint32_t i = 0; i < k && p.n_handles; ++i) { --p.n_handles; s->graph.pop_manifold(((uint64_t)cid << 8) | p.n_handles); ++s->stats.pops; } };
        if (flags & AVN_CP_DISJOINT_AABB) { if (generates) pop(p.n_handles); removed.push_back(cid); if (p.owner == s->rank) s->removed_local.push_back(cid); }
        else if (flags & AVN_CP_STARTED_TOUCHING) { if (generates) push(kv.second.manifold_count); }
        else if (flags & AVN_CP_STOPPED_TOUCHING) { if (generates) pop(p.n_handles); }
        else if (touching && (flags & AVN_CP_STARTED_GENERATING_CONSTRAINTS)) push(kv.second.manifold_count);
        else if (touching && generates && kv.second.manifold_count_change > 0) push((uint32_t)kv.second.manifold_count_change);
        else if (touching && generates && kv.second.manifold_count_change < 0) pop((uint32_t)(-kv.second.manifold_count_change));
    }
    for (uint32_t cid : s->removed_local) s->active.erase(std::remove(s->active.begin(), s->active.end(), cid), s->active.end());
    for (uint32_t cid : removed) { s->pairs.erase(cid); s->free_ids.insert(cid); }
    s->stats.pairs_removed += (uint32_t)removed.size(); s->stats.next_id = s->next_id; s->stats.n_free = (uint32_t)s->free_ids.size(); s->stats.last_status_changes = (uint32_t)n;
    s->glob_handles.clear(); s->loc_handles.clear();
    for (int c = 0; c < AVN_GRAPH_COLOR_COUNT; ++c) {
        s->glob_off[c] = (uint32_t)s->glob_handles.size(); s->loc_off[c] = (uint32_t)s->loc_handles.size();
        for (const auto& h : s->graph.colors[c].manifold_handles) {
            const uint32_t cid = (uint32_t)(h.handle >> 8);
            s->glob_handles.push_back(cid);
            if (s->pairs.at(cid).owner == s->rank) s->loc_handles.push_back(cid);
        }
    }
    s->glob_off[AVN_GRAPH_COLOR_COUNT] = (uint32_t)s->glob_handles.size(); s->loc_off[AVN_GRAPH_COLOR_COUNT] = (uint32_t)s->loc_handles.size();
    return AVN_OK;
}
avn_status avo_shard_removed_local(avn_shard* s, const uint32_t** ids, size_t* n) { if (!s || !ids || !n) return AVN_ERR_BAD_ARG; *ids = s->removed_local.data(); *n = s->removed_local.size(); return AVN_OK; }
avn_status avo_shard_handles(avn_shard* s, int global, uint32_t* offsets, const uint32_t** ids, size_t* n) {
    if (!s || !offsets || !ids || !n) return AVN_ERR_BAD_ARG;
    std::memcpy(offsets, global ? s->glob_off : s->loc_off, sizeof s->loc_off);
    const std::vector<uint32_t>& v = global ? s->glob_handles : s->loc_handles;
    *ids = v.data(); *n = v.size();
    return AVN_OK;
}
avn_status avo_shard_stats_get(avn_shard* s, avn_shard_stats* o) { if (!s || !o) return AVN_ERR_BAD_ARG; *o = s->stats; return AVN_OK; }
}
