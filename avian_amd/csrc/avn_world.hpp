// avn_world.hpp — host side of the MI355X physics step (C++), mirroring the reference's plugin/system
// structure for the hot path:
//
//   reference plugin (file)                                   here
//   ---------------------------------------------------------+-----------------------------------------------
//   SolverBodyPlugin   (solver/solver_body/plugin.rs:21-122)  World::prepare_solver_bodies / writeback_solver_bodies
//   IntegratorPlugin   (integrator/mod.rs:45-88)              World::pre_process_velocity_increments / integrate_*
//   SolverPlugin       (solver/plugin.rs:88-151)              World::prepare_contact_constraints / warm_start /
//                                                             solve_contacts / solve_restitution / store_contact_impulses
//   XpbdSolverPlugin   (solver/xpbd/plugin.rs:21-110)         World::prepare_joints / xpbd_solve / xpbd_velocity_projection
//   BroadPhasePlugin   (collision/broad_phase.rs:33-170)      World::update_aabb / collect_collision_pairs
//   SolverSchedulePlugin (solver/schedule.rs:17-72)           World::substep / World::solver (system ORDER only)
//
// Everything runs on one HIP stream owned by the world; the substep loop can be replayed from a hipGraph.
// There is NO CPU fallback: without a gfx950 device world creation fails with AVN_ERR_NO_DEVICE.
#pragma once
#include <cstdlib>
#include <hip/hip_runtime.h>

#include <string>
#include <unordered_map>
#include <vector>

#include "avn_kernels.h"

namespace avn {

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    ~DevBuf();
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    // grows (never shrinks); returns true when the pointer changed; contents are NOT preserved unless keep
    bool ensure(size_t bytes, hipError_t& err, bool keep = false, hipStream_t s = nullptr);
    template <class U> U* as() const { return (U*)p; }
};

// Host-built execution schedule that makes a serial joint loop parallel without changing its result
// (see k_xpbd.hip header).
struct JointSchedule {
    std::vector<uint32_t> comp_level_begin, level_offsets, order;
    // the same items ordered by LEVEL only (items of one level never share a key, whatever their component): for kernels
    // that run one launch per level over the whole device instead of one workgroup per component
    std::vector<uint32_t> glevel_offsets, gorder;
    DevBuf d_gorder;
    uint32_t n_components = 0;
    bool touches_dummy = false;
    DevBuf d_comp_level_begin, d_level_offsets, d_order;
    DevBuf d_rec;   // joints only: (joint, body1, body2, local slots) per schedule slot, so a level needs no index walk through memory (k_xpbd.hip)
    // the LDS form of the walk (k_joint_schedule_lds): bodies per component (0xFFFFFFFF = global walk), the dynamic LDS the largest staged component needs (0: none staged)
    std::vector<uint32_t> comp_bodies;
    DevBuf d_comp_bodies;
    uint32_t lds_bytes = 0;
    // bodies: per joint the two scheduling keys (body index, or -1 = does not serialise), in joint order
    // levels_only: fill glevel_offsets / gorder only (no components: n_components stays 0)
    void build(const std::vector<uint32_t>& joints, const std::vector<int32_t>& key1, const std::vector<int32_t>& key2, uint32_t n_keys, bool levels_only = false);
};

struct WorldBase {
    std::string error;
    virtual ~WorldBase() {}
    virtual void bind() = 0;  // make the world's device current on the calling thread
    virtual void touched() {}  // called by every entry point that can change the world's state other than avn_step (ends the sleeping world's identity steps)
    virtual avn_status config_set(const avn_config*) = 0;
    virtual avn_status bodies_upload(const avn_bodies*) = 0;
    virtual avn_status bodies_download(const avn_bodies_out*) = 0;
    virtual avn_status solver_bodies_download(const avn_solver_bodies_out*) = 0;
    virtual avn_status manifolds_upload(const avn_manifolds*) = 0;
    virtual avn_status impulses_download(const avn_impulses_out*) = 0;
    virtual avn_status constraints_download(const avn_constraints_out*) = 0;
    virtual avn_status distance_joints_upload(const avn_distance_joints*) = 0;
    virtual avn_status joints_upload(const avn_joints*) = 0;
    virtual avn_status joints_download(const avn_joints_out*) = 0;
    virtual avn_status colliders_upload(const avn_colliders*) = 0;
    virtual avn_status collider_transforms_upload(const avn_collider_transforms*) = 0;
    virtual avn_status local_accelerations_upload(uint32_t count, const void* linear, const void* angular) = 0;
    virtual avn_status existing_pairs_upload(const uint64_t*, size_t) = 0;
    virtual avn_status pairs_get(const avn_pair**, size_t*) = 0;
    virtual avn_status aabbs_download(void*, void*, uint32_t*, size_t*) = 0;
    virtual avn_status run_system(avn_system) = 0;
    virtual avn_status step() = 0;
    virtual avn_status synchronize() = 0;
    virtual avn_status timers(avn_timers*) = 0;
    virtual avn_status diagnostics(avn_diagnostics*) = 0;
    virtual avn_status profile_system(avn_system, uint32_t, double*, uint32_t*) = 0;
    virtual avn_status dynamic_bounds(double*, double*) = 0;
    virtual avn_status contact_manifolds(const avn_shape_pairs*, const avn_query_manifolds_out*) = 0;
    virtual avn_status collider_materials_upload(const avn_collider_materials*) = 0;
    virtual avn_status contact_pairs_add(const avn_contact_pairs*) = 0;
    virtual avn_status contact_pairs_remove(const uint32_t*, size_t) = 0;
    virtual avn_status active_pairs_set(const uint32_t*, size_t) = 0;
    virtual avn_status contact_changes_get(const avn_contact_change**, size_t*) = 0;
    virtual avn_status manifold_handles_upload(const uint32_t*, const uint32_t*) = 0;
    virtual avn_status contacts_download(const uint32_t*, size_t, const avn_contacts_out*) = 0;
    virtual avn_status contacts_upload(const uint32_t*, size_t, const avn_contacts_in*) = 0;
    virtual avn_status pipeline_enable(int) = 0;
    virtual avn_status pipeline_stats_get(avn_pipeline_stats*) = 0;
    virtual avn_status pipeline_handles_get(uint32_t*, const uint32_t**, size_t*) = 0;
    virtual avn_status pipeline_new_pair_ids_get(const uint32_t**, size_t*) = 0;
    virtual avn_status host_shapes_set(avn_host_aabb_fn, avn_host_manifolds_fn, void*) = 0;
    virtual avn_status host_shape_stats_get(avn_host_shape_stats*) = 0;
    virtual avn_status collision_hooks_set(avn_filter_pairs_fn, avn_modify_contacts_fn, void*) = 0;
    virtual avn_status collision_hook_stats_get(avn_collision_hook_stats*) = 0;
    virtual avn_status halo_plan_upload(const avn_halo_plan*) = 0;
    virtual avn_status halo_overflow_levels_upload(uint32_t, const uint32_t*, size_t) = 0;
    virtual avn_status halo_joint_slot_set(uint32_t, uint32_t) = 0;
    virtual avn_status run_color_pass(avn_system, uint32_t) = 0;
    virtual avn_status halo_pack(uint32_t, uint32_t, void*, size_t*) = 0;
    virtual avn_status halo_unpack(uint32_t, uint32_t, const void*, size_t) = 0;
    virtual avn_status comm_init(const uint8_t*, int, int) = 0;
    virtual avn_status bounds_exchange(double*, uint32_t, uint32_t*, uint32_t*, uint32_t, uint32_t*) = 0;
    virtual avn_status islands_get(uint32_t*, uint32_t*) = 0;
    virtual avn_status sleep_update(const avn_sleep_params*, avn_sleep_stats*) = 0;
    virtual avn_status sleep_get(const avn_sleep_out*) = 0;
    virtual avn_status sleep_reset(const uint32_t*, size_t) = 0;
    virtual avn_status sleeping_enable(const avn_sleep_params*) = 0;
    virtual avn_status sleeping_stats_get(avn_sleeping_stats*) = 0;
    virtual avn_status sleeping_state_get(const avn_sleeping_out*) = 0;
    virtual avn_status wake_bodies(const uint32_t*, size_t) = 0;
    virtual avn_status despawn(const avn_despawn_list*) = 0;
    virtual avn_status dshard_enable(const avn_dshard_config*) = 0;
    virtual avn_status dshard_bodies_pack(void*, size_t, size_t*) = 0;
    virtual avn_status dshard_bodies_unpack(uint32_t, const void*, size_t) = 0;
    virtual avn_status dshard_stats_get(avn_dshard_stats*) = 0;
};

// RCCL transport of the level-2 halo exchange (avn_comm.cpp; librccl is opened on first use)
struct CommXfer { void* ptr; size_t bytes; int peer; };
struct Comm {
    void* handle = nullptr;   // ncclComm_t
    int n_ranks = 0, rank = 0;
    ~Comm();
    avn_status init(const uint8_t* unique_id, int n_ranks, int rank, std::string& err);
    avn_status exchange(const CommXfer* sends, size_t n_sends, const CommXfer* recvs, size_t n_recvs, hipStream_t s, std::string& err);
    avn_status all_gather(const void* send, void* recv, size_t bytes_per_rank, hipStream_t s, std::string& err);
};
avn_status comm_unique_id(uint8_t* out, std::string& err);

WorldBase* make_world_f32(const avn_config* cfg, avn_status* st, std::string* err);
WorldBase* make_world_f64(const avn_config* cfg, avn_status* st, std::string* err);

// Host ConstraintGraph (reference solver/constraint_graph.rs:129-296)
struct ConstraintGraphHost {
    struct Handle { uint64_t handle; uint32_t body1, body2; };
    struct Color { std::vector<uint64_t> body_bits; std::vector<Handle> manifold_handles; };
    struct Loc { uint8_t color; uint32_t local_index; };
    Color colors[AVN_GRAPH_COLOR_COUNT];
    std::unordered_map<uint64_t, Loc> where;
    int push_manifold(uint64_t handle, uint32_t body1, uint32_t body2, bool is_static1, bool is_static2);
    bool pop_manifold(uint64_t handle);
    void clear() { for (auto& c : colors) { c.body_bits.clear(); c.manifold_handles.clear(); } where.clear(); }
};

}  // namespace avn
