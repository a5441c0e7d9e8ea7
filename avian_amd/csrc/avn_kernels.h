// avn_kernels.h — launch interface of the gfx950 kernels (implemented in k_*.hip).
#pragma once
#include <type_traits>

#include "avn_device.h"

namespace avn {

// broad-phase state resident in HBM
template <class T> struct BP {
    using Key = typename std::conditional<sizeof(T) == 4, uint32_t, uint64_t>::type;
    uint32_t n_colliders, n_intervals;
    uint4* col_info;       // (entity index, body index, shape | collider_flags << 8, 0) per collider slot
    Vec4<T>* col_he;       // (half_extents.xyz, collision_margin)
    T* col_spec;           // SpeculativeMargin (< 0 = absent)
    uint2* col_layers;     // (memberships, filters)
    Vec4<T>* aabb_min;     // ColliderAabb per collider slot
    Vec4<T>* aabb_max;
    uint32_t* iv_collider; // AabbIntervals: collider slot per interval, persistent sorted order
    T* s_minx;             // sorted interval records: min.x (the sort key), max.x
    T* s_maxx;
    Vec4<T>* s_yz;         // (min.y, max.y, min.z, max.z)
    uint32_t* s_end;       // end(i): first j > i with min_x[j] > max_x[i]  (i + 1 for dropped / long intervals)
    uint4* s_info;         // (entity, body, memberships, filters)
    uint32_t* s_flags;     // AabbIntervalFlags | AVN_IV_DROPPED | AVN_IV_LONG
    uint64_t* pair_set;    // ContactGraph::pair_set as an open-addressing hash set (EMPTY = ~0)
    uint32_t pair_set_cap; // power of two, 0 = no set
    uint64_t* disabled_set;  // body pairs whose joints disable collision
    uint32_t disabled_cap;
    Vec4<T>* s_bb;         // per group of 8 consecutive sorted records: (min of min.y, max of max.y, min of min.z, max of max.z) -- the sweep's batch cull
    Vec4<T>* s_bb2;        // the same bounds per 64 consecutive sorted records (second level of the cull)
    // child colliders (avn_collider_transforms_upload): ColliderTransform per collider slot; nullptr = every collider sits on its body's entity
    Vec4<T>* col_lpos;     // (translation.xyz, 1 = child | 0)
    Vec4<T>* col_lrot;     // rotation xyzw
};
// update_child_collider_position (collision/collider/collider_transform/plugin.rs:62-91): a collider's Position / Rotation from its body's
template <class T> __device__ __forceinline__ void collider_pose(const BP<T>& bp, uint32_t slot, V3<T> body_pos, Q4<T> body_rot, V3<T>& pos, Q4<T>& rot, bool* child = nullptr) {
    pos = body_pos; rot = body_rot;
    if (child) *child = false;
    if (!bp.col_lpos) return;
    const Vec4<T> lp = bp.col_lpos[slot];
    if (lp.w == T(0)) return;
    pos = body_pos + qrot(body_rot, xyz<T>(lp));
    rot = qnormalize(qmul(body_rot, quat<T>(bp.col_lrot[slot])));
    if (child) *child = true;
}
#define AVN_IV_DROPPED 0x80000000u
#define AVN_IV_LONG 0x40000000u   // > SW_CAP sweep candidates: swept by k_sweep_long in chunks

// scratch of the sweep's long-interval path
struct SweepScratch {
    void* long_items;       // LongItem[long_cap]
    uint32_t* long_counts;  // [long_cap] pairs found per chunk
    uint32_t* long_off;     // [long_cap] chunk offset inside its interval's output range
    uint32_t* n_long;       // [2] device counters: chunks used, overflow flag
    uint32_t long_cap;
    uint32_t* hits;         // [sweep_hit_words(n)] per wave of k_sweep: (pairs found by the count pass, the first few as (lane, j) records)
};
size_t sweep_hit_words(uint32_t n_intervals);

enum { PASS_WARM_START = 0, PASS_SOLVE_BIAS = 1, PASS_SOLVE_RELAX = 2, PASS_RESTITUTION_ = 3, PASS_WARM_START_COLORS = 4 /* manifold-centric, one launch per colour */, PASS_MEMORY_SKELETON = 5 /* loads + stores of the solve pass, no solve */ };

// k_bodies.hip
template <class T> void launch_prepare_solver_bodies(const DW<T>&, hipStream_t);
template <class T> void launch_pre_process_increments(const DW<T>&, const StepParams<T>&, hipStream_t);
template <class T> void launch_clear_increments(const DW<T>&, hipStream_t);
template <class T> void launch_integrate_velocities(const DW<T>&, const StepParams<T>&, hipStream_t);
template <class T> void launch_integrate_positions(const DW<T>&, const StepParams<T>&, hipStream_t);
template <class T> void launch_writeback_solver_bodies(const DW<T>&, hipStream_t);
template <class T> void launch_xpbd_snapshot(const DW<T>&, hipStream_t);
template <class T> void launch_xpbd_velocity_projection(const DW<T>&, const StepParams<T>&, hipStream_t);
// k_contacts.hip
// the contact table as k_prepare_contact_constraints reads it in handle mode (manifold m <- row handles[m]); CT / BP are declared further down
template <class T> struct RowsView {
    const uint32_t* handles; const uint4* meta; const uint4* col_info; const Vec4<T>* rows;   // CT<T>::rows: 16 records per row (n | tv | a1[4] | a2[4] | w[4] | fid)
};
template <class T> void launch_prepare_contact_constraints(const DW<T>&, const StepParams<T>&, hipStream_t, bool count_clean = false /* *DW::constraint_count is known to be zero */,
                                                           const RowsView<T>* rows = nullptr /* handle mode: read the ContactGraph side from the table (no k_gather_manifolds) */);
template <class T> void launch_store_contact_impulses(const DW<T>&, hipStream_t);
// body-centric warm start over the incidence CSR (DW::inc_off / inc_ent), optionally preceded by integrate_velocities
template <class T> void launch_body_warm_start(const DW<T>&, const StepParams<T>&, bool fuse_integrate_velocities, bool quads /* four lanes per body: the device closed loop */, hipStream_t);
// (re)build DW::inc_slot from DW::m_bodies / color_offsets (memset + one kernel)
template <class T> void launch_build_incidence_slots(const DW<T>&, hipStream_t, bool cleared = false /* the table has been set to EMPTY already */);
uint32_t color_grid_blocks(uint32_t count);
// level schedule of the overflow colour (device arrays; see k_overflow_pass): manifold indices in `order`
struct OverflowSchedule {
    uint32_t n_components; const uint32_t *comp_level_begin, *level_offsets, *order;   // one workgroup per component (device arrays)
    const uint32_t* gorder; const uint32_t* glevel_offsets /* HOST array */; uint32_t n_glevels;  // or: one launch per level (n_glevels != 0)
};
// grid_blocks[c] = captured grid of colour c (0 = colour skipped); arg_offsets: the 25 colour offsets to pass in the kernel
// arguments, or nullptr = the kernels read the live ranges from DW::color_offsets; returns the number of launches issued
// oct_mask: bit c = colour c runs the eight-lanes-per-manifold kernel (f32 biased solve / relax only; bit-identical: a scheduling choice)
template <class T> uint32_t launch_contact_pass(const DW<T>&, const StepParams<T>&, int pass, const uint32_t* grid_blocks, const uint32_t* arg_offsets, const OverflowSchedule&, hipStream_t, uint32_t oct_mask = 0u);
// Island blocks (k_island_substeps): block b owns bodies[body_off[b] .. body_off[b+1]) (world body indices; LDS slot = position in
// the range) and, per colour slot (0 = the overflow colour, 1 + c = colour c: solve order), the entries
// ent[col_off[24 b + slot] .. col_off[24 b + slot + 1]) = (manifold index, LDS slot of body1 | LDS slot of body2 << 16); a side
// without a SolverBody points at slot 0 and is masked by the constraint's NOBODY flag.
#define ISLAND_THREADS 256
#define ISLAND_MAX_BODIES 512
#define ISLAND_LDS_VEC4 4088u   // 64 KB of f32 Vec4: 6 records per body (+ 20 per manifold when the constraint records are staged too)
// cache_records: every block satisfies 6 max_bodies + 20 max_manifolds <= ISLAND_LDS_VEC4 (max_* = the largest block's counts; the LDS
// layout is the same for all blocks): the kernel then stages the block's constraint records and entries in LDS as well
struct IslandBlocks { const uint32_t *body_off, *bodies, *col_off; const uint2* ent; uint32_t n_blocks, max_bodies, max_manifolds, cache_records; };
void launch_island_substeps(const DW<float>&, const StepParams<float>&, const IslandBlocks&, uint32_t substeps, uint32_t iterations, hipStream_t);
// k_xpbd.hip
template <class T> void launch_prepare_joints(const DW<T>&, hipStream_t);
// the same walk with a component's bodies and joints staged in LDS (k_xpbd.hip, round 6): rec.w = local slot of body1 | body2 << 16, comp_bodies[c] = bodies of component c
// (0xFFFFFFFF: too large, global walk), lds_bytes = the largest staged component
template <class T> void launch_joint_schedule_lds(const DW<T>&, const StepParams<T>&, uint32_t n_components, const uint32_t* comp_level_begin, const uint32_t* level_offsets, const int4* rec,
                                                  const uint32_t* comp_bodies, uint32_t lds_bytes, hipStream_t);
template <class T> void launch_joint_schedule(const DW<T>&, const StepParams<T>&, int op, uint32_t n_components, const uint32_t* comp_level_begin,
                                              const uint32_t* level_offsets, const int4* rec, hipStream_t);
template <class T> void launch_writeback_joint_forces(const DW<T>&, const StepParams<T>&, hipStream_t);
// k_broadphase.hip
// zero_words[0 .. n_zero) (n_zero <= 256) are cleared by the kernel: the step-scoped counters of what follows; returns whether a kernel ran
template <class T> bool launch_update_aabb(const DW<T>&, const BP<T>&, const StepParams<T>&, hipStream_t, uint32_t* zero_words = nullptr, uint32_t n_zero = 0);
// counters_clean: n_dropped[0..2) is known to be zero (no memset launch)
template <class T> void launch_interval_keys(const DW<T>&, const BP<T>&, typename BP<T>::Key* keys, uint32_t* vals, uint32_t* n_dropped, hipStream_t, bool counters_clean = false);
// partial: 6 * ceil(C / 256) scalars; returns the number of partial records written
template <class T> uint32_t launch_dynamic_bounds(const DW<T>&, const BP<T>&, T* partial, hipStream_t);
uint32_t radix_blocks(uint32_t n);
uint32_t radix_pass_launches(uint32_t n);  // kernels per 8-bit pass of launch_radix_sort
uint32_t radix_sort_launches(uint32_t n, uint32_t key_bytes);  // kernels launch_radix_sort issues in total
uint32_t scan_block_sums_needed(uint32_t n);
uint32_t exclusive_scan_launches(uint32_t n);  // kernels launch_exclusive_scan issues for n items
// `enabled` (device flag, may be null): when it reads 0 every kernel of the call returns immediately
template <class K> void launch_radix_sort(K* keys_a, uint32_t* vals_a, K* keys_b, uint32_t* vals_b, uint32_t n, uint32_t* hist, uint32_t* block_sums,
                                          const uint32_t* enabled, K** keys_out, uint32_t** vals_out /* where the result is: one of the two buffer pairs */, hipStream_t);
void launch_exclusive_scan(const uint32_t* in, uint32_t* out, uint32_t n, uint32_t* block_sums, uint32_t* total, hipStream_t, const uint32_t* enabled = nullptr);
template <class T> void launch_gather_sorted(const DW<T>&, const BP<T>&, const uint32_t* sorted_collider, uint32_t n, hipStream_t);
template <class T> void launch_sweep_ranges(const BP<T>&, uint32_t n, const SweepScratch&, hipStream_t, bool counters_clean = false /* SweepScratch::n_long[0..2) is known to be zero */);
template <class T> void launch_sweep(const BP<T>&, uint32_t n, bool emit, const SweepScratch&, uint32_t* counts, const uint32_t* offsets, avn_pair* out, hipStream_t);
size_t sweep_long_item_bytes();
template <class T> void launch_bounds_reduce(const T* partial, uint32_t n_partials, double* out6, hipStream_t);   // k_dynamic_bounds' partials -> one (min, max)
uint32_t sweep_pad_records();
uint32_t sweep_bounds_group();  // sorted records per y/z bounds group of the sweep's batch cull
uint32_t sweep_bounds_words(uint32_t n_records);          // Vec4 records of BP::s_bb for both cull levels
uint32_t sweep_bounds_level2_offset(uint32_t n_records);  // BP::s_bb2 = BP::s_bb + this
uint32_t sweep_count_slots();  // counts / offsets entries per interval (the sweep keeps one per candidate-range quarter)
void launch_hs_insert(uint64_t* tab, uint32_t cap, const uint64_t* keys, uint32_t n, hipStream_t);
void launch_hs_insert_pairs(uint64_t* tab, uint32_t cap, const avn_pair* pairs, uint32_t n, hipStream_t);
// The ContactGraph side of the narrow phase in HBM: rows indexed by ContactId, one manifold per (convex) pair.  Round 4: the manifold's records are
// ONE BLOCK PER ROW (16 Vec4<T>: 256 B in f32, 512 B in f64) instead of [p][row] planes: only the ~17 % of the rows that hold a manifold are ever
// read or written, and their consumers -- the narrow phase's match_contacts, the constraint generation through the handle lists, the impulse
// write-back -- name rows in no particular order, so a plane layout cost a whole 128-byte line per 16-byte record (PMC, settled cfg2:
// k_prepare_contact_constraints fetched 461 MB for 92 MB of records).  `meta` and `dcount`, which every launch sweeps densely, stay flat arrays.
#define AVN_CT_ROW_V4 16u
template <class T> struct CT {
    uint32_t cap;      // rows allocated
    uint4* meta;       // (collider slot 1, collider slot 2, AVN_CP_* flags, n_manifolds | point_count << 8)
    int32_t* dcount;   // ContactPair::manifold_count_change
    Vec4<T>* rows;     // [cap][16]: n | tv | a1[4] | a2[4] | w[4] | fid[4] (uint2, in the last two records)
    __host__ __device__ __forceinline__ Vec4<T>* row(uint32_t c) const { return rows + (size_t)c * AVN_CT_ROW_V4; }
    __host__ __device__ __forceinline__ Vec4<T>& n(uint32_t c) const { return row(c)[0]; }                 // (normal.xyz, friction)
    __host__ __device__ __forceinline__ Vec4<T>& tv(uint32_t c) const { return row(c)[1]; }                // (tangent_velocity.xyz, restitution)
    __host__ __device__ __forceinline__ Vec4<T>& a1(uint32_t c, uint32_t k) const { return row(c)[2 + k]; }    // (anchor1.xyz, penetration)
    __host__ __device__ __forceinline__ Vec4<T>& a2(uint32_t c, uint32_t k) const { return row(c)[6 + k]; }    // (anchor2.xyz, normal_speed)
    __host__ __device__ __forceinline__ Vec4<T>& w(uint32_t c, uint32_t k) const { return row(c)[10 + k]; }    // (warm_start_normal, warm_start_tangent.x, .y, normal_impulse)
    __host__ __device__ __forceinline__ uint2& fid(uint32_t c, uint32_t k) const { return reinterpret_cast<uint2*>(row(c) + 14)[k]; }   // (feature_id1, feature_id2)
    Vec4<T>* col_mat;  // per collider slot: (friction, restitution, bits(friction_combine | restitution_combine << 8), 0)
    // the narrow phase's hand-over between its two kernels: the cuboid pairs that survive the SAT (k_narrow.hip)
    uint32_t* np_row;  // [cap + slack] row ids: 64 lists, arbitrary order inside a list
    T* np_axis;        // [3 (cap + slack)] the separating direction the SAT found for np_row[k]
    uint32_t* np_ctr;  // per list, a cache line apart: (entries, workgroups of the second kernel that are done); zero between launches
};
size_t np_survivor_list_slack();     // entries np_row / np_axis need beyond `cap`
size_t np_survivor_counter_bytes();  // size of np_ctr
template <class T> void launch_init_contact_rows(const CT<T>&, const uint32_t* ids, const uint32_t* slot1, const uint32_t* slot2, const uint32_t* pair_flags, uint32_t n, hipStream_t);
template <class T> void launch_clear_contact_rows(const CT<T>&, const uint32_t* ids, uint32_t n, hipStream_t);
// host shapes (include/avian_mi355x.h): where the light kernel leaves the queries of pairs with an AVN_SHAPE_HOST collider; queries == nullptr: the world holds none
// (the plain kernels run).  host_only: a retry after the list overflowed -- only those pairs are visited (they wrote nothing the first time), everything else is skipped
// collision hooks (include/avian_mi355x.h): CollisionHooks::modify_contacts for the pairs flagged AVN_CP_MODIFY_CONTACTS that have a manifold.  Three passes over such a pair:
// phase 1 (inside the step's ordinary launches): the pair stops at the hook point, writes NOTHING but AVN_CP_ROW_HOOK_PENDING and is counted; phase 2 (the same launches
// again, visiting only pending rows): the ContactPair as the hook sees it goes to records[0 .. *count) (the host sized the list from phase 1's count); phase 3
// (launch_narrow_phase_hooked): the pair again from the top with the record the hook returned in the place of its manifold -- match_contacts, status change, row.
// count == nullptr: no modify hook is registered (the plain kernels run unless the world holds host shapes).
struct NpHookList { void* records = nullptr; uint32_t* count = nullptr; uint32_t cap = 0; uint32_t phase = 0; };
// locals: the world holds child colliders (BP::col_lpos): their poses are computed by the HS instantiations only
struct NpHostList { void* queries = nullptr; uint32_t* count = nullptr; uint32_t cap = 0; uint32_t host_only = 0; NpHookList hook; bool locals = false;
                    bool any() const { return queries != nullptr || hook.count != nullptr || locals; } };
template <class T> void launch_narrow_phase_hooked(const DW<T>&, const BP<T>&, const CT<T>&, const StepParams<T>&, bool dense, avn_contact_change* changes, uint32_t* n_changes, uint32_t* chg,
                                                   uint32_t* has, const void* records /* sorted by contact id, as the hook left them */, uint32_t n, hipStream_t);
template <class T> void launch_narrow_phase_host(const DW<T>&, const BP<T>&, const CT<T>&, const StepParams<T>&, bool dense, avn_contact_change* changes, uint32_t* n_changes, uint32_t* chg,
                                                 uint32_t* has, const void* queries /* sorted by contact id */, const void* manifolds, uint32_t n, hipStream_t, const NpHookList& hook = NpHookList());
template <class T> void launch_host_aabb_queries(const DW<T>&, const BP<T>&, const StepParams<T>&, const uint32_t* slots, uint32_t n, void* out, hipStream_t);
template <class T> void launch_host_aabb_apply(const BP<T>&, const StepParams<T>&, const uint32_t* slots, uint32_t n, const void* in, hipStream_t);
// NarrowPhase::update_contacts over the active pairs; changes[0..*n_changes) in arbitrary order (the host sorts by id)
template <class T> void launch_narrow_phase(const DW<T>&, const BP<T>&, const CT<T>&, const StepParams<T>&, const uint32_t* active, uint32_t n_active,
                                            avn_contact_change* changes, uint32_t* n_changes, hipStream_t, const NpHostList& hl = NpHostList());
// dense form (device closed loop): every row id < n_rows with AVN_CP_ROW_USED is a pair; the row's status change goes to chg[id] / has[id]
// (id order = the order NarrowPhase::update walks the status bits) and rows that must be removed are counted in *n_remove
template <class T> void launch_narrow_phase_dense(const DW<T>&, const BP<T>&, const CT<T>&, const StepParams<T>&, uint32_t n_rows, uint32_t* chg, uint32_t* has,
                                                  uint32_t* n_remove, hipStream_t, bool reset_counter = true /* false: *n_remove is known to be zero (no memset launch) */, const NpHostList& hl = NpHostList());
template <class T> void launch_narrow_phase_rows(const DW<T>&, const BP<T>&, const CT<T>&, const StepParams<T>&, const uint32_t* list, uint32_t n_list, uint32_t range_base, uint32_t n_range,
                                                 uint32_t* chg, uint32_t* has, uint32_t* n_remove, hipStream_t, const NpHostList& hl = NpHostList());   // rows added this step (counter not reset)
// manifold m of the solver-side arrays <- row handles[m] of the contact table (GraphColor::manifold_handles indirection)
// store_contact_impulses' write into the ContactGraph (plugin.rs:744-749): table row <- DW::mp_w
template <class T> void launch_scatter_impulses(const DW<T>&, const CT<T>&, const uint32_t* handles, hipStream_t);
template <class T> struct ContactsStage {
    uint32_t* flags; uint8_t* point_count;
    T *normal, *friction, *restitution, *anchor1, *anchor2, *penetration, *normal_speed, *warm_n, *warm_t, *normal_impulse;
    uint32_t *feature_id1, *feature_id2;
};
template <class T> void launch_unpack_contacts(const CT<T>&, const uint32_t* ids, uint32_t n, const ContactsStage<T>&, hipStream_t);
template <class T> void launch_remap_row_slots(const CT<T>&, const uint32_t* map, uint32_t n_old, uint32_t apply, uint32_t* n_orphans, hipStream_t);   // collider slots of live rows after a re-upload
template <class T> void launch_pack_contacts(const CT<T>&, const uint32_t* ids, uint32_t n, const ContactsStage<T>&, uint32_t* error, hipStream_t);   // stage -> rows (ids without a live row: skipped, *error |= 4)
void launch_hs_remove(uint64_t* tab, uint32_t cap, const uint64_t* keys, uint32_t n, hipStream_t);
// CollisionHooks::filter_pairs in the device closed loop (k_graph.hip): the emitted pairs that ask for the filter; the emission list without the rejected ones
void launch_hook_filter_collect(const avn_pair* pairs, uint32_t total, avn_hook_pair* out /* [total] */, uint32_t* count, hipStream_t);
void launch_hook_filter_compact(const avn_pair* in, avn_pair* out, uint32_t total, const uint32_t* rejected /* ascending emission indices */, uint32_t n_rejected, hipStream_t);

// ---- k_graph.hip: the closed loop's integer bookkeeping ON THE DEVICE -------------------------------------------------------
// ContactGraph edge list + IdPool (data_structures/id_pool.rs:31-40), the status-change loop of NarrowPhase::update
// (collision/narrow_phase/system_param.rs:141-389) and the ConstraintGraph (solver/constraint_graph.rs:163-296), all in HBM, so
// that a step of the closed loop moves only a few counters over the bus.
#define PG_NONE 0xFFFFFFFFu
#define AVN_CP_ROW_USED 0x40000000u   // internal row flag (never reported through the ABI): the ContactId is live
#define AVN_CP_ROW_HOOK_PENDING 0x10000000u   // internal row flag, alive only inside one narrow phase: the pair waits for CollisionHooks::modify_contacts (k_narrow.hip)
#define AVN_CP_ROW_SLEEPING 0x20000000u   // internal row flag: ContactEdgeFlags::SLEEPING -- the pair is in ContactGraph::sleeping_pairs, the narrow phase does not update it
// counters block (uint32 words of PG::ctr)
enum { PGC_FREE_HEAD = 0, PGC_N_FREE = 1, PGC_NEXT_ID = 2, PGC_N_OPS = 3, PGC_N_REM = 4, PGC_ERROR = 5, PGC_TILE = 6 /* dynamic tile ids of k_pg_color */,
       PGC_N_PUSH = 7, PGC_N_POP = 8, PGC_REM_TOTAL = 9, PGC_ADD_DONE = 10 /* workgroups of k_pg_add_pairs that are done */, PGC_N_SLEEP_OPS = 11 /* status changes of this step that name a Sleeping body (sleeping on) */, PGC_LLEN = 228 /* [24] sharded closed loop: this rank's share of GraphColor::manifold_handles.len() */, PGC_SEQ = 12 /* [2] 64-bit count of pairs ever added: the edges' insertion stamps */,
       PGC_COLLECT = 14 /* k_pg_collect_edges: records written */, PGC_LEN = 32 /* [24] GraphColor::manifold_handles.len() */,
       PGC_BUCKET = 64 /* [26] ops per colour of this step -> offsets */, PGC_OFFSETS = 96 /* [25] colour offsets of the concatenated handles */,
       PGC_DBG = 130 /* [96] k_pg_replay diagnostics */, PGC_OVF_TILE = 256 /* [512] dynamic tile ids of the overflow passes of a step */, PGC_OVF_TILES = 512,
       PGC_SORT_DUP = 768 /* [24] k_pg_build_handles: manifolds of a colour whose key body was already taken (never, while the colouring's invariant holds) */, PGC_WORDS = 1024 };
struct PG {
    uint32_t rows;          // capacity of the per-row arrays (= CT::cap)
    int2* bodies;           // [rows] ContactPair::body1 / body2
    uint32_t* color;        // [rows] colour of the row's constraint handle (0..23) | PG_NONE
    uint32_t* lpos;         // [rows] ContactConstraintHandle::local_index
    uint32_t* lists;        // [24][list_stride] GraphColor::manifold_handles (contact ids; manifold index 0: convex pairs)
    uint32_t list_stride;
    uint32_t* bcol;         // [n_bodies] bit c set = the body is in GraphColor c's body_set
    uint32_t* free_ids;     // IdPool: free ids ascending, live part [ctr[FREE_HEAD], +ctr[N_FREE])
    uint32_t* free_alt;     // merge target (ping-pong)
    uint32_t* ctr;          // counters block
    const uint32_t* ent2slot;  // collider Entity::index() -> slot
    // per-step scratch
    uint32_t* chg;          // [rows] packed status change of the row (0 = none): flags | n_manifolds << 16 | (dcount + 128) << 24
    uint32_t* has;          // [rows] 0 | 1
    uint32_t* off;          // [rows] exclusive scan of has = op index
    uint32_t* op_cid;       // [ops]
    uint32_t* op_info;      // [ops] kind (0 none, 1 push, 2 pop) | static1 << 2 | static2 << 3 | remove << 4 | colour << 8
    int2* op_bodies;        // [ops]
    uint32_t* ekey_a, *eval_a, *ekey_b, *eval_b;   // [2 ops] (body | n_bodies = none, 2 op + side): sorted by body, stable
    uint32_t* epos;         // [2 ops] sorted position of entry 2 op + side
    uint32_t* popbefore;    // [2 ops] colours freed on the entry's body by this step's earlier pops
    uint32_t* prevpush;     // [2 ops] 1 + sorted position of the previous push entry of the same body | 0
    uint32_t* est;          // [2 ops] dataflow state of a push entry: colours taken on the body by this step's pushes so far | DONE
    uint32_t* tile_agg;     // [5 tiles] segmented-scan tile aggregates
    uint32_t* ckey_a, *cval_a, *ckey_b, *cval_b;   // [ops] (colour, op): bucketed by colour, stable
    uint32_t* rem_flag, *rem_off, *rem_ids;        // [ops]
    uint32_t* op_chg;       // [ops] the row's packed status change (PG::chg) next to op_cid: what the host-side island manager reads (sleeping enabled)
    uint32_t* new_ids;      // [new pairs of the step] the ContactId k_pg_add_pairs gave the i-th new pair (NULL: not recorded)
    unsigned long long* seq; // [rows] insertion stamp of the row's ContactEdge (the n-th pair ever added): a collider's edge list is its live edges by DESCENDING stamp
};
// avn_despawn: the edges of the removed colliders (rm_rank[slot] != PG_NONE); one record of 8 words per edge, any order
struct PGEdgeRec { uint32_t cid, slot1, slot2, flags, color, seq_lo, seq_hi, pad; };
template <class T> void launch_pg_collect_edges(const PG&, const CT<T>&, uint32_t n_rows, const uint32_t* rm_rank, PGEdgeRec* out, uint32_t cap, hipStream_t);
// rows `ids` (ascending) leave the table: cleared, PairKeys tombstoned, PG::rem_ids <- ids (for launch_pg_merge_free)
template <class T> void launch_pg_remove_list(const PG&, const CT<T>&, const BP<T>&, const uint32_t* ids, uint32_t n, hipStream_t);
// body renumbering after a despawn: PG::bodies of the live rows, joint bodies, and compaction of per-body arrays (dst[new_index[i]] = src[i])
template <class T> void launch_pg_renumber_rows(const PG&, const CT<T>&, uint32_t n_rows, const uint32_t* new_index, hipStream_t);
void launch_renumber_int2(int2* v, uint32_t n, const uint32_t* new_index, hipStream_t);
void launch_compact_u32(const uint32_t* src, uint32_t* dst, const uint32_t* new_index, uint32_t n_old, hipStream_t);
void launch_compact_u8(const uint8_t* src, uint8_t* dst, const uint32_t* new_index, uint32_t n_old, hipStream_t);
#define PG_EST_DONE 0x80000000u
// ids + rows + PairKeys of the step's new pairs and the IdPool counters, one launch (pair_set must have room: pg_pair_set_reserve)
template <class T> void launch_pg_add_pairs(const PG&, const CT<T>&, const avn_pair* pairs, uint32_t total, uint64_t* pair_set, uint32_t pair_set_cap, hipStream_t);
// exclusive scan of PG::has (= op index per changed row) + the classification of the changed rows, one launch; ctr[PGC_N_OPS] <- changes.
// PGC_BUCKET must be zero when it starts (k_pg_build_handles leaves it so).
// Sharded device closed loop (avn_dshard_enable, k_graph.hip / k_transfer.hip): this rank's share of every colour list, order kept (ctr[PGC_LLEN + c] <- its length; error bit 16:
// a manifold joins bodies of two ranks); FOREIGN flags from the owner array; the own bodies' components packed for / unpacked from the per-step all-gather
void launch_pg_local_lists(const PG&, const uint4* ct_meta, const int32_t* owner, uint32_t rank, uint32_t* local_lists, uint32_t* chunk_counts /* [24][n_chunks] */, uint32_t n_chunks /* of 2 048 entries: covers the longest list */, hipStream_t);
template <class T> void launch_dsh_set_foreign(const DW<T>&, const int32_t* owner, uint32_t rank, hipStream_t);
template <class T> void launch_dsh_pack(const DW<T>&, const uint32_t* bodies, uint32_t n, void* out, hipStream_t);
template <class T> void launch_dsh_unpack(const DW<T>&, const uint32_t* bodies, uint32_t n, const void* in, hipStream_t);
void launch_pg_scan_classify(const PG&, uint32_t n_rows, uint32_t n_bodies, uint32_t* scan_state, hipStream_t, const uint32_t* bmeta_if_sleeping = nullptr);
// split_island's neighbour lists as a CSR over bodies, from the rows (k_graph.hip, round 6): off[n_bodies + 2] (off[n_bodies + 1] = entries found), adj[n]
struct IslAdj { uint32_t cap; uint32_t *e_key2, *e_other, *e_body, *k_a, *v_a, *k_b, *v_b; /* [cap] each */ };
void launch_isl_adjacency(const PG&, const uint4* ct_meta, const uint32_t* bmeta, const uint32_t* slot_rank, const uint32_t* handles, uint32_t n_handles, uint32_t n_bodies, uint32_t seq_bits, uint32_t rank_bits,
                          uint32_t pad_key2, const IslAdj&, uint32_t* hist, uint32_t* block_sums, uint32_t* off, uint32_t* adj, hipStream_t);
// an op batch from a LIST instead of from the rows' status changes (SleepIslands / WakeIslands: pops and pushes in the island manager's order):
// fills the same op arrays as k_pg_classify for ops (cids[k], kinds[k] = 1 push | 2 pop); the rest of the pipeline is the status loop's
template <class T> void launch_pg_ops_from_list(const PG&, const CT<T>&, const uint32_t* cids, const uint32_t* kinds, uint32_t n, uint32_t n_bodies, hipStream_t);
// SLEEPING bit of contact rows, Sleeping flag of bodies (+ SleepTimer = 0 for woken bodies)
template <class T> void launch_rows_set_sleeping(const CT<T>&, const uint32_t* cids, uint32_t n, uint32_t sleeping, hipStream_t);
template <class T> void launch_bodies_set_sleeping(const DW<T>&, const uint32_t* bodies, uint32_t n, uint32_t sleeping, float* timer, hipStream_t);
uint32_t pg_scan_tiles(uint32_t n_entries);
void launch_pg_entry_scan(const PG&, const uint32_t* keys, const uint32_t* vals, uint32_t n_entries, uint32_t n_bodies, hipStream_t);
void launch_pg_color(const PG&, uint32_t n_ops, hipStream_t);
void launch_pg_apply_masks(const PG&, const uint32_t* keys, const uint32_t* vals, uint32_t n_entries, uint32_t n_bodies, hipStream_t);
void launch_pg_replay(const PG&, uint32_t n_ops, hipStream_t);   // stable partition of the ops by colour + the exact push / swap_remove replay of every colour
template <class T> void launch_pg_remove(const PG&, const CT<T>&, const BP<T>&, uint32_t n_ops, hipStream_t);
void launch_pg_merge_free(const PG&, uint32_t head, uint32_t n_free, uint32_t n_rem, hipStream_t);
// handles <- the colours' lists, concatenated.  sort_tab != NULL: colours 0..22 in KEY-BODY order instead of list order (the solver's second order, k_graph.hip;
// sort_tab: [23][pg_sort_stride(n_bodies)] words, all PG_NONE on entry and on exit; sort_cnt: [23][stride / 2048] words of scratch; 3 launches instead of 1)
uint32_t pg_sort_stride(uint32_t n_bodies);
// (lens_at: PGC_LEN, or PGC_LLEN with PG::lists = the rank's local lists)
void launch_pg_build_handles(const PG&, uint32_t* handles, uint32_t* color_offsets, uint32_t total, const uint4* ct_meta, uint32_t* sort_tab, uint32_t* sort_cnt, uint32_t n_bodies, hipStream_t, uint32_t lens_at = PGC_LEN);
template <class T> void launch_pg_rebuild_pair_set(const CT<T>&, const BP<T>&, uint32_t n_rows, hipStream_t);
// overflow colour on the device: incidence CSR of the body-centric warm start + per-body ranks of the dataflow passes
struct OverflowFlow { const uint32_t* rank; /* [2 n23] rank of the manifold among its body's overflow entries | PG_NONE */ uint32_t* ticket; /* [n_bodies] */ uint32_t* tiles; /* PGC_OVF_TILE words */ uint32_t* error; uint32_t poll_sleep = 8; /* x 64 clocks between polling rounds */ };
template <class T> void launch_ovf_entries(const DW<T>&, uint32_t o0, uint32_t n23, uint32_t* keys, uint32_t* vals, hipStream_t);
template <class T> void launch_ovf_csr(const DW<T>&, uint32_t o0, uint32_t n23, const uint32_t* keys, const uint32_t* vals, uint32_t* inc_off, uint32_t* inc_ent, uint32_t* rank, hipStream_t);
void launch_overflow_reset(uint32_t* ticket, uint32_t n_ticket, uint32_t* tiles, uint32_t n_tiles, hipStream_t);
template <class T> void launch_overflow_flow(const DW<T>&, const StepParams<T>&, int pass, const OverflowFlow&, uint32_t epoch, uint32_t grid_blocks, hipStream_t);
// radix sort on the low `bits` bits of the keys (stable LSD, 8 bits per pass); the result is in (*keys_out, *vals_out)
void launch_radix_sort_bits(uint32_t* keys_a, uint32_t* vals_a, uint32_t* keys_b, uint32_t* vals_b, uint32_t n, uint32_t bits, uint32_t* hist, uint32_t* block_sums,
                            uint32_t** keys_out, uint32_t** vals_out, hipStream_t);
// k_narrow.hip: device staging copies of avn_shape_pairs / avn_query_manifolds_out (host layout, nullptr = not wanted)
template <class T> struct QueryStage {
    const uint8_t *shape1, *shape2;
    const T *half_extents1, *position1, *rotation1, *half_extents2, *position2, *rotation2, *prediction;
    uint8_t* point_count;
    T *normal, *anchor1, *anchor2, *point, *penetration;
    uint32_t *feature_id1, *feature_id2;
};
template <class T> void launch_contact_manifolds_query(const QueryStage<T>&, uint32_t n, hipStream_t);
// k_transfer.hip: host-layout (interleaved xyz) <-> device Vec4 records
template <class T> struct BodyStage {  // device staging copies of the avn_bodies arrays (nullptr = absent)
    const T *position, *rotation, *linear_velocity, *angular_velocity, *inv_mass, *inv_inertia_local, *center_of_mass, *linear_damping,
        *angular_damping, *gravity_scale, *accel_linear, *accel_angular, *max_linear_speed, *max_angular_speed;
    const uint8_t *rb_type, *locked_axes, *body_flags;
    const int8_t* dominance;
};
template <class T> void launch_pack_bodies(const DW<T>&, const BodyStage<T>&, hipStream_t);
// islands and sleeping (k_islands.hip)
template <class T> struct SleepParams {
    T length_unit_squared, lin_threshold_squared, ang_threshold_squared; float delta_secs, time_to_sleep;
    const float *body_lin, *body_ang; const uint8_t* body_disabled;   // optional per-body SleepThreshold / SleepingDisabled (device copies), nullptr = world level
};
template <class T> void launch_islands_validate(const DW<T>&, const uint32_t* label, uint32_t* invalid, hipStream_t, uint32_t solver_nodes);   // last step's labels against this step's edges
template <class T> void launch_islands(const DW<T>&, uint32_t* parent, uint32_t* label, uint32_t* ctr /* [0] islands, [1] island bodies */, hipStream_t,
                                       uint32_t solver_nodes = 0u /* 1: only bodies with a SolverBody connect (island-block builder) */);
// components over the contact-table rows that hold constraint handles (+ joints): labels as launch_islands writes them
template <class T> void launch_islands_rows(const DW<T>&, const int2* row_bodies, const uint32_t* row_color, const uint32_t* handles, uint32_t n_handles, uint32_t* parent, uint32_t* label, uint32_t* ctr, hipStream_t);
template <class T> void launch_sleep_update(const DW<T>&, const SleepParams<T>&, const uint32_t* label, float* timer, uint32_t* awake, uint8_t* rests, uint8_t* wakes,
                                            uint32_t* ctr /* [2] resting islands, [3] resting bodies, [4] waking islands, [5] their sleeping bodies, [6] sleeping bodies */, hipStream_t);
// update_sleeping_states, body half, for the closed loop with persistent islands: timer[b] updated, flags[b] = 1 took part | 2 SleepingDisabled | 4 owns a SolverBody
template <class T> void launch_sleep_timers_flags(const DW<T>&, const SleepParams<T>&, float* timer, uint8_t* flags, hipStream_t);
void launch_sleep_reset(float* timer, const uint32_t* bodies, uint32_t n, uint32_t n_bodies, hipStream_t);
// level-2 sharding: (SolverBody linear | angular velocity records) of a list of bodies <-> a contiguous buffer of 2 records per body
template <class T> void launch_halo_pack(const DW<T>&, const int32_t* bodies, uint32_t n, Vec4<T>* out, hipStream_t);
template <class T> void launch_halo_unpack(const DW<T>&, const int32_t* bodies, uint32_t n, const Vec4<T>* in, hipStream_t);
template <class T> void launch_halo_pack_joint(const DW<T>&, const int32_t* bodies, uint32_t n, Vec4<T>* out /* 4 records per body */, hipStream_t);
template <class T> void launch_halo_unpack_joint(const DW<T>&, const int32_t* bodies, uint32_t n, const Vec4<T>* in, hipStream_t);
template <class T> struct ManifoldStage {
    const int32_t *body1, *body2;
    const T *normal, *friction, *restitution, *tangent_velocity, *anchor1, *anchor2, *penetration, *normal_speed, *warm_n, *warm_t;
    const uint8_t *point_count, *manifold_flags;
};
template <class T> void launch_pack_manifolds(const DW<T>&, const ManifoldStage<T>&, hipStream_t);
template <class T> struct JointStage {  // device staging copies of the avn_joints arrays (nullptr = absent)
    const uint8_t *joint_type, *limit_flags;
    const int32_t *body1, *body2;
    const T *local_anchor1, *local_anchor2, *local_basis1, *local_basis2, *axis, *limit_min, *limit_max, *limit2_min, *limit2_max,
        *compliance /* [3J] */, *damping_linear, *damping_angular;
};
template <class T> void launch_pack_joints(const DW<T>&, const JointStage<T>&, hipStream_t);
template <class T> struct ColliderStage {
    const uint32_t *entity, *memberships, *filters;
    const int32_t* body;
    const uint8_t *shape, *cflags;
    const T *half_extents, *collision_margin, *speculative_margin;
};
template <class T> void launch_pack_colliders(const BP<T>&, const ColliderStage<T>&, hipStream_t);
// unpack: device records -> planar staging arrays laid out like the *_out structs (nullptr = skip)
template <class T> void launch_unpack_bodies(const DW<T>&, T* position, T* rotation, T* linear_velocity, T* angular_velocity, hipStream_t);
template <class T> void launch_pack_local_accelerations(Vec4<T>* lin, Vec4<T>* ang, const T* s_lin, const T* s_ang, uint32_t n, hipStream_t);
template <class T> struct SolverBodiesStage {
    T *linear_velocity, *angular_velocity, *delta_position, *delta_rotation, *inv_mass, *inv_inertia_world, *linear_increment, *angular_increment,
        *linear_damping_rhs, *angular_damping_rhs;
    uint32_t* flags;
    int16_t* dominance;
};
template <class T> void launch_unpack_solver_bodies(const DW<T>&, const SolverBodiesStage<T>&, hipStream_t);
template <class T> void launch_unpack_impulses(const DW<T>&, T* warm_n, T* warm_t, T* normal_impulse, hipStream_t);
template <class T> struct ConstraintsStage {
    uint8_t *point_count, *softness_non_dynamic;
    int16_t* relative_dominance;
    T *tangent1, *anchor1, *initial_separation, *normal_impulse, *total_impulse, *normal_effective_mass, *tangent_impulse, *tangent_k;
};
template <class T> void launch_unpack_constraints(const DW<T>&, const ConstraintsStage<T>&, hipStream_t);
template <class T> void launch_unpack_joints(const DW<T>&, T* r1, T* r2, T* cd, T* lag, T* force, T* rot_lag, T* torque, hipStream_t);
template <class T> void launch_unpack_aabbs(const BP<T>&, T* mn, T* mx, uint32_t* interval_entities, hipStream_t);

}  // namespace avn
