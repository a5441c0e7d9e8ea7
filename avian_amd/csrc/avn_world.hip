// avn_world.hip — host orchestration of the MI355X physics step (see avn_world.hpp).
#include "avn_world.hpp"
#include "avn_islands.hpp"

#include <algorithm>
#include <array>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <numeric>
#include <chrono>
#include <queue>
#include <unordered_set>

namespace avn {

#define HIPCHK(call)                                                                          \
    do {                                                                                      \
        hipError_t e_ = (call);                                                               \
        if (e_ != hipSuccess) {                                                               \
            error = std::string(#call) + ": " + hipGetErrorName(e_) + " (" + hipGetErrorString(e_) + ")"; \
            return e_ == hipErrorOutOfMemory ? AVN_ERR_OOM : AVN_ERR_HIP;                      \
        }                                                                                     \
    } while (0)

DevBuf::~DevBuf() { if (p) (void)hipFree(p); }
bool DevBuf::ensure(size_t bytes, hipError_t& err, bool keep, hipStream_t s) {
    err = hipSuccess;
    if (bytes <= cap) return false;
    size_t ncap = std::max(bytes, cap + cap / 2);
    ncap = (ncap + 255) & ~(size_t)255;
    void* np = nullptr;
    err = hipMalloc(&np, ncap);
    if (err != hipSuccess) return false;
    if (p) {
        if (keep) { err = hipMemcpyAsync(np, p, cap, hipMemcpyDeviceToDevice, s); if (err == hipSuccess) err = hipStreamSynchronize(s); }
        (void)hipFree(p);
    }
    p = np;
    cap = ncap;
    return true;
}

// ---------------------------------------------------------------------------------------------------------
void JointSchedule::build(const std::vector<uint32_t>& joints, const std::vector<int32_t>& key1, const std::vector<int32_t>& key2, uint32_t n_keys, bool levels_only) {
    comp_level_begin.clear(); level_offsets.clear(); order.clear(); glevel_offsets.clear(); gorder.clear();
    n_components = 0;
    size_t J = joints.size();
    if (J == 0) { comp_level_begin.push_back(0); level_offsets.push_back(0); glevel_offsets.push_back(0); return; }
    // level(k) = 1 + max level of the earlier items that share a key with k
    std::vector<uint32_t> level(J);
    std::vector<uint32_t> last(n_keys, 0);
    uint32_t max_level = 0;
    for (size_t k = 0; k < J; ++k) {
        int32_t a = key1[k], b = key2[k];
        uint32_t lv = 1 + std::max(a >= 0 ? last[a] : 0u, b >= 0 ? last[b] : 0u);
        if (a >= 0) last[a] = lv;
        if (b >= 0) last[b] = lv;
        level[k] = lv;
        max_level = std::max(max_level, lv);
    }
    // by level only: counting sort (levels are 1..max_level), stable in the original order
    glevel_offsets.assign((size_t)max_level + 1, 0u);
    for (size_t k = 0; k < J; ++k) ++glevel_offsets[level[k]];
    { uint32_t run = 0; for (uint32_t l = 1; l <= max_level; ++l) { uint32_t c = glevel_offsets[l]; glevel_offsets[l] = run; run += c; } glevel_offsets[0] = 0; }
    gorder.resize(J);
    { std::vector<uint32_t> cur(glevel_offsets.begin(), glevel_offsets.end()); for (size_t k = 0; k < J; ++k) gorder[cur[level[k]]++] = joints[k]; }
    glevel_offsets.erase(glevel_offsets.begin());   // offsets of levels 1..max_level, then the end
    glevel_offsets.push_back((uint32_t)J);
    if (levels_only) { comp_level_begin.push_back(0); level_offsets.push_back(0); return; }
    // connected components (union-find over the keys), numbered in order of first appearance
    std::vector<int32_t> parent(n_keys);
    std::iota(parent.begin(), parent.end(), 0);
    auto find = [&](int32_t x) { while (parent[x] != x) { parent[x] = parent[parent[x]]; x = parent[x]; } return x; };
    for (size_t k = 0; k < J; ++k) {
        int32_t a = key1[k], b = key2[k];
        if (a >= 0 && b >= 0) { int32_t ra = find(a), rb = find(b); if (ra != rb) parent[ra] = rb; }
    }
    std::vector<uint32_t> comp(J);
    std::vector<uint32_t> comp_of_root(n_keys, 0xFFFFFFFFu);
    for (size_t k = 0; k < J; ++k) {
        int32_t a = key1[k], b = key2[k];
        int32_t root = a >= 0 ? find(a) : (b >= 0 ? find(b) : -1);
        if (root < 0) comp[k] = n_components++;  // touches no scheduled key: its own component
        else {
            if (comp_of_root[root] == 0xFFFFFFFFu) comp_of_root[root] = n_components++;
            comp[k] = comp_of_root[root];
        }
    }
    // sort item slots by (component, level, original order)
    std::vector<uint32_t> idx(J);
    std::iota(idx.begin(), idx.end(), 0u);
    std::stable_sort(idx.begin(), idx.end(), [&](uint32_t x, uint32_t y) { return comp[x] != comp[y] ? comp[x] < comp[y] : level[x] < level[y]; });
    order.resize(J);
    comp_level_begin.assign(1, 0u);
    uint32_t cur_comp = comp[idx[0]], cur_level = 0;
    for (size_t k = 0; k < J; ++k) {
        uint32_t s = idx[k];
        if (comp[s] != cur_comp) { comp_level_begin.push_back((uint32_t)level_offsets.size()); cur_comp = comp[s]; cur_level = 0; }
        if (level[s] != cur_level) { level_offsets.push_back((uint32_t)k); cur_level = level[s]; }
        order[k] = joints[s];
    }
    comp_level_begin.push_back((uint32_t)level_offsets.size());
    level_offsets.push_back((uint32_t)J);
}

// ---------------------------------------------------------------------------------------------------------
template <class T> struct World : WorldBase {
    using V = Vec4<T>;
    using Key = typename BP<T>::Key;
    avn_config cfg;
    StepParams<T> params;
    hipStream_t stream = nullptr;
    // The broad phase of a step only READS the body components, which the solver rewrites at the very end (write-back): with
    // host-uploaded manifolds the two are independent until then, so avn_step runs the broad phase on a second stream next
    // to the solver's latency-bound colour launches (which leave most of the chip idle) and joins before the write-back.
    hipStream_t stream_bp = nullptr, bs = nullptr;  // bs: the stream the broad-phase functions launch on (stream | stream_bp)
    hipEvent_t ev_bp_done = nullptr, ev_bp_t0 = nullptr, ev_bp_t1 = nullptr;
    bool overlap_bp = true, bp_timed = false;
    hipEvent_t ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    static constexpr uint32_t BIAS_EV = 16;      // substeps whose biased-solve pass is bracketed by events (direct launches only)
    hipEvent_t ev_bias[2 * BIAS_EV] = {nullptr};
    uint32_t bias_timed = 0, bias_launches = 0;  // substeps timed in the last step; launches of one pass
    uint32_t substep_index = 0;
    bool ev_valid = false;
    // SolverDiagnostics / CollisionDiagnostics stamps (avn_diagnostics_get): events on the world's stream
    enum { DG_BP0 = 0, DG_BP1, DG_NP1, DG_PREP1, DG_INC1, DG_SUB1, DG_REST1, DG_FIN1, DG_STORE1, DG_STEP_COUNT, DG_SUBSTEPS = 16, DG_PER = 5 };
    hipEvent_t ev_dg[DG_STEP_COUNT] = {nullptr};
    hipEvent_t ev_dgs[DG_SUBSTEPS * DG_PER] = {nullptr};   // per substep: start, after warm start, after solve, after positions, end
    bool dg_stamped[DG_STEP_COUNT] = {false};
    uint32_t dg_substeps = 0; bool dg_np = false;
    const unsigned EV_FLAGS = avn_env("AVN_EVENT_SYSTEM_FENCE") ? 0u : (unsigned)hipEventDisableSystemFence;   // (A/B: the default, fencing events)
    void stamp(int id) { if (ev_dg[id]) { (void)hipEventRecord(ev_dg[id], stream); dg_stamped[id] = true; } }
    DW<T> dw;
    BP<T> bp;
    // capacities
    uint32_t cap_bodies = 0, cap_manifolds = 0, cap_joints = 0, cap_colliders = 0;
    // body buffers (Vec4 each)
    DevBuf b_pos, b_rot, b_lvel, b_avel, b_com, b_iloc_a, b_iloc_b, b_acc_l, b_acc_a, b_bmeta;
    DevBuf b_lacc_l, b_lacc_a;   // AccumulatedLocalAcceleration (avn_local_accelerations_upload); dw.lacc_l == nullptr while no body has one
    DevBuf b_sb_vel, b_sb_delta, b_si, b_vid_l, b_vid_a, b_pre_dp, b_pre_dq, b_sb_flags;
    DevBuf b_m_bodies, b_m_n, b_m_tv, b_m_meta, b_mp_a1, b_mp_a2, b_mp_w, b_c_h1, b_c_pa, b_c_pb, b_c_pc, b_c_pd, b_c_reldom, b_misc;
    DevBuf b_j_bodies, b_j_a1, b_j_a2, b_j_par, b_j_b1, b_j_b2, b_j_ax, b_j_l2, b_j_r1, b_j_r2, b_j_cd, b_j_lag, b_j_s0, b_j_s1, b_j_s2, b_j_s3, b_j_rl0, b_j_rl1, b_j_force,
        b_j_torque;
    DevBuf b_col_info, b_col_he, b_col_spec, b_col_layers, b_aabb_min, b_aabb_max, b_iv, b_s_minx, b_s_maxx, b_s_yz, b_s_bb, b_s_end, b_s_info, b_s_flags;
    DevBuf b_keys_a, b_keys_b, b_vals_a, b_vals_b, b_hist, b_block_sums, b_counts, b_offsets, b_pairs, b_pair_set, b_disabled_set, b_pair_keys, b_long_items, b_long_counts, b_long_off, b_sweep_hits;
    DevBuf b_inc_off, b_inc_ent, b_inc_slot;
    bool overflow_csr_nonzero = true;  // the device CSR offsets may be non-zero (first build uploads them)
    uint32_t overflow_csr_bodies = 0;
    // ---- narrow phase: the ContactGraph side on device (CT) + host mirrors of what the host structures of the reference hold ----
    CT<T> ct;
    DevBuf b_ct_meta, b_ct_dcount, b_ct_rows, b_col_mat, b_active, b_changes, b_handles, b_np_row, b_np_axis, b_np_ctr;
    std::unordered_map<uint32_t, uint32_t> entity_slot;   // collider Entity::index() -> slot (last colliders_upload)
    std::vector<int32_t> h_col_body;                       // body of each collider slot
    std::vector<uint8_t> h_ct_used;
    std::vector<uint32_t> h_ct_c1, h_ct_c2;                // collider entities of each row
    std::vector<int32_t> h_ct_b1, h_ct_b2;                 // ... and the bodies they sit on
    std::vector<avn_contact_change> h_changes;
    uint32_t n_active = 0;
    bool use_handles = false, materials_restitution = false, contact_keys_live = false;
    std::unordered_set<uint64_t> h_live_keys;              // pair keys of the live rows (pair-set rebuilds after removals)
    // ---- standalone closed loop (avn_pipeline_enable): the host structures an Avian integration would own ----
    struct PipePair { uint32_t c1 = 0, c2 = 0; int32_t b1 = -1, b2 = -1; uint32_t n_handles = 0; uint32_t active_pos = 0; uint32_t color_pos = 0; int8_t color = -1; bool used = false; };
    bool pipe_on = false, pipe_handles_dirty = true, pipe_active_dirty = false;
    std::priority_queue<uint32_t, std::vector<uint32_t>, std::greater<uint32_t>> pipe_free_ids;  // IdPool: lowest free id first
    uint32_t pipe_next_id = 0;
    std::vector<PipePair> pipe_pairs;          // indexed by ContactId
    std::vector<uint32_t> pipe_active;         // ContactGraph::active_pairs (iteration order is irrelevant to results)
    // ConstraintGraph (constraint_graph.rs:163-296) with dense per-contact bookkeeping: one manifold per pair, so the handle
    // is the ContactId and (colour, index in the colour's list) live in PipePair
    struct PipeColor { std::vector<uint64_t> body_bits; std::vector<uint32_t> handles; };
    PipeColor pipe_colors[AVN_GRAPH_COLOR_COUNT];
    std::vector<uint32_t> pipe_handles;        // colour-major contact ids (GraphColor::manifold_handles)
    uint32_t pipe_offsets[AVN_GRAPH_COLOR_COUNT + 1];
    // pinned host staging (grow-only): island block arrays on their way up, narrow-phase change list on its way down
    struct Pinned {
        void* p = nullptr; size_t cap = 0;
        ~Pinned() { if (p) (void)hipHostFree(p); }
        hipError_t ensure(size_t bytes) {
            if (bytes <= cap) return hipSuccess;
            if (p) (void)hipHostFree(p);
            p = nullptr; cap = 0;
            size_t c = (bytes + bytes / 2 + 4095) & ~(size_t)4095;
            hipError_t e = hipHostMalloc(&p, c, hipHostMallocDefault);
            if (e == hipSuccess) cap = c;
            return e;
        }
    };
    avn_pipeline_stats pipe_stats;
    // ---- the same closed loop with the bookkeeping ON THE DEVICE (k_graph.hip): the host keeps exact mirrors of a few counters ----
    bool pipe_dev = false;               // avn_pipeline_enable(1): device bookkeeping (AVN_PIPELINE_HOST=1 or enable(2): the host structures above)
    PG pg;
    DevBuf b_pg_bodies, b_pg_color, b_pg_lpos, b_pg_lists, b_pg_bcol, b_pg_free_a, b_pg_free_b, b_pg_ctr, b_pg_ent2slot;
    DevBuf b_pg_chg, b_pg_has, b_pg_off, b_pg_op_cid, b_pg_op_info, b_pg_op_bodies, b_pg_ekey_a, b_pg_eval_a, b_pg_ekey_b, b_pg_eval_b, b_pg_epos, b_pg_popbefore, b_pg_prevpush,
        b_pg_est, b_pg_tile_agg, b_pg_ckey_a, b_pg_cval_a, b_pg_ckey_b, b_pg_cval_b, b_pg_rem_flag, b_pg_rem_off, b_pg_rem_ids, b_pg_hist, b_pg_sums;
    DevBuf b_pg_op_chg, b_pg_new_ids, b_pg_seq;
    DevBuf b_ovf_keys_a, b_ovf_vals_a, b_ovf_keys_b, b_ovf_vals_b, b_ovf_rank, b_ovf_ticket;
    uint32_t pg_rows = 0, pg_ops_cap = 0, pg_ovf_cap = 0;
    uint32_t pgm_head = 0, pgm_n_free = 0, pgm_next_id = 0, pgm_live = 0, pgm_tomb = 0;   // exact host mirrors of the device counters
    uint32_t pgm_len[AVN_GRAPH_COLOR_COUNT] = {0};
    uint32_t ovf_epoch = 0, ovf_epoch_after_substeps = 0;

    uint64_t pg_dump_step = 0;
    Pinned pin_ctr;
    SweepScratch sweep_scratch{nullptr, nullptr, nullptr, nullptr, 0, nullptr};
    DevBuf stage;  // staging arena for uploads/downloads
    size_t stage_off = 0;
    // host state
    uint32_t color_offsets[AVN_GRAPH_COLOR_COUNT + 1];
    uint32_t grid_blocks[AVN_GRAPH_COLOR_COUNT];      // launch grids (captured into the graph with slack)
    std::vector<int32_t> h_j_body1, h_j_body2;
    std::vector<uint8_t> h_j_damped, h_j_collision_disabled, h_j_type;
    std::vector<uint8_t> h_body_has_sb;
    std::vector<int32_t> h_m_body1, h_m_body2;  // ContactPair bodies of the uploaded manifolds (incidence CSR source)
    bool incidence_dirty = true;
    std::vector<uint32_t> inc_off_h, inc_cursor_h, inc_ent_h;  // host scratch of rebuild_incidence (kept: the closed loop rebuilds every step)
    bool joint_schedule_dirty = true;
    JointSchedule sched_solve, sched_damp, sched_overflow;
    size_t overflow_level_threshold = 4096;  // overflow manifolds above which the colour runs one launch per level (AVN_OVERFLOW_LEVEL_THRESHOLD overrides: tests)
    // island blocks (k_island_substeps): the whole substep loop in one launch when the contact graph is many small islands
    bool island_enabled = true, island_mode = false;
    size_t island_max_manifolds = 65536;  // above this the colour launches are throughput- not latency-bound (1 wave per SIMD = 65k manifolds): keep the device-wide path
    uint32_t island_pack_bodies = 256;    // islands are packed into one block up to this many bodies (a single island may reach ISLAND_MAX_BODIES)
    std::vector<uint32_t> isl_parent, isl_island_of, isl_count, isl_block_of_island, isl_slot, isl_body_off, isl_bodies, isl_col_off, isl_cursor, isl_ent, isl_mcount, isl_root_island, isl_roots;
    bool isl_labels_step_valid = false;   // b_isl_label / isl_roots hold the labels of an earlier closed-loop step (reused while they still group the manifolds)
    DevBuf b_isl_bodies;   // [body_off | bodies | col_off | ent], 256-byte aligned parts
    IslandBlocks islands{nullptr, nullptr, nullptr, nullptr, 0, 0, 0, 0};
    bool island_cache_records = true;    // AVN_ISLAND_CACHE_RECORDS=0: bodies only in LDS (A/B runs, tests)
    bool islands_dirty = false;          // the manifold set changed since the blocks were built
    Pinned pin_islands, pin_changes;
    static constexpr uint32_t CHANGES_PREFIX = 2048;   // status changes fetched together with their count (one round trip)
    bool any_damped = false;
    bool any_restitution = false;  // some manifold has restitution != 0 (else apply_restitution early-outs for all, contact/mod.rs:366-369)
    std::vector<uint32_t> slot_entity;  // collider entity per slot (last upload)
    uint32_t n_pair_keys = 0;           // keys currently in the device pair set
    std::vector<avn_pair> h_pairs;
    bool have_colliders = false, have_bodies = false;
    avn_timers last_timers;
    uint32_t launches = 0;
    // graph
    hipGraph_t graph = nullptr;
    hipGraphExec_t graph_exec = nullptr;
    bool graph_valid = false;

    World() {
        std::memset(&dw, 0, sizeof dw);
        std::memset(&bp, 0, sizeof bp);
        std::memset(&ct, 0, sizeof ct);
        std::memset(&pg, 0, sizeof pg);
        std::memset(&pipe_stats, 0, sizeof pipe_stats);
        std::memset(pipe_offsets, 0, sizeof pipe_offsets);
        std::memset(&last_timers, 0, sizeof last_timers);
        std::memset(color_offsets, 0, sizeof color_offsets);
        std::memset(grid_blocks, 0, sizeof grid_blocks);
    }
    ~World() override {
        if (stream) (void)hipStreamSynchronize(stream);
        if (stream_bp) (void)hipStreamSynchronize(stream_bp);
        drop_graph();
        for (hipEvent_t e : {ev_bp_done, ev_bp_t0, ev_bp_t1, ev_spin}) if (e) (void)hipEventDestroy(e);
        if (stream_bp) (void)hipStreamDestroy(stream_bp);
        for (auto& e : ev) if (e) (void)hipEventDestroy(e);
        for (auto& e : ev_bias) if (e) (void)hipEventDestroy(e);
        for (auto& e : ev_dg) if (e) (void)hipEventDestroy(e);
        for (auto& e : ev_dgs) if (e) (void)hipEventDestroy(e);
        if (ev_counters) (void)hipEventDestroy(ev_counters);
        if (h_counters) (void)hipHostFree(h_counters);
        if (h_pg_error) (void)hipHostFree(h_pg_error);
        for (hipEvent_t e : {ev_np_fork, ev_np_old, ev_slot_clear}) if (e) (void)hipEventDestroy(e);
        for (hipEvent_t e : {ev_side_fork, ev_side_done}) if (e) (void)hipEventDestroy(e);
        if (stream_side) { (void)hipStreamSynchronize(stream_side); (void)hipStreamDestroy(stream_side); }
        if (stream) (void)hipStreamDestroy(stream);
    }
    void bind() override { (void)hipSetDevice(cfg.device); }
    void drop_graph() {
        if (graph_exec) { (void)hipGraphExecDestroy(graph_exec); graph_exec = nullptr; }
        if (graph) { (void)hipGraphDestroy(graph); graph = nullptr; }
        graph_valid = false;
    }

    avn_status init(const avn_config* c) {
        int ndev = 0;
        hipError_t e = hipGetDeviceCount(&ndev);
        if (e != hipSuccess || ndev == 0) { error = "no HIP device visible: the MI355X path has no CPU fallback"; return AVN_ERR_NO_DEVICE; }
        if (c->device < 0 || c->device >= ndev) { error = "config.device out of range"; return AVN_ERR_BAD_ARG; }
        HIPCHK(hipSetDevice(c->device));
        cfg.device = c->device;
        if (const char* e = avn_env("AVN_OVERFLOW_LEVEL_THRESHOLD")) overflow_level_threshold = (size_t)strtoull(e, nullptr, 10);
        if (const char* e = avn_env("AVN_ISLAND_BLOCKS")) island_enabled = atoi(e) != 0;                                  // 0: always the device-wide colour launches
        if (const char* e = avn_env("AVN_ISLAND_CACHE_RECORDS")) island_cache_records = atoi(e) != 0;
        if (const char* e = avn_env("AVN_ISLAND_MAX_MANIFOLDS")) island_max_manifolds = (size_t)strtoull(e, nullptr, 10);
        if (const char* e = avn_env("AVN_ISLAND_MAX_BODIES_TOTAL")) island_max_bodies_total = (size_t)strtoull(e, nullptr, 10);
        if (const char* e = avn_env("AVN_ISLAND_PACK_BODIES")) island_pack_bodies = std::min<uint32_t>(ISLAND_MAX_BODIES, std::max<uint32_t>(1u, (uint32_t)strtoul(e, nullptr, 10)));
        // (CU masks -- 64 CUs for the broad phase, 192 for the solver -- were tried for the overlap below and lost: a colour launch
        //  on 192 CUs is 12 % slower than on 256, more than the contention it avoids; tools/cumask_probe.hip)
        HIPCHK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
        HIPCHK(hipStreamCreateWithFlags(&stream_bp, hipStreamNonBlocking));
        bs = stream;
        // Events that order streams of THIS device or only measure time carry hipEventDisableSystemFence: a default event makes the device write
        // back and invalidate its caches when it is recorded (system-scope release), and the kernels behind it start cold -- a dozen records
        // per step cost cfg2 ~0.1 ms of 1.6.  Kernel boundaries already release at agent scope, which is all a second stream needs.  (Events
        // behind which the HOST reads pinned memory -- ev_counters, ev_spin -- keep the default.)
        HIPCHK(hipEventCreateWithFlags(&ev_bp_done, hipEventDisableTiming | EV_FLAGS));
        HIPCHK(hipEventCreateWithFlags(&ev_bp_t0, EV_FLAGS)); HIPCHK(hipEventCreateWithFlags(&ev_bp_t1, EV_FLAGS));
        if (avn_env("AVN_NO_BP_OVERLAP")) overlap_bp = false;
#ifdef AVN_MEASURE
        if (avn_env("AVN_NO_OCT")) oct_enabled = false;           // (A/B: every colour on the lane-per-manifold kernel)
        if (avn_env("AVN_NO_HANDLE_SORT")) handle_sort = false;   // (A/B: the solver's arrays in the bookkeeping's list order, as before round 5)
#endif
        for (auto& x : ev) HIPCHK(hipEventCreateWithFlags(&x, EV_FLAGS));
        for (auto& x : ev_dg) HIPCHK(hipEventCreateWithFlags(&x, EV_FLAGS));
        for (auto& x : ev_dgs) HIPCHK(hipEventCreateWithFlags(&x, EV_FLAGS));
        for (auto& x : ev_bias) HIPCHK(hipEventCreateWithFlags(&x, EV_FLAGS));
        hipError_t err;
        b_misc.ensure(4096, err);
        if (err != hipSuccess) { error = "hipMalloc failed"; return AVN_ERR_OOM; }
        HIPCHK(hipMemsetAsync(b_misc.p, 0, 4096, stream));
        // misc layout: [0..25) color offsets, [32] constraint count, [33] n_dropped, [34] unsorted flag, [35] pair total,
        // [36] long chunks used, [37] long chunk overflow
        dw.color_offsets = b_misc.as<uint32_t>();
        dw.constraint_count = b_misc.as<uint32_t>() + 32;
        return config_set(c);
    }

    static T as_secs_adjusted(uint64_t ns) {
        if (sizeof(T) == 8) return (T)((double)(ns / 1000000000ull) + (double)(ns % 1000000000ull) / 1e9);
        return (T)((float)(ns / 1000000000ull) + (float)(ns % 1000000000ull) / 1e9f);  // Duration::as_secs_f32
    }
    static double as_secs_f64(uint64_t ns) { return (double)(ns / 1000000000ull) + (double)(ns % 1000000000ull) / 1e9; }
    // reference solver/softness_parameters/mod.rs:36-41,64-79
    static SoftCoef<T> softness(T damping_ratio, T frequency_hz, T delta_secs) {
        const T TAU = T(6.283185307179586476925286766559);
        T double_damping_ratio = T(2) * damping_ratio;
        T angular_frequency = TAU * frequency_hz;
        T a1 = double_damping_ratio + angular_frequency * delta_secs;
        T a2 = angular_frequency * delta_secs * a1;
        T a3 = T(1) / (T(1) + a2);
        return {angular_frequency / a1, a2 * a3, a3};
    }
    avn_status config_set(const avn_config* c) override {
        if (!c || c->struct_size != sizeof(avn_config)) { error = "config: struct_size mismatch"; return AVN_ERR_BAD_ARG; }
        if (c->substeps == 0 || c->dt_ns == 0) { error = "config: substeps and dt_ns must be > 0"; return AVN_ERR_BAD_ARG; }
        if (c->scalar_bits != sizeof(T) * 8) { error = "config: scalar_bits cannot change after world creation"; return AVN_ERR_BAD_ARG; }
        cfg = *c;
        slp_world_asleep = slp_world_idle = false;
        if (cfg.solver_iterations == 0) cfg.solver_iterations = 1;
        // Duration arithmetic of run_physics_schedule / run_substep_schedule (reference schedule/mod.rs:240-284,
        // solver/schedule.rs:194-200): sub_delta = delta.div_f64(substeps), rounded to the nearest nanosecond.
        uint64_t dt_ns = cfg.dt_ns;
        uint64_t h_ns = (uint64_t)std::llround(as_secs_f64(dt_ns) / (double)cfg.substeps * 1e9);
        params.dt_f64cast = (T)as_secs_f64(dt_ns);
        params.h_f64cast = (T)as_secs_f64(h_ns);
        params.dt_adj = as_secs_adjusted(dt_ns);
        params.h_adj = as_secs_adjusted(h_ns);
        for (int k = 0; k < 3; ++k) params.gravity[k] = (T)cfg.gravity[k];
        params.max_overlap_solve_speed = (T)cfg.max_overlap_solve_speed * (T)cfg.length_unit;
        params.warm_start_coefficient = (T)cfg.warm_start_coefficient;
        params.restitution_threshold = (T)cfg.restitution_threshold * (T)cfg.length_unit;
        params.contact_tolerance = (T)cfg.length_unit * (T)cfg.contact_tolerance;
        T dsm = cfg.default_speculative_margin >= (double)std::numeric_limits<T>::max() ? std::numeric_limits<T>::max() : (T)cfg.default_speculative_margin;
        params.default_speculative_margin = (T)cfg.length_unit * dsm;
        params.substeps_as_scalar = (T)cfg.substeps;
        params.length_unit = (T)cfg.length_unit;
        params.restitution_iterations = cfg.restitution_iterations;
        params.match_contacts = cfg.match_contacts;
#ifdef AVN_MEASURE   // measurement build only (make measure): cut-offs of the narrow phase's kernels (tools/np_phases.sh)
        params.np_debug = avn_env("AVN_NP_DEBUG") ? (uint32_t)atoi(avn_env("AVN_NP_DEBUG")) : 0u;
#else
        params.np_debug = 0u;
#endif
        // update_contact_softness, reference solver/plugin.rs:326-350
        T dt = params.dt_f64cast, h = params.h_f64cast;
        T max_hz = T(1) / (dt * T(2));
        T hz = (T)cfg.contact_frequency_factor * smin(max_hz, T(0.25) / h);
        params.soft_dynamic = softness((T)cfg.contact_damping_ratio, hz, h);
        params.soft_non_dynamic = softness((T)cfg.contact_damping_ratio, T(2) * hz, h);
        graph_valid = false;
        return AVN_OK;
    }

    // ---- staging arena ---------------------------------------------------------------------------------
    avn_status stage_reserve(size_t bytes) {
        hipError_t err;
        HIPCHK(hipStreamSynchronize(stream));  // arena reuse: previous users must be done
        if (stream_bp) HIPCHK(hipStreamSynchronize(stream_bp));
        stage.ensure(bytes + 4096, err);
        if (err != hipSuccess) { error = "staging allocation failed"; return AVN_ERR_OOM; }
        stage_off = 0;
        return AVN_OK;
    }
    template <class U> U* stage_alloc(size_t count) {
        stage_off = (stage_off + 63) & ~(size_t)63;
        U* r = (U*)((char*)stage.p + stage_off);
        stage_off += count * sizeof(U);
        return r;
    }
    template <class U> avn_status stage_in(const void* host, size_t count, const U** out) {
        if (!host || count == 0) { *out = nullptr; return AVN_OK; }
        U* d = stage_alloc<U>(count);
        HIPCHK(hipMemcpyAsync(d, host, count * sizeof(U), hipMemcpyHostToDevice, stream));
        *out = d;
        return AVN_OK;
    }
    template <class U> avn_status stage_out(void* host, const U* dev, size_t count) {
        if (!host || !dev || count == 0) return AVN_OK;
        HIPCHK(hipMemcpyAsync(host, dev, count * sizeof(U), hipMemcpyDeviceToHost, stream));
        return AVN_OK;
    }
    static size_t al(size_t b) { return (b + 63) & ~(size_t)63; }

    template <class U> avn_status grow(DevBuf& b, size_t count, U** field, bool& moved) {
        hipError_t err;
        if (b.ensure(count * sizeof(U), err)) moved = true;
        if (err != hipSuccess) { error = std::string("hipMalloc: ") + hipGetErrorName(err); return AVN_ERR_OOM; }
        *field = b.as<U>();
        return AVN_OK;
    }
#define GROW(buf, count, field) do { avn_status s_ = grow(buf, count, &(field), moved); if (s_ != AVN_OK) return s_; } while (0)

#include "world/bodies.hpp"
#include "world/manifolds.hpp"
#include "world/joints.hpp"
#include "world/broad_phase_data.hpp"
#include "world/host_shapes.hpp"
#include "world/hooks.hpp"
#include "world/contacts.hpp"
#include "world/pipeline_host.hpp"
#include "world/pipeline_device.hpp"
#include "world/broad_phase.hpp"
#include "world/systems.hpp"
#include "world/level2.hpp"
#include "world/islands.hpp"
#include "world/sleeping.hpp"
#include "world/despawn.hpp"
#include "world/dshard.hpp"
#include "world/timers.hpp"
};

template <class T> avn_status World<T>::diagnostics(avn_diagnostics* d) {
    if (!d) return AVN_ERR_BAD_ARG;
    std::memset(d, 0, sizeof *d);
    HIPCHK(hipStreamSynchronize(stream));
    HIPCHK(hipStreamSynchronize(stream_bp));
    if (!ev_valid) return AVN_OK;
    auto ms = [&](hipEvent_t a, hipEvent_t b) { float f = 0; return hipEventElapsedTime(&f, a, b) == hipSuccess ? (double)f : 0.0; };
    // step-level events of timers(): ev[0] step start, ev[1] after the broad phase, ev[2] before the substep loop, ev[3] after it, ev[4] end
    d->broad_phase_ms = bp_timed ? ms(ev_bp_t0, ev_bp_t1) : ms(ev[0], ev[1]);
    const bool np = dg_np && dg_stamped[DG_NP1];
    if (np) d->narrow_phase_ms = ms(ev[1], ev_dg[DG_NP1]);
    if (dg_stamped[DG_PREP1]) d->prepare_constraints_ms = ms(np ? ev_dg[DG_NP1] : ev[1], ev_dg[DG_PREP1]);
    if (dg_stamped[DG_INC1]) d->update_velocity_increments_ms = ms(ev_dg[DG_PREP1], ev_dg[DG_INC1]);
    d->substeps_ms = ms(ev[2], ev[3]);
    if (dg_stamped[DG_REST1]) d->apply_restitution_ms = ms(ev_dg[DG_SUB1], ev_dg[DG_REST1]);
    if (dg_stamped[DG_FIN1]) d->finalize_ms = ms(ev_dg[DG_REST1], ev_dg[DG_FIN1]);
    if (dg_stamped[DG_STORE1]) d->store_impulses_ms = ms(ev_dg[DG_FIN1], ev_dg[DG_STORE1]);
    if (dg_substeps && !cfg.use_graph && !islands_active()) {
        for (uint32_t s = 0; s < dg_substeps; ++s) {
            hipEvent_t* e = ev_dgs + (size_t)s * DG_PER;
            d->warm_start_ms += ms(e[0], e[1]); d->solve_constraints_ms += ms(e[1], e[2]);
            d->integrate_positions_ms += ms(e[2], e[3]); d->relax_velocities_ms += ms(e[3], e[4]);
        }
        d->per_system_valid = 1;
    }
    uint32_t cc = 0;
    HIPCHK(hipMemcpyAsync(&cc, dw.constraint_count, 4, hipMemcpyDeviceToHost, stream));
    HIPCHK(hipStreamSynchronize(stream));
    d->contact_constraint_count = cc;
    d->contact_count = pipe_on ? (pipe_dev ? pgm_live : (uint32_t)pipe_active.size()) : last_timers.pair_count;
    return AVN_OK;
}

template <class T> static WorldBase* make_world(const avn_config* cfg, avn_status* st, std::string* err) {
    World<T>* w = new (std::nothrow) World<T>();
    if (!w) { *st = AVN_ERR_OOM; *err = "out of host memory"; return nullptr; }
    *st = w->init(cfg);
    if (*st != AVN_OK) { *err = w->error; delete w; return nullptr; }
    return w;
}
WorldBase* make_world_f32(const avn_config* cfg, avn_status* st, std::string* err) { return make_world<float>(cfg, st, err); }
WorldBase* make_world_f64(const avn_config* cfg, avn_status* st, std::string* err) { return make_world<double>(cfg, st, err); }

// ---- ConstraintGraphHost (reference solver/constraint_graph.rs:163-296) ------------------------------------
static inline bool bit_get(const std::vector<uint64_t>& s, uint32_t i) { return (i >> 6) < s.size() && ((s[i >> 6] >> (i & 63)) & 1ull); }
static inline void bit_set_and_grow(std::vector<uint64_t>& s, uint32_t i) { if ((i >> 6) >= s.size()) s.resize((i >> 6) + 1, 0ull); s[i >> 6] |= 1ull << (i & 63); }
static inline void bit_unset(std::vector<uint64_t>& s, uint32_t i) { if ((i >> 6) < s.size()) s[i >> 6] &= ~(1ull << (i & 63)); }
int ConstraintGraphHost::push_manifold(uint64_t handle, uint32_t body1, uint32_t body2, bool is_static1, bool is_static2) {
    if (where.count(handle)) return -1;
    int color_index = AVN_COLOR_OVERFLOW_INDEX;
    if (!is_static1 && !is_static2) {
        // dynamic-vs-dynamic constraints only use colours 0..19
        for (int i = 0; i < AVN_DYNAMIC_COLOR_COUNT; ++i) {
            Color& c = colors[i];
            if (bit_get(c.body_bits, body1) || bit_get(c.body_bits, body2)) continue;
            bit_set_and_grow(c.body_bits, body1);
            bit_set_and_grow(c.body_bits, body2);
            color_index = i;
            break;
        }
    } else if (!is_static1 || !is_static2) {
        // static colours are filled from the end (22 down to 1); only the non-static body is marked
        uint32_t body = !is_static1 ? body1 : body2;
        for (int i = AVN_COLOR_OVERFLOW_INDEX - 1; i >= 1; --i) {
            Color& c = colors[i];
            if (bit_get(c.body_bits, body)) continue;
            bit_set_and_grow(c.body_bits, body);
            color_index = i;
            break;
        }
    }
    Color& c = colors[color_index];
    where[handle] = Loc{(uint8_t)color_index, (uint32_t)c.manifold_handles.size()};
    c.manifold_handles.push_back(Handle{handle, body1, body2});
    return color_index;
}
bool ConstraintGraphHost::pop_manifold(uint64_t handle) {
    auto it = where.find(handle);
    if (it == where.end()) return false;
    Loc loc = it->second;
    where.erase(it);
    Color& c = colors[loc.color];
    Handle h = c.manifold_handles[loc.local_index];
    if (loc.color != AVN_COLOR_OVERFLOW_INDEX) { bit_unset(c.body_bits, h.body1); bit_unset(c.body_bits, h.body2); }
    uint32_t moved_index = (uint32_t)c.manifold_handles.size() - 1;
    c.manifold_handles[loc.local_index] = c.manifold_handles[moved_index];
    c.manifold_handles.pop_back();
    if (moved_index != loc.local_index) where[c.manifold_handles[loc.local_index].handle].local_index = loc.local_index;
    return true;
}

}  // namespace avn
