// avn_world.hip — host orchestration of the MI355X physics step (see avn_world.hpp).
#include "avn_world.hpp"

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
    void stamp(int id) { if (ev_dg[id]) { (void)hipEventRecord(ev_dg[id], stream); dg_stamped[id] = true; } }
    DW<T> dw;
    BP<T> bp;
    // capacities
    uint32_t cap_bodies = 0, cap_manifolds = 0, cap_joints = 0, cap_colliders = 0;
    // body buffers (Vec4 each)
    DevBuf b_pos, b_rot, b_lvel, b_avel, b_com, b_iloc_a, b_iloc_b, b_acc_l, b_acc_a, b_bmeta;
    DevBuf b_sb_vel, b_sb_delta, b_si, b_vid_l, b_vid_a, b_pre_dp, b_pre_dq, b_sb_flags;
    DevBuf b_m_bodies, b_m_n, b_m_tv, b_m_meta, b_mp_a1, b_mp_a2, b_mp_w, b_c_h1, b_c_pa, b_c_pb, b_c_pc, b_c_pd, b_c_reldom, b_misc;
    DevBuf b_j_bodies, b_j_a1, b_j_a2, b_j_par, b_j_b1, b_j_b2, b_j_ax, b_j_l2, b_j_r1, b_j_r2, b_j_cd, b_j_lag, b_j_s0, b_j_s1, b_j_s2, b_j_s3, b_j_rl0, b_j_rl1, b_j_force,
        b_j_torque;
    DevBuf b_col_info, b_col_he, b_col_spec, b_col_layers, b_aabb_min, b_aabb_max, b_iv, b_s_minx, b_s_maxx, b_s_yz, b_s_bb, b_s_end, b_s_info, b_s_flags;
    DevBuf b_keys_a, b_keys_b, b_vals_a, b_vals_b, b_hist, b_block_sums, b_counts, b_offsets, b_pairs, b_pair_set, b_disabled_set, b_pair_keys, b_long_items, b_long_counts, b_long_off;
    DevBuf b_inc_off, b_inc_ent, b_inc_slot;
    bool overflow_csr_nonzero = true;  // the device CSR offsets may be non-zero (first build uploads them)
    uint32_t overflow_csr_bodies = 0;
    // ---- narrow phase: the ContactGraph side on device (CT) + host mirrors of what the host structures of the reference hold ----
    CT<T> ct;
    DevBuf b_ct_meta, b_ct_dcount, b_ct_n, b_ct_tv, b_ct_a1, b_ct_a2, b_ct_w, b_ct_fid, b_col_mat, b_active, b_changes, b_handles;
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
    DevBuf b_ovf_keys_a, b_ovf_vals_a, b_ovf_keys_b, b_ovf_vals_b, b_ovf_rank, b_ovf_ticket;
    uint32_t pg_rows = 0, pg_ops_cap = 0, pg_ovf_cap = 0;
    uint32_t pgm_head = 0, pgm_n_free = 0, pgm_next_id = 0, pgm_live = 0, pgm_tomb = 0;   // exact host mirrors of the device counters
    uint32_t pgm_len[AVN_GRAPH_COLOR_COUNT] = {0};
    uint32_t ovf_epoch = 0, ovf_epoch_after_substeps = 0;

    uint64_t pg_dump_step = 0;
    Pinned pin_ctr;
    SweepScratch sweep_scratch{nullptr, nullptr, nullptr, nullptr, 0};
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
    std::vector<uint32_t> isl_parent, isl_island_of, isl_count, isl_block_of_island, isl_slot, isl_body_off, isl_bodies, isl_col_off, isl_cursor, isl_ent, isl_mcount;
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
        for (hipEvent_t e : {ev_bp_done, ev_bp_t0, ev_bp_t1}) if (e) (void)hipEventDestroy(e);
        if (stream_bp) (void)hipStreamDestroy(stream_bp);
        for (auto& e : ev) if (e) (void)hipEventDestroy(e);
        for (auto& e : ev_bias) if (e) (void)hipEventDestroy(e);
        for (auto& e : ev_dg) if (e) (void)hipEventDestroy(e);
        for (auto& e : ev_dgs) if (e) (void)hipEventDestroy(e);
        if (ev_counters) (void)hipEventDestroy(ev_counters);
        if (h_counters) (void)hipHostFree(h_counters);
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
        if (const char* e = getenv("AVN_OVERFLOW_LEVEL_THRESHOLD")) overflow_level_threshold = (size_t)strtoull(e, nullptr, 10);
        if (const char* e = getenv("AVN_ISLAND_BLOCKS")) island_enabled = atoi(e) != 0;                                  // 0: always the device-wide colour launches
        if (const char* e = getenv("AVN_ISLAND_CACHE_RECORDS")) island_cache_records = atoi(e) != 0;
        if (const char* e = getenv("AVN_ISLAND_MAX_MANIFOLDS")) island_max_manifolds = (size_t)strtoull(e, nullptr, 10);
        if (const char* e = getenv("AVN_ISLAND_PACK_BODIES")) island_pack_bodies = std::min<uint32_t>(ISLAND_MAX_BODIES, std::max<uint32_t>(1u, (uint32_t)strtoul(e, nullptr, 10)));
        // (CU masks -- 64 CUs for the broad phase, 192 for the solver -- were tried for the overlap below and lost: a colour launch
        //  on 192 CUs is 12 % slower than on 256, more than the contention it avoids; tools/cumask_probe.hip)
        HIPCHK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
        HIPCHK(hipStreamCreateWithFlags(&stream_bp, hipStreamNonBlocking));
        bs = stream;
        HIPCHK(hipEventCreateWithFlags(&ev_bp_done, hipEventDisableTiming));
        HIPCHK(hipEventCreate(&ev_bp_t0)); HIPCHK(hipEventCreate(&ev_bp_t1));
        if (getenv("AVN_NO_BP_OVERLAP")) overlap_bp = false;
        for (auto& x : ev) HIPCHK(hipEventCreate(&x));
        for (auto& x : ev_dg) HIPCHK(hipEventCreate(&x));
        for (auto& x : ev_dgs) HIPCHK(hipEventCreate(&x));
        for (auto& x : ev_bias) HIPCHK(hipEventCreate(&x));
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
        params.np_debug = getenv("AVN_NP_DEBUG") ? (uint32_t)atoi(getenv("AVN_NP_DEBUG")) : 0u;
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

    // ---- bodies ------------------------------------------------------------------------------------------
    static constexpr uint32_t DUMMY_SLOTS = 2 * AVN_JOINT_TYPE_COUNT;  // joint_damping::<T>: two fresh DUMMY SolverBodies per joint type
    avn_status bodies_upload(const avn_bodies* b) override {
        if (!b || (b->count && (!b->position || !b->rotation || !b->linear_velocity || !b->angular_velocity || !b->inv_mass || !b->inv_inertia_local || !b->rb_type))) {
            error = "bodies_upload: null array"; return AVN_ERR_BAD_ARG;
        }
        uint32_t n = b->count;
        bool moved = false;
        if (n + DUMMY_SLOTS > cap_bodies || !have_bodies) {
            HIPCHK(hipStreamSynchronize(stream));
            size_t c = (size_t)std::max<uint32_t>(n + DUMMY_SLOTS, cap_bodies + cap_bodies / 2);  // + the virtual DUMMY bodies of joint_damping
            GROW(b_pos, c, dw.pos); GROW(b_rot, c, dw.rot); GROW(b_lvel, c, dw.lvel); GROW(b_avel, c, dw.avel); GROW(b_com, c, dw.com);
            GROW(b_iloc_a, c, dw.iloc_a); GROW(b_iloc_b, c, dw.iloc_b); GROW(b_acc_l, c, dw.acc_l); GROW(b_acc_a, c, dw.acc_a); GROW(b_bmeta, c, dw.bmeta);
            GROW(b_sb_vel, 2 * c, dw.sb_lin.p); dw.sb_ang.p = dw.sb_lin.p + 1;   // Pair2 slots (avn_device.h)
            GROW(b_sb_delta, 2 * c, dw.sb_dp.p); dw.sb_dq.p = dw.sb_dp.p + 1;
            GROW(b_si, 2 * c, dw.si_a.p); dw.si_b.p = dw.si_a.p + 1;
            GROW(b_vid_l, c, dw.vid_l); GROW(b_vid_a, c, dw.vid_a);
            GROW(b_pre_dp, c, dw.pre_dp); GROW(b_pre_dq, c, dw.pre_dq); GROW(b_sb_flags, c, dw.sb_flags);
            cap_bodies = (uint32_t)c;
        }
        if (moved || dw.n_bodies != n) graph_valid = false;
        if (have_bodies && n < dw.n_bodies) {
            // fewer bodies than before: everything that may still index a body >= n is dropped (the host re-uploads it; nothing
            // may gather or schedule out of range meanwhile) -- uploaded manifolds, joints, colliders whose body is gone
            bool bad_m = false, bad_j = false, bad_c = false;
            if (!use_handles) for (size_t i = 0; i < h_m_body1.size() && !bad_m; ++i) bad_m = (uint32_t)h_m_body1[i] >= n || (uint32_t)h_m_body2[i] >= n;
            for (size_t i = 0; i < h_j_body1.size() && !bad_j; ++i) bad_j = (h_j_body1[i] >= 0 && (uint32_t)h_j_body1[i] >= n) || (h_j_body2[i] >= 0 && (uint32_t)h_j_body2[i] >= n);
            for (size_t i = 0; i < h_col_body.size() && !bad_c; ++i) bad_c = h_col_body[i] >= 0 && (uint32_t)h_col_body[i] >= n;
            if (bad_m || (use_handles && dw.n_manifolds)) {
                uint32_t zero[AVN_GRAPH_COLOR_COUNT + 1] = {0};
                dw.n_manifolds = 0; h_m_body1.clear(); h_m_body2.clear();
                set_color_offsets(zero);
                HIPCHK(hipMemcpyAsync(dw.color_offsets, zero, sizeof zero, hipMemcpyHostToDevice, stream));
                island_mode = false; islands_dirty = false;
            }
            if (bad_j) { dw.n_joints = 0; h_j_body1.clear(); h_j_body2.clear(); h_j_damped.clear(); h_j_collision_disabled.clear(); h_j_type.clear(); any_damped = false; }
            // a halo plan (level-2 sharding) names local body indices too: one that reaches past the new count is dropped with the rest
            bool bad_h = false;
            for (int32_t b : halo.send) bad_h = bad_h || (uint32_t)b >= n;
            for (int32_t b : halo.recv) bad_h = bad_h || (uint32_t)b >= n;
            if (bad_h) { halo = HaloPlan(); halo_on = false; }
            if (bad_c || pipe_on) { bp.n_colliders = 0; bp.n_intervals = 0; have_colliders = false; slot_entity.clear(); entity_slot.clear(); h_col_body.clear(); pipe_on = false; pipe_dev = false; }
        }
        dw.n_bodies = n;
        size_t total = 0;
        total += al(sizeof(T) * 3 * n) * 7 + al(sizeof(T) * 4 * n) + al(sizeof(T) * 6 * n) + al(sizeof(T) * n) * 6 + al(n) * 4;
        avn_status st = stage_reserve(total + 64 * 32);
        if (st != AVN_OK) return st;
        BodyStage<T> s;
        std::memset(&s, 0, sizeof s);
#define SIN(field, src, cnt, U) do { st = stage_in<U>(src, cnt, &s.field); if (st != AVN_OK) return st; } while (0)
        SIN(position, b->position, 3 * (size_t)n, T); SIN(rotation, b->rotation, 4 * (size_t)n, T);
        SIN(linear_velocity, b->linear_velocity, 3 * (size_t)n, T); SIN(angular_velocity, b->angular_velocity, 3 * (size_t)n, T);
        SIN(inv_mass, b->inv_mass, n, T); SIN(inv_inertia_local, b->inv_inertia_local, 6 * (size_t)n, T);
        SIN(center_of_mass, b->center_of_mass, 3 * (size_t)n, T); SIN(linear_damping, b->linear_damping, n, T);
        SIN(angular_damping, b->angular_damping, n, T); SIN(gravity_scale, b->gravity_scale, n, T);
        SIN(accel_linear, b->accel_linear, 3 * (size_t)n, T); SIN(accel_angular, b->accel_angular, 3 * (size_t)n, T);
        SIN(max_linear_speed, b->max_linear_speed, n, T); SIN(max_angular_speed, b->max_angular_speed, n, T);
        SIN(rb_type, b->rb_type, n, uint8_t); SIN(locked_axes, b->locked_axes, n, uint8_t); SIN(body_flags, b->body_flags, n, uint8_t);
        SIN(dominance, b->dominance, n, int8_t);
        launch_pack_bodies<T>(dw, s, stream);
        HIPCHK(hipGetLastError());
        // host copy of "has SolverBody" for the joint schedules
        h_body_has_sb.resize(n);
        for (uint32_t i = 0; i < n; ++i) {
            uint8_t fl = b->body_flags ? b->body_flags[i] : 0;
            h_body_has_sb[i] = b->rb_type[i] != AVN_RB_STATIC && !(fl & (AVN_BODY_SLEEPING | AVN_BODY_DISABLED));
        }
        joint_schedule_dirty = true;
        incidence_dirty = true;
        have_bodies = true;
        HIPCHK(hipStreamSynchronize(stream));  // host arrays are only borrowed for the call
        return AVN_OK;
    }
    avn_status bodies_download(const avn_bodies_out* o) override {
        if (!o) return AVN_ERR_BAD_ARG;
        size_t n = dw.n_bodies;
        avn_status st = stage_reserve(al(sizeof(T) * 3 * n) * 3 + al(sizeof(T) * 4 * n) + 1024);
        if (st != AVN_OK) return st;
        T* p = o->position ? stage_alloc<T>(3 * n) : nullptr;
        T* r = o->rotation ? stage_alloc<T>(4 * n) : nullptr;
        T* l = o->linear_velocity ? stage_alloc<T>(3 * n) : nullptr;
        T* a = o->angular_velocity ? stage_alloc<T>(3 * n) : nullptr;
        launch_unpack_bodies<T>(dw, p, r, l, a, stream);
        HIPCHK(hipGetLastError());
        if ((st = stage_out<T>(o->position, p, 3 * n)) != AVN_OK) return st;
        if ((st = stage_out<T>(o->rotation, r, 4 * n)) != AVN_OK) return st;
        if ((st = stage_out<T>(o->linear_velocity, l, 3 * n)) != AVN_OK) return st;
        if ((st = stage_out<T>(o->angular_velocity, a, 3 * n)) != AVN_OK) return st;
        HIPCHK(hipStreamSynchronize(stream));
        return AVN_OK;
    }
    avn_status solver_bodies_download(const avn_solver_bodies_out* o) override {
        if (!o) return AVN_ERR_BAD_ARG;
        size_t n = dw.n_bodies;
        avn_status st = stage_reserve(al(sizeof(T) * 3 * n) * 5 + al(sizeof(T) * 4 * n) + al(sizeof(T) * 6 * n) + al(sizeof(T) * n) * 3 + al(4 * n) + al(2 * n) + 4096);
        if (st != AVN_OK) return st;
        SolverBodiesStage<T> s;
        s.linear_velocity = o->linear_velocity ? stage_alloc<T>(3 * n) : nullptr;
        s.angular_velocity = o->angular_velocity ? stage_alloc<T>(3 * n) : nullptr;
        s.delta_position = o->delta_position ? stage_alloc<T>(3 * n) : nullptr;
        s.delta_rotation = o->delta_rotation ? stage_alloc<T>(4 * n) : nullptr;
        s.flags = o->flags ? stage_alloc<uint32_t>(n) : nullptr;
        s.inv_mass = o->inv_mass ? stage_alloc<T>(n) : nullptr;
        s.inv_inertia_world = o->inv_inertia_world ? stage_alloc<T>(6 * n) : nullptr;
        s.dominance = o->dominance ? stage_alloc<int16_t>(n) : nullptr;
        s.linear_increment = o->linear_increment ? stage_alloc<T>(3 * n) : nullptr;
        s.angular_increment = o->angular_increment ? stage_alloc<T>(3 * n) : nullptr;
        s.linear_damping_rhs = o->linear_damping_rhs ? stage_alloc<T>(n) : nullptr;
        s.angular_damping_rhs = o->angular_damping_rhs ? stage_alloc<T>(n) : nullptr;
        launch_unpack_solver_bodies<T>(dw, s, stream);
        HIPCHK(hipGetLastError());
#define SOUT(dst, src, cnt, U) do { if ((st = stage_out<U>(dst, src, cnt)) != AVN_OK) return st; } while (0)
        SOUT(o->linear_velocity, s.linear_velocity, 3 * n, T); SOUT(o->angular_velocity, s.angular_velocity, 3 * n, T);
        SOUT(o->delta_position, s.delta_position, 3 * n, T); SOUT(o->delta_rotation, s.delta_rotation, 4 * n, T);
        SOUT(o->flags, s.flags, n, uint32_t); SOUT(o->inv_mass, s.inv_mass, n, T); SOUT(o->inv_inertia_world, s.inv_inertia_world, 6 * n, T);
        SOUT(o->dominance, s.dominance, n, int16_t); SOUT(o->linear_increment, s.linear_increment, 3 * n, T);
        SOUT(o->angular_increment, s.angular_increment, 3 * n, T); SOUT(o->linear_damping_rhs, s.linear_damping_rhs, n, T);
        SOUT(o->angular_damping_rhs, s.angular_damping_rhs, n, T);
        HIPCHK(hipStreamSynchronize(stream));
        return AVN_OK;
    }

    // ---- manifolds ---------------------------------------------------------------------------------------
    avn_status manifolds_upload(const avn_manifolds* m) override {
        if (!have_bodies) { error = "manifolds_upload before bodies_upload"; return AVN_ERR_STATE; }
        if (!m || !m->color_offsets || (m->count && (!m->body1 || !m->body2 || !m->normal || !m->friction || !m->restitution || !m->point_count ||
                                                    !m->anchor1 || !m->anchor2 || !m->penetration || !m->normal_speed))) {
            error = "manifolds_upload: null array"; return AVN_ERR_BAD_ARG;
        }
        uint32_t M = m->count;
        if (m->color_offsets[0] != 0 || m->color_offsets[AVN_GRAPH_COLOR_COUNT] != M) { error = "manifolds_upload: bad color_offsets"; return AVN_ERR_BAD_ARG; }
        for (int c = 0; c < AVN_GRAPH_COLOR_COUNT; ++c)
            if (m->color_offsets[c] > m->color_offsets[c + 1]) { error = "manifolds_upload: color_offsets not monotone"; return AVN_ERR_BAD_ARG; }
        for (uint32_t i = 0; i < M; ++i) {
            if (m->body1[i] < 0 || m->body2[i] < 0 || (uint32_t)m->body1[i] >= dw.n_bodies || (uint32_t)m->body2[i] >= dw.n_bodies) { error = "manifolds_upload: body index out of range"; return AVN_ERR_BAD_ARG; }
            if (m->point_count[i] > AVN_MAX_MANIFOLD_POINTS) { error = "manifolds_upload: point_count > 4"; return AVN_ERR_BAD_ARG; }
        }
        any_restitution = false;
        for (uint32_t i = 0; i < M && !any_restitution; ++i) any_restitution = !(((const T*)m->restitution)[i] == T(0));
        if (use_handles) graph_valid = false;
        use_handles = false;  // the manifolds come from the host again
        avn_status st0 = ensure_manifold_capacity(M);
        if (st0 != AVN_OK) return st0;
        if (dw.n_manifolds != M) graph_valid = false;
        dw.n_manifolds = M;
        set_color_offsets(m->color_offsets);
        size_t total = al(4 * (size_t)M) * 2 + al(sizeof(T) * 3 * M) * 2 + al(sizeof(T) * M) * 2 + al(M) * 2 + al(sizeof(T) * 12 * M) * 2 + al(sizeof(T) * 4 * M) * 3 + al(sizeof(T) * 8 * M);
        avn_status st = stage_reserve(total + 64 * 32);
        if (st != AVN_OK) return st;
        HIPCHK(hipMemcpyAsync(dw.color_offsets, color_offsets, sizeof color_offsets, hipMemcpyHostToDevice, stream));
        ManifoldStage<T> s;
        std::memset(&s, 0, sizeof s);
        SIN(body1, m->body1, M, int32_t); SIN(body2, m->body2, M, int32_t); SIN(normal, m->normal, 3 * (size_t)M, T);
        SIN(friction, m->friction, M, T); SIN(restitution, m->restitution, M, T); SIN(tangent_velocity, m->tangent_velocity, 3 * (size_t)M, T);
        SIN(point_count, m->point_count, M, uint8_t); SIN(manifold_flags, m->manifold_flags, M, uint8_t);
        SIN(anchor1, m->anchor1, 12 * (size_t)M, T); SIN(anchor2, m->anchor2, 12 * (size_t)M, T);
        SIN(penetration, m->penetration, 4 * (size_t)M, T); SIN(normal_speed, m->normal_speed, 4 * (size_t)M, T);
        SIN(warm_n, m->warm_start_normal_impulse, 4 * (size_t)M, T); SIN(warm_t, m->warm_start_tangent_impulse, 8 * (size_t)M, T);
        launch_pack_manifolds<T>(dw, s, stream);
        HIPCHK(hipGetLastError());
        h_m_body1.assign(m->body1, m->body1 + M);
        h_m_body2.assign(m->body2, m->body2 + M);
        incidence_dirty = true;
        HIPCHK(hipStreamSynchronize(stream));
        return AVN_OK;
    }
    // Incidence CSR of the body-centric warm start: per body that has a SolverBody, its (manifold, side) entries in SOLVE
    // order = overflow colour first, then colours 0..22 (reference plugin.rs:461-470), list order inside a colour.
    // Incidence of the body-centric warm start.  Colours 0..22: the slot table is (re)built ON THE DEVICE from the manifold arrays
    // (launch_build_incidence_slots, run after the manifolds are in place).  Host part: only the overflow colour -- its
    // per-body entry lists (CSR, list order) and its level schedule.
    bool slots_dirty = true;
    bool ovf_csr_dirty = false;
    avn_status rebuild_incidence() {
        if (!incidence_dirty) return AVN_OK;
        if (pipe_dev) { ovf_csr_dirty = true; return rebuild_incidence_device(); }
        uint32_t N = dw.n_bodies, M = dw.n_manifolds;
        if (M == 0) { incidence_dirty = false; island_mode = false; islands_dirty = false; return AVN_OK; }
        if (h_body_has_sb.size() != N || h_m_body1.size() != M) { error = "incidence: bodies / manifolds out of sync"; return AVN_ERR_STATE; }
        HIPCHK(hipStreamSynchronize(stream));
        hipError_t err;
        {   // slot table storage: 23 colour planes of cap_bodies entries
            bool moved = b_inc_slot.ensure((size_t)AVN_COLOR_OVERFLOW_INDEX * cap_bodies * sizeof(uint32_t), err);
            if (err != hipSuccess) { error = "hipMalloc failed"; return AVN_ERR_OOM; }
            if (moved || dw.inc_stride != cap_bodies) graph_valid = false;
            dw.inc_slot = b_inc_slot.as<uint32_t>();
            dw.inc_stride = cap_bodies;
            slots_dirty = true;
        }
        const uint32_t o0 = color_offsets[AVN_COLOR_OVERFLOW_INDEX], o1 = color_offsets[AVN_COLOR_OVERFLOW_INDEX + 1];
        std::vector<uint32_t>& off = inc_off_h; std::vector<uint32_t>& cursor = inc_cursor_h; std::vector<uint32_t>& ent = inc_ent_h;
        off.assign((size_t)N + 1, 0u);
        for (uint32_t m = o0; m < o1; ++m) {
            if (h_body_has_sb[h_m_body1[m]]) ++off[(size_t)h_m_body1[m] + 1];
            if (h_body_has_sb[h_m_body2[m]]) ++off[(size_t)h_m_body2[m] + 1];
        }
        for (uint32_t i = 0; i < N; ++i) off[i + 1] += off[i];
        cursor.assign(off.begin(), off.end() - 1);
        ent.resize(off[N]);
        for (uint32_t m = o0; m < o1; ++m) {
            uint32_t a = (uint32_t)h_m_body1[m], b = (uint32_t)h_m_body2[m];
            if (h_body_has_sb[a]) ent[cursor[a]++] = m;
            if (h_body_has_sb[b]) ent[cursor[b]++] = m | 0x80000000u;
        }
        bool moved = b_inc_off.ensure(((size_t)N + 1) * 4, err);
        if (err != hipSuccess) { error = "hipMalloc failed"; return AVN_ERR_OOM; }
        moved |= b_inc_ent.ensure(std::max<size_t>(ent.size(), 1) * sizeof(uint32_t), err);
        if (err != hipSuccess) { error = "hipMalloc failed"; return AVN_ERR_OOM; }
        if (moved || !dw.inc_off) graph_valid = false;
        dw.inc_off = b_inc_off.as<uint32_t>();
        dw.inc_ent = b_inc_ent.as<uint32_t>();
        if (o1 > o0 || overflow_csr_nonzero) {   // an all-zero offset array stays valid while the overflow colour is empty
            HIPCHK(hipMemcpyAsync(b_inc_off.p, off.data(), off.size() * 4, hipMemcpyHostToDevice, stream));
            if (!ent.empty()) HIPCHK(hipMemcpyAsync(b_inc_ent.p, ent.data(), ent.size() * sizeof(uint32_t), hipMemcpyHostToDevice, stream));
            HIPCHK(hipStreamSynchronize(stream));
            overflow_csr_nonzero = o1 > o0;
        } else if (moved || overflow_csr_bodies != N) {
            HIPCHK(hipMemsetAsync(b_inc_off.p, 0, ((size_t)N + 1) * 4, stream));
        }
        overflow_csr_bodies = N;
        {   // level schedule of the overflow colour (k_overflow_pass): keys = the bodies a manifold can modify
            std::vector<uint32_t> ms(o1 - o0);
            std::vector<int32_t> k1(o1 - o0), k2(o1 - o0);
            for (uint32_t m = o0; m < o1; ++m) {
                ms[m - o0] = m;
                k1[m - o0] = h_body_has_sb[h_m_body1[m]] ? h_m_body1[m] : -1;
                k2[m - o0] = h_body_has_sb[h_m_body2[m]] ? h_m_body2[m] : -1;
            }
            uint32_t before = sched_overflow.n_components;
            sched_overflow.build(ms, k1, k2, N, ms.size() > overflow_level_threshold);
            void* p0 = sched_overflow.d_order.p; void* p1 = sched_overflow.d_level_offsets.p; void* p2 = sched_overflow.d_comp_level_begin.p;
            avn_status st;
            if ((st = upload_u32(sched_overflow.d_comp_level_begin, sched_overflow.comp_level_begin)) != AVN_OK) return st;
            if ((st = upload_u32(sched_overflow.d_level_offsets, sched_overflow.level_offsets)) != AVN_OK) return st;
            if ((st = upload_u32(sched_overflow.d_order, sched_overflow.order)) != AVN_OK) return st;
            if ((st = upload_u32(sched_overflow.d_gorder, sched_overflow.gorder)) != AVN_OK) return st;
            HIPCHK(hipStreamSynchronize(stream));
            if (o1 > o0) graph_valid = false;  // level sizes are captured launch parameters
            if (before != sched_overflow.n_components || p0 != sched_overflow.d_order.p || p1 != sched_overflow.d_level_offsets.p || p2 != sched_overflow.d_comp_level_begin.p) graph_valid = false;
        }
        islands_dirty = true;   // rebuilt by solver_front AFTER the prepare kernels are enqueued (host work overlaps them)
        incidence_dirty = false;
        return AVN_OK;
    }
    // Island blocks (k_island_substeps).  Islands = connected components of the bodies that have a SolverBody under "share a
    // manifold" (a body without one -- static, sleeping, disabled -- is never written by the solver and joins nothing;
    // kinematic bodies DO have a SolverBody that the solver reads and re-writes, so they merge like dynamic ones).  Eligible
    // when f32, no joints, few enough manifolds for the colour launches to be latency-bound and every island fits a block.
    bool island_candidate(size_t M) const { return sizeof(T) == 4 && island_enabled && M != 0 && M <= island_max_manifolds; }
    avn_status rebuild_island_blocks() {
        island_mode = false;
        const uint32_t N = dw.n_bodies, M = dw.n_manifolds;
        if (!island_candidate(M) || dw.n_joints) return AVN_OK;
        auto has_sb = [&](int32_t b) { return b >= 0 && (uint32_t)b < N && h_body_has_sb[(uint32_t)b]; };
        std::vector<uint32_t>& parent = isl_parent;
        parent.resize(N);
        bool labelled = false;
        static const bool host_labels = getenv("AVN_ISLAND_LABELS_HOST") != nullptr;   // A/B: the host union-find below
        if (pipe_dev && !host_labels) {
            // device closed loop: the manifolds' bodies are on the device already -- label the islands there (k_islands.hip: lock-free
            // union-find, root = lowest body index, only bodies with a SolverBody connect) and fetch 4 bytes per body; parent[] then holds
            // roots directly
            avn_status st = island_buffers();
            if (st != AVN_OK) return st;
            HIPCHK(hipMemsetAsync(b_isl_ctr.p, 0, 64, stream));
            launch_islands<T>(dw, b_isl_parent.as<uint32_t>(), b_isl_label.as<uint32_t>(), b_isl_ctr.as<uint32_t>(), stream, 1u);
            launches += 3;
            HIPCHK(hipMemcpyAsync(parent.data(), b_isl_label.p, (size_t)N * 4, hipMemcpyDeviceToHost, stream));
            HIPCHK(hipStreamSynchronize(stream));
            for (uint32_t i = 0; i < N; ++i) if (parent[i] == 0xFFFFFFFFu) parent[i] = i;   // (bodies without a SolverBody: never asked)
            labelled = true;
        } else for (uint32_t i = 0; i < N; ++i) parent[i] = i;
        auto find = [&](uint32_t x) { while (parent[x] != x) { parent[x] = parent[parent[x]]; x = parent[x]; } return x; };
        for (uint32_t m = 0; m < M && !labelled; ++m) {
            int32_t a = h_m_body1[m], b = h_m_body2[m];
            if (has_sb(a) && has_sb(b)) { uint32_t ra = find((uint32_t)a), rb = find((uint32_t)b); if (ra != rb) parent[std::max(ra, rb)] = std::min(ra, rb); }
        }
        // islands numbered by their lowest body (the root: unions keep the smaller index on top), bodies inside in index order
        std::vector<uint32_t>& island_of = isl_island_of; std::vector<uint32_t>& count = isl_count;
        island_of.assign(N, 0xFFFFFFFFu);
        count.clear();
        for (uint32_t i = 0; i < N; ++i) {
            if (!h_body_has_sb[i]) continue;
            uint32_t r = find(i);
            if (r == i) { island_of[i] = (uint32_t)count.size(); count.push_back(0u); }
            island_of[i] = island_of[r];   // r <= i: already numbered
            if (++count[island_of[i]] > ISLAND_MAX_BODIES) return AVN_OK;   // an island too big for one workgroup's LDS: device-wide path
        }
        const uint32_t n_islands = (uint32_t)count.size();
        if (!n_islands) return AVN_OK;
        // manifolds per island (a manifold belongs to the island of its body that has a SolverBody)
        std::vector<uint32_t>& mcount = isl_mcount;
        mcount.assign(n_islands, 0u);
        for (uint32_t m = 0; m < M; ++m) {
            int32_t a = h_m_body1[m], b = h_m_body2[m];
            if (has_sb(a)) ++mcount[island_of[(uint32_t)a]]; else if (has_sb(b)) ++mcount[island_of[(uint32_t)b]]; else ++mcount[0];
        }
        // blocks = runs of consecutive islands of at most island_pack_bodies bodies (one island may exceed that, up to the LDS cap);
        // a run is also closed when its bodies + constraint records would no longer fit the LDS staging of the kernel's CACHE variant
        std::vector<uint32_t>& block_of = isl_block_of_island; std::vector<uint32_t>& body_off = isl_body_off;
        block_of.resize(n_islands);
        body_off.assign(1, 0u);
        std::vector<uint32_t>& cursor = isl_cursor;   // per island: next LDS slot
        cursor.resize(n_islands);
        // pack target: enough blocks to cover the 256 CUs before blocks grow (a block's pass time is flat up to ~256 manifolds per colour)
        uint32_t n_sb = 0;
        for (uint32_t k = 0; k < n_islands; ++k) n_sb += count[k];
        const uint32_t pack = std::min<uint32_t>(island_pack_bodies, std::max<uint32_t>(64u, (n_sb + 255u) / 256u));
        uint32_t in_block = 0, m_in_block = 0, max_bodies = 0, max_manifolds = 0;
        auto close_block = [&]() { max_bodies = std::max(max_bodies, in_block); max_manifolds = std::max(max_manifolds, m_in_block); body_off.push_back(body_off.back() + in_block); in_block = 0; m_in_block = 0; };
        for (uint32_t k = 0; k < n_islands; ++k) {
            if (in_block && (in_block + count[k] > pack || 6u * (in_block + count[k]) + 20u * (m_in_block + mcount[k]) > ISLAND_LDS_VEC4)) close_block();
            block_of[k] = (uint32_t)body_off.size() - 1;
            cursor[k] = in_block;
            in_block += count[k];
            m_in_block += mcount[k];
        }
        close_block();
        // the LDS layout is the same for every block (sized by the largest body and manifold counts)
        const uint32_t lm_pad = (max_manifolds + 1u) & ~1u;   // (the entry list behind the records is uint2: keep it 16-byte aligned)
        const bool cache_records = island_cache_records && 6u * max_bodies + 20u * lm_pad <= ISLAND_LDS_VEC4;
        const uint32_t n_blocks = (uint32_t)body_off.size() - 1;
        std::vector<uint32_t>& slot = isl_slot; std::vector<uint32_t>& bodies = isl_bodies;
        slot.assign(N, 0u);
        bodies.resize(body_off.back());
        for (uint32_t i = 0; i < N; ++i) {
            uint32_t k = island_of[i];
            if (k == 0xFFFFFFFFu) continue;
            slot[i] = cursor[k]++;
            bodies[body_off[block_of[k]] + slot[i]] = i;
        }
        // entries: counting sort of the manifolds by (block, colour slot), ascending manifold index inside (= list order: the
        // overflow colour's serial order); colour slot 0 = overflow (solved first), 1 + c = colour c
        std::vector<uint32_t>& col_off = isl_col_off; std::vector<uint32_t>& ent = isl_ent;
        col_off.assign((size_t)n_blocks * AVN_GRAPH_COLOR_COUNT + 1, 0u);
        auto key_of = [&](uint32_t m, uint32_t c) -> size_t {
            int32_t a = h_m_body1[m], b = h_m_body2[m];
            uint32_t blk = has_sb(a) ? block_of[island_of[(uint32_t)a]] : has_sb(b) ? block_of[island_of[(uint32_t)b]] : 0u;  // (no SolverBody on either side: touches no body, any block)
            return (size_t)blk * AVN_GRAPH_COLOR_COUNT + (c == AVN_COLOR_OVERFLOW_INDEX ? 0u : c + 1u);
        };
        for (uint32_t c = 0; c < AVN_GRAPH_COLOR_COUNT; ++c)
            for (uint32_t m = color_offsets[c]; m < color_offsets[c + 1]; ++m) ++col_off[key_of(m, c) + 1];
        for (size_t i = 1; i < col_off.size(); ++i) col_off[i] += col_off[i - 1];
        std::vector<uint32_t> next(col_off.begin(), col_off.end() - 1);
        ent.resize((size_t)M * 2);
        for (uint32_t c = 0; c < AVN_GRAPH_COLOR_COUNT; ++c)
            for (uint32_t m = color_offsets[c]; m < color_offsets[c + 1]; ++m) {
                int32_t a = h_m_body1[m], b = h_m_body2[m];
                uint32_t e = next[key_of(m, c)]++;
                ent[2 * (size_t)e] = m;
                ent[2 * (size_t)e + 1] = (has_sb(a) ? slot[(uint32_t)a] : 0u) | (has_sb(b) ? slot[(uint32_t)b] : 0u) << 16;
            }
        // one pinned staging block -> one async copy on the solver's stream (the consumer); no synchronisation: the stream
        // was idle when the previous block went up (rebuild_incidence / the narrow-phase read-back synchronise it every step)
        const size_t w0 = body_off.size(), w1 = bodies.size(), w2 = col_off.size(), w3 = ent.size();
        const size_t o1 = (w0 + 63) & ~(size_t)63, o2 = o1 + ((w1 + 63) & ~(size_t)63), o3 = o2 + ((w2 + 63) & ~(size_t)63), words = o3 + w3;
        if (pin_islands.ensure(words * 4) != hipSuccess) { error = "hipHostMalloc failed"; return AVN_ERR_OOM; }
        uint32_t* h = (uint32_t*)pin_islands.p;
        std::memcpy(h, body_off.data(), w0 * 4); std::memcpy(h + o1, bodies.data(), w1 * 4);
        std::memcpy(h + o2, col_off.data(), w2 * 4); std::memcpy(h + o3, ent.data(), w3 * 4);
        if (words * 4 > b_isl_bodies.cap) HIPCHK(hipStreamSynchronize(stream));   // growing frees the old block: nothing may still read it
        hipError_t err;
        b_isl_bodies.ensure(words * 4, err);
        if (err != hipSuccess) { error = "hipMalloc failed"; return AVN_ERR_OOM; }
        HIPCHK(hipMemcpyAsync(b_isl_bodies.p, h, words * 4, hipMemcpyHostToDevice, stream));
        uint32_t* d = b_isl_bodies.as<uint32_t>();
        islands = IslandBlocks{d, d + o1, d + o2, (const uint2*)(d + o3), n_blocks, max_bodies, lm_pad, cache_records ? 1u : 0u};
        island_mode = true;
        return AVN_OK;
    }
    avn_status impulses_download(const avn_impulses_out* o) override {
        if (!o) return AVN_ERR_BAD_ARG;
        size_t M = dw.n_manifolds;
        avn_status st = stage_reserve(al(sizeof(T) * 4 * M) * 2 + al(sizeof(T) * 8 * M) + 1024);
        if (st != AVN_OK) return st;
        T* a = o->warm_start_normal_impulse ? stage_alloc<T>(4 * M) : nullptr;
        T* b = o->warm_start_tangent_impulse ? stage_alloc<T>(8 * M) : nullptr;
        T* c = o->normal_impulse ? stage_alloc<T>(4 * M) : nullptr;
        launch_unpack_impulses<T>(dw, a, b, c, stream);
        HIPCHK(hipGetLastError());
        SOUT(o->warm_start_normal_impulse, a, 4 * M, T); SOUT(o->warm_start_tangent_impulse, b, 8 * M, T); SOUT(o->normal_impulse, c, 4 * M, T);
        HIPCHK(hipStreamSynchronize(stream));
        return AVN_OK;
    }
    avn_status constraints_download(const avn_constraints_out* o) override {
        if (!o) return AVN_ERR_BAD_ARG;
        size_t M = dw.n_manifolds;
        avn_status st = stage_reserve(al(M) * 2 + al(2 * M) + al(sizeof(T) * 3 * M) + al(sizeof(T) * 12 * M) * 2 + al(sizeof(T) * 4 * M) * 4 + al(sizeof(T) * 8 * M) + 4096);
        if (st != AVN_OK) return st;
        ConstraintsStage<T> s;
        s.point_count = o->point_count ? stage_alloc<uint8_t>(M) : nullptr;
        s.softness_non_dynamic = o->softness_non_dynamic ? stage_alloc<uint8_t>(M) : nullptr;
        s.relative_dominance = o->relative_dominance ? stage_alloc<int16_t>(M) : nullptr;
        s.tangent1 = o->tangent1 ? stage_alloc<T>(3 * M) : nullptr;
        s.anchor1 = o->anchor1 ? stage_alloc<T>(12 * M) : nullptr;
        s.initial_separation = o->initial_separation ? stage_alloc<T>(4 * M) : nullptr;
        s.normal_impulse = o->normal_impulse ? stage_alloc<T>(4 * M) : nullptr;
        s.total_impulse = o->total_impulse ? stage_alloc<T>(4 * M) : nullptr;
        s.normal_effective_mass = o->normal_effective_mass ? stage_alloc<T>(4 * M) : nullptr;
        s.tangent_impulse = o->tangent_impulse ? stage_alloc<T>(8 * M) : nullptr;
        s.tangent_k = o->tangent_effective_inverse_mass ? stage_alloc<T>(12 * M) : nullptr;
        if (stage_off) HIPCHK(hipMemsetAsync(stage.p, 0, stage_off, stream));  // absent constraints read back as zeros
        launch_unpack_constraints<T>(dw, s, stream);
        HIPCHK(hipGetLastError());
        SOUT(o->point_count, s.point_count, M, uint8_t); SOUT(o->softness_non_dynamic, s.softness_non_dynamic, M, uint8_t);
        SOUT(o->relative_dominance, s.relative_dominance, M, int16_t); SOUT(o->tangent1, s.tangent1, 3 * M, T);
        SOUT(o->anchor1, s.anchor1, 12 * M, T); SOUT(o->initial_separation, s.initial_separation, 4 * M, T);
        SOUT(o->normal_impulse, s.normal_impulse, 4 * M, T); SOUT(o->total_impulse, s.total_impulse, 4 * M, T);
        SOUT(o->normal_effective_mass, s.normal_effective_mass, 4 * M, T); SOUT(o->tangent_impulse, s.tangent_impulse, 8 * M, T);
        SOUT(o->tangent_effective_inverse_mass, s.tangent_k, 12 * M, T);
        HIPCHK(hipStreamSynchronize(stream));
        return AVN_OK;
    }

    // ---- joints ----------------------------------------------------------------------------------------
    avn_status distance_joints_upload(const avn_distance_joints* j) override {
        if (!j || (j->count && (!j->body1 || !j->body2 || !j->local_anchor1 || !j->local_anchor2 || !j->limit_min || !j->limit_max || !j->compliance))) {
            error = "distance_joints_upload: null array"; return AVN_ERR_BAD_ARG;
        }
        // the special case joint_type = DISTANCE of joints_upload
        std::vector<uint8_t> types(j->count, (uint8_t)AVN_JOINT_DISTANCE);
        std::vector<T> comp(3 * (size_t)j->count, T(0));
        for (size_t i = 0; i < j->count; ++i) comp[3 * i] = ((const T*)j->compliance)[i];
        avn_joints g;
        std::memset(&g, 0, sizeof g);
        g.count = j->count; g.joint_type = types.data(); g.body1 = j->body1; g.body2 = j->body2;
        g.local_anchor1 = j->local_anchor1; g.local_anchor2 = j->local_anchor2; g.limit_min = j->limit_min; g.limit_max = j->limit_max;
        g.compliance = comp.data(); g.damping_linear = j->damping_linear; g.damping_angular = j->damping_angular;
        g.collision_disabled = j->collision_disabled;
        return joints_upload(&g);
    }
    avn_status joints_upload(const avn_joints* j) override {
        if (!have_bodies) { error = "joints_upload before bodies_upload"; return AVN_ERR_STATE; }
        if (!j || (j->count && (!j->joint_type || !j->body1 || !j->body2 || !j->local_anchor1 || !j->local_anchor2 || !j->compliance))) {
            error = "joints_upload: null array"; return AVN_ERR_BAD_ARG;
        }
        uint32_t J = j->count;
        for (uint32_t i = 0; i < J; ++i) {
            if (j->joint_type[i] >= AVN_JOINT_TYPE_COUNT) { error = "joints_upload: bad joint_type"; return AVN_ERR_BAD_ARG; }
            if (j->body1[i] < 0 || j->body2[i] < 0 || (uint32_t)j->body1[i] >= dw.n_bodies || (uint32_t)j->body2[i] >= dw.n_bodies || j->body1[i] == j->body2[i]) {
                error = "joints_upload: bad body index"; return AVN_ERR_BAD_ARG;
            }
        }
        bool moved = false;
        if (J > cap_joints) {
            HIPCHK(hipStreamSynchronize(stream));
            size_t c = std::max<size_t>(J, cap_joints + cap_joints / 2);
            GROW(b_j_bodies, c, dw.j_bodies); GROW(b_j_a1, c, dw.j_a1); GROW(b_j_a2, c, dw.j_a2); GROW(b_j_par, c, dw.j_par);
            GROW(b_j_b1, c, dw.j_b1); GROW(b_j_b2, c, dw.j_b2); GROW(b_j_ax, c, dw.j_ax); GROW(b_j_l2, c, dw.j_l2);
            GROW(b_j_r1, c, dw.j_r1); GROW(b_j_r2, c, dw.j_r2); GROW(b_j_cd, c, dw.j_cd); GROW(b_j_lag, c, dw.j_lag);
            GROW(b_j_s0, c, dw.j_s0); GROW(b_j_s1, c, dw.j_s1); GROW(b_j_s2, c, dw.j_s2); GROW(b_j_s3, c, dw.j_s3);
            GROW(b_j_rl0, c, dw.j_rl0); GROW(b_j_rl1, c, dw.j_rl1); GROW(b_j_force, c, dw.j_force); GROW(b_j_torque, c, dw.j_torque);
            cap_joints = (uint32_t)c;
        }
        if (moved || dw.n_joints != J) graph_valid = false;
        dw.n_joints = J;
        avn_status st = stage_reserve(al(4 * (size_t)J) * 2 + al(J) * 2 + al(sizeof(T) * 3 * J) * 4 + al(sizeof(T) * 4 * J) * 2 + al(sizeof(T) * J) * 6 + 8192);
        if (st != AVN_OK) return st;
        JointStage<T> s;
        std::memset(&s, 0, sizeof s);
        SIN(joint_type, j->joint_type, J, uint8_t); SIN(limit_flags, j->limit_flags, J, uint8_t);
        SIN(body1, j->body1, J, int32_t); SIN(body2, j->body2, J, int32_t);
        SIN(local_anchor1, j->local_anchor1, 3 * (size_t)J, T); SIN(local_anchor2, j->local_anchor2, 3 * (size_t)J, T);
        SIN(local_basis1, j->local_basis1, 4 * (size_t)J, T); SIN(local_basis2, j->local_basis2, 4 * (size_t)J, T);
        SIN(axis, j->axis, 3 * (size_t)J, T);
        SIN(limit_min, j->limit_min, J, T); SIN(limit_max, j->limit_max, J, T); SIN(limit2_min, j->limit2_min, J, T); SIN(limit2_max, j->limit2_max, J, T);
        SIN(compliance, j->compliance, 3 * (size_t)J, T);
        SIN(damping_linear, j->damping_linear, J, T); SIN(damping_angular, j->damping_angular, J, T);
        launch_pack_joints<T>(dw, s, stream);
        HIPCHK(hipGetLastError());
        h_j_body1.assign(j->body1, j->body1 + J);
        h_j_body2.assign(j->body2, j->body2 + J);
        h_j_type.assign(j->joint_type, j->joint_type + J);
        bool damp = j->damping_linear && j->damping_angular;
        h_j_damped.assign(J, damp ? 1 : 0);
        any_damped = damp && J > 0;
        // body pairs whose joints disable collision (reference broad_phase.rs:423-428)
        std::vector<uint64_t> disabled;
        for (uint32_t i = 0; i < J; ++i)
            if (j->collision_disabled && j->collision_disabled[i]) {
                uint32_t a = (uint32_t)j->body1[i], b = (uint32_t)j->body2[i];
                disabled.push_back(a < b ? ((uint64_t)a << 32) | b : ((uint64_t)b << 32) | a);
            }
        st = build_hash_set(b_disabled_set, bp.disabled_set, bp.disabled_cap, disabled.data(), (uint32_t)disabled.size());
        if (st != AVN_OK) return st;
        joint_schedule_dirty = true;
        HIPCHK(hipStreamSynchronize(stream));
        return AVN_OK;
    }
    avn_status joints_download(const avn_joints_out* o) override {
        if (!o) return AVN_ERR_BAD_ARG;
        size_t J = dw.n_joints;
        avn_status st = stage_reserve(al(sizeof(T) * 3 * J) * 7 + 1024);
        if (st != AVN_OK) return st;
        T* a = o->world_r1 ? stage_alloc<T>(3 * J) : nullptr;
        T* b = o->world_r2 ? stage_alloc<T>(3 * J) : nullptr;
        T* c = o->center_difference ? stage_alloc<T>(3 * J) : nullptr;
        T* d = o->total_lagrange ? stage_alloc<T>(3 * J) : nullptr;
        T* e = o->force ? stage_alloc<T>(3 * J) : nullptr;
        T* f = o->total_rotation_lagrange ? stage_alloc<T>(3 * J) : nullptr;
        T* g = o->torque ? stage_alloc<T>(3 * J) : nullptr;
        launch_unpack_joints<T>(dw, a, b, c, d, e, f, g, stream);
        HIPCHK(hipGetLastError());
        SOUT(o->world_r1, a, 3 * J, T); SOUT(o->world_r2, b, 3 * J, T); SOUT(o->center_difference, c, 3 * J, T);
        SOUT(o->total_lagrange, d, 3 * J, T); SOUT(o->force, e, 3 * J, T);
        SOUT(o->total_rotation_lagrange, f, 3 * J, T); SOUT(o->torque, g, 3 * J, T);
        HIPCHK(hipStreamSynchronize(stream));
        return AVN_OK;
    }
    avn_status upload_u32(DevBuf& b, const std::vector<uint32_t>& v) {
        hipError_t err;
        b.ensure(std::max<size_t>(v.size(), 1) * 4, err);
        if (err != hipSuccess) { error = "hipMalloc failed"; return AVN_ERR_OOM; }
        if (!v.empty()) HIPCHK(hipMemcpyAsync(b.p, v.data(), v.size() * 4, hipMemcpyHostToDevice, stream));
        return AVN_OK;
    }
    avn_status rebuild_joint_schedules() {
        if (!joint_schedule_dirty) return AVN_OK;
        HIPCHK(hipStreamSynchronize(stream));
        uint32_t J = dw.n_joints, N = dw.n_bodies;
        // the reference's serial order: one system per joint type in the order of xpbd/plugin.rs:77-82 (= the AVN_JOINT_* ids),
        // each iterating its joints in array (= spawn) order
        std::vector<uint32_t> all(J);
        std::iota(all.begin(), all.end(), 0u);
        std::stable_sort(all.begin(), all.end(), [&](uint32_t a, uint32_t b) { return h_j_type[a] < h_j_type[b]; });
        std::vector<int32_t> k1(J), k2(J);
        for (uint32_t k = 0; k < J; ++k) {
            uint32_t i = all[k];
            // bodies without a SolverBody are DUMMY in solve_xpbd_joint: never modified => they do not serialise joints
            k1[k] = h_body_has_sb[h_j_body1[i]] ? h_j_body1[i] : -1;
            k2[k] = h_body_has_sb[h_j_body2[i]] ? h_j_body2[i] : -1;
        }
        sched_solve.build(all, k1, k2, N);
        std::vector<uint32_t> damped;
        std::vector<int32_t> d1, d2;
        sched_damp.touches_dummy = false;
        for (uint32_t k = 0; k < J; ++k) {
            uint32_t i = all[k];
            if (h_j_damped[i]) {
                damped.push_back(i);
                // joint_damping's DUMMY bodies are shared by the joints of ONE type and mutable: virtual bodies N + 2t, N + 2t + 1
                bool m1 = !h_body_has_sb[h_j_body1[i]], m2 = !h_body_has_sb[h_j_body2[i]];
                d1.push_back(m1 ? (int32_t)(N + 2u * h_j_type[i]) : h_j_body1[i]);
                d2.push_back(m2 ? (int32_t)(N + 2u * h_j_type[i] + 1u) : h_j_body2[i]);
                if (m1 || m2) sched_damp.touches_dummy = true;
            }
        }
        sched_damp.build(damped, d1, d2, N + DUMMY_SLOTS);
        avn_status st;
        for (JointSchedule* s : {&sched_solve, &sched_damp}) {
            if ((st = upload_u32(s->d_comp_level_begin, s->comp_level_begin)) != AVN_OK) return st;
            if ((st = upload_u32(s->d_level_offsets, s->level_offsets)) != AVN_OK) return st;
            if ((st = upload_u32(s->d_order, s->order)) != AVN_OK) return st;
            std::vector<uint32_t> rec(4 * s->order.size());
            for (size_t k = 0; k < s->order.size(); ++k) {
                const uint32_t j = s->order[k];
                rec[4 * k] = j; rec[4 * k + 1] = (uint32_t)h_j_body1[j]; rec[4 * k + 2] = (uint32_t)h_j_body2[j]; rec[4 * k + 3] = 0u;
            }
            if ((st = upload_u32(s->d_rec, rec)) != AVN_OK) return st;
            HIPCHK(hipStreamSynchronize(stream));   // (`rec` is a local: the copy must have left it)
        }
        HIPCHK(hipStreamSynchronize(stream));
        joint_schedule_dirty = false;
        graph_valid = false;
        return AVN_OK;
    }

    // ---- broad phase -------------------------------------------------------------------------------------
    avn_status build_hash_set(DevBuf& buf, uint64_t*& tab, uint32_t& cap, const uint64_t* host_keys, uint32_t n) {
        if (n == 0) { if (cap) graph_valid = false; cap = 0; return AVN_OK; }
        uint32_t need = 64;
        while (need < 2 * n + 16) need <<= 1;
        hipError_t err;
        buf.ensure((size_t)need * 8, err);
        if (err != hipSuccess) { error = "hipMalloc failed"; return AVN_ERR_OOM; }
        tab = buf.as<uint64_t>();
        cap = need;
        avn_status st = stage_reserve((size_t)n * 8 + 1024);
        if (st != AVN_OK) return st;
        HIPCHK(hipMemsetAsync(tab, 0xFF, (size_t)cap * 8, stream));
        const uint64_t* d;
        if ((st = stage_in<uint64_t>(host_keys, n, &d)) != AVN_OK) return st;
        launch_hs_insert(tab, cap, d, n, stream);
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(stream));
        return AVN_OK;
    }
    avn_status existing_pairs_upload(const uint64_t* keys, size_t n) override {
        if (n && !keys) return AVN_ERR_BAD_ARG;
        hipError_t err;
        b_pair_keys.ensure(std::max<size_t>(n, 1) * 8, err);
        if (err != hipSuccess) { error = "hipMalloc failed"; return AVN_ERR_OOM; }
        if (n) HIPCHK(hipMemcpyAsync(b_pair_keys.p, keys, n * 8, hipMemcpyHostToDevice, stream));
        n_pair_keys = (uint32_t)n;
        return rebuild_pair_set((uint32_t)n);
    }
    // (re)build the device pair set from the key list with room for `expect` keys
    avn_status rebuild_pair_set(uint32_t expect) {
        if (contact_keys_live) {
            // rows have been removed since the key list was built (contact_pairs_remove): rebuild from the live keys only
            std::vector<uint64_t> keys(h_live_keys.begin(), h_live_keys.end());
            hipError_t e2;
            b_pair_keys.ensure(std::max<size_t>(keys.size(), 1) * 8, e2);
            if (e2 != hipSuccess) { error = "hipMalloc failed"; return AVN_ERR_OOM; }
            HIPCHK(hipStreamSynchronize(bs));
            if (!keys.empty()) HIPCHK(hipMemcpy(b_pair_keys.p, keys.data(), keys.size() * 8, hipMemcpyHostToDevice));
            n_pair_keys = (uint32_t)keys.size();
            expect = std::max(expect, n_pair_keys + n_pair_keys / 2);
        }
        uint32_t need = 1024;
        while (need < 2 * (expect + 16)) need <<= 1;
        hipError_t err;
        b_pair_set.ensure((size_t)need * 8, err);
        if (err != hipSuccess) { error = "hipMalloc failed"; return AVN_ERR_OOM; }
        bp.pair_set = b_pair_set.as<uint64_t>();
        bp.pair_set_cap = need;
        HIPCHK(hipMemsetAsync(bp.pair_set, 0xFF, (size_t)need * 8, bs));
        launch_hs_insert(bp.pair_set, need, b_pair_keys.as<uint64_t>(), n_pair_keys, bs);
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(bs));
        return AVN_OK;
    }
    avn_status colliders_upload(const avn_colliders* c) override {
        if (!have_bodies) { error = "colliders_upload before bodies_upload"; return AVN_ERR_STATE; }
        if (!c || (c->count && (!c->entity_index || !c->body || !c->shape || !c->half_extents))) { error = "colliders_upload: null array"; return AVN_ERR_BAD_ARG; }
        uint32_t C = c->count;
        for (uint32_t i = 0; i < C; ++i)
            if (c->body[i] < 0 || (uint32_t)c->body[i] >= dw.n_bodies) { error = "colliders_upload: body index out of range"; return AVN_ERR_BAD_ARG; }
        HIPCHK(hipStreamSynchronize(stream));
        HIPCHK(hipStreamSynchronize(stream_bp));
        bool same = slot_entity.size() == C && (C == 0 || std::memcmp(slot_entity.data(), c->entity_index, C * 4) == 0);
        std::vector<uint32_t> new_iv;
        std::vector<V> keep_min, keep_max;
        if (!same) {
            std::unordered_map<uint32_t, uint32_t> next_slot;
            next_slot.reserve(C * 2);
            for (uint32_t i = 0; i < C; ++i)
                if (!next_slot.emplace(c->entity_index[i], i).second) { error = "colliders_upload: duplicate entity_index"; return AVN_ERR_BAD_ARG; }
            // retain_mut (reference broad_phase.rs:230-279) on the current device order, then append the new ones
            std::vector<uint32_t> old_iv(bp.n_intervals);
            if (bp.n_intervals) HIPCHK(hipMemcpy(old_iv.data(), bp.iv_collider, (size_t)bp.n_intervals * 4, hipMemcpyDeviceToHost));
            std::vector<uint8_t> known(C, 0);
            // carry the ColliderAabb component of surviving colliders over to their new slot
            std::vector<V> omin(bp.n_colliders), omax(bp.n_colliders);
            if (bp.n_colliders) {
                HIPCHK(hipMemcpy(omin.data(), bp.aabb_min, (size_t)bp.n_colliders * sizeof(V), hipMemcpyDeviceToHost));
                HIPCHK(hipMemcpy(omax.data(), bp.aabb_max, (size_t)bp.n_colliders * sizeof(V), hipMemcpyDeviceToHost));
            }
            keep_min.assign(C, make4<T>(0, 0, 0, 0));
            keep_max.assign(C, make4<T>(0, 0, 0, 0));
            for (uint32_t s = 0; s < bp.n_colliders && s < slot_entity.size(); ++s) {
                auto it = next_slot.find(slot_entity[s]);
                if (it != next_slot.end()) { keep_min[it->second] = omin[s]; keep_max[it->second] = omax[s]; }
            }
            for (uint32_t iv : old_iv) {
                auto it = next_slot.find(slot_entity[iv]);
                if (it == next_slot.end()) continue;
                new_iv.push_back(it->second);
                known[it->second] = 1;
            }
            for (uint32_t i = 0; i < C; ++i)
                if (!known[i]) new_iv.push_back(i);  // add_new_aabb_intervals: appended at the END in upload order
            slot_entity.assign(c->entity_index, c->entity_index + C);
        }
        bool moved = false;
        if (C > cap_colliders) {
            size_t cc = std::max<size_t>(C, cap_colliders + cap_colliders / 2);
            GROW(b_col_info, cc, bp.col_info); GROW(b_col_he, cc, bp.col_he); GROW(b_col_spec, cc, bp.col_spec); GROW(b_col_layers, cc, bp.col_layers);
            GROW(b_aabb_min, cc, bp.aabb_min); GROW(b_aabb_max, cc, bp.aabb_max); GROW(b_iv, cc, bp.iv_collider);
            GROW(b_s_minx, cc, bp.s_minx); GROW(b_s_maxx, cc, bp.s_maxx); GROW(b_s_yz, cc + sweep_pad_records(), bp.s_yz); GROW(b_s_bb, cc / sweep_bounds_group() + 2, bp.s_bb); GROW(b_s_end, cc, bp.s_end);
            GROW(b_s_info, cc, bp.s_info); GROW(b_s_flags, cc, bp.s_flags);
            Key* dummy_k; uint32_t* dummy_u;
            GROW(b_keys_a, cc, dummy_k); GROW(b_keys_b, cc, dummy_k); GROW(b_vals_a, cc, dummy_u); GROW(b_vals_b, cc, dummy_u);
            GROW(b_hist, (size_t)256 * radix_blocks((uint32_t)cc) + 256, dummy_u);
            GROW(b_block_sums, std::max<size_t>(scan_block_sums_needed(256 * radix_blocks((uint32_t)cc)), scan_block_sums_needed((uint32_t)cc * sweep_count_slots())) + 16, dummy_u);
            GROW(b_counts, cc * sweep_count_slots() + 1, dummy_u); GROW(b_offsets, cc * sweep_count_slots() + 1, dummy_u);
            {   // long-interval chunks: every interval may need one slot, plus room for the chunks of scene-spanning ones
                size_t lcap = cc + 65536;
                if (const char* e = getenv("AVN_SWEEP_LONG_CAP")) lcap = std::max<size_t>(8, (size_t)strtoull(e, nullptr, 10));   // (tests: force the grow-and-retry path)
                uint8_t* dummy_b;
                GROW(b_long_items, lcap * sweep_long_item_bytes(), dummy_b);
                GROW(b_long_counts, lcap, dummy_u); GROW(b_long_off, lcap, dummy_u);
                sweep_scratch.long_items = b_long_items.p; sweep_scratch.long_counts = b_long_counts.as<uint32_t>();
                sweep_scratch.long_off = b_long_off.as<uint32_t>(); sweep_scratch.long_cap = (uint32_t)lcap;
            }
            cap_colliders = (uint32_t)cc;
        }
        bp.n_colliders = C;
        if (!same) {
            bp.n_intervals = (uint32_t)new_iv.size();
            if (!new_iv.empty()) HIPCHK(hipMemcpy(bp.iv_collider, new_iv.data(), new_iv.size() * 4, hipMemcpyHostToDevice));
            if (C) {
                HIPCHK(hipMemcpy(bp.aabb_min, keep_min.data(), (size_t)C * sizeof(V), hipMemcpyHostToDevice));
                HIPCHK(hipMemcpy(bp.aabb_max, keep_max.data(), (size_t)C * sizeof(V), hipMemcpyHostToDevice));
            }
        }
        avn_status st = stage_reserve(al(4 * (size_t)C) * 4 + al(C) * 2 + al(sizeof(T) * 3 * C) + al(sizeof(T) * C) * 2 + 4096);
        if (st != AVN_OK) return st;
        ColliderStage<T> s;
        std::memset(&s, 0, sizeof s);
        SIN(entity, c->entity_index, C, uint32_t); SIN(body, c->body, C, int32_t); SIN(shape, c->shape, C, uint8_t);
        SIN(half_extents, c->half_extents, 3 * (size_t)C, T); SIN(memberships, c->memberships, C, uint32_t); SIN(filters, c->filters, C, uint32_t);
        SIN(cflags, c->collider_flags, C, uint8_t); SIN(collision_margin, c->collision_margin, C, T); SIN(speculative_margin, c->speculative_margin, C, T);
        launch_pack_colliders<T>(bp, s, stream);
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(stream));
        entity_slot.clear();
        entity_slot.reserve((size_t)C * 2);
        h_col_body.assign(c->body, c->body + C);
        for (uint32_t i = 0; i < C; ++i) entity_slot.emplace(c->entity_index[i], i);
        {   // Friction / Restitution defaults until collider_materials_upload: DefaultFriction 0.5, DefaultRestitution 0, Average
            hipError_t err;
            b_col_mat.ensure(std::max<size_t>(C, 1) * sizeof(V), err);
            if (err != hipSuccess) { error = "hipMalloc failed"; return AVN_ERR_OOM; }
            ct.col_mat = b_col_mat.as<V>();
            std::vector<V> mats(C, make4<T>(T(0.5), T(0), bits_to_scalar((uint32_t)AVN_COMBINE_AVERAGE | ((uint32_t)AVN_COMBINE_AVERAGE << 8), T(0)), T(0)));
            if (C) HIPCHK(hipMemcpy(ct.col_mat, mats.data(), (size_t)C * sizeof(V), hipMemcpyHostToDevice));
            materials_restitution = false;
        }
        have_colliders = true;
        return AVN_OK;
    }
    // ---- narrow phase, part 2 ------------------------------------------------------------------------------------------
    avn_status collider_materials_upload(const avn_collider_materials* m) override {
        if (!m || m->count != bp.n_colliders) { error = "collider_materials_upload: count must equal the collider count"; return AVN_ERR_BAD_ARG; }
        uint32_t C = m->count;
        std::vector<V> mats(C);
        materials_restitution = false;
        for (uint32_t i = 0; i < C; ++i) {
            T fr = m->friction ? ((const T*)m->friction)[i] : T(0.5), re = m->restitution ? ((const T*)m->restitution)[i] : T(0);
            uint32_t fc = m->friction_combine ? m->friction_combine[i] : (uint32_t)AVN_COMBINE_AVERAGE, rc = m->restitution_combine ? m->restitution_combine[i] : (uint32_t)AVN_COMBINE_AVERAGE;
            if (fc < AVN_COMBINE_AVERAGE || fc > AVN_COMBINE_MAX || rc < AVN_COMBINE_AVERAGE || rc > AVN_COMBINE_MAX) { error = "collider_materials_upload: bad combine rule"; return AVN_ERR_BAD_ARG; }
            mats[i] = make4<T>(fr, re, bits_to_scalar(fc | (rc << 8), T(0)), T(0));
            if (!(re == T(0))) materials_restitution = true;
        }
        HIPCHK(hipStreamSynchronize(stream));
        if (C) HIPCHK(hipMemcpy(ct.col_mat, mats.data(), (size_t)C * sizeof(V), hipMemcpyHostToDevice));
        if (use_handles) any_restitution = materials_restitution;
        return AVN_OK;
    }
    avn_status ensure_contact_rows(uint32_t rows) {
        if (rows <= ct.cap) return AVN_OK;
        HIPCHK(hipStreamSynchronize(stream));
        uint32_t old = ct.cap;
        size_t c = std::max<size_t>(rows, (size_t)old + old / 2);
        c = (c + 63) & ~(size_t)63;
        // grow with contents: the rows are persistent state.  The point planes are [p][row]: re-lay them out for the new stride.
        auto grow_flat = [&](DevBuf& b, size_t elem, void** field) -> avn_status {
            hipError_t err;
            b.ensure(c * elem, err, true, stream);
            if (err != hipSuccess) { error = "hipMalloc failed"; return AVN_ERR_OOM; }
            *field = b.p;
            return AVN_OK;
        };
        auto grow_planes = [&](DevBuf& b, size_t elem, void** field) -> avn_status {
            void* np = nullptr;
            if (hipMalloc(&np, 4 * c * elem) != hipSuccess) { error = "hipMalloc failed"; return AVN_ERR_OOM; }
            for (int k = 0; k < 4 && old; ++k)
                if (hipMemcpy((char*)np + (size_t)k * c * elem, (char*)b.p + (size_t)k * old * elem, (size_t)old * elem, hipMemcpyDeviceToDevice) != hipSuccess) { error = "hipMemcpy failed"; return AVN_ERR_HIP; }
            if (b.p) (void)hipFree(b.p);
            b.p = np; b.cap = 4 * c * elem;
            *field = np;
            return AVN_OK;
        };
        avn_status st;
        if ((st = grow_flat(b_ct_meta, sizeof(uint4), (void**)&ct.meta)) != AVN_OK) return st;
        if ((st = grow_flat(b_ct_dcount, sizeof(int32_t), (void**)&ct.dcount)) != AVN_OK) return st;
        if ((st = grow_flat(b_ct_n, sizeof(V), (void**)&ct.n)) != AVN_OK) return st;
        if ((st = grow_flat(b_ct_tv, sizeof(V), (void**)&ct.tv)) != AVN_OK) return st;
        if ((st = grow_planes(b_ct_a1, sizeof(V), (void**)&ct.a1)) != AVN_OK) return st;
        if ((st = grow_planes(b_ct_a2, sizeof(V), (void**)&ct.a2)) != AVN_OK) return st;
        if ((st = grow_planes(b_ct_w, sizeof(V), (void**)&ct.w)) != AVN_OK) return st;
        if ((st = grow_planes(b_ct_fid, sizeof(uint2), (void**)&ct.fid)) != AVN_OK) return st;
        HIPCHK(hipMemset((char*)ct.meta + (size_t)old * sizeof(uint4), 0, (c - old) * sizeof(uint4)));
        ct.cap = (uint32_t)c;
        h_ct_used.resize(c, 0); h_ct_c1.resize(c, 0); h_ct_c2.resize(c, 0); h_ct_b1.resize(c, -1); h_ct_b2.resize(c, -1);
        if (pipe_dev) return pg_ensure_rows(ct.cap);
        return AVN_OK;
    }
    avn_status contact_pairs_add(const avn_contact_pairs* p) override {
        if (!p || (p->count && (!p->contact_id || !p->collider1 || !p->collider2 || !p->pair_flags))) { error = "contact_pairs_add: null array"; return AVN_ERR_BAD_ARG; }
        uint32_t n = p->count;
        if (!n) return AVN_OK;
        uint32_t max_id = 0;
        for (uint32_t i = 0; i < n; ++i) max_id = std::max(max_id, p->contact_id[i]);
        avn_status st = ensure_contact_rows(max_id + 1);
        if (st != AVN_OK) return st;
        std::vector<uint32_t> s1(n), s2(n);
        for (uint32_t i = 0; i < n; ++i) {
            auto a = entity_slot.find(p->collider1[i]), b = entity_slot.find(p->collider2[i]);
            if (a == entity_slot.end() || b == entity_slot.end()) { error = "contact_pairs_add: unknown collider"; return AVN_ERR_BAD_ARG; }
            if (h_ct_used[p->contact_id[i]]) { error = "contact_pairs_add: contact id in use"; return AVN_ERR_STATE; }
            s1[i] = a->second; s2[i] = b->second;
        }
        for (uint32_t i = 0; i < n; ++i) {
            uint32_t id = p->contact_id[i];
            h_ct_used[id] = 1; h_ct_c1[id] = p->collider1[i]; h_ct_c2[id] = p->collider2[i];
            h_ct_b1[id] = h_col_body[s1[i]]; h_ct_b2[id] = h_col_body[s2[i]];
            uint32_t x = p->collider1[i], y = p->collider2[i];
            h_live_keys.insert(x < y ? ((uint64_t)x << 32) | y : ((uint64_t)y << 32) | x);
        }
        contact_keys_live = true;
        if ((st = stage_reserve(al(4 * (size_t)n) * 4 + 1024)) != AVN_OK) return st;
        const uint32_t *d_id, *d_s1, *d_s2, *d_pf;
        if ((st = stage_in<uint32_t>(p->contact_id, n, &d_id)) != AVN_OK) return st;
        if ((st = stage_in<uint32_t>(s1.data(), n, &d_s1)) != AVN_OK) return st;
        if ((st = stage_in<uint32_t>(s2.data(), n, &d_s2)) != AVN_OK) return st;
        if ((st = stage_in<uint32_t>(p->pair_flags, n, &d_pf)) != AVN_OK) return st;
        launch_init_contact_rows<T>(ct, d_id, d_s1, d_s2, d_pf, n, stream);
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(stream));
        return AVN_OK;
    }
    avn_status contact_pairs_remove(const uint32_t* ids, size_t n) override {
        if (n && !ids) return AVN_ERR_BAD_ARG;
        if (!n) return AVN_OK;
        std::vector<uint64_t> keys(n);
        for (size_t i = 0; i < n; ++i) {
            if (ids[i] >= ct.cap || !h_ct_used[ids[i]]) { error = "contact_pairs_remove: no such contact"; return AVN_ERR_STATE; }
            uint32_t x = h_ct_c1[ids[i]], y = h_ct_c2[ids[i]];
            keys[i] = x < y ? ((uint64_t)x << 32) | y : ((uint64_t)y << 32) | x;
        }
        for (size_t i = 0; i < n; ++i) { h_ct_used[ids[i]] = 0; h_live_keys.erase(keys[i]); }
        avn_status st = stage_reserve(al(4 * n) + al(8 * n) + 1024);
        if (st != AVN_OK) return st;
        const uint32_t* d_id; const uint64_t* d_keys;
        if ((st = stage_in<uint32_t>(ids, n, &d_id)) != AVN_OK) return st;
        if ((st = stage_in<uint64_t>(keys.data(), n, &d_keys)) != AVN_OK) return st;
        launch_clear_contact_rows<T>(ct, d_id, (uint32_t)n, stream);
        launch_hs_remove(bp.pair_set, bp.pair_set_cap, d_keys, (uint32_t)n, stream);
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(stream));
        return AVN_OK;
    }
    avn_status active_pairs_set(const uint32_t* ids, size_t n) override {
        if (n && !ids) return AVN_ERR_BAD_ARG;
        for (size_t i = 0; i < n; ++i)
            if (ids[i] >= ct.cap || !h_ct_used[ids[i]]) { error = "active_pairs_set: no such contact"; return AVN_ERR_STATE; }
        HIPCHK(hipStreamSynchronize(stream));
        hipError_t err;
        b_active.ensure(std::max<size_t>(n, 1) * 4, err);
        if (err != hipSuccess) { error = "hipMalloc failed"; return AVN_ERR_OOM; }
        b_changes.ensure(std::max<size_t>(n, 1) * sizeof(avn_contact_change) + 64, err);
        if (err != hipSuccess) { error = "hipMalloc failed"; return AVN_ERR_OOM; }
        if (n) HIPCHK(hipMemcpy(b_active.p, ids, n * 4, hipMemcpyHostToDevice));
        n_active = (uint32_t)n;
        return AVN_OK;
    }
    avn_status narrow_phase() {
        h_changes.clear();
        if (!n_active) return AVN_OK;
        uint32_t* d_count = b_misc.as<uint32_t>() + 40;
        launch_narrow_phase<T>(dw, bp, ct, params, b_active.as<uint32_t>(), n_active, b_changes.as<avn_contact_change>(), d_count, stream);
        ++launches;
        HIPCHK(hipGetLastError());
        // the count and the first CHANGES_PREFIX changes come back in one round trip (pinned memory, one synchronisation);
        // only a step with more changes than that pays a second copy
        const uint32_t prefix = std::min<uint32_t>(CHANGES_PREFIX, n_active);
        HIPCHK(pin_changes.ensure(64 + (size_t)CHANGES_PREFIX * sizeof(avn_contact_change)));
        uint32_t* h_cnt = (uint32_t*)pin_changes.p;
        avn_contact_change* h_pre = (avn_contact_change*)((char*)pin_changes.p + 64);
        HIPCHK(hipMemcpyAsync(h_cnt, d_count, 4, hipMemcpyDeviceToHost, stream));
        HIPCHK(hipMemcpyAsync(h_pre, b_changes.p, (size_t)prefix * sizeof(avn_contact_change), hipMemcpyDeviceToHost, stream));
        HIPCHK(hipStreamSynchronize(stream));
        const uint32_t cnt = *h_cnt;
        if (cnt) {
            h_changes.resize(cnt);
            std::memcpy(h_changes.data(), h_pre, (size_t)std::min(cnt, prefix) * sizeof(avn_contact_change));
            if (cnt > prefix)
                HIPCHK(hipMemcpy(h_changes.data() + prefix, b_changes.as<avn_contact_change>() + prefix, (size_t)(cnt - prefix) * sizeof(avn_contact_change), hipMemcpyDeviceToHost));
            // ContactStatusBits are walked in ascending contact id (system_param.rs:141-145)
            std::sort(h_changes.begin(), h_changes.end(), [](const avn_contact_change& a, const avn_contact_change& b) { return a.contact_id < b.contact_id; });
        }
        return AVN_OK;
    }
    avn_status contact_changes_get(const avn_contact_change** out, size_t* n) override {
        if (!out || !n) return AVN_ERR_BAD_ARG;
        *out = h_changes.data(); *n = h_changes.size();
        return AVN_OK;
    }
    avn_status ensure_manifold_capacity(uint32_t M) {
        bool moved = false;
        if (M > cap_manifolds) {
            HIPCHK(hipStreamSynchronize(stream));
            size_t c = std::max<size_t>(M, cap_manifolds + cap_manifolds / 2);
            c = (c + 63) & ~(size_t)63;  // keep every point plane 1 KiB aligned
            GROW(b_m_bodies, c, dw.m_bodies); GROW(b_m_n, c, dw.m_n); GROW(b_m_tv, c, dw.m_tv); GROW(b_m_meta, c, dw.m_meta);
            GROW(b_mp_a1, 4 * c, dw.mp_a1); GROW(b_mp_a2, 4 * c, dw.mp_a2); GROW(b_mp_w, 4 * c, dw.mp_w);
            GROW(b_c_h1, c, dw.c_h1); GROW(b_c_pa, 4 * c, dw.c_pa); GROW(b_c_pb, 4 * c, dw.c_pb); GROW(b_c_pc, 4 * c, dw.c_pc); GROW(b_c_pd, 4 * c, dw.c_pd);
            GROW(b_c_reldom, c, dw.c_reldom);
            cap_manifolds = (uint32_t)c;
            dw.m_stride = cap_manifolds;
        }
        if (moved) graph_valid = false;
        return AVN_OK;
    }
    uint32_t ovf_grid_blocks = 0;   // device closed loop: captured grid of the overflow colour's dataflow pass (with slack, like the colours')
    void set_color_offsets(const uint32_t* offsets) {
        if (!use_handles && std::memcmp(color_offsets, offsets, sizeof color_offsets) != 0) graph_valid = false;  // (ranges captured as kernel arguments; handle mode reads them from the device)
        std::memcpy(color_offsets, offsets, sizeof color_offsets);
        {
            const uint32_t n23 = color_offsets[AVN_COLOR_OVERFLOW_INDEX + 1] - color_offsets[AVN_COLOR_OVERFLOW_INDEX];
            const uint32_t need = (n23 + 63u) / 64u;
            if (pipe_dev && (need > ovf_grid_blocks || ovf_grid_blocks > 4 * need + 64)) { ovf_grid_blocks = n23 ? (n23 + n23 / 4 + 64 + 63u) / 64u : 0u; graph_valid = false; }
        }
        // launch grids per colour: the kernels read the live colour ranges from device memory, so a captured grid stays
        // valid while it still covers the colour; grids are captured with 25 % slack and re-captured when outgrown
        for (int c = 0; c < AVN_GRAPH_COLOR_COUNT; ++c) {
            uint32_t cnt = color_offsets[c + 1] - color_offsets[c];
            if (c == AVN_COLOR_OVERFLOW_INDEX) {  // serial kernel: the "grid" is only an on/off flag
                if (cnt && !grid_blocks[c]) { grid_blocks[c] = 8; graph_valid = false; }
                continue;
            }
            uint32_t need = cnt ? color_grid_blocks(cnt) : 0u;
            if (need > grid_blocks[c] || grid_blocks[c] > 4 * need + 64) {
                grid_blocks[c] = cnt ? color_grid_blocks(cnt + cnt / 4 + 64) : 0u;
                graph_valid = false;
            }
        }
    }
    avn_status manifold_handles_upload(const uint32_t* offsets, const uint32_t* ids) override {
        if (!have_bodies) { error = "manifold_handles_upload before bodies_upload"; return AVN_ERR_STATE; }
        if (!offsets || offsets[0] != 0) { error = "manifold_handles_upload: bad offsets"; return AVN_ERR_BAD_ARG; }
        for (int c = 0; c < AVN_GRAPH_COLOR_COUNT; ++c) if (offsets[c] > offsets[c + 1]) { error = "manifold_handles_upload: offsets not monotone"; return AVN_ERR_BAD_ARG; }
        uint32_t M = offsets[AVN_GRAPH_COLOR_COUNT];
        if (M && !ids) return AVN_ERR_BAD_ARG;
        h_m_body1.resize(M); h_m_body2.resize(M);
        for (uint32_t i = 0; i < M; ++i)
            if (ids[i] >= ct.cap || !h_ct_used[ids[i]]) { error = "manifold_handles_upload: no such contact"; return AVN_ERR_STATE; }
        // the host only needs the bodies of the OVERFLOW colour's manifolds (entry lists + level schedule); the incidence of
        // colours 0..22 is built on the device
        // (... and ALL of them when the set is small enough for the island blocks, whose entry lists are host-built)
        for (uint32_t i = island_candidate(M) ? 0u : offsets[AVN_COLOR_OVERFLOW_INDEX]; i < M; ++i) { h_m_body1[i] = h_ct_b1[ids[i]]; h_m_body2[i] = h_ct_b2[ids[i]]; }
        HIPCHK(hipStreamSynchronize(stream));
        avn_status st = ensure_manifold_capacity(M);
        if (st != AVN_OK) return st;
        if (dw.n_manifolds != M) graph_valid = false;
        dw.n_manifolds = M;
        set_color_offsets(offsets);
        hipError_t err;
        if (b_handles.ensure(std::max<size_t>(M, 1) * 4, err)) graph_valid = false;
        if (err != hipSuccess) { error = "hipMalloc failed"; return AVN_ERR_OOM; }
        HIPCHK(hipMemcpy(dw.color_offsets, color_offsets, sizeof color_offsets, hipMemcpyHostToDevice));
        if (M) HIPCHK(hipMemcpy(b_handles.p, ids, (size_t)M * 4, hipMemcpyHostToDevice));
        if (!use_handles) graph_valid = false;
        use_handles = true;
        any_restitution = materials_restitution;
        incidence_dirty = true;
        return AVN_OK;
    }
    avn_status contacts_download(const uint32_t* ids, size_t n, const avn_contacts_out* o) override {
        if (!o || (n && !ids)) return AVN_ERR_BAD_ARG;
        for (size_t i = 0; i < n; ++i)
            if (ids[i] >= ct.cap || (!pipe_dev && !h_ct_used[ids[i]])) { error = "contacts_download: no such contact"; return AVN_ERR_STATE; }   // (device closed loop: liveness is a row flag)
        avn_status st = stage_reserve(al(4 * n) * 4 + al(n) + al(sizeof(T) * 3 * n) + al(sizeof(T) * n) * 2 + al(sizeof(T) * 12 * n) * 2 + al(sizeof(T) * 4 * n) * 4 + al(sizeof(T) * 8 * n) + al(16 * n) * 2 + 4096);
        if (st != AVN_OK) return st;
        const uint32_t* d_id;
        if ((st = stage_in<uint32_t>(ids, n, &d_id)) != AVN_OK) return st;
        ContactsStage<T> s;
        s.flags = o->flags ? stage_alloc<uint32_t>(n) : nullptr; s.point_count = o->point_count ? stage_alloc<uint8_t>(n) : nullptr;
        s.normal = o->normal ? stage_alloc<T>(3 * n) : nullptr; s.friction = o->friction ? stage_alloc<T>(n) : nullptr; s.restitution = o->restitution ? stage_alloc<T>(n) : nullptr;
        s.anchor1 = o->anchor1 ? stage_alloc<T>(12 * n) : nullptr; s.anchor2 = o->anchor2 ? stage_alloc<T>(12 * n) : nullptr;
        s.penetration = o->penetration ? stage_alloc<T>(4 * n) : nullptr; s.normal_speed = o->normal_speed ? stage_alloc<T>(4 * n) : nullptr;
        s.warm_n = o->warm_start_normal_impulse ? stage_alloc<T>(4 * n) : nullptr; s.warm_t = o->warm_start_tangent_impulse ? stage_alloc<T>(8 * n) : nullptr;
        s.normal_impulse = o->normal_impulse ? stage_alloc<T>(4 * n) : nullptr;
        s.feature_id1 = o->feature_id1 ? stage_alloc<uint32_t>(4 * n) : nullptr; s.feature_id2 = o->feature_id2 ? stage_alloc<uint32_t>(4 * n) : nullptr;
        launch_unpack_contacts<T>(ct, d_id, (uint32_t)n, s, stream);
        HIPCHK(hipGetLastError());
        SOUT(o->flags, s.flags, n, uint32_t); SOUT(o->point_count, s.point_count, n, uint8_t); SOUT(o->normal, s.normal, 3 * n, T);
        SOUT(o->friction, s.friction, n, T); SOUT(o->restitution, s.restitution, n, T); SOUT(o->anchor1, s.anchor1, 12 * n, T); SOUT(o->anchor2, s.anchor2, 12 * n, T);
        SOUT(o->penetration, s.penetration, 4 * n, T); SOUT(o->normal_speed, s.normal_speed, 4 * n, T); SOUT(o->warm_start_normal_impulse, s.warm_n, 4 * n, T);
        SOUT(o->warm_start_tangent_impulse, s.warm_t, 8 * n, T); SOUT(o->normal_impulse, s.normal_impulse, 4 * n, T);
        SOUT(o->feature_id1, s.feature_id1, 4 * n, uint32_t); SOUT(o->feature_id2, s.feature_id2, 4 * n, uint32_t);
        HIPCHK(hipStreamSynchronize(stream));
        return AVN_OK;
    }
    avn_status contacts_upload(const uint32_t* ids, size_t n, const avn_contacts_in* in) override {
        if (!in || (n && !ids)) return AVN_ERR_BAD_ARG;
        if (n && (!in->flags || !in->point_count || !in->normal || !in->friction || !in->restitution || !in->anchor1 || !in->anchor2 || !in->penetration || !in->normal_speed ||
                  !in->warm_start_normal_impulse || !in->warm_start_tangent_impulse || !in->normal_impulse || !in->feature_id1 || !in->feature_id2)) {
            error = "contacts_upload: every field of avn_contacts_in is required"; return AVN_ERR_BAD_ARG;
        }
        for (size_t i = 0; i < n; ++i) {
            if (ids[i] >= ct.cap || (!pipe_dev && !h_ct_used[ids[i]])) { error = "contacts_upload: no such contact (avn_contact_pairs_add first)"; return AVN_ERR_STATE; }
            if (in->point_count[i] > AVN_MAX_MANIFOLD_POINTS) { error = "contacts_upload: point_count > 4"; return AVN_ERR_BAD_ARG; }
        }
        if (!n) return AVN_OK;
        avn_status st = stage_reserve(al(4 * n) * 2 + al(n) + al(sizeof(T) * 3 * n) + al(sizeof(T) * n) * 2 + al(sizeof(T) * 12 * n) * 2 + al(sizeof(T) * 4 * n) * 4 + al(sizeof(T) * 8 * n) + al(16 * n) * 2 + 4096);
        if (st != AVN_OK) return st;
        const uint32_t *d_id, *d_flags, *d_f1, *d_f2; const uint8_t* d_pc;
        const T *d_n, *d_fr, *d_re, *d_a1, *d_a2, *d_pen, *d_ns, *d_wn, *d_wt, *d_ni;
        if ((st = stage_in<uint32_t>(ids, n, &d_id)) != AVN_OK || (st = stage_in<uint32_t>(in->flags, n, &d_flags)) != AVN_OK || (st = stage_in<uint8_t>(in->point_count, n, &d_pc)) != AVN_OK ||
            (st = stage_in<T>((const T*)in->normal, 3 * n, &d_n)) != AVN_OK || (st = stage_in<T>((const T*)in->friction, n, &d_fr)) != AVN_OK ||
            (st = stage_in<T>((const T*)in->restitution, n, &d_re)) != AVN_OK || (st = stage_in<T>((const T*)in->anchor1, 12 * n, &d_a1)) != AVN_OK ||
            (st = stage_in<T>((const T*)in->anchor2, 12 * n, &d_a2)) != AVN_OK || (st = stage_in<T>((const T*)in->penetration, 4 * n, &d_pen)) != AVN_OK ||
            (st = stage_in<T>((const T*)in->normal_speed, 4 * n, &d_ns)) != AVN_OK || (st = stage_in<T>((const T*)in->warm_start_normal_impulse, 4 * n, &d_wn)) != AVN_OK ||
            (st = stage_in<T>((const T*)in->warm_start_tangent_impulse, 8 * n, &d_wt)) != AVN_OK || (st = stage_in<T>((const T*)in->normal_impulse, 4 * n, &d_ni)) != AVN_OK ||
            (st = stage_in<uint32_t>(in->feature_id1, 4 * n, &d_f1)) != AVN_OK || (st = stage_in<uint32_t>(in->feature_id2, 4 * n, &d_f2)) != AVN_OK)
            return st;
        ContactsStage<T> s;   // read-only here; the struct is shared with the download direction
        s.flags = const_cast<uint32_t*>(d_flags); s.point_count = const_cast<uint8_t*>(d_pc); s.normal = const_cast<T*>(d_n); s.friction = const_cast<T*>(d_fr);
        s.restitution = const_cast<T*>(d_re); s.anchor1 = const_cast<T*>(d_a1); s.anchor2 = const_cast<T*>(d_a2); s.penetration = const_cast<T*>(d_pen);
        s.normal_speed = const_cast<T*>(d_ns); s.warm_n = const_cast<T*>(d_wn); s.warm_t = const_cast<T*>(d_wt); s.normal_impulse = const_cast<T*>(d_ni);
        s.feature_id1 = const_cast<uint32_t*>(d_f1); s.feature_id2 = const_cast<uint32_t*>(d_f2);
        launch_pack_contacts<T>(ct, d_id, (uint32_t)n, s, stream);
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(stream));   // the staging buffer is reused by the next call
        return AVN_OK;
    }
    avn_status pairs_get(const avn_pair** out, size_t* n) override {
        if (!out || !n) return AVN_ERR_BAD_ARG;
        *out = h_pairs.data();
        *n = h_pairs.size();
        return AVN_OK;
    }
    avn_status aabbs_download(void* mn, void* mx, uint32_t* ents, size_t* n_iv) override {
        size_t C = bp.n_colliders, I = bp.n_intervals;
        avn_status st = stage_reserve(al(sizeof(T) * 3 * C) * 2 + al(4 * I) + 1024);
        if (st != AVN_OK) return st;
        T* a = mn ? stage_alloc<T>(3 * C) : nullptr;
        T* b = mx ? stage_alloc<T>(3 * C) : nullptr;
        uint32_t* e = ents ? stage_alloc<uint32_t>(I) : nullptr;
        launch_unpack_aabbs<T>(bp, a, b, e, stream);
        HIPCHK(hipGetLastError());
        SOUT(mn, a, 3 * C, T); SOUT(mx, b, 3 * C, T); SOUT(ents, e, I, uint32_t);
        HIPCHK(hipStreamSynchronize(stream));
        if (n_iv) *n_iv = I;
        return AVN_OK;
    }
    avn_status dynamic_bounds(double* mn, double* mx) override {
        if (!mn || !mx) return AVN_ERR_BAD_ARG;
        const double inf = std::numeric_limits<double>::infinity();
        for (int k = 0; k < 3; ++k) { mn[k] = inf; mx[k] = -inf; }
        uint32_t nb = (bp.n_colliders + 255) / 256;
        if (!nb) return AVN_OK;
        avn_status st = stage_reserve((size_t)nb * 6 * sizeof(T) + 1024);
        if (st != AVN_OK) return st;
        T* part = stage_alloc<T>((size_t)nb * 6);
        launch_dynamic_bounds<T>(dw, bp, part, stream);
        HIPCHK(hipGetLastError());
        std::vector<T> h((size_t)nb * 6);
        HIPCHK(hipMemcpyAsync(h.data(), part, h.size() * sizeof(T), hipMemcpyDeviceToHost, stream));
        HIPCHK(hipStreamSynchronize(stream));
        for (uint32_t b = 0; b < nb; ++b)
            for (int k = 0; k < 3; ++k) { mn[k] = std::min(mn[k], (double)h[b * 6 + k]); mx[k] = std::max(mx[k], (double)h[b * 6 + 3 + k]); }
        return AVN_OK;
    }
    // batch contact_query::contact_manifolds (k_narrow.hip)
    avn_status contact_manifolds(const avn_shape_pairs* p, const avn_query_manifolds_out* o) override {
        if (!p || !o || (p->count && (!p->shape1 || !p->shape2 || !p->half_extents1 || !p->half_extents2 || !p->position1 || !p->position2 ||
                                      !p->rotation1 || !p->rotation2 || !p->prediction_distance))) { error = "contact_manifolds: null array"; return AVN_ERR_BAD_ARG; }
        size_t n = p->count;
        for (size_t i = 0; i < n; ++i)
            if (p->shape1[i] > AVN_SHAPE_BALL || p->shape2[i] > AVN_SHAPE_BALL) { error = "contact_manifolds: unknown shape"; return AVN_ERR_BAD_ARG; }
        const size_t Q = AVN_MAX_QUERY_POINTS;
        avn_status st = stage_reserve(al(n) * 3 + al(sizeof(T) * 3 * n) * 5 + al(sizeof(T) * 4 * n) * 2 + al(sizeof(T) * n) + al(sizeof(T) * 3 * Q * n) * 3 +
                                      al(sizeof(T) * Q * n) + al(4 * Q * n) * 2 + 64 * 32);
        if (st != AVN_OK) return st;
        QueryStage<T> s;
        std::memset(&s, 0, sizeof s);
        SIN(shape1, p->shape1, n, uint8_t); SIN(shape2, p->shape2, n, uint8_t);
        SIN(half_extents1, p->half_extents1, 3 * n, T); SIN(position1, p->position1, 3 * n, T); SIN(rotation1, p->rotation1, 4 * n, T);
        SIN(half_extents2, p->half_extents2, 3 * n, T); SIN(position2, p->position2, 3 * n, T); SIN(rotation2, p->rotation2, 4 * n, T);
        SIN(prediction, p->prediction_distance, n, T);
        s.point_count = o->point_count ? stage_alloc<uint8_t>(n) : nullptr;
        s.normal = o->normal ? stage_alloc<T>(3 * n) : nullptr;
        s.anchor1 = o->anchor1 ? stage_alloc<T>(3 * Q * n) : nullptr;
        s.anchor2 = o->anchor2 ? stage_alloc<T>(3 * Q * n) : nullptr;
        s.point = o->point ? stage_alloc<T>(3 * Q * n) : nullptr;
        s.penetration = o->penetration ? stage_alloc<T>(Q * n) : nullptr;
        s.feature_id1 = o->feature_id1 ? stage_alloc<uint32_t>(Q * n) : nullptr;
        s.feature_id2 = o->feature_id2 ? stage_alloc<uint32_t>(Q * n) : nullptr;
        launch_contact_manifolds_query<T>(s, (uint32_t)n, stream);
        HIPCHK(hipGetLastError());
        SOUT(o->point_count, s.point_count, n, uint8_t); SOUT(o->normal, s.normal, 3 * n, T);
        SOUT(o->anchor1, s.anchor1, 3 * Q * n, T); SOUT(o->anchor2, s.anchor2, 3 * Q * n, T); SOUT(o->point, s.point, 3 * Q * n, T);
        SOUT(o->penetration, s.penetration, Q * n, T); SOUT(o->feature_id1, s.feature_id1, Q * n, uint32_t); SOUT(o->feature_id2, s.feature_id2, Q * n, uint32_t);
        HIPCHK(hipStreamSynchronize(stream));
        return AVN_OK;
    }
    // ---- standalone closed loop ------------------------------------------------------------------------------------------
    avn_status pipeline_enable(int on) override {
        if (on && !have_colliders) { error = "pipeline_enable: upload bodies and colliders first"; return AVN_ERR_STATE; }
        if (on && pipe_on) return AVN_OK;
        if (pipe_on && pipe_dev) {   // leaving the device closed loop: its rows and keys go with it
            HIPCHK(hipStreamSynchronize(stream));
            if (ct.cap) HIPCHK(hipMemset(ct.meta, 0, (size_t)ct.cap * sizeof(uint4)));
            pipe_dev = false; pipe_on = false;
            contact_keys_live = false; h_live_keys.clear();
            avn_status st = rebuild_pair_set(n_pair_keys);   // only the keys the host uploaded / collected outside the closed loop remain
            if (st != AVN_OK) return st;
            if (!on) return AVN_OK;
        }
        // on == 1: the bookkeeping runs on the device (k_graph.hip); on == 2 or AVN_PIPELINE_HOST=1: host structures (round-1 path, kept for A/B runs)
        const bool want_dev = on == 1 && !getenv("AVN_PIPELINE_HOST");
        if (want_dev) {
            for (uint32_t id = 0; id < pipe_pairs.size(); ++id)
                if (pipe_pairs[id].used) { uint32_t cid = id; avn_status st = contact_pairs_remove(&cid, 1); if (st != AVN_OK) return st; }
            pipe_pairs.clear(); pipe_active.clear(); pipe_handles.clear();
            avn_status st = pipeline_device_reset();
            if (st != AVN_OK) return st;
            pipe_on = true; pipe_dev = true;
            return AVN_OK;
        }
        pipe_on = on != 0;
        // a fresh ContactGraph / ConstraintGraph: rows, ids, colour lists and the broad phase's pair set start empty
        for (uint32_t id = 0; id < pipe_pairs.size(); ++id)
            if (pipe_pairs[id].used) { uint32_t cid = id; avn_status st = contact_pairs_remove(&cid, 1); if (st != AVN_OK) return st; }
        pipe_pairs.clear(); pipe_active.clear(); pipe_handles.clear();
        for (auto& c : pipe_colors) { c.body_bits.clear(); c.handles.clear(); }
        pipe_free_ids = decltype(pipe_free_ids)();
        pipe_next_id = 0; pipe_handles_dirty = true; pipe_active_dirty = true;
        std::memset(&pipe_stats, 0, sizeof pipe_stats);
        std::memset(pipe_offsets, 0, sizeof pipe_offsets);
        return AVN_OK;
    }
    avn_status pipeline_stats_get(avn_pipeline_stats* o) override {
        if (!o) return AVN_ERR_BAD_ARG;
        if (pipe_dev) { pipe_stats.active_pairs = pgm_live; pipe_stats.manifolds = dw.n_manifolds; *o = pipe_stats; return AVN_OK; }
        pipe_stats.active_pairs = (uint32_t)pipe_active.size();
        pipe_stats.manifolds = (uint32_t)pipe_handles.size();
        *o = pipe_stats;
        return AVN_OK;
    }
    avn_status pipeline_handles_get(uint32_t* off, const uint32_t** ids, size_t* n) override {
        if (!off || !ids || !n) return AVN_ERR_BAD_ARG;
        if (pipe_dev) {   // the lists live on the device: fetched on request (tests, inspection)
            HIPCHK(hipStreamSynchronize(stream));
            std::memcpy(pipe_offsets, color_offsets, sizeof pipe_offsets);
            pipe_handles.resize(dw.n_manifolds);
            if (dw.n_manifolds) {
                HIPCHK(hipMemcpyAsync(pipe_handles.data(), b_handles.p, (size_t)dw.n_manifolds * 4, hipMemcpyDeviceToHost, stream));
                HIPCHK(hipStreamSynchronize(stream));
            }
        }
        std::memcpy(off, pipe_offsets, sizeof pipe_offsets);
        *ids = pipe_handles.data(); *n = pipe_handles.size();
        return AVN_OK;
    }
    static bool pbit_get(const std::vector<uint64_t>& s, uint32_t i) { return (i >> 6) < s.size() && ((s[i >> 6] >> (i & 63)) & 1ull); }
    static void pbit_set(std::vector<uint64_t>& s, uint32_t i) { if ((i >> 6) >= s.size()) s.resize((i >> 6) + 1, 0ull); s[i >> 6] |= 1ull << (i & 63); }
    static void pbit_unset(std::vector<uint64_t>& s, uint32_t i) { if ((i >> 6) < s.size()) s[i >> 6] &= ~(1ull << (i & 63)); }
    void pipe_push(uint32_t cid, uint32_t flags) {   // ConstraintGraph::push_manifold (constraint_graph.rs:163-236)
        PipePair& p = pipe_pairs[cid];
        if (p.n_handles) return;  // (one manifold per convex pair)
        const bool s1 = flags & AVN_CP_STATIC1, s2 = flags & AVN_CP_STATIC2;
        const uint32_t b1 = (uint32_t)p.b1, b2 = (uint32_t)p.b2;
        int color = AVN_COLOR_OVERFLOW_INDEX;
        if (!s1 && !s2) {
            for (int i = 0; i < AVN_DYNAMIC_COLOR_COUNT; ++i) {
                PipeColor& c = pipe_colors[i];
                if (pbit_get(c.body_bits, b1) || pbit_get(c.body_bits, b2)) continue;
                pbit_set(c.body_bits, b1); pbit_set(c.body_bits, b2);
                color = i;
                break;
            }
        } else if (!s1 || !s2) {
            const uint32_t body = !s1 ? b1 : b2;
            for (int i = AVN_COLOR_OVERFLOW_INDEX - 1; i >= 1; --i) {
                PipeColor& c = pipe_colors[i];
                if (pbit_get(c.body_bits, body)) continue;
                pbit_set(c.body_bits, body);
                color = i;
                break;
            }
        }
        p.color = (int8_t)color; p.color_pos = (uint32_t)pipe_colors[color].handles.size();
        pipe_colors[color].handles.push_back(cid);
        p.n_handles = 1; pipe_handles_dirty = true; ++pipe_stats.manifolds_pushed;
    }
    void pipe_pop(uint32_t cid) {                      // ConstraintGraph::pop_manifold (:245-296): swap-remove
        PipePair& p = pipe_pairs[cid];
        if (!p.n_handles) return;
        PipeColor& c = pipe_colors[p.color];
        if (p.color != AVN_COLOR_OVERFLOW_INDEX) { pbit_unset(c.body_bits, (uint32_t)p.b1); pbit_unset(c.body_bits, (uint32_t)p.b2); }
        uint32_t moved = c.handles.back();
        c.handles[p.color_pos] = moved; pipe_pairs[moved].color_pos = p.color_pos;
        c.handles.pop_back();
        p.n_handles = 0; p.color = -1; pipe_handles_dirty = true; ++pipe_stats.manifolds_popped;
    }
    avn_status pipeline_step() {
        avn_status st;
        launches = 0;
        HIPCHK(hipEventRecord(ev[0], stream));
        if ((st = update_aabb()) != AVN_OK) return st;
        if ((st = collect_collision_pairs()) != AVN_OK) return st;   // new pairs in h_pairs (emission order)
        HIPCHK(hipEventRecord(ev[1], stream));
        auto t0 = std::chrono::steady_clock::now();
        // ContactGraph::add_edge_and_key_with + IdPool::alloc_id for every new pair, in emission order
        if (!h_pairs.empty()) {
            size_t n = h_pairs.size();
            std::vector<uint32_t> ids(n), c1(n), c2(n), fl(n);
            for (size_t i = 0; i < n; ++i) {
                uint32_t id;
                if (!pipe_free_ids.empty()) { id = pipe_free_ids.top(); pipe_free_ids.pop(); } else id = pipe_next_id++;
                if (id >= pipe_pairs.size()) pipe_pairs.resize(std::max<size_t>((size_t)id + 1, pipe_pairs.size() + pipe_pairs.size() / 2));
                const avn_pair& pr = h_pairs[i];
                PipePair& p = pipe_pairs[id];
                p.c1 = pr.collider1; p.c2 = pr.collider2; p.b1 = pr.body1; p.b2 = pr.body2; p.n_handles = 0; p.used = true;
                p.active_pos = (uint32_t)pipe_active.size();
                pipe_active.push_back(id);
                ids[i] = id; c1[i] = pr.collider1; c2[i] = pr.collider2; fl[i] = pr.flags;
            }
            avn_contact_pairs cp{(uint32_t)n, ids.data(), c1.data(), c2.data(), fl.data()};
            if ((st = contact_pairs_add(&cp)) != AVN_OK) return st;
            pipe_stats.pairs_added += n;
            pipe_active_dirty = true;
        }
        if (pipe_active_dirty) {
            if ((st = active_pairs_set(pipe_active.data(), pipe_active.size())) != AVN_OK) return st;
            pipe_active_dirty = false;
        }
        double host_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        if ((st = narrow_phase()) != AVN_OK) return st;
        t0 = std::chrono::steady_clock::now();
        // the status-change loop of NarrowPhase::update (system_param.rs:141-389), ascending ContactId
        std::vector<uint32_t> removed;
        for (const avn_contact_change& c : h_changes) {
            const uint32_t cid = c.contact_id, flags = c.flags;
            const bool generates = flags & AVN_CP_GENERATE_CONSTRAINTS, touching = flags & AVN_CP_TOUCHING;
            PipePair& p = pipe_pairs[cid];
            if (flags & AVN_CP_DISJOINT_AABB) {
                if (generates) while (p.n_handles) pipe_pop(cid);
                removed.push_back(cid);
            } else if (flags & AVN_CP_STARTED_TOUCHING) {
                if (generates) for (uint32_t k = 0; k < c.manifold_count; ++k) pipe_push(cid, flags);
            } else if (flags & AVN_CP_STOPPED_TOUCHING) {
                if (generates) while (p.n_handles) pipe_pop(cid);
            } else if (touching && (flags & AVN_CP_STARTED_GENERATING_CONSTRAINTS)) {
                for (uint32_t k = 0; k < c.manifold_count; ++k) pipe_push(cid, flags);
            } else if (touching && generates && c.manifold_count_change > 0) {
                for (int32_t k = 0; k < c.manifold_count_change; ++k) pipe_push(cid, flags);
            } else if (touching && generates && c.manifold_count_change < 0) {
                for (int32_t k = 0; k < -c.manifold_count_change; ++k) pipe_pop(cid);
            }
        }
        pipe_stats.last_status_changes = (uint32_t)h_changes.size();
        if (!removed.empty()) {   // ContactGraph::remove_edge_by_id + IdPool::free_id
            if ((st = contact_pairs_remove(removed.data(), removed.size())) != AVN_OK) return st;
            for (uint32_t cid : removed) {
                PipePair& p = pipe_pairs[cid];
                uint32_t last = pipe_active.back();
                pipe_active[p.active_pos] = last; pipe_pairs[last].active_pos = p.active_pos; pipe_active.pop_back();
                p = PipePair();
                pipe_free_ids.push(cid);
            }
            pipe_stats.pairs_removed += removed.size();
            pipe_active_dirty = true;
        }
        if (pipe_handles_dirty) {
            size_t n = 0;
            for (int c = 0; c < AVN_GRAPH_COLOR_COUNT; ++c) { pipe_offsets[c] = (uint32_t)n; n += pipe_colors[c].handles.size(); }
            pipe_offsets[AVN_GRAPH_COLOR_COUNT] = (uint32_t)n;
            pipe_handles.resize(n);
            for (int c = 0; c < AVN_GRAPH_COLOR_COUNT; ++c)
                if (!pipe_colors[c].handles.empty()) std::memcpy(pipe_handles.data() + pipe_offsets[c], pipe_colors[c].handles.data(), pipe_colors[c].handles.size() * 4);
            if ((st = manifold_handles_upload(pipe_offsets, pipe_handles.data())) != AVN_OK) return st;
            pipe_handles_dirty = false;
        }
        pipe_stats.last_overflow_manifolds = pipe_offsets[AVN_COLOR_OVERFLOW_INDEX + 1] - pipe_offsets[AVN_COLOR_OVERFLOW_INDEX];
        pipe_stats.last_host_ms = host_ms + std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        stamp(DG_NP1); dg_np = true;
        if ((st = solver()) != AVN_OK) return st;
        HIPCHK(hipEventRecord(ev[4], stream));
        ev_valid = true;
        last_timers.kernel_launches = launches;
        return AVN_OK;
    }
    // ---- closed loop, bookkeeping on the device -------------------------------------------------------------------------------
    template <class U> avn_status pg_buf(DevBuf& b, size_t count, U** field, bool keep = false) {
        hipError_t err;
        b.ensure(std::max<size_t>(count, 1) * sizeof(U), err, keep, stream);
        if (err != hipSuccess) { error = "hipMalloc failed"; return AVN_ERR_OOM; }
        *field = b.as<U>();
        return AVN_OK;
    }
    // per-row arrays follow CT::cap (contents kept: they are persistent state); per-op scratch is sized for one op per row
    avn_status pg_ensure_rows(uint32_t rows) {
        if (rows <= pg_rows) return AVN_OK;
        HIPCHK(hipStreamSynchronize(stream));
        const uint32_t old = pg_rows;
        avn_status st;
#define PGB(buf, cnt, field, keep) do { if ((st = pg_buf(buf, cnt, &(field), keep)) != AVN_OK) return st; } while (0)
        PGB(b_pg_bodies, rows, pg.bodies, true); PGB(b_pg_color, rows, pg.color, true); PGB(b_pg_lpos, rows, pg.lpos, true);
        PGB(b_pg_free_a, rows, pg.free_ids, true); PGB(b_pg_free_b, rows, pg.free_alt, true);
        {   // colour lists: [24][stride] re-laid out for the new stride
            uint32_t* nl = nullptr;
            if (hipMalloc((void**)&nl, (size_t)AVN_GRAPH_COLOR_COUNT * rows * 4) != hipSuccess) { error = "hipMalloc failed"; return AVN_ERR_OOM; }
            for (int c = 0; c < AVN_GRAPH_COLOR_COUNT && old; ++c)
                if (pgm_len[c]) HIPCHK(hipMemcpy(nl + (size_t)c * rows, pg.lists + (size_t)c * old, (size_t)pgm_len[c] * 4, hipMemcpyDeviceToDevice));
            if (b_pg_lists.p) (void)hipFree(b_pg_lists.p);
            b_pg_lists.p = nl; b_pg_lists.cap = (size_t)AVN_GRAPH_COLOR_COUNT * rows * 4;
            pg.lists = nl; pg.list_stride = rows;
        }
        PGB(b_pg_chg, rows, pg.chg, false); PGB(b_pg_has, rows, pg.has, false); PGB(b_pg_off, rows + 1, pg.off, false);
        PGB(b_pg_op_cid, rows, pg.op_cid, false); PGB(b_pg_op_info, rows, pg.op_info, false); PGB(b_pg_op_bodies, rows, pg.op_bodies, false);
        PGB(b_pg_ekey_a, 2 * (size_t)rows, pg.ekey_a, false); PGB(b_pg_eval_a, 2 * (size_t)rows, pg.eval_a, false);
        PGB(b_pg_ekey_b, 2 * (size_t)rows, pg.ekey_b, false); PGB(b_pg_eval_b, 2 * (size_t)rows, pg.eval_b, false);
        PGB(b_pg_epos, 2 * (size_t)rows, pg.epos, false); PGB(b_pg_popbefore, 2 * (size_t)rows, pg.popbefore, false);
        PGB(b_pg_prevpush, 2 * (size_t)rows, pg.prevpush, false); PGB(b_pg_est, 2 * (size_t)rows, pg.est, false);
        PGB(b_pg_tile_agg, 5 * (size_t)pg_scan_tiles(2 * rows) + 8, pg.tile_agg, false);
        PGB(b_pg_ckey_a, rows, pg.ckey_a, false); PGB(b_pg_cval_a, rows, pg.cval_a, false); PGB(b_pg_ckey_b, rows, pg.ckey_b, false); PGB(b_pg_cval_b, rows, pg.cval_b, false);
        PGB(b_pg_rem_flag, rows, pg.rem_flag, false); PGB(b_pg_rem_off, rows + 1, pg.rem_off, false); PGB(b_pg_rem_ids, rows, pg.rem_ids, false);
        uint32_t* dummy;
        PGB(b_pg_hist, (size_t)256 * radix_blocks(2 * rows) + 256, dummy, false);
        PGB(b_pg_sums, std::max<size_t>(scan_block_sums_needed(256 * radix_blocks(2 * rows)), scan_block_sums_needed(2 * rows)) + 16, dummy, false);
#undef PGB
        pg.rows = rows; pg_rows = rows;
        graph_valid = false;
        return AVN_OK;
    }
    avn_status pipeline_device_reset() {
        HIPCHK(hipStreamSynchronize(stream));
        HIPCHK(hipStreamSynchronize(stream_bp));
        avn_status st;
        hipError_t err;
        b_pg_ctr.ensure(PGC_WORDS * 4, err);
        if (err != hipSuccess) { error = "hipMalloc failed"; return AVN_ERR_OOM; }
        pg.ctr = b_pg_ctr.as<uint32_t>();
        HIPCHK(hipMemset(pg.ctr, 0, PGC_WORDS * 4));
        if ((st = pg_buf(b_pg_bcol, (size_t)cap_bodies + 1, &pg.bcol)) != AVN_OK) return st;
        HIPCHK(hipMemset(pg.bcol, 0, ((size_t)cap_bodies + 1) * 4));
        {   // collider entity -> slot (dense: Entity::index() values are small integers)
            uint32_t max_ent = 0;
            for (uint32_t e : slot_entity) max_ent = std::max(max_ent, e);
            if (max_ent > (1u << 27)) { error = "pipeline_enable: collider entity indices above 2^27 need the host bookkeeping (AVN_PIPELINE_HOST=1)"; return AVN_ERR_CAPACITY; }
            std::vector<uint32_t> e2s((size_t)max_ent + 1, 0u);
            for (uint32_t i = 0; i < slot_entity.size(); ++i) e2s[slot_entity[i]] = i;
            uint32_t* d;
            if ((st = pg_buf(b_pg_ent2slot, e2s.size(), &d)) != AVN_OK) return st;
            HIPCHK(hipMemcpy(d, e2s.data(), e2s.size() * 4, hipMemcpyHostToDevice));
            pg.ent2slot = d;
        }
        if (ct.cap) HIPCHK(hipMemset(ct.meta, 0, (size_t)ct.cap * sizeof(uint4)));
        if ((st = ensure_contact_rows(std::max<uint32_t>(ct.cap, 1024u))) != AVN_OK) return st;
        pg_rows = 0;   // (re)allocate everything for the table's capacity
        std::memset(pgm_len, 0, sizeof pgm_len);
        if ((st = pg_ensure_rows(ct.cap)) != AVN_OK) return st;
        HIPCHK(hipMemset(pg.color, 0xFF, (size_t)pg_rows * 4));
        contact_keys_live = false; h_live_keys.clear();   // (the pair set keeps the keys the host announced: existing pairs stay existing)
        pgm_head = pgm_n_free = pgm_next_id = pgm_live = pgm_tomb = 0;
        std::memset(&pipe_stats, 0, sizeof pipe_stats);
        std::memset(pipe_offsets, 0, sizeof pipe_offsets);
        uint32_t zero[AVN_GRAPH_COLOR_COUNT + 1] = {0};
        dw.n_manifolds = 0;
        set_color_offsets(zero);
        HIPCHK(hipMemcpy(dw.color_offsets, zero, sizeof zero, hipMemcpyHostToDevice));
        use_handles = true; any_restitution = materials_restitution;
        incidence_dirty = true; graph_valid = false;
        if (pin_ctr.ensure(4096) != hipSuccess) { error = "hipHostMalloc failed"; return AVN_ERR_OOM; }
        return AVN_OK;
    }
    static uint32_t bits_for(uint32_t max_value) { uint32_t b = 1; while (b < 32 && (max_value >> b)) ++b; return b; }
    // ContactGraph::pair_set with room for `expect` more keys: rebuilt from the live rows when it would pass half full (tombstones count)
    avn_status pg_pair_set_reserve(uint32_t n_rows_now, uint32_t incoming) {
        const uint64_t need_keys = (uint64_t)n_pair_keys + pgm_live + pgm_tomb + incoming + 16;
        if (bp.pair_set_cap && 2 * need_keys <= bp.pair_set_cap) return AVN_OK;
        uint32_t need = 1024;
        while ((uint64_t)need < 4 * ((uint64_t)n_pair_keys + pgm_live + incoming + 16)) need <<= 1;
        HIPCHK(hipStreamSynchronize(bs));
        hipError_t err;
        b_pair_set.ensure((size_t)need * 8, err);
        if (err != hipSuccess) { error = "hipMalloc failed"; return AVN_ERR_OOM; }
        bp.pair_set = b_pair_set.as<uint64_t>();
        bp.pair_set_cap = need;
        HIPCHK(hipMemsetAsync(bp.pair_set, 0xFF, (size_t)need * 8, bs));
        launch_hs_insert(bp.pair_set, need, b_pair_keys.as<uint64_t>(), n_pair_keys, bs);   // keys announced by the host (avn_existing_pairs_upload, pairs collected outside the loop)
        launch_pg_rebuild_pair_set<T>(ct, bp, n_rows_now, bs);
        HIPCHK(hipGetLastError());
        pgm_tomb = 0;
        graph_valid = false;
        return AVN_OK;
    }
    avn_status pipeline_step_device() {
        avn_status st;
        launches = 0;
        double host_ms = 0;
        auto t0 = std::chrono::steady_clock::now();
        auto lap = [&]() { auto t1 = std::chrono::steady_clock::now(); host_ms += std::chrono::duration<double, std::milli>(t1 - t0).count(); };
        HIPCHK(hipEventRecord(ev[0], stream));
        if ((st = update_aabb()) != AVN_OK) return st;
        if ((st = collect_launch()) != AVN_OK) return st;
        lap();
        // ---- new pairs (emission order) -> ids, rows, pair keys: all on the device; the host reads the pair COUNT ----
        uint32_t total = 0;
        if (collect_pending) {
            collect_pending = false;
            HIPCHK(hipEventSynchronize(ev_counters));
            t0 = std::chrono::steady_clock::now();
            if (h_counters[4]) {   // more long-interval chunks than slots: grow to the requested count and run the count pass again
                if ((st = grow_long_chunks(h_counters[3])) != AVN_OK) return st;
                if ((st = collect_launch()) != AVN_OK) return st;
                collect_pending = false;
                HIPCHK(hipEventSynchronize(ev_counters));
                if (h_counters[4]) { error = "collect_collision_pairs: long-interval chunk capacity exceeded"; return AVN_ERR_CAPACITY; }
            }
            const uint32_t dropped = h_counters[0];
            total = h_counters[2];
            if (total) {
                hipError_t err;
                b_pairs.ensure((size_t)total * sizeof(avn_pair), err);
                if (err != hipSuccess) { error = "pair buffer allocation failed"; return AVN_ERR_OOM; }
                launch_sweep<T>(bp, collect_n, true, sweep_scratch, b_counts.as<uint32_t>(), b_offsets.as<uint32_t>(), b_pairs.as<avn_pair>(), bs);
                launches += 2;
                const uint32_t fresh = total > pgm_n_free ? total - pgm_n_free : 0u;
                if ((st = ensure_contact_rows(pgm_next_id + fresh)) != AVN_OK) return st;
                if ((st = pg_pair_set_reserve(pgm_next_id, total)) != AVN_OK) return st;
                launch_hs_insert_pairs(bp.pair_set, bp.pair_set_cap, b_pairs.as<avn_pair>(), total, bs);   // add_edge_and_key_with: the keys join the pair set
                launch_pg_add_pairs<T>(pg, ct, b_pairs.as<avn_pair>(), total, bs);
                launches += 3;
                HIPCHK(hipGetLastError());
                const uint32_t used = std::min(total, pgm_n_free);
                pgm_head += used; pgm_n_free -= used; pgm_next_id += total - used; pgm_live += total;
                pipe_stats.pairs_added += total;
            }
            bp.n_intervals = collect_n - dropped;
            last_timers.pair_count = total;
        }
        HIPCHK(hipEventRecord(ev[1], stream));
        // ---- narrow phase over every live row; changes numbered in ascending ContactId ----
        const uint32_t n_rows = pgm_next_id;
        uint32_t n_ops = 0, n_rem = 0;
        if (n_rows) {
            launch_narrow_phase_dense<T>(dw, bp, ct, params, n_rows, pg.chg, pg.has, pg.ctr + PGC_N_REM, stream);
            launch_exclusive_scan(pg.has, pg.off, n_rows, b_pg_sums.as<uint32_t>(), pg.ctr + PGC_N_OPS, stream);
            launches += 1 + exclusive_scan_launches(n_rows);
            HIPCHK(hipGetLastError());
            uint32_t* h = (uint32_t*)pin_ctr.p;
            HIPCHK(hipMemcpyAsync(h, pg.ctr + PGC_N_OPS, 3 * 4, hipMemcpyDeviceToHost, stream));   // N_OPS, N_REM, ERROR
            lap();
            HIPCHK(hipStreamSynchronize(stream));
            t0 = std::chrono::steady_clock::now();
            n_ops = h[0]; n_rem = h[1];
            if (h[2]) { error = "device constraint graph: a dataflow wait timed out in the previous step"; return AVN_ERR_STATE; }
        }
        pipe_stats.last_status_changes = n_ops;
        ++pg_dump_step;
        if (n_ops) {
            // ---- the status-change loop: decisions, colours, handle lists ----
            HIPCHK(hipMemsetAsync(pg.ctr + PGC_BUCKET, 0, 32 * 4, stream));
            launch_pg_classify(pg, n_rows, dw.n_bodies, stream);
            uint32_t *ek, *evv;
            launch_radix_sort_bits(pg.ekey_a, pg.eval_a, pg.ekey_b, pg.eval_b, 2 * n_ops, bits_for(dw.n_bodies), b_pg_hist.as<uint32_t>(), b_pg_sums.as<uint32_t>(), &ek, &evv, stream);
            launch_pg_entry_scan(pg, ek, evv, 2 * n_ops, dw.n_bodies, stream);
            launch_pg_color(pg, n_ops, stream);
            launch_pg_apply_masks(pg, ek, evv, 2 * n_ops, dw.n_bodies, stream);
            launch_pg_bucket_keys(pg, n_ops, stream);
            uint32_t *ck, *order;
            launch_radix_sort_bits(pg.ckey_a, pg.cval_a, pg.ckey_b, pg.cval_b, n_ops, 5, b_pg_hist.as<uint32_t>(), b_pg_sums.as<uint32_t>(), &ck, &order, stream);
            launch_pg_replay(pg, order, n_ops, stream);
            launches += 8 + ((bits_for(dw.n_bodies) + 7) / 8 + 1) * radix_pass_launches(2 * n_ops);
            if (n_rem) {   // ContactGraph::remove_edge_by_id + IdPool::free_id
                launch_exclusive_scan(pg.rem_flag, pg.rem_off, n_ops, b_pg_sums.as<uint32_t>(), nullptr, stream);
                launch_pg_remove<T>(pg, ct, bp, n_ops, stream);
                launch_pg_merge_free(pg, pgm_head, pgm_n_free, n_rem, stream);
                HIPCHK(hipMemcpyAsync(pg.free_ids, pg.free_alt, ((size_t)pgm_n_free + n_rem) * 4, hipMemcpyDeviceToDevice, stream));
                pgm_head = 0; pgm_n_free += n_rem; pgm_live -= n_rem; pgm_tomb += n_rem;
                pipe_stats.pairs_removed += n_rem;
                launches += 5;
            }
            HIPCHK(hipGetLastError());
            uint32_t* h = (uint32_t*)pin_ctr.p + 16;
            HIPCHK(hipMemcpyAsync(h, pg.ctr + PGC_LEN, AVN_GRAPH_COLOR_COUNT * 4, hipMemcpyDeviceToHost, stream));
            HIPCHK(hipMemcpyAsync(h + 32, pg.ctr + PGC_ERROR, 4 * 4, hipMemcpyDeviceToHost, stream));   // ERROR, TILE, N_PUSH, N_POP
            lap();
            HIPCHK(hipStreamSynchronize(stream));
            t0 = std::chrono::steady_clock::now();
            if (h[32]) { error = "device constraint graph: the colouring's dataflow wait timed out"; return AVN_ERR_STATE; }
            if (getenv("AVN_PG_REPLAY_STATS")) {
                uint32_t d[96];
                HIPCHK(hipMemcpy(d, pg.ctr + PGC_DBG, sizeof d, hipMemcpyDeviceToHost));
                std::fprintf(stderr, "[avn replay] colour: ops/iterations/serial/reloads:");
                for (int c = 0; c < AVN_GRAPH_COLOR_COUNT; ++c) if (d[72 + c]) std::fprintf(stderr, " %d:%u/%u/%u/%u", c, d[72 + c], d[c], d[24 + c], d[48 + c]);
                std::fprintf(stderr, "\n");
            }
            if (const char* dir = getenv("AVN_PG_DUMP")) {   // debugging aid (tools/debug_pg.py): this step's ops as the device saw them
                std::vector<uint32_t> a(n_ops), b(n_ops), o(n_ops), cnt(32);
                std::vector<int2> bd(n_ops);
                HIPCHK(hipMemcpy(a.data(), pg.op_cid, (size_t)n_ops * 4, hipMemcpyDeviceToHost));
                HIPCHK(hipMemcpy(b.data(), pg.op_info, (size_t)n_ops * 4, hipMemcpyDeviceToHost));
                HIPCHK(hipMemcpy(bd.data(), pg.op_bodies, (size_t)n_ops * 8, hipMemcpyDeviceToHost));
                HIPCHK(hipMemcpy(o.data(), order, (size_t)n_ops * 4, hipMemcpyDeviceToHost));
                HIPCHK(hipMemcpy(cnt.data(), pg.ctr + PGC_BUCKET, 32 * 4, hipMemcpyDeviceToHost));
                char path[512];
                std::snprintf(path, sizeof path, "%s/step_%04llu.bin", dir, (unsigned long long)pg_dump_step);
                if (FILE* f = std::fopen(path, "wb")) {
                    uint32_t hdr[4] = {n_ops, n_rem, 0, 0};
                    std::fwrite(hdr, 4, 4, f); std::fwrite(a.data(), 4, n_ops, f); std::fwrite(b.data(), 4, n_ops, f); std::fwrite(bd.data(), 8, n_ops, f);
                    std::fwrite(o.data(), 4, n_ops, f); std::fwrite(cnt.data(), 4, 32, f);
                    std::fclose(f);
                }
            }
            pipe_stats.manifolds_pushed = h[34]; pipe_stats.manifolds_popped = h[35];
            uint32_t offs[AVN_GRAPH_COLOR_COUNT + 1];
            uint32_t M = 0;
            for (int c = 0; c < AVN_GRAPH_COLOR_COUNT; ++c) { pgm_len[c] = h[c]; offs[c] = M; M += h[c]; }
            offs[AVN_GRAPH_COLOR_COUNT] = M;
            if ((st = ensure_manifold_capacity(M)) != AVN_OK) return st;
            if ((dw.n_manifolds == 0) != (M == 0)) graph_valid = false;   // (no captured kernel reads DW::n_manifolds; only "any manifolds at all" shapes the substep)
            dw.n_manifolds = M;
            set_color_offsets(offs);
            hipError_t err;
            if (b_handles.ensure(std::max<size_t>(M, 1) * 4, err)) graph_valid = false;
            if (err != hipSuccess) { error = "hipMalloc failed"; return AVN_ERR_OOM; }
            launch_pg_build_handles(pg, b_handles.as<uint32_t>(), dw.color_offsets, M, stream);
            ++launches;
            HIPCHK(hipGetLastError());
            incidence_dirty = true;
        }
        pipe_stats.last_overflow_manifolds = color_offsets[AVN_COLOR_OVERFLOW_INDEX + 1] - color_offsets[AVN_COLOR_OVERFLOW_INDEX];
        lap();
        pipe_stats.last_host_ms = host_ms;
        stamp(DG_NP1); dg_np = true;
        if ((st = solver()) != AVN_OK) return st;
        HIPCHK(hipEventRecord(ev[4], stream));
        ev_valid = true;
        last_timers.kernel_launches = launches;
        return AVN_OK;
    }
    // the overflow colour's CSR + ranks, and the slot table of the other colours, from the gathered manifold arrays (all on the device)
    avn_status rebuild_incidence_device() {
        const uint32_t N = dw.n_bodies, M = dw.n_manifolds;
        incidence_dirty = false;
        island_mode = false; islands_dirty = false;
        if (M == 0) return AVN_OK;
        hipError_t err;
        bool moved = b_inc_slot.ensure((size_t)AVN_COLOR_OVERFLOW_INDEX * cap_bodies * sizeof(uint32_t), err);
        if (err != hipSuccess) { error = "hipMalloc failed"; return AVN_ERR_OOM; }
        if (moved || dw.inc_stride != cap_bodies) graph_valid = false;
        dw.inc_slot = b_inc_slot.as<uint32_t>(); dw.inc_stride = cap_bodies;
        slots_dirty = true;
        const uint32_t o0 = color_offsets[AVN_COLOR_OVERFLOW_INDEX], n23 = color_offsets[AVN_COLOR_OVERFLOW_INDEX + 1] - o0;
        moved = b_inc_off.ensure(((size_t)N + 2) * 4, err);
        if (err != hipSuccess) { error = "hipMalloc failed"; return AVN_ERR_OOM; }
        if (n23 > pg_ovf_cap) {
            HIPCHK(hipStreamSynchronize(stream));
            const size_t c = std::max<size_t>(2 * (size_t)n23 + 1024, (size_t)pg_ovf_cap * 3);
            for (DevBuf* b : {&b_inc_ent, &b_ovf_keys_a, &b_ovf_vals_a, &b_ovf_keys_b, &b_ovf_vals_b, &b_ovf_rank}) {
                b->ensure(c * 4, err);
                if (err != hipSuccess) { error = "hipMalloc failed"; return AVN_ERR_OOM; }
            }
            pg_ovf_cap = (uint32_t)(c / 2);
            moved = true;
        }
        if (b_inc_ent.cap == 0) { b_inc_ent.ensure(1024, err); b_ovf_rank.ensure(1024, err); moved = true; }
        if (b_ovf_ticket.ensure(((size_t)cap_bodies + 1) * 4, err)) moved = true;
        if (err != hipSuccess) { error = "hipMalloc failed"; return AVN_ERR_OOM; }
        if (moved || !dw.inc_off) graph_valid = false;
        dw.inc_off = b_inc_off.as<uint32_t>(); dw.inc_ent = b_inc_ent.as<uint32_t>();
        islands_dirty = island_candidate(M) && dw.n_joints == 0;
        return AVN_OK;
    }
    // after k_gather_manifolds (the CSR reads DW::m_bodies of the overflow range)
    void overflow_csr_device() {
        const uint32_t o0 = color_offsets[AVN_COLOR_OVERFLOW_INDEX], n23 = color_offsets[AVN_COLOR_OVERFLOW_INDEX + 1] - o0;
        uint32_t *k = b_ovf_keys_a.as<uint32_t>(), *v = b_ovf_vals_a.as<uint32_t>();
        if (n23) {
            launch_ovf_entries<T>(dw, o0, n23, k, v, stream);
            launch_radix_sort_bits(k, v, b_ovf_keys_b.as<uint32_t>(), b_ovf_vals_b.as<uint32_t>(), 2 * n23, bits_for(dw.n_bodies), b_pg_hist.as<uint32_t>(), b_pg_sums.as<uint32_t>(), &k, &v, stream);
            launches += 1 + ((bits_for(dw.n_bodies) + 7) / 8) * radix_pass_launches(2 * n23);
        }
        launch_ovf_csr<T>(dw, o0, n23, k, v, b_inc_off.as<uint32_t>(), b_inc_ent.as<uint32_t>(), b_ovf_rank.as<uint32_t>(), stream);
        launches += 2;
    }
    avn_status update_aabb() {
        launch_update_aabb<T>(dw, bp, params, bs);
        ++launches;
        HIPCHK(hipGetLastError());
        return AVN_OK;
    }
    // COLLECT_COLLISION_PAIRS is split in two so that avn_step can overlap the host round trip of the pair count with the
    // solver launches: collect_launch() enqueues sort + ranges + count pass + scan and an async read-back of the counters
    // into pinned memory (event-tracked); collect_finish() waits for THAT copy only, and runs the emit pass when new pairs exist.
    uint32_t* h_counters = nullptr;  // pinned: [dropped, unsorted, total, long chunks, long overflow]
    hipEvent_t ev_counters = nullptr;
    uint32_t collect_n = 0;
    bool collect_pending = false;
    avn_status collect_launch() {
        uint32_t n = bp.n_intervals;
        h_pairs.clear();
        last_timers.pair_count = 0;
        collect_n = n;
        collect_pending = false;
        if (n == 0) return AVN_OK;
        if (n > (1u << 26)) { error = "collect_collision_pairs: more than 2^26 intervals"; return AVN_ERR_CAPACITY; }
        if (!h_counters) {
            HIPCHK(hipHostMalloc((void**)&h_counters, 8 * sizeof(uint32_t), hipHostMallocDefault));
            HIPCHK(hipEventCreateWithFlags(&ev_counters, hipEventDisableTiming));
        }
        uint32_t* misc = b_misc.as<uint32_t>();
        uint32_t* d_dropped = misc + 33;   // [33] dropped, [34] unsorted
        uint32_t* d_total = misc + 35;
        sweep_scratch.n_long = misc + 36;  // [36] chunks, [37] overflow
        Key* keys_a = b_keys_a.as<Key>(); Key* keys_b = b_keys_b.as<Key>();
        uint32_t* vals_a = b_vals_a.as<uint32_t>(); uint32_t* vals_b = b_vals_b.as<uint32_t>();
        launch_interval_keys<T>(dw, bp, keys_a, vals_a, d_dropped, bs);
        launch_radix_sort<Key>(keys_a, vals_a, keys_b, vals_b, n, b_hist.as<uint32_t>(), b_block_sums.as<uint32_t>(), d_dropped + 1, bs);
        launch_gather_sorted<T>(dw, bp, vals_a, n, bs);
        launch_sweep_ranges<T>(bp, n, sweep_scratch, bs);
        launch_sweep<T>(bp, n, false, sweep_scratch, b_counts.as<uint32_t>(), nullptr, nullptr, bs);
        launch_exclusive_scan(b_counts.as<uint32_t>(), b_offsets.as<uint32_t>(), n * sweep_count_slots(), b_block_sums.as<uint32_t>(), d_total, bs);
        launches += 3 + radix_sort_launches(n, (uint32_t)sizeof(Key)) + 4 + exclusive_scan_launches(n * sweep_count_slots());
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(h_counters, d_dropped, 5 * sizeof(uint32_t), hipMemcpyDeviceToHost, bs));
        HIPCHK(hipEventRecord(ev_counters, bs));
        collect_pending = true;
        return AVN_OK;
    }
    avn_status grow_long_chunks(uint32_t chunks_needed) {
        HIPCHK(hipStreamSynchronize(bs));
        const size_t lcap = (size_t)chunks_needed + chunks_needed / 4 + 65536;
        bool moved = false;
        uint8_t* dummy_b; uint32_t* dummy_u;
        GROW(b_long_items, lcap * sweep_long_item_bytes(), dummy_b);
        GROW(b_long_counts, lcap, dummy_u); GROW(b_long_off, lcap, dummy_u);
        sweep_scratch.long_items = b_long_items.p; sweep_scratch.long_counts = b_long_counts.as<uint32_t>();
        sweep_scratch.long_off = b_long_off.as<uint32_t>(); sweep_scratch.long_cap = (uint32_t)lcap;
        return AVN_OK;
    }
    avn_status collect_finish() {
        if (!collect_pending) return AVN_OK;
        collect_pending = false;
        uint32_t n = collect_n;
        HIPCHK(hipEventSynchronize(ev_counters));
        if (h_counters[4]) {   // more long-interval chunks than slots: grow to the requested count and run the count pass again
            avn_status st = grow_long_chunks(h_counters[3]);
            if (st != AVN_OK) return st;
            if ((st = collect_launch()) != AVN_OK) return st;
            collect_pending = false;
            HIPCHK(hipEventSynchronize(ev_counters));
            if (h_counters[4]) { error = "collect_collision_pairs: long-interval chunk capacity exceeded"; return AVN_ERR_CAPACITY; }
        }
        uint32_t dropped = h_counters[0], total = h_counters[2];
        if (total) {
            hipError_t err;
            b_pairs.ensure((size_t)total * sizeof(avn_pair), err);
            if (err != hipSuccess) { error = "pair buffer allocation failed"; return AVN_ERR_OOM; }
            launch_sweep<T>(bp, n, true, sweep_scratch, b_counts.as<uint32_t>(), b_offsets.as<uint32_t>(), b_pairs.as<avn_pair>(), bs);
            launches += 2;
            HIPCHK(hipGetLastError());
            h_pairs.resize(total);
            HIPCHK(hipMemcpyAsync(h_pairs.data(), b_pairs.p, (size_t)total * sizeof(avn_pair), hipMemcpyDeviceToHost, bs));
            // add_edge_and_key_with (reference contact_graph.rs:521-566): the new keys join the pair set
            HIPCHK(hipStreamSynchronize(bs));
            std::vector<uint64_t> nk(total);
            for (uint32_t i = 0; i < total; ++i) { uint32_t a = h_pairs[i].collider1, b = h_pairs[i].collider2; nk[i] = a < b ? ((uint64_t)a << 32) | b : ((uint64_t)b << 32) | a; }
            if (contact_keys_live) h_live_keys.insert(nk.begin(), nk.end());
            b_pair_keys.ensure(((size_t)n_pair_keys + total) * 8, err, true, bs);
            if (err != hipSuccess) { error = "pair key list allocation failed"; return AVN_ERR_OOM; }
            HIPCHK(hipMemcpyAsync(b_pair_keys.as<uint64_t>() + n_pair_keys, nk.data(), (size_t)total * 8, hipMemcpyHostToDevice, bs));
            n_pair_keys += total;
            if (bp.pair_set_cap < 2 * (n_pair_keys + 16)) { avn_status st = rebuild_pair_set(n_pair_keys + n_pair_keys / 2); if (st != AVN_OK) return st; }
            else { launch_hs_insert(bp.pair_set, bp.pair_set_cap, b_pair_keys.as<uint64_t>() + (n_pair_keys - total), total, bs); HIPCHK(hipGetLastError()); HIPCHK(hipStreamSynchronize(bs)); }
        }
        bp.n_intervals = n - dropped;  // dropped intervals were sorted to the end
        last_timers.pair_count = total;
        return AVN_OK;
    }
    avn_status collect_collision_pairs() {
        avn_status st = collect_launch();
        if (st != AVN_OK) return st;
        return collect_finish();
    }

    // ---- systems -------------------------------------------------------------------------------------------
    avn_status need_bodies() { if (!have_bodies) { error = "no bodies uploaded"; return AVN_ERR_STATE; } return AVN_OK; }
    void prepare_solver_bodies() { launch_prepare_solver_bodies<T>(dw, stream); ++launches; }
    void prepare_joints() { if (dw.n_joints) { launch_prepare_joints<T>(dw, stream); ++launches; } }
    void prepare_contact_constraints() {
        // GraphColor::manifold_handles indirection (plugin.rs:389-398): the colours' manifolds are fetched from the contact table
        if (use_handles && dw.n_manifolds) { launch_gather_manifolds<T>(dw, bp, ct, b_handles.as<uint32_t>(), stream); ++launches; }
        if (pipe_dev && ovf_csr_dirty && dw.n_manifolds) { overflow_csr_device(); ovf_csr_dirty = false; }
        launch_prepare_contact_constraints<T>(dw, params, stream); ++launches;
    }
    void store_contact_impulses() {
        launch_store_contact_impulses<T>(dw, stream); ++launches;
        if (use_handles && dw.n_manifolds) { launch_scatter_impulses<T>(dw, ct, b_handles.as<uint32_t>(), stream); ++launches; }
    }
    void pre_process_velocity_increments() { launch_pre_process_increments<T>(dw, params, stream); ++launches; }
    void integrate_velocities() { launch_integrate_velocities<T>(dw, params, stream); ++launches; }
    // warm start of ALL colours in one body-centric launch; `fused` also runs integrate_velocities for the body first
    void warm_start(bool fused) {
        if (dw.n_manifolds) {
            if (slots_dirty && dw.inc_slot) { launch_build_incidence_slots<T>(dw, stream); launches += 2; slots_dirty = false; }  // (normally done by prepare)
            launch_body_warm_start<T>(dw, params, fused, stream); ++launches;
        }
        else if (fused) integrate_velocities();
    }
    void integrate_positions() { launch_integrate_positions<T>(dw, params, stream); ++launches; }
    void contact_pass(int pass) {
        if (!dw.n_manifolds) return;
        if (bias_skeleton && pass == PASS_SOLVE_BIAS) pass = PASS_MEMORY_SKELETON;   // AVN_BIAS_SKELETON=1: measurement aid, state unchanged
        if (pipe_dev) {   // overflow colour first (one dataflow launch), then colours 0..22
            // (launched whenever a grid is captured for it, whatever the colour's current population: the captured graph must not
            //  depend on the step's counts; an empty colour costs one launch of idle lanes)
            if (ovf_grid_blocks && ovf_epoch < PGC_OVF_TILES) {
                OverflowFlow of{b_ovf_rank.as<uint32_t>(), b_ovf_ticket.as<uint32_t>(), pg.ctr + PGC_OVF_TILE, pg.ctr + PGC_ERROR};
                launch_overflow_flow<T>(dw, params, pass, of, ovf_epoch, ovf_grid_blocks, stream);
                ++ovf_epoch; ++launches;
            }
            uint32_t gb[AVN_GRAPH_COLOR_COUNT];
            std::memcpy(gb, grid_blocks, sizeof gb);
            gb[AVN_COLOR_OVERFLOW_INDEX] = 0;
            OverflowSchedule none{0, nullptr, nullptr, nullptr, nullptr, nullptr, 0};
            launches += launch_contact_pass<T>(dw, params, pass, gb, nullptr, none, stream);
            return;
        }
        OverflowSchedule ovf{sched_overflow.n_components, sched_overflow.d_comp_level_begin.as<uint32_t>(), sched_overflow.d_level_offsets.as<uint32_t>(),
                             sched_overflow.d_order.as<uint32_t>(), nullptr, nullptr, 0};
        if (sched_overflow.gorder.size() > overflow_level_threshold) {  // a big overflow colour: one device-wide launch per level instead of one workgroup per component
            ovf.gorder = sched_overflow.d_gorder.as<uint32_t>();
            ovf.glevel_offsets = sched_overflow.glevel_offsets.data();
            ovf.n_glevels = (uint32_t)sched_overflow.glevel_offsets.size() - 1;
        }
        launches += launch_contact_pass<T>(dw, params, pass, grid_blocks, use_handles ? nullptr : color_offsets, ovf, stream);
    }
    // The reference runs the snapshot and the velocity projection over ALL active bodies whenever XpbdSolverPlugin is
    // installed (xpbd/plugin.rs:61-76,192-240).  With no joints the projection adds 2 * (dq * conj(dq)).xyz / h:
    //  - f32 (glam's SIMD `Quat`, pairwise sums): every xyz component cancels exactly, e.g. y = (-wy + xz) + (yw - zx) is
    //    a + (-a) = 0, so the two systems are exact no-ops (up to the sign of a zero) and are skipped;
    //  - f64 (scalar `DQuat`, left-to-right sums): y = ((-wy + xz) + yw) - zx leaves a rounding residual of order
    //    ulp(wy), so the reference really perturbs omega every substep — replicate, don't "fix" (found by the cfg5 test).
    bool xpbd_body_passes_needed() const { return dw.n_joints != 0 || sizeof(T) == 8; }
    void xpbd_solve(bool snapshot) {
        if (snapshot && xpbd_body_passes_needed()) { launch_xpbd_snapshot<T>(dw, stream); ++launches; }
        if (!dw.n_joints) return;
        launch_joint_schedule<T>(dw, params, 0, (uint32_t)sched_solve.n_components, sched_solve.d_comp_level_begin.as<uint32_t>(),
                                 sched_solve.d_level_offsets.as<uint32_t>(), sched_solve.d_rec.as<int4>(), stream);
        ++launches;
    }
    void xpbd_velocity_projection() { if (xpbd_body_passes_needed()) { launch_xpbd_velocity_projection<T>(dw, params, stream); ++launches; } }
    void joint_damping() {
        if (!any_damped || !sched_damp.n_components) return;
        if (sched_damp.touches_dummy) {
            // reset the two virtual SolverBody::DUMMY slots (all-zero bit pattern = zero velocities)
            (void)hipMemsetAsync(&dw.sb_lin[dw.n_bodies], 0, 2 * DUMMY_SLOTS * sizeof(V), stream);  // DUMMY_SLOTS bodies x (lin | ang) slot
        }
        launch_joint_schedule<T>(dw, params, 1, (uint32_t)sched_damp.n_components, sched_damp.d_comp_level_begin.as<uint32_t>(),
                                 sched_damp.d_level_offsets.as<uint32_t>(), sched_damp.d_rec.as<int4>(), stream);
        ++launches;
    }
    void substep() {  // SubstepSchedule order (reference solver/schedule.rs:59-69, xpbd/plugin.rs:30-40)
        const bool dg = !cfg.use_graph && substep_index < DG_SUBSTEPS;   // (events captured into a hipGraph cannot be read back)
        hipEvent_t* de = ev_dgs + (size_t)substep_index * DG_PER;
        if (dg) (void)hipEventRecord(de[0], stream);
        warm_start(true);  // integrate_velocities + warm_start
        if (dg) (void)hipEventRecord(de[1], stream);
        // measurement hook: the dominant kernel's launches inside the step.  Direct launches only: events recorded as nodes of a
        // captured graph cannot be read back with hipEventElapsedTime on this runtime (hipErrorInvalidHandle).
        const bool timed = substep_index < BIAS_EV && dw.n_manifolds != 0 && !cfg.use_graph;
        if (timed) { (void)hipEventRecord(ev_bias[2 * substep_index], stream); bias_launches = launches; }
        for (uint32_t it = 0; it < cfg.solver_iterations; ++it) contact_pass(PASS_SOLVE_BIAS);
        if (timed) { (void)hipEventRecord(ev_bias[2 * substep_index + 1], stream); bias_launches = launches - bias_launches; bias_timed = substep_index + 1; }
        ++substep_index;
        if (dg) (void)hipEventRecord(de[2], stream);
        integrate_positions();
        if (dg) (void)hipEventRecord(de[3], stream);
        for (uint32_t it = 0; it < cfg.solver_iterations; ++it) contact_pass(PASS_SOLVE_RELAX);
        for (uint32_t it = 0; it < cfg.solver_iterations; ++it) xpbd_solve(it == 0);
        xpbd_velocity_projection();
        joint_damping();
        if (dg) { (void)hipEventRecord(de[4], stream); dg_substeps = substep_index; }
    }
    bool islands_active() const { return island_mode && dw.n_joints == 0 && dw.n_manifolds != 0 && !halo_on; }
    avn_status run_substeps() {
        substep_index = 0;
        bias_timed = 0;
        dg_substeps = 0;
        if (halo_on) {   // level-2 sharding: direct launches, the exchanges are RCCL calls on the same stream
            if (!comm.handle) { error = "a halo plan is set but no communicator: call avn_comm_init, or drive the colours through avn_run_color_pass"; return AVN_ERR_STATE; }
            return level2_substeps();
        }
        if constexpr (sizeof(T) == 4) {
            if (islands_active()) {   // every substep of every island block in ONE launch (k_island_substeps)
                launch_island_substeps(dw, params, islands, cfg.substeps, cfg.solver_iterations, stream); ++launches;
                // (device closed loop: the restitution pass after the loop still runs colour by colour; its overflow pass starts a fresh epoch count)
                if (pipe_dev && ovf_grid_blocks) { launch_overflow_reset(b_ovf_ticket.as<uint32_t>(), dw.n_bodies + 1, pg.ctr + PGC_OVF_TILE, PGC_OVF_TILES, stream); ++launches; }
                ovf_epoch = 0;
                return AVN_OK;
            }
        }
        // the body-centric warm start's slot table (not needed by the island blocks); outside the capture below
        if (slots_dirty && dw.n_manifolds && dw.inc_slot) { launch_build_incidence_slots<T>(dw, stream); launches += 2; slots_dirty = false; }
        const bool flow = pipe_dev && dw.n_manifolds && ovf_grid_blocks;
        if (flow && (uint64_t)cfg.substeps * 2 * cfg.solver_iterations + 2 > PGC_OVF_TILES) { error = "device closed loop: too many contact passes per step for the overflow tickets"; return AVN_ERR_CAPACITY; }
        if (!cfg.use_graph) {
            if (flow) { launch_overflow_reset(b_ovf_ticket.as<uint32_t>(), dw.n_bodies + 1, pg.ctr + PGC_OVF_TILE, PGC_OVF_TILES, stream); ++launches; }
            ovf_epoch = 0;
            for (uint32_t s = 0; s < cfg.substeps; ++s) substep();
            ovf_epoch_after_substeps = ovf_epoch;
            return AVN_OK;
        }
        if (!graph_valid) {
            if (getenv("AVN_DBG_CAPTURE")) std::fprintf(stderr, "[avn] substep graph re-captured (M %u, overflow grid %u)\n", dw.n_manifolds, ovf_grid_blocks);
            drop_graph();
            uint32_t before = launches;
            HIPCHK(hipStreamBeginCapture(stream, hipStreamCaptureModeThreadLocal));
            // the overflow passes' tickets and tile counters restart with every step (a kernel node, replayed first)
            if (flow) { launch_overflow_reset(b_ovf_ticket.as<uint32_t>(), dw.n_bodies + 1, pg.ctr + PGC_OVF_TILE, PGC_OVF_TILES, stream); ++launches; }
            ovf_epoch = 0;
            for (uint32_t s = 0; s < cfg.substeps; ++s) substep();
            ovf_epoch_after_substeps = ovf_epoch;
            // whatever went wrong inside the capture, the stream must leave capture mode and the partial graph must not survive
            hipError_t ce = hipStreamEndCapture(stream, &graph);
            graph_launches = launches - before;
            launches = before;
            if (ce == hipSuccess) ce = hipGraphInstantiate(&graph_exec, graph, nullptr, nullptr, 0);
            if (ce != hipSuccess) {
                (void)hipGetLastError();
                drop_graph();
                error = std::string("substep graph capture failed: ") + hipGetErrorName(ce);
                return AVN_ERR_HIP;
            }
            graph_valid = true;
        }
        HIPCHK(hipGraphLaunch(graph_exec, stream));
        launches += graph_launches;
        ovf_epoch = ovf_epoch_after_substeps;   // (the restitution pass after the loop continues the step's epochs)
        return AVN_OK;
    }
    uint32_t graph_launches = 0;
    avn_status solver_front() {   // everything that only READS the rigid-body components
        avn_status st = need_bodies();
        if (st != AVN_OK) return st;
        if ((st = rebuild_joint_schedules()) != AVN_OK) return st;
        if ((st = rebuild_incidence()) != AVN_OK) return st;
        prepare_solver_bodies();
        prepare_joints();
        prepare_contact_constraints();
        stamp(DG_PREP1);
        pre_process_velocity_increments();
        stamp(DG_INC1);
        // host work that only the substep loop needs, done while the prepare kernels above run
        if (islands_dirty) {
            islands_dirty = false;
            if (pipe_dev) {   // the island builder is host code: fetch the (small) gathered body pairs
                const uint32_t M = dw.n_manifolds;
                std::vector<int2> mb(M);
                HIPCHK(hipMemcpyAsync(mb.data(), dw.m_bodies, (size_t)M * sizeof(int2), hipMemcpyDeviceToHost, stream));
                HIPCHK(hipStreamSynchronize(stream));
                h_m_body1.resize(M); h_m_body2.resize(M);
                for (uint32_t m = 0; m < M; ++m) { h_m_body1[m] = mb[m].x; h_m_body2[m] = mb[m].y; }
            }
            if ((st = rebuild_island_blocks()) != AVN_OK) return st;
        }
        HIPCHK(hipEventRecord(ev[2], stream));
        if ((st = run_substeps()) != AVN_OK) return st;
        HIPCHK(hipEventRecord(ev[3], stream));
        stamp(DG_SUB1);
        launch_clear_increments<T>(dw, stream); ++launches;
        // restitution == 0 everywhere: every manifold would early-out.  (Level 2: the exchanges are collective and `any_restitution` is a
        // per-rank fact, so the pass always runs there; a rank without restitution launches kernels whose lanes all early-out.)
        if (halo_on) { if ((st = level2_pass(PASS_RESTITUTION_)) != AVN_OK) return st; }
        else if (any_restitution) contact_pass(PASS_RESTITUTION_);
        stamp(DG_REST1);
        HIPCHK(hipGetLastError());
        return AVN_OK;
    }
    avn_status solver_back() {    // the write-back into Position / Rotation / velocities and the ContactGraph
        launch_writeback_solver_bodies<T>(dw, stream); ++launches;
        if (dw.n_joints) { launch_writeback_joint_forces<T>(dw, params, stream); ++launches; }
        stamp(DG_FIN1);
        store_contact_impulses();
        stamp(DG_STORE1);
        HIPCHK(hipGetLastError());
        return AVN_OK;
    }
    avn_status solver() {
        avn_status st = solver_front();
        if (st != AVN_OK) return st;
        return solver_back();
    }
    avn_status dispatch_system(avn_system sys) {
        avn_status st = AVN_OK;
        switch (sys) {
            case AVN_SYS_UPDATE_AABB: if ((st = update_aabb()) != AVN_OK) return st; break;
            case AVN_SYS_COLLECT_COLLISION_PAIRS: if ((st = collect_collision_pairs()) != AVN_OK) return st; break;
            case AVN_SYS_PREPARE_SOLVER_BODIES: prepare_solver_bodies(); break;
            case AVN_SYS_PREPARE_JOINTS: prepare_joints(); break;
            case AVN_SYS_PREPARE_CONTACT_CONSTRAINTS: prepare_contact_constraints(); break;
            case AVN_SYS_PRE_PROCESS_VELOCITY_INCREMENTS: pre_process_velocity_increments(); break;
            case AVN_SYS_INTEGRATE_VELOCITIES: integrate_velocities(); break;
            case AVN_SYS_WARM_START: warm_start(false); break;
            case AVN_SYS_SOLVE_CONTACTS_BIAS: contact_pass(PASS_SOLVE_BIAS); break;
            case AVN_SYS_INTEGRATE_POSITIONS: integrate_positions(); break;
            case AVN_SYS_SOLVE_CONTACTS_RELAX: contact_pass(PASS_SOLVE_RELAX); break;
            case AVN_SYS_XPBD_SOLVE: xpbd_solve(true); break;
            case AVN_SYS_XPBD_VELOCITY_PROJECTION: xpbd_velocity_projection(); break;
            case AVN_SYS_JOINT_DAMPING: joint_damping(); break;
            case AVN_SYS_CLEAR_VELOCITY_INCREMENTS: launch_clear_increments<T>(dw, stream); ++launches; break;
            case AVN_SYS_SOLVE_RESTITUTION: if (any_restitution) contact_pass(PASS_RESTITUTION_); break;
            case AVN_SYS_WRITEBACK_SOLVER_BODIES:
                launch_writeback_solver_bodies<T>(dw, stream); ++launches;
                if (dw.n_joints) { launch_writeback_joint_forces<T>(dw, params, stream); ++launches; }
                break;
            case AVN_SYS_STORE_CONTACT_IMPULSES: store_contact_impulses(); break;
            case AVN_SYS_NARROW_PHASE: if ((st = narrow_phase()) != AVN_OK) return st; break;
            case AVN_SYS_SUBSTEP: substep(); break;
            case AVN_SYS_SOLVER: {
                HIPCHK(hipEventRecord(ev[0], stream)); HIPCHK(hipEventRecord(ev[1], stream));
                if ((st = solver()) != AVN_OK) return st;
                HIPCHK(hipEventRecord(ev[4], stream));
                ev_valid = true;
                break;
            }
            default: error = "run_system: unknown system"; return AVN_ERR_BAD_ARG;
        }
        HIPCHK(hipGetLastError());
        return AVN_OK;
    }
    // single systems run outside avn_step: the dataflow passes' per-step state has to be fresh
    avn_status flow_begin_standalone() {
        if (!dw.n_manifolds || !(pipe_dev && ovf_grid_blocks)) return AVN_OK;
        launch_overflow_reset(b_ovf_ticket.as<uint32_t>(), dw.n_bodies + 1, pg.ctr + PGC_OVF_TILE, PGC_OVF_TILES, stream);
        ovf_epoch = 0;
        return AVN_OK;
    }
    avn_status run_system(avn_system sys) override {
        avn_status st = need_bodies();
        if (st != AVN_OK) return st;
        if ((st = rebuild_joint_schedules()) != AVN_OK) return st;
        if ((st = rebuild_incidence()) != AVN_OK) return st;
        if (sys != AVN_SYS_SOLVER && (st = flow_begin_standalone()) != AVN_OK) return st;
        if ((st = dispatch_system(sys)) != AVN_OK) return st;
        HIPCHK(hipStreamSynchronize(stream));
        return AVN_OK;
    }
    avn_status profile_system(avn_system sys, uint32_t repeats, double* total_ms, uint32_t* n_launches) override {
        avn_status st = need_bodies();
        if (st != AVN_OK) return st;
        if ((st = rebuild_joint_schedules()) != AVN_OK) return st;
        if ((st = rebuild_incidence()) != AVN_OK) return st;
        if ((st = flow_begin_standalone()) != AVN_OK) return st;
        hipEvent_t a, b;
        HIPCHK(hipEventCreate(&a)); HIPCHK(hipEventCreate(&b));
        HIPCHK(hipStreamSynchronize(stream));
        uint32_t before = launches;
        HIPCHK(hipEventRecord(a, stream));  // events on the stream the kernels are launched on
        for (uint32_t r = 0; r < repeats; ++r)
            if ((st = dispatch_system(sys)) != AVN_OK) break;
        HIPCHK(hipEventRecord(b, stream));
        HIPCHK(hipStreamSynchronize(stream));
        float ms = 0;
        HIPCHK(hipEventElapsedTime(&ms, a, b));
        (void)hipEventDestroy(a); (void)hipEventDestroy(b);
        if (total_ms) *total_ms = ms;
        if (n_launches) *n_launches = launches - before;
        return st;
    }
    avn_status step() override {
        avn_status st = need_bodies();
        if (st != AVN_OK) return st;
        for (bool& b : dg_stamped) b = false;
        dg_np = false;
        if (pipe_on) return pipe_dev ? pipeline_step_device() : pipeline_step();
        launches = 0;
        HIPCHK(hipEventRecord(ev[0], stream));
        const bool overlap = overlap_bp && have_colliders;
        bp_timed = false;
        if (have_colliders) {
            if (overlap) {
                HIPCHK(hipStreamWaitEvent(stream_bp, ev[0], 0));  // after the previous step's write-back
                bs = stream_bp;
                HIPCHK(hipEventRecord(ev_bp_t0, stream_bp));
            }
            st = update_aabb();
            if (st == AVN_OK) st = collect_launch();
            if (overlap) { (void)hipEventRecord(ev_bp_t1, stream_bp); bp_timed = true; }
            if (st != AVN_OK) { bs = stream; return st; }
        }
        HIPCHK(hipEventRecord(ev[1], stream));
        st = solver_front();                              // enqueued while the broad phase runs / its pair counters travel back
        if (st == AVN_OK) st = collect_finish();          // (emit pass only when the step found new pairs)
        if (overlap) {
            (void)hipEventRecord(ev_bp_done, stream_bp);
            (void)hipStreamWaitEvent(stream, ev_bp_done, 0);  // the write-back must not overtake k_update_aabb's reads
            bs = stream;
        }
        if (st != AVN_OK) return st;
        if ((st = solver_back()) != AVN_OK) return st;
        HIPCHK(hipEventRecord(ev[4], stream));
        ev_valid = true;
        last_timers.kernel_launches = launches;
        return AVN_OK;
    }
    avn_status synchronize() override { HIPCHK(hipStreamSynchronize(stream)); HIPCHK(hipStreamSynchronize(stream_bp)); return AVN_OK; }

    // ---- level-2 sharding (include/avian_mi355x.h: avn_halo_plan) ------------------------------------------------------------------------
    // One contact island over several worlds: global colouring, and after every colour launch the (linear, angular) velocity records of the
    // shared bodies this world's manifolds moved go to the other holders.  Exchange records of one colour are contiguous over the peers
    // (list k = colour * n_peers + peer), so a colour costs one pack launch, one grouped RCCL send/recv and one unpack launch.
    struct HaloPlan {
        std::vector<int32_t> peers, send, recv;
        std::vector<uint32_t> send_off, recv_off;   // [24 * n_peers + 1]
    } halo;
    DevBuf b_halo_send, b_halo_recv, b_halo_out, b_halo_in;
    bool halo_on = false;
    bool bias_skeleton = getenv("AVN_BIAS_SKELETON") != nullptr && getenv("AVN_BIAS_SKELETON")[0] == '1';
    Comm comm;
    std::vector<CommXfer> xf_send, xf_recv;
    avn_status halo_plan_upload(const avn_halo_plan* p) override {
        if (!p) { error = "halo_plan_upload: null plan"; return AVN_ERR_BAD_ARG; }
        const size_t n = (size_t)AVN_GRAPH_COLOR_COUNT * p->n_peers;
        if (p->n_peers && (!p->peer_rank || !p->send_offsets || !p->recv_offsets)) { error = "halo_plan_upload: null array"; return AVN_ERR_BAD_ARG; }
        HaloPlan h;
        if (p->n_peers) {
            h.peers.assign(p->peer_rank, p->peer_rank + p->n_peers);
            h.send_off.assign(p->send_offsets, p->send_offsets + n + 1); h.recv_off.assign(p->recv_offsets, p->recv_offsets + n + 1);
            for (size_t k = 0; k < n; ++k)
                if (h.send_off[k] > h.send_off[k + 1] || h.recv_off[k] > h.recv_off[k + 1]) { error = "halo_plan_upload: offsets must ascend"; return AVN_ERR_BAD_ARG; }
            if (h.send_off[0] || h.recv_off[0]) { error = "halo_plan_upload: offsets must start at 0"; return AVN_ERR_BAD_ARG; }
            if ((h.send_off[n] && !p->send_bodies) || (h.recv_off[n] && !p->recv_bodies)) { error = "halo_plan_upload: null body list"; return AVN_ERR_BAD_ARG; }
            h.send.assign(p->send_bodies, p->send_bodies + h.send_off[n]); h.recv.assign(p->recv_bodies, p->recv_bodies + h.recv_off[n]);
            const int64_t nb = have_bodies ? (int64_t)dw.n_bodies : INT32_MAX;
            for (int32_t b : h.send) if (b < 0 || b >= nb) { error = "halo_plan_upload: body index out of range"; return AVN_ERR_BAD_ARG; }
            for (int32_t b : h.recv) if (b < 0 || b >= nb) { error = "halo_plan_upload: body index out of range"; return AVN_ERR_BAD_ARG; }
        }
        HIPCHK(hipStreamSynchronize(stream));
        halo = std::move(h);
        halo_on = !halo.peers.empty();
        drop_graph();
        if (!halo_on) return AVN_OK;
        hipError_t err;
        b_halo_send.ensure(std::max<size_t>(halo.send.size(), 1) * 4, err); if (err != hipSuccess) { error = "hipMalloc failed"; return AVN_ERR_OOM; }
        b_halo_recv.ensure(std::max<size_t>(halo.recv.size(), 1) * 4, err); if (err != hipSuccess) { error = "hipMalloc failed"; return AVN_ERR_OOM; }
        b_halo_out.ensure(std::max<size_t>(halo.send.size(), 1) * 2 * sizeof(V), err); if (err != hipSuccess) { error = "hipMalloc failed"; return AVN_ERR_OOM; }
        b_halo_in.ensure(std::max<size_t>(halo.recv.size(), 1) * 2 * sizeof(V), err); if (err != hipSuccess) { error = "hipMalloc failed"; return AVN_ERR_OOM; }
        if (!halo.send.empty()) HIPCHK(hipMemcpyAsync(b_halo_send.p, halo.send.data(), halo.send.size() * 4, hipMemcpyHostToDevice, stream));
        if (!halo.recv.empty()) HIPCHK(hipMemcpyAsync(b_halo_recv.p, halo.recv.data(), halo.recv.size() * 4, hipMemcpyHostToDevice, stream));
        HIPCHK(hipStreamSynchronize(stream));
        return AVN_OK;
    }
    // one colour of one contact pass, in the single-world launch shape (overflow colour: the host schedule's launches)
    avn_status color_pass_enqueue(int pass, uint32_t color) {
        if (!dw.n_manifolds || !grid_blocks[color]) return AVN_OK;
        if (pipe_dev) { error = "level-2 colour passes need host-uploaded manifolds (not the device closed loop)"; return AVN_ERR_STATE; }
        uint32_t gb[AVN_GRAPH_COLOR_COUNT] = {0};
        gb[color] = grid_blocks[color];
        OverflowSchedule ovf{0, nullptr, nullptr, nullptr, nullptr, nullptr, 0};
        if (color == AVN_COLOR_OVERFLOW_INDEX) {
            ovf = OverflowSchedule{sched_overflow.n_components, sched_overflow.d_comp_level_begin.as<uint32_t>(), sched_overflow.d_level_offsets.as<uint32_t>(),
                                   sched_overflow.d_order.as<uint32_t>(), nullptr, nullptr, 0};
            if (sched_overflow.gorder.size() > overflow_level_threshold) {
                ovf.gorder = sched_overflow.d_gorder.as<uint32_t>();
                ovf.glevel_offsets = sched_overflow.glevel_offsets.data();
                ovf.n_glevels = (uint32_t)sched_overflow.glevel_offsets.size() - 1;
            }
        }
        launches += launch_contact_pass<T>(dw, params, pass, gb, use_handles ? nullptr : color_offsets, ovf, stream);
        return AVN_OK;
    }
    static int color_pass_of(avn_system sys) {
        switch (sys) {
            case AVN_SYS_WARM_START: return PASS_WARM_START_COLORS;
            case AVN_SYS_SOLVE_CONTACTS_BIAS: return PASS_SOLVE_BIAS;
            case AVN_SYS_SOLVE_CONTACTS_RELAX: return PASS_SOLVE_RELAX;
            case AVN_SYS_SOLVE_RESTITUTION: return PASS_RESTITUTION_;
            default: return -1;
        }
    }
    avn_status run_color_pass(avn_system sys, uint32_t color) override {
        if (color >= AVN_GRAPH_COLOR_COUNT) { error = "run_color_pass: colour out of range"; return AVN_ERR_BAD_ARG; }
        const int pass = color_pass_of(sys);
        if (pass < 0) { error = "run_color_pass: not a contact pass"; return AVN_ERR_BAD_ARG; }
        avn_status st = need_bodies();
        if (st != AVN_OK) return st;
        if ((st = rebuild_incidence()) != AVN_OK) return st;
        if (pass == PASS_RESTITUTION_ && !any_restitution) return AVN_OK;
        if ((st = color_pass_enqueue(pass, color)) != AVN_OK) return st;
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(stream));
        return AVN_OK;
    }
    avn_status halo_list(uint32_t color, uint32_t peer, const std::vector<uint32_t>& off, size_t* b0, size_t* b1) {
        if (color >= AVN_GRAPH_COLOR_COUNT || peer >= halo.peers.size()) { error = "halo: colour or peer out of range"; return AVN_ERR_BAD_ARG; }
        const size_t k = (size_t)color * halo.peers.size() + peer;
        *b0 = off[k]; *b1 = off[k + 1];
        return AVN_OK;
    }
    avn_status halo_pack(uint32_t color, uint32_t peer, void* out, size_t* count) override {
        if (!count) { error = "halo_pack: null count"; return AVN_ERR_BAD_ARG; }
        size_t b0, b1;
        avn_status st = halo_list(color, peer, halo.send_off, &b0, &b1);
        if (st != AVN_OK) return st;
        *count = b1 - b0;
        if (b1 == b0) return AVN_OK;
        if (!out) { error = "halo_pack: null output"; return AVN_ERR_BAD_ARG; }
        if ((st = need_bodies()) != AVN_OK) return st;
        launch_halo_pack<T>(dw, b_halo_send.as<int32_t>() + b0, (uint32_t)(b1 - b0), b_halo_out.as<V>() + 2 * b0, stream); ++launches;
        HIPCHK(hipMemcpyAsync(out, b_halo_out.as<V>() + 2 * b0, (b1 - b0) * 2 * sizeof(V), hipMemcpyDeviceToHost, stream));
        HIPCHK(hipStreamSynchronize(stream));
        return AVN_OK;
    }
    avn_status halo_unpack(uint32_t color, uint32_t peer, const void* in, size_t count) override {
        size_t b0, b1;
        avn_status st = halo_list(color, peer, halo.recv_off, &b0, &b1);
        if (st != AVN_OK) return st;
        if (count != b1 - b0 || (count && !in)) { error = "halo_unpack: count does not match the plan"; return AVN_ERR_BAD_ARG; }
        if (!count) return AVN_OK;
        if ((st = need_bodies()) != AVN_OK) return st;
        HIPCHK(hipMemcpyAsync(b_halo_in.as<V>() + 2 * b0, in, count * 2 * sizeof(V), hipMemcpyHostToDevice, stream));
        launch_halo_unpack<T>(dw, b_halo_recv.as<int32_t>() + b0, (uint32_t)count, b_halo_in.as<V>() + 2 * b0, stream); ++launches;
        HIPCHK(hipStreamSynchronize(stream));
        return AVN_OK;
    }
    avn_status comm_init(const uint8_t* unique_id, int n_ranks, int rank) override {
        HIPCHK(hipStreamSynchronize(stream));
        return comm.init(unique_id, n_ranks, rank, error);
    }
    // the exchange after colour c inside avn_step: everything is enqueued on the world's stream, no host code waits
    avn_status halo_exchange(uint32_t c) {
        const size_t np = halo.peers.size(), k0 = (size_t)c * np;
        const size_t s0 = halo.send_off[k0], s1 = halo.send_off[k0 + np], r0 = halo.recv_off[k0], r1 = halo.recv_off[k0 + np];
        if (s1 == s0 && r1 == r0) return AVN_OK;
        if (s1 > s0) { launch_halo_pack<T>(dw, b_halo_send.as<int32_t>() + s0, (uint32_t)(s1 - s0), b_halo_out.as<V>() + 2 * s0, stream); ++launches; }
        xf_send.clear(); xf_recv.clear();
        for (size_t p = 0; p < np; ++p) {
            const size_t a = halo.send_off[k0 + p], b = halo.send_off[k0 + p + 1], ra = halo.recv_off[k0 + p], rb = halo.recv_off[k0 + p + 1];
            if (b > a) xf_send.push_back(CommXfer{b_halo_out.as<V>() + 2 * a, (b - a) * 2 * sizeof(V), halo.peers[p]});
            if (rb > ra) xf_recv.push_back(CommXfer{b_halo_in.as<V>() + 2 * ra, (rb - ra) * 2 * sizeof(V), halo.peers[p]});
        }
        avn_status st = comm.exchange(xf_send.data(), xf_send.size(), xf_recv.data(), xf_recv.size(), stream, error);
        if (st != AVN_OK) return st;
        ++halo_exchanges;
        if (r1 > r0) { launch_halo_unpack<T>(dw, b_halo_recv.as<int32_t>() + r0, (uint32_t)(r1 - r0), b_halo_in.as<V>() + 2 * r0, stream); ++launches; }
        return AVN_OK;
    }
    uint32_t halo_exchanges = 0;
    // one contact pass in level-2 form: colours in solve order (overflow first), exchange after each
    avn_status level2_pass(int pass) {
        static const auto order = [] { std::array<uint32_t, AVN_GRAPH_COLOR_COUNT> o; o[0] = AVN_COLOR_OVERFLOW_INDEX; for (uint32_t c = 0; c < AVN_COLOR_OVERFLOW_INDEX; ++c) o[c + 1] = c; return o; }();
        for (uint32_t c : order) {
            avn_status st = color_pass_enqueue(pass, c);
            if (st == AVN_OK) st = halo_exchange(c);
            if (st != AVN_OK) return st;
        }
        return AVN_OK;
    }
    avn_status level2_substeps() {   // SubstepSchedule with the contact passes split by colour (avian_amd/shard.py: level2_solver)
        for (uint32_t s = 0; s < cfg.substeps; ++s) {
            integrate_velocities();
            avn_status st = level2_pass(PASS_WARM_START_COLORS);
            for (uint32_t it = 0; it < cfg.solver_iterations && st == AVN_OK; ++it) st = level2_pass(PASS_SOLVE_BIAS);
            if (st != AVN_OK) return st;
            integrate_positions();
            for (uint32_t it = 0; it < cfg.solver_iterations && st == AVN_OK; ++it) st = level2_pass(PASS_SOLVE_RELAX);
            if (st != AVN_OK) return st;
            for (uint32_t it = 0; it < cfg.solver_iterations; ++it) xpbd_solve(it == 0);
            xpbd_velocity_projection();
            joint_damping();
        }
        return AVN_OK;
    }
    // ---- islands and sleeping (include/avian_mi355x.h: avn_islands_get / avn_sleep_update; k_islands.hip) --------------------------------
    DevBuf b_isl_parent, b_isl_label, b_isl_ctr, b_sleep_timer, b_isl_awake, b_isl_rests, b_isl_wakes;
    uint32_t sleep_n = 0;       // body count the timers belong to (a different count restarts them)
    bool islands_fresh = false; // labels on the device describe the current constraint graph
    avn_status island_buffers() {
        const size_t n = std::max<uint32_t>(dw.n_bodies, 1);
        hipError_t err;
        for (DevBuf* b : {&b_isl_parent, &b_isl_label, &b_isl_awake}) { b->ensure(n * 4, err); if (err != hipSuccess) { error = "hipMalloc failed"; return AVN_ERR_OOM; } }
        b_isl_rests.ensure(n, err); if (err != hipSuccess) { error = "hipMalloc failed"; return AVN_ERR_OOM; }
        b_isl_wakes.ensure(n, err); if (err != hipSuccess) { error = "hipMalloc failed"; return AVN_ERR_OOM; }
        b_isl_ctr.ensure(64, err); if (err != hipSuccess) { error = "hipMalloc failed"; return AVN_ERR_OOM; }
        const bool grown = b_sleep_timer.ensure(n * 4, err); if (err != hipSuccess) { error = "hipMalloc failed"; return AVN_ERR_OOM; }
        if (grown || sleep_n != dw.n_bodies) { HIPCHK(hipMemsetAsync(b_sleep_timer.p, 0, n * 4, stream)); sleep_n = dw.n_bodies; }
        return AVN_OK;
    }
    avn_status islands_compute() {
        avn_status st = need_bodies();
        if (st != AVN_OK) return st;
        if ((st = island_buffers()) != AVN_OK) return st;
        HIPCHK(hipMemsetAsync(b_isl_ctr.p, 0, 64, stream));
        launch_islands<T>(dw, b_isl_parent.as<uint32_t>(), b_isl_label.as<uint32_t>(), b_isl_ctr.as<uint32_t>(), stream);
        launches += 2 + (dw.n_manifolds ? 1 : 0) + (dw.n_joints ? 1 : 0);
        HIPCHK(hipGetLastError());
        return AVN_OK;
    }
    avn_status islands_get(uint32_t* island_of_body, uint32_t* n_islands) override {
        avn_status st = islands_compute();
        if (st != AVN_OK) return st;
        uint32_t ctr[2] = {0, 0};
        HIPCHK(hipMemcpyAsync(ctr, b_isl_ctr.p, 8, hipMemcpyDeviceToHost, stream));
        if (island_of_body && dw.n_bodies) HIPCHK(hipMemcpyAsync(island_of_body, b_isl_label.p, (size_t)dw.n_bodies * 4, hipMemcpyDeviceToHost, stream));
        HIPCHK(hipStreamSynchronize(stream));
        if (n_islands) *n_islands = ctr[0];
        return AVN_OK;
    }
    avn_status sleep_update(const avn_sleep_params* sp, avn_sleep_stats* out) override {
        if (!sp || sp->struct_size != sizeof(avn_sleep_params)) { error = "sleep_update: bad params"; return AVN_ERR_BAD_ARG; }
        avn_status st = islands_compute();
        if (st != AVN_OK) return st;
        SleepParams<T> k;
        k.length_unit_squared = (T)sp->length_unit * (T)sp->length_unit;
        k.lin_threshold_squared = (T)(sp->linear_threshold * std::fabs(sp->linear_threshold));   // f32 product, "keep signs", then `as Scalar`
        k.ang_threshold_squared = (T)(sp->angular_threshold * std::fabs(sp->angular_threshold));
        k.delta_secs = sp->delta_secs; k.time_to_sleep = sp->time_to_sleep;
        k.body_lin = nullptr; k.body_ang = nullptr; k.body_disabled = nullptr;
        if (sp->body_linear_threshold || sp->body_angular_threshold || sp->body_sleeping_disabled) {
            const size_t n = dw.n_bodies;
            if ((st = stage_reserve(al(4 * n) * 2 + al(n) + 1024)) != AVN_OK) return st;
            if ((st = stage_in<float>(sp->body_linear_threshold, n, &k.body_lin)) != AVN_OK) return st;
            if ((st = stage_in<float>(sp->body_angular_threshold, n, &k.body_ang)) != AVN_OK) return st;
            if ((st = stage_in<uint8_t>(sp->body_sleeping_disabled, n, &k.body_disabled)) != AVN_OK) return st;
        }
        HIPCHK(hipMemsetAsync(b_isl_awake.p, 0, (size_t)std::max<uint32_t>(dw.n_bodies, 1) * 4, stream));
        launch_sleep_update<T>(dw, k, b_isl_label.as<uint32_t>(), b_sleep_timer.as<float>(), b_isl_awake.as<uint32_t>(), b_isl_rests.as<uint8_t>(), b_isl_wakes.as<uint8_t>(), b_isl_ctr.as<uint32_t>(), stream);
        launches += 2;
        HIPCHK(hipGetLastError());
        if (out) {
            uint32_t ctr[8] = {0};
            HIPCHK(hipMemcpyAsync(ctr, b_isl_ctr.p, 32, hipMemcpyDeviceToHost, stream));
            HIPCHK(hipStreamSynchronize(stream));
            out->n_islands = ctr[0]; out->n_island_bodies = ctr[1]; out->n_sleeping_bodies = ctr[6]; out->n_awake_bodies = ctr[1] - ctr[6];
            out->n_resting_islands = ctr[2]; out->n_resting_bodies = ctr[3]; out->n_waking_islands = ctr[4]; out->n_waking_bodies = ctr[5];
        }
        return AVN_OK;
    }
    avn_status sleep_get(const avn_sleep_out* o) override {
        if (!o) { error = "sleep_get: null"; return AVN_ERR_BAD_ARG; }
        avn_status st = need_bodies();
        if (st != AVN_OK) return st;
        if (sleep_n != dw.n_bodies || !b_isl_rests.p) { error = "sleep_get: call avn_sleep_update first"; return AVN_ERR_STATE; }
        const size_t n = dw.n_bodies;
        if (o->sleep_timer) HIPCHK(hipMemcpyAsync(o->sleep_timer, b_sleep_timer.p, n * 4, hipMemcpyDeviceToHost, stream));
        if (o->island) HIPCHK(hipMemcpyAsync(o->island, b_isl_label.p, n * 4, hipMemcpyDeviceToHost, stream));
        if (o->island_rests) HIPCHK(hipMemcpyAsync(o->island_rests, b_isl_rests.p, n, hipMemcpyDeviceToHost, stream));
        if (o->island_wakes) HIPCHK(hipMemcpyAsync(o->island_wakes, b_isl_wakes.p, n, hipMemcpyDeviceToHost, stream));
        HIPCHK(hipStreamSynchronize(stream));
        return AVN_OK;
    }
    avn_status sleep_reset(const uint32_t* bodies, size_t n) override {
        avn_status st = need_bodies();
        if (st != AVN_OK) return st;
        if ((st = island_buffers()) != AVN_OK) return st;
        if (!bodies || n == 0) { launch_sleep_reset(b_sleep_timer.as<float>(), nullptr, dw.n_bodies, dw.n_bodies, stream); HIPCHK(hipStreamSynchronize(stream)); return AVN_OK; }
        if ((st = stage_reserve(al(4 * n) + 1024)) != AVN_OK) return st;
        const uint32_t* d = nullptr;
        if ((st = stage_in<uint32_t>(bodies, n, &d)) != AVN_OK) return st;
        launch_sleep_reset(b_sleep_timer.as<float>(), d, (uint32_t)n, dw.n_bodies, stream);
        HIPCHK(hipStreamSynchronize(stream));
        return AVN_OK;
    }
    avn_status diagnostics(avn_diagnostics* d) override;
    avn_status timers(avn_timers* t) override {
        if (!t) return AVN_ERR_BAD_ARG;
        HIPCHK(hipStreamSynchronize(stream));
        HIPCHK(hipStreamSynchronize(stream_bp));
        if (ev_valid) {
            float a = 0, b = 0, c = 0, d = 0, e = 0;
            HIPCHK(hipEventElapsedTime(&a, ev[0], ev[1]));
            HIPCHK(hipEventElapsedTime(&b, ev[1], ev[2]));
            HIPCHK(hipEventElapsedTime(&c, ev[2], ev[3]));
            HIPCHK(hipEventElapsedTime(&d, ev[3], ev[4]));
            HIPCHK(hipEventElapsedTime(&e, ev[0], ev[4]));
            // overlapped broad phase: its own duration on its own stream (it is NOT a term of step_ms then)
            if (bp_timed) HIPCHK(hipEventElapsedTime(&a, ev_bp_t0, ev_bp_t1));
            last_timers.broad_phase_ms = a; last_timers.prepare_ms = b; last_timers.substeps_ms = c; last_timers.finalize_ms = d;
            last_timers.step_ms = e;
            last_timers.bias_pass_ms = 0; last_timers.bias_pass_launches = 0;
            last_timers.island_blocks = islands_active() ? islands.n_blocks : 0u; last_timers.reserved0 = 0;
            if (bias_timed) {   // mean over the step's substeps
                double sum = 0;
                for (uint32_t k = 0; k < bias_timed; ++k) { float f = 0; HIPCHK(hipEventElapsedTime(&f, ev_bias[2 * k], ev_bias[2 * k + 1])); sum += f; }
                last_timers.bias_pass_ms = sum / bias_timed; last_timers.bias_pass_launches = bias_launches;
            }
        }
        uint32_t cc = 0;
        HIPCHK(hipMemcpy(&cc, dw.constraint_count, 4, hipMemcpyDeviceToHost));
        last_timers.contact_constraint_count = cc;
        *t = last_timers;
        return AVN_OK;
    }
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
