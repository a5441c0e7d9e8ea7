// k_graph.hip — the closed loop's integer bookkeeping on the device: IdPool, ContactGraph rows, the status-change loop of
// NarrowPhase::update and the ConstraintGraph (greedy colouring + per-colour handle lists), bit-identical to the reference's
// serial loops.
//
// Reference (paths relative to /root/reference/src):
//   data_structures/id_pool.rs:31-40                     IdPool::alloc_id / free_id (lowest free id first)
//   collision/contact_types/contact_graph.rs:521-631     add_edge_and_key_with / remove_edge_by_id
//   collision/narrow_phase/system_param.rs:141-389       the status-change loop (ascending ContactId)
//   dynamics/solver/constraint_graph.rs:163-296          push_manifold (greedy colouring) / pop_manifold (swap_remove)
//
// The reference walks the changed contacts serially.  What makes the walk order-dependent is (a) the greedy colour choice of
// a push, which depends on the colours taken by the EARLIER pushes on the same two bodies, and (b) the position of every
// handle inside its colour's Vec (push = append, pop = swap_remove).  Both are replayed exactly:
//
//   1. k_narrow_phase<DENSE> leaves one packed change word per row; an exclusive scan over the rows numbers the changes in
//      ascending ContactId = the reference's processing order ("ops").
//   2. k_pg_classify turns a change into PUSH | POP | nothing (+ REMOVE) -- the if/else chain of system_param.rs:155-373 --
//      and emits one (body, op) entry per non-static side; a stable radix sort by body gives every body its ops in order.
//   3. A pop's colour is known (stored on the row), so the bits it frees in a body's colour mask are STATIC: a segmented scan
//      over the sorted entries gives every entry the colours freed on its body by earlier pops (k_pg_scan_*).  Only pushes
//      depend on each other: k_pg_color is a dataflow kernel, one lane per op in op order, that waits (agent-scope loads) for
//      the previous push entry of each of its bodies, picks the reference's colour (lowest free of 0..19 for two non-static
//      bodies, highest free of 22..1 next to a static one, else overflow) and publishes the bits.  Waves take their tile from
//      an atomic counter, so a lane only ever waits for lanes of waves that already run: no deadlock; spins are bounded.
//   4. The ops are bucketed by colour (stable) and k_pg_replay replays each colour's push / swap_remove sequence with one wave,
//      64 ops at a time (tools/experiments/replay_batches.py is the CPU model of the rule that makes a batch parallel).
//   5. Removed pairs free their ids into the sorted free list (k_pg_merge_free), rows are cleared, keys leave the pair set.
//
// Per-body colour masks (PG::bcol, bit c = "in GraphColor c's body_set") replace the 24 BitVecs: same information, one load.
#include "avn_kernels.h"
#include "avn_scan.h"

namespace avn {

#define PG_KIND_NONE 0u
#define PG_KIND_PUSH 1u
#define PG_KIND_POP 2u
#define PG_SPIN_LIMIT (1u << 22)

__device__ __forceinline__ uint32_t ld_agent(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agent(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// same-wave read-after-write through memory (k_pg_replay): bypass the CU's L1
__device__ __forceinline__ uint32_t ld_wg(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

// A global store the compiler's s_waitcnt insertion does not see (k_pg_replay): hipcc drains ALL outstanding memory operations
// (vmcnt(0)) in front of the stores of a loop body that also holds rarely-taken loads, i.e. every batch would wait ~2 us for the
// previous batch's scattered stores.  The kernel drains them itself where it matters (chunk start, window reload).
__device__ __forceinline__ void st_async(uint32_t* p, uint32_t v) { asm volatile("global_store_dword %0, %1, off" : : "v"(p), "v"(v) : "memory"); }

__device__ __forceinline__ int wave_incl_add(int v, uint32_t lane) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { int u = __shfl_up(v, off); if ((int)lane >= off) v += u; }
    return v;
}
__device__ __forceinline__ uint32_t wave_incl_min(uint32_t v, uint32_t lane) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { uint32_t u = (uint32_t)__shfl_up((int)v, off); if ((int)lane >= off) v = u < v ? u : v; }
    return v;
}
__device__ __forceinline__ uint32_t wave_incl_max(uint32_t v, uint32_t lane) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { uint32_t u = (uint32_t)__shfl_up((int)v, off); if ((int)lane >= off) v = u > v ? u : v; }
    return v;
}

// ---- new pairs: IdPool::alloc_id in emission order + ContactGraph::add_edge_and_key_with --------------------------------------
// One launch (round 4; was k_hs_insert_pairs + k_pg_add_pairs + k_pg_after_add): every pair takes its id, initialises its row and puts its
// PairKey into ContactGraph::pair_set; the LAST workgroup to finish (ticket in ctr[PGC_ADD_DONE]) moves the IdPool's counters -- every
// workgroup has read them by then.
__device__ __forceinline__ uint64_t pg_hs_mix(uint64_t k) { k ^= k >> 33; k *= 0xff51afd7ed558ccdull; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ull; k ^= k >> 33; return k; }
template <class T>
__global__ __launch_bounds__(256) void k_pg_add_pairs(PG pg, CT<T> ct, const avn_pair* __restrict__ pairs, uint32_t total, uint64_t* __restrict__ pair_set, uint32_t pair_set_mask) {
    __shared__ uint32_t s_last;
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    const uint32_t head = pg.ctr[PGC_FREE_HEAD], n_free = pg.ctr[PGC_N_FREE], next = pg.ctr[PGC_NEXT_ID];
    const unsigned long long seq0 = ((unsigned long long)pg.ctr[PGC_SEQ + 1] << 32) | pg.ctr[PGC_SEQ];
    if (i < total) {
        const uint32_t id = i < n_free ? pg.free_ids[head + i] : next + (i - n_free);   // the i-th lowest free id, then fresh ids
        const avn_pair pr = pairs[i];
        const uint32_t f = pr.flags;
        const uint32_t flags = ((f & AVN_PAIR_GENERATE_CONSTRAINTS) ? (uint32_t)AVN_CP_GENERATE_CONSTRAINTS : 0u) | ((f & AVN_PAIR_MODIFY_CONTACTS) ? (uint32_t)AVN_CP_MODIFY_CONTACTS : 0u) |
                               ((f & AVN_PAIR_CONTACT_EVENTS) ? (uint32_t)AVN_CP_CONTACT_EVENTS : 0u) | AVN_CP_ROW_USED;
        ct.meta[id] = make_uint4(pg.ent2slot[pr.collider1], pg.ent2slot[pr.collider2], flags, 0u);
        ct.dcount[id] = 0;
        pg.bodies[id] = make_int2(pr.body1, pr.body2);
        pg.color[id] = PG_NONE;
        if (pg.new_ids) pg.new_ids[i] = id;
        pg.seq[id] = seq0 + i;   // the edge's place in its colliders' edge lists (newest first = descending stamp)
        // add_edge_and_key_with (contact_graph.rs:521-566): the key joins the pair set
        const uint32_t a = pr.collider1, b = pr.collider2;
        const uint64_t key = a < b ? ((uint64_t)a << 32) | b : ((uint64_t)b << 32) | a;
        uint32_t h = (uint32_t)pg_hs_mix(key) & pair_set_mask;
        for (;;) {
            const unsigned long long prev = atomicCAS((unsigned long long*)&pair_set[h], ~0ull, (unsigned long long)key);
            if (prev == ~0ull || prev == key) break;
            h = (h + 1u) & pair_set_mask;
        }
    }
    __syncthreads();   // every thread of the workgroup has read the counters
    if (threadIdx.x == 0) { __threadfence(); s_last = atomicAdd(&pg.ctr[PGC_ADD_DONE], 1u) == gridDim.x - 1u ? 1u : 0u; }
    __syncthreads();
    if (s_last && threadIdx.x == 0) {
        const uint32_t used = total < n_free ? total : n_free;
        pg.ctr[PGC_FREE_HEAD] = head + used;
        pg.ctr[PGC_N_FREE] = n_free - used;
        pg.ctr[PGC_NEXT_ID] = next + (total - used);
        const unsigned long long seq1 = seq0 + total;
        pg.ctr[PGC_SEQ] = (uint32_t)seq1; pg.ctr[PGC_SEQ + 1] = (uint32_t)(seq1 >> 32);
        pg.ctr[PGC_ADD_DONE] = 0u;
    }
}
template <class T> void launch_pg_add_pairs(const PG& pg, const CT<T>& ct, const avn_pair* pairs, uint32_t total, uint64_t* pair_set, uint32_t pair_set_cap, hipStream_t s) {
    if (!total) return;
    hipLaunchKernelGGL(k_pg_add_pairs<T>, dim3((total + 255) / 256), dim3(256), 0, s, pg, ct, pairs, total, pair_set, pair_set_cap - 1u);
}

// ---- collision hooks: CollisionHooks::filter_pairs (broad_phase.rs:431-439; include/avian_mi355x.h "collision hooks") ----------------------
// The emitted pairs one of whose colliders asks for the filter, with their place in the emission order (arbitrary list order: the host sorts by it) ...
__global__ __launch_bounds__(256) void k_hook_filter_collect(const avn_pair* __restrict__ pairs, uint32_t total, avn_hook_pair* __restrict__ out, uint32_t* __restrict__ count) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const avn_pair pr = pairs[i];
    if (!(pr.flags & AVN_PAIR_NEEDS_CUSTOM_FILTER)) return;
    const uint32_t k = atomicAdd(count, 1u);   // (k < total: the list is sized for every pair)
    out[k] = avn_hook_pair{i, pr.collider1, pr.collider2};
}
// ... and the emission list without the pairs the hook rejected (rej: ascending emission indices), order kept
__global__ __launch_bounds__(256) void k_hook_filter_compact(const avn_pair* __restrict__ in, avn_pair* __restrict__ out, uint32_t total, const uint32_t* __restrict__ rej, uint32_t n_rej) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    uint32_t lo = 0, hi = n_rej;   // rejected indices below i
    while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (rej[mid] < i) lo = mid + 1; else hi = mid; }
    if (lo < n_rej && rej[lo] == i) return;
    out[i - lo] = in[i];
}
void launch_hook_filter_collect(const avn_pair* pairs, uint32_t total, avn_hook_pair* out, uint32_t* count, hipStream_t s) {
    if (total) hipLaunchKernelGGL(k_hook_filter_collect, dim3((total + 255) / 256), dim3(256), 0, s, pairs, total, out, count);
}
void launch_hook_filter_compact(const avn_pair* in, avn_pair* out, uint32_t total, const uint32_t* rej, uint32_t n_rej, hipStream_t s) {
    if (total) hipLaunchKernelGGL(k_hook_filter_compact, dim3((total + 255) / 256), dim3(256), 0, s, in, out, total, rej, n_rej);
}

// ---- the status-change loop, decision part (system_param.rs:155-373) ------------------------------------------------------------
// one changed row -> op k (k = the number of changed rows with a lower ContactId: the reference's processing order)
__device__ __forceinline__ void pg_classify_row(const PG& pg, uint32_t c, uint32_t k, uint32_t n_bodies, uint32_t* hist, const uint32_t* __restrict__ bmeta) {
    const uint32_t w = pg.chg[c];
    const uint32_t flags = w & 0xFFFFu, n_manifolds = (w >> 16) & 0xFFu;
    const int dcount = (int)((w >> 24) & 0xFFu) - 128;
    const bool generates = flags & AVN_CP_GENERATE_CONSTRAINTS, touching = flags & AVN_CP_TOUCHING;
    const uint32_t col = pg.color[c];
    const bool has_handle = col != PG_NONE;   // ContactEdge::constraint_handles is non-empty (one manifold per convex pair)
    uint32_t kind = PG_KIND_NONE, remove = 0;
    if (flags & AVN_CP_DISJOINT_AABB) { if (generates && has_handle) kind = PG_KIND_POP; remove = 1; }
    else if (flags & AVN_CP_STARTED_TOUCHING) { if (generates && n_manifolds && !has_handle) kind = PG_KIND_PUSH; }
    else if (flags & AVN_CP_STOPPED_TOUCHING) { if (generates && has_handle) kind = PG_KIND_POP; }
    else if (touching && (flags & AVN_CP_STARTED_GENERATING_CONSTRAINTS)) { if (n_manifolds && !has_handle) kind = PG_KIND_PUSH; }
    else if (touching && generates && dcount > 0) { if (!has_handle) kind = PG_KIND_PUSH; }
    else if (touching && generates && dcount < 0) { if (has_handle) kind = PG_KIND_POP; }
    const bool s1 = flags & AVN_CP_STATIC1, s2 = flags & AVN_CP_STATIC2;
    if (kind == PG_KIND_PUSH && s1 && s2) kind = PG_KIND_NONE;   // (debug_assert in the reference: never both)
    const int2 b = pg.bodies[c];
    // sleeping on: a status change on a body that sleeps may link into / unlink from a sleeping island, i.e. queue a WakeIslands (system_param.rs:391-398);
    // a step without such a change cannot wake anything, and the host enqueues the solver BEFORE its island manager digests the changes
    if (bmeta && ((b.x >= 0 && (uint32_t)b.x < n_bodies && (meta_flags(bmeta[b.x]) & AVN_BODY_SLEEPING)) || (b.y >= 0 && (uint32_t)b.y < n_bodies && (meta_flags(bmeta[b.y]) & AVN_BODY_SLEEPING))))
        atomicAdd(&pg.ctr[PGC_N_SLEEP_OPS], 1u);
    const uint32_t opcol = kind == PG_KIND_POP ? col : 0xFFu;
    pg.op_cid[k] = c;
    pg.op_chg[k] = w;
    pg.op_info[k] = kind | (s1 ? 4u : 0u) | (s2 ? 8u : 0u) | (remove << 4) | (opcol << 8);
    pg.op_bodies[k] = b;
    pg.rem_flag[k] = remove;
    // one entry per side whose body's colour mask the op reads or writes: non-static sides of pushes and of pops of colours 0..22
    const bool masks = kind == PG_KIND_PUSH || (kind == PG_KIND_POP && col < (uint32_t)AVN_COLOR_OVERFLOW_INDEX);
    pg.ekey_a[2 * k] = (masks && !s1) ? (uint32_t)b.x : n_bodies;
    pg.ekey_a[2 * k + 1] = (masks && !s2) ? (uint32_t)b.y : n_bodies;
    pg.eval_a[2 * k] = 2 * k; pg.eval_a[2 * k + 1] = 2 * k + 1;
    if (kind == PG_KIND_POP) atomicAdd(&hist[col], 1u);
}
// The exclusive scan over the rows' "has a change" flags (= the op index of every changed row, ascending ContactId) and the
// classification of the changed rows in ONE launch (round 4: was k_scan_sums -> k_scan_top -> k_scan_apply -> k_pg_classify, the last of
// them a 1.2 M-row gather for ~30 k changes): the chained scan of avn_scan.h with k_pg_classify's body as its apply stage.  Thread t of a
// tile owns rows base + 8 t .. + 7.  ctr[PGC_N_OPS] <- the number of changes.
__global__ __launch_bounds__(256) void k_pg_scan_classify(PG pg, uint32_t n_rows, uint32_t n_bodies, uint32_t* __restrict__ st, const uint32_t* __restrict__ bmeta) {
    __shared__ uint32_t hist[AVN_GRAPH_COLOR_COUNT];
    const uint32_t nb = gridDim.x, t = threadIdx.x;
    if (t < AVN_GRAPH_COLOR_COUNT) hist[t] = 0;
    const uint32_t tile = sc_take_tile(st);   // (barrier: hist is cleared)
    constexpr uint32_t per = SC_TILE / 256;
    const uint32_t c0 = tile * SC_TILE + t * per;
    uint32_t h[per], s = 0;
#pragma unroll
    for (uint32_t k = 0; k < per; ++k) { h[k] = (c0 + k < n_rows && pg.has[c0 + k]) ? 1u : 0u; s += h[k]; }
    uint32_t tile_sum;
    uint32_t excl = sc_block_excl(s, &tile_sum);
    excl += sc_lookback(st, tile, nb, tile_sum);
    if (s) {
#pragma unroll
        for (uint32_t k = 0; k < per; ++k) if (h[k]) { pg_classify_row(pg, c0 + k, excl, n_bodies, hist, bmeta); ++excl; }
    }
    if (tile == nb - 1u && t == 255u) pg.ctr[PGC_N_OPS] = excl;
    __syncthreads();
    if (t < AVN_GRAPH_COLOR_COUNT && hist[t]) { atomicAdd(&pg.ctr[PGC_BUCKET + t], hist[t]); atomicAdd(&pg.ctr[PGC_N_POP], hist[t]); }
}
void launch_pg_scan_classify(const PG& pg, uint32_t n_rows, uint32_t n_bodies, uint32_t* scan_state, hipStream_t s, const uint32_t* bmeta_if_sleeping) {
    if (n_rows) hipLaunchKernelGGL(k_pg_scan_classify, dim3((n_rows + SC_TILE - 1) / SC_TILE), dim3(256), 0, s, pg, n_rows, n_bodies, scan_state, bmeta_if_sleeping);
}

// An op batch from a list (SleepIslands / WakeIslands of the island manager): the arrays k_pg_classify fills, for ops given as
// (contact id, kind).  A pop of a row without a handle and a push of a row that has one are no-ops, like pop_manifold / push_manifold
// on the host structures.
template <class T>
__global__ __launch_bounds__(256) void k_pg_ops_from_list(PG pg, CT<T> ct, const uint32_t* __restrict__ cids, const uint32_t* __restrict__ kinds, uint32_t n, uint32_t n_bodies) {
    __shared__ uint32_t hist[AVN_GRAPH_COLOR_COUNT];
    if (threadIdx.x < AVN_GRAPH_COLOR_COUNT) hist[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t k = blockIdx.x * 256 + threadIdx.x;
    if (k < n) {
        const uint32_t c = cids[k];
        const uint32_t flags = ct.meta[c].z;
        const uint32_t col = pg.color[c];
        const bool has_handle = col != PG_NONE;
        uint32_t kind = kinds[k];
        if (kind == PG_KIND_POP && !has_handle) kind = PG_KIND_NONE;
        if (kind == PG_KIND_PUSH && has_handle) kind = PG_KIND_NONE;
        const bool s1 = flags & AVN_CP_STATIC1, s2 = flags & AVN_CP_STATIC2;
        if (kind == PG_KIND_PUSH && s1 && s2) kind = PG_KIND_NONE;
        const int2 b = pg.bodies[c];
        const uint32_t opcol = kind == PG_KIND_POP ? col : 0xFFu;
        pg.op_cid[k] = c;
        pg.op_chg[k] = 0u;
        pg.op_info[k] = kind | (s1 ? 4u : 0u) | (s2 ? 8u : 0u) | (opcol << 8);
        pg.op_bodies[k] = b;
        pg.rem_flag[k] = 0u;
        const bool masks = kind == PG_KIND_PUSH || (kind == PG_KIND_POP && col < (uint32_t)AVN_COLOR_OVERFLOW_INDEX);
        pg.ekey_a[2 * k] = (masks && !s1) ? (uint32_t)b.x : n_bodies;
        pg.ekey_a[2 * k + 1] = (masks && !s2) ? (uint32_t)b.y : n_bodies;
        pg.eval_a[2 * k] = 2 * k; pg.eval_a[2 * k + 1] = 2 * k + 1;
        if (kind == PG_KIND_POP) atomicAdd(&hist[col], 1u);
    }
    __syncthreads();
    if (threadIdx.x < AVN_GRAPH_COLOR_COUNT && hist[threadIdx.x]) { atomicAdd(&pg.ctr[PGC_BUCKET + threadIdx.x], hist[threadIdx.x]); atomicAdd(&pg.ctr[PGC_N_POP], hist[threadIdx.x]); }
}
template <class T> void launch_pg_ops_from_list(const PG& pg, const CT<T>& ct, const uint32_t* cids, const uint32_t* kinds, uint32_t n, uint32_t n_bodies, hipStream_t s) {
    if (n) hipLaunchKernelGGL(k_pg_ops_from_list<T>, dim3((n + 255) / 256), dim3(256), 0, s, pg, ct, cids, kinds, n, n_bodies);
}
template void launch_pg_ops_from_list<float>(const PG&, const CT<float>&, const uint32_t*, const uint32_t*, uint32_t, uint32_t, hipStream_t);
template void launch_pg_ops_from_list<double>(const PG&, const CT<double>&, const uint32_t*, const uint32_t*, uint32_t, uint32_t, hipStream_t);

// ---- segmented scan over the body-sorted entries: colours freed by earlier pops, previous push entry ----------------------------
#define PGS_TILE 256
__device__ __forceinline__ void pg_entry_fields(const PG& pg, uint32_t val, uint32_t& popbit, uint32_t& is_push) {
    const uint32_t info = pg.op_info[val >> 1], kind = info & 3u, col = (info >> 8) & 0xFFu;
    popbit = (kind == PG_KIND_POP && col < (uint32_t)AVN_COLOR_OVERFLOW_INDEX) ? (1u << col) : 0u;
    is_push = kind == PG_KIND_PUSH;
}
// in-tile INCLUSIVE segmented scan (keys sorted: "same segment" == same key); pop: OR of pop bits, lp: 1 + position of the latest push entry
__device__ __forceinline__ void pg_tile_scan(uint32_t key, uint32_t& pop, uint32_t& lp, uint32_t* s_key, uint32_t* s_pop, uint32_t* s_lp) {
    const uint32_t t = threadIdx.x;
    s_key[t] = key; s_pop[t] = pop; s_lp[t] = lp;
    __syncthreads();
    for (uint32_t off = 1; off < PGS_TILE; off <<= 1) {
        uint32_t ap = 0, al = 0;
        if (t >= off && s_key[t - off] == key) { ap = s_pop[t - off]; al = s_lp[t - off]; }
        __syncthreads();
        pop |= ap; lp = al > lp ? al : lp;
        s_pop[t] = pop; s_lp[t] = lp;
        __syncthreads();
    }
}
// tile aggregates: [first key, last key, pop of the trailing segment, lp of the trailing segment, uniform]
__global__ __launch_bounds__(PGS_TILE) void k_pg_scan_tiles(PG pg, const uint32_t* __restrict__ keys, const uint32_t* __restrict__ vals, uint32_t n) {
    __shared__ uint32_t s_key[PGS_TILE], s_pop[PGS_TILE], s_lp[PGS_TILE];
    const uint32_t t = threadIdx.x, e = blockIdx.x * PGS_TILE + t;
    const bool valid = e < n;
    uint32_t key = valid ? keys[e] : 0xFFFFFFFFu, pop = 0, lp = 0;
    if (valid) { uint32_t ip; pg_entry_fields(pg, vals[e], pop, ip); lp = ip ? e + 1 : 0u; }
    pg_tile_scan(key, pop, lp, s_key, s_pop, s_lp);
    const uint32_t last = min(n - blockIdx.x * PGS_TILE, (uint32_t)PGS_TILE) - 1;
    if (t == last) {
        uint32_t* a = pg.tile_agg + 5 * (size_t)blockIdx.x;
        a[0] = s_key[0]; a[1] = key; a[2] = pop; a[3] = lp; a[4] = s_key[0] == key;
    }
}
__global__ __launch_bounds__(PGS_TILE) void k_pg_scan_apply(PG pg, const uint32_t* __restrict__ keys, const uint32_t* __restrict__ vals, uint32_t n, uint32_t n_bodies) {
    __shared__ uint32_t s_key[PGS_TILE], s_pop[PGS_TILE], s_lp[PGS_TILE];
    __shared__ uint32_t c_pop, c_lp;
    const uint32_t t = threadIdx.x, e = blockIdx.x * PGS_TILE + t;
    const bool valid = e < n;
    const uint32_t key = valid ? keys[e] : 0xFFFFFFFFu;
    const uint32_t val = valid ? vals[e] : 0u;
    uint32_t own_pop = 0, own_push = 0;
    if (valid) pg_entry_fields(pg, val, own_pop, own_push);
    uint32_t pop = own_pop, lp = own_push ? e + 1 : 0u;
    if (t == 0) {   // carry of the tile's first segment from the tiles before it (not for the "no body" key: nobody reads those)
        uint32_t cp = 0, cl = 0;
        if (blockIdx.x && key < n_bodies) {
            for (uint32_t b = blockIdx.x; b-- > 0;) {
                const uint32_t* a = pg.tile_agg + 5 * (size_t)b;
                if (a[1] != key) break;
                cp |= a[2]; if (!cl) cl = a[3];
                if (!a[4]) break;
            }
        }
        c_pop = cp; c_lp = cl;
    }
    pg_tile_scan(key, pop, lp, s_key, s_pop, s_lp);
    if (!valid) return;
    // inclusive -> exclusive inside the segment
    uint32_t ex_pop = 0, ex_lp = 0;
    if (t > 0 && s_key[t - 1] == key) { ex_pop = s_pop[t - 1]; ex_lp = s_lp[t - 1]; }
    if (s_key[0] == key) { ex_pop |= c_pop; if (!ex_lp) ex_lp = c_lp; }   // still the tile's first segment
    pg.epos[val] = e;
    pg.popbefore[e] = ex_pop;
    pg.prevpush[e] = ex_lp;
    pg.est[e] = 0u;
}
uint32_t pg_scan_tiles(uint32_t n_entries) { return (n_entries + PGS_TILE - 1) / PGS_TILE; }
void launch_pg_entry_scan(const PG& pg, const uint32_t* keys, const uint32_t* vals, uint32_t n, uint32_t n_bodies, hipStream_t s) {
    if (!n) return;
    const uint32_t nb = pg_scan_tiles(n);
    hipLaunchKernelGGL(k_pg_scan_tiles, dim3(nb), dim3(PGS_TILE), 0, s, pg, keys, vals, n);
    hipLaunchKernelGGL(k_pg_scan_apply, dim3(nb), dim3(PGS_TILE), 0, s, pg, keys, vals, n, n_bodies);
}

// ---- ConstraintGraph::push_manifold: the greedy colour of every push, in op order (dataflow) ----------------------------------
__device__ __forceinline__ uint32_t pg_pick_color(bool s1, bool s2, uint32_t mask1, uint32_t mask2) {
    if (!s1 && !s2) {   // lowest colour of 0..DYNAMIC_COLOR_COUNT-1 free on both bodies (constraint_graph.rs:181-196)
        const uint32_t free_bits = ~(mask1 | mask2) & ((1u << AVN_DYNAMIC_COLOR_COUNT) - 1u);
        return free_bits ? (uint32_t)__ffs((int)free_bits) - 1u : (uint32_t)AVN_COLOR_OVERFLOW_INDEX;
    }
    // next to a static body: highest free colour of COLOR_OVERFLOW_INDEX-1 .. 1 on the non-static body (:197-222)
    const uint32_t m = s1 ? mask2 : mask1;
    const uint32_t free_bits = ~m & (((1u << AVN_COLOR_OVERFLOW_INDEX) - 1u) & ~1u);
    return free_bits ? 31u - (uint32_t)__clz((int)free_bits) : (uint32_t)AVN_COLOR_OVERFLOW_INDEX;
}
__global__ __launch_bounds__(64) void k_pg_color(PG pg, uint32_t n_ops) {
    __shared__ uint32_t s_tile;
    const uint32_t lane = threadIdx.x;
    if (lane == 0) s_tile = atomicAdd(&pg.ctr[PGC_TILE], 1u);   // tiles are taken in start order: a lane only waits for lanes of running waves
    __syncthreads();
    const uint32_t k = s_tile * 64u + lane;
    uint32_t info = 0;
    if (k < n_ops) info = pg.op_info[k];
    const bool active = (info & 3u) == PG_KIND_PUSH;
    const bool s1 = info & 4u, s2 = info & 8u;
    uint32_t e[2] = {0, 0}, pp[2] = {0, 0}, base[2] = {0, 0};
    bool side[2] = {false, false};
    if (active) {
        const int2 b = pg.op_bodies[k];
        side[0] = !s1; side[1] = !s2;
        for (int s = 0; s < 2; ++s)
            if (side[s]) {
                e[s] = pg.epos[2 * k + s];
                pp[s] = pg.prevpush[e[s]];
                base[s] = pg.bcol[s ? b.y : b.x] & ~pg.popbefore[e[s]];   // the body's mask with this step's earlier pops applied
            }
    }
    bool done = !active;
    uint32_t color = 0xFFu;
    for (uint32_t it = 0;; ++it) {
        if (!done) {
            uint32_t taken[2] = {0, 0};
            bool ready = true;
            for (int s = 0; s < 2; ++s)
                if (side[s] && pp[s]) {
                    const uint32_t v = ld_agent(&pg.est[pp[s] - 1u]);
                    if (!(v & PG_EST_DONE)) ready = false;
                    taken[s] = v & ~PG_EST_DONE;
                }
            if (ready) {
                color = pg_pick_color(s1, s2, base[0] | taken[0], base[1] | taken[1]);
                const uint32_t bit = color < (uint32_t)AVN_COLOR_OVERFLOW_INDEX ? (1u << color) : 0u;   // the overflow colour has no body set
                for (int s = 0; s < 2; ++s)
                    if (side[s]) st_agent(&pg.est[e[s]], taken[s] | bit | PG_EST_DONE);
                pg.op_info[k] = (info & 0xFFu) | (color << 8);
                done = true;
            }
        }
        if (__all(done)) break;
        if (it > PG_SPIN_LIMIT) { if (lane == 0) atomicOr(&pg.ctr[PGC_ERROR], 1u); break; }
        __builtin_amdgcn_s_sleep(2);
    }
    // this step's pushes per colour (with the pops counted by k_pg_classify: the bucket sizes of the replay)
    for (uint32_t c = 0; c < AVN_GRAPH_COLOR_COUNT; ++c) {
        const unsigned long long m = __ballot(active && color == c);
        if (m && lane == 0) { atomicAdd(&pg.ctr[PGC_BUCKET + c], (uint32_t)__popcll(m)); atomicAdd(&pg.ctr[PGC_N_PUSH], (uint32_t)__popcll(m)); }
    }
}
void launch_pg_color(const PG& pg, uint32_t n_ops, hipStream_t s) {
    // (ctr[PGC_TILE] is zero here: k_pg_build_handles, the last kernel of every op batch, leaves it so)
    if (n_ops) hipLaunchKernelGGL(k_pg_color, dim3((n_ops + 63) / 64), dim3(64), 0, s, pg, n_ops);
}
// body masks after the step's ops: (mask & ~freed by pops) | taken by pushes -- a push only ever takes a bit that is free at its
// time and a pop only frees a bit an OLDER manifold holds (every contact id has at most one op per step), so the order inside
// the step does not matter for the final mask
__global__ __launch_bounds__(256) void k_pg_apply_masks(PG pg, const uint32_t* __restrict__ keys, const uint32_t* __restrict__ vals, uint32_t n, uint32_t n_bodies) {
    const uint32_t e = blockIdx.x * 256 + threadIdx.x;
    if (e >= n) return;
    const uint32_t key = keys[e];
    if (key >= n_bodies) return;
    if (e + 1 < n && keys[e + 1] == key) return;   // not the segment's last entry
    uint32_t own_pop, own_push;
    pg_entry_fields(pg, vals[e], own_pop, own_push);
    const uint32_t freed = pg.popbefore[e] | own_pop;
    const uint32_t lp = own_push ? e + 1 : pg.prevpush[e];
    const uint32_t taken = lp ? (pg.est[lp - 1u] & ~PG_EST_DONE) : 0u;
    pg.bcol[key] = (pg.bcol[key] & ~freed) | taken;
}
void launch_pg_apply_masks(const PG& pg, const uint32_t* keys, const uint32_t* vals, uint32_t n, uint32_t n_bodies, hipStream_t s) {
    if (n) hipLaunchKernelGGL(k_pg_apply_masks, dim3((n + 255) / 256), dim3(256), 0, s, pg, keys, vals, n, n_bodies);
}

// ---- per-colour op sequences + exact replay of push / swap_remove ----------------------------------------------------------------
// One wave per colour.  See tools/experiments/replay_batches.py for the rule: with h_t the list length before op t of a batch,
// a push writes position h_t and a pop vacates position h_t - 1 (its content fills the hole the popped handle leaves).  While
// every pop's handle sits BELOW the lowest position the batch's pushes / vacates touch, holes and moving tail never meet: the
// filler of a pop is the list entry at h_t - 1 as of the batch start, or the handle of the latest earlier push of the batch at
// that height, and all lanes apply their op at once.  The first op that breaks the rule runs alone, serially.
// The batch loop touches no global memory on its critical path: the ops of a chunk (RP_CHUNK) and the current positions of its pops
// are staged in LDS, so is a window of the list around its moving end, and a small LDS hash (pop handle -> index in the chunk)
// patches the staged position of a handle that moves before its own pop comes up.  Global stores are fire-and-forget; they are
// drained once per chunk / window reload, before the next staging loads.
#define RP_CHUNK 1024u
#define RP_HASH 2048u
#define RP_WIN 2048u
__global__ __launch_bounds__(64) void k_pg_replay(PG pg, const uint32_t* __restrict__ rx) {
    __shared__ uint32_t cx[RP_CHUNK], cP[RP_CHUNK], hkey[RP_HASH], hval[RP_HASH], win[RP_WIN], s_cons[64];
    // (the block is ONE wave: LDS operations execute in program order, so the phases below need no s_barrier -- and must not have
    //  __syncthreads(), whose fence would drain the scattered global stores of every batch, ~2 us each, before the next phase)
    const uint32_t c = blockIdx.x, lane = threadIdx.x;
    uint32_t b0 = 0;
    for (uint32_t i = 0; i < c; ++i) b0 += pg.ctr[PGC_BUCKET + i];
    const uint32_t b1 = b0 + pg.ctr[PGC_BUCKET + c];
    uint32_t* __restrict__ list = pg.lists + (size_t)c * pg.list_stride;
    uint32_t L = pg.ctr[PGC_LEN + c];
    uint32_t dbg_iter = 0, dbg_serial = 0, dbg_reload = 0;
    uint32_t wbase = 0; bool wvalid = false;   // the window: list[wbase .. wbase + RP_WIN) as the replay has left it so far
    auto hash_patch = [&](uint32_t y, uint32_t newP) {   // handle y moved to newP: if it is popped later in this chunk, that pop must find it there
        uint32_t hsl = (y * 2654435761u) >> 21;   // 11 bits
        for (;;) {
            const uint32_t k = hkey[hsl];
            if (k == y) { cP[hval[hsl]] = newP; return; }
            if (k == 0xFFFFFFFFu) return;
            hsl = (hsl + 1u) & (RP_HASH - 1u);
        }
    };
    for (uint32_t c0 = b0; c0 < b1; c0 += RP_CHUNK) {
        const uint32_t nC = min(RP_CHUNK, b1 - c0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the previous chunk's stores are performed before this chunk's positions are read
        for (uint32_t i = lane; i < RP_HASH; i += 64) hkey[i] = 0xFFFFFFFFu;
        __builtin_amdgcn_wave_barrier();
        {   // two memory levels for the whole chunk (ops, then the pops' positions), every load of a level in flight at once
            uint32_t vv[RP_CHUNK / 64], pp[RP_CHUNK / 64];
#pragma unroll
            for (uint32_t j = 0; j < RP_CHUNK / 64; ++j) { const uint32_t i = j * 64 + lane; vv[j] = i < nC ? rx[c0 + i] : 0x80000000u; }
#pragma unroll
            for (uint32_t j = 0; j < RP_CHUNK / 64; ++j) pp[j] = (vv[j] >> 31) ? 0u : ld_agent(&pg.lpos[vv[j]]);
#pragma unroll
            for (uint32_t j = 0; j < RP_CHUNK / 64; ++j) {
                const uint32_t i = j * 64 + lane, v = vv[j];
                if (i < nC) {
                    cx[i] = v;
                    if (!(v >> 31)) {
                        cP[i] = pp[j];
                        uint32_t hsl = (v * 2654435761u) >> 21;
                        for (;;) { const uint32_t prev = atomicCAS(&hkey[hsl], 0xFFFFFFFFu, v); if (prev == 0xFFFFFFFFu) break; hsl = (hsl + 1u) & (RP_HASH - 1u); }
                        hval[hsl] = i;
                    }
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        for (uint32_t cur = 0; cur < nC;) {
            const uint32_t n = min(64u, nC - cur);
            const bool valid = lane < n;
            const uint32_t v = valid ? cx[cur + lane] : 0u;
            const uint32_t x = v & 0x7FFFFFFFu;
            const bool push = valid && (v >> 31), pop = valid && !(v >> 31);
            const int delta = valid ? (push ? 1 : -1) : 0;
            // prefix sum of the +1 / -1 deltas from two ballots (no cross-lane scan: a ds_bpermute chain costs more than the whole batch)
            const unsigned long long pushes_all = __ballot(push), pops_all = __ballot(pop);
            const unsigned long long le_mask = lane == 63 ? ~0ull : ((2ull << lane) - 1ull);
            const int incl = __popcll(pushes_all & le_mask) - __popcll(pops_all & le_mask);
            const uint32_t h = (uint32_t)((int)L + incl - delta);              // list length before this lane's op
            const uint32_t touch = valid ? (push ? h : h - 1u) : 0xFFFFFFFFu;
            const uint32_t P = pop ? cP[cur + lane] : 0u;                      // ContactConstraintHandle::local_index, current
            // conflict-free prefix.  Every position the batch touches is >= L - 64, so a batch whose pops all sit below that is
            // conflict-free as a whole (the common case: one ballot); otherwise the exact rule with prefix min / max scans
            uint32_t f = n;
            if (__ballot(pop && P + 64u >= L)) {
                const uint32_t lomin = wave_incl_min(touch, lane), pmax = wave_incl_max(pop ? P + 1u : 0u, lane);
                const unsigned long long cf = __ballot(valid && pmax > lomin);
                f = cf ? (uint32_t)__ffsll((long long)cf) - 1u : n;   // ops [0, f) are conflict-free
            }
            // the window must hold every position this iteration reads or writes at the list's end (all within 64 of L).  Invariant:
            // win mirrors list[wbase .. current length) -- loaded below, or written since by a push / a hole fill
            if (!wvalid || L < wbase + 64u || L + 64u >= wbase + RP_WIN) {
                if (!wvalid || (wbase != 0u && L < wbase + 64u) || L + 64u >= wbase + RP_WIN) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // earlier iterations' list stores are performed before the reload
                    wbase = L > RP_WIN / 2u ? L - RP_WIN / 2u : 0u;
                    const uint32_t wend = min(L, wbase + RP_WIN);
                    for (uint32_t i0 = 0; wbase + i0 < wend; i0 += 16 * 64) {   // 16 loads per lane in flight
                        uint32_t t[16];
#pragma unroll
                        for (uint32_t j = 0; j < 16; ++j) { const uint32_t i = i0 + j * 64 + lane; t[j] = wbase + i < wend ? ld_agent(&list[wbase + i]) : 0u; }
#pragma unroll
                        for (uint32_t j = 0; j < 16; ++j) { const uint32_t i = i0 + j * 64 + lane; if (wbase + i < wend) win[i] = t[j]; }
                    }
                    wvalid = true; ++dbg_reload;
                    __builtin_amdgcn_wave_barrier();
                }
            }
            ++dbg_iter;
            if (f == 0) {   // the batch's first op alone: the serial statement of the reference
                ++dbg_serial;
                if (lane == 0) {
                    if (push) { st_async(&list[L], x); st_async(&pg.lpos[x], L); st_async(&pg.color[x], c); win[L - wbase] = x; }
                    else {
                        const uint32_t last = win[L - 1u - wbase];
                        if (P != L - 1u) {   // swap_remove + "fix moved manifold handle"
                            st_async(&list[P], last); st_async(&pg.lpos[last], P);
                            if (P >= wbase) win[P - wbase] = last;
                            hash_patch(last, P);
                        }
                        st_async(&pg.color[x], PG_NONE);
                    }
                }
                L = (uint32_t)((int)L + __shfl(delta, 0));
                cur += 1;
                __builtin_amdgcn_wave_barrier();
                continue;
            }
            const bool in = lane < f;
            // stack matching: the latest earlier push of the batch that wrote the position this pop vacates
            int match = -1;
            const unsigned long long fmask = f == 64 ? ~0ull : ((1ull << f) - 1ull);
            const unsigned long long pushes = pushes_all & fmask, pops = pops_all & fmask;
            if (pushes && pops) {
                const uint32_t s_lo = (uint32_t)__ffsll((long long)pushes) - 1u, s_hi = 63u - (uint32_t)__clzll((long long)pops);   // only pushes before the last pop can match
                for (uint32_t s = s_lo; s < s_hi; ++s) {
                    if (!((pushes >> s) & 1ull)) continue;   // (wave-uniform)
                    const uint32_t hs = (uint32_t)__builtin_amdgcn_readlane((int)h, (int)s);
                    if (in && pop && s < lane && hs == h - 1u) match = (int)s;
                }
            }
            s_cons[lane] = 0u;
            __builtin_amdgcn_wave_barrier();
            if (in && pop && match >= 0) s_cons[match] = 1u;
            const uint32_t ym = (uint32_t)__shfl((int)x, match >= 0 ? match : 0);
            // every filler is read from the list AS OF THE BATCH START: all reads of all lanes before any lane writes
            uint32_t y = ym;
            if (in && pop && match < 0) y = win[h - 1u - wbase];
            __builtin_amdgcn_wave_barrier();
            if (in && pop) {
                st_async(&list[P], y); st_async(&pg.lpos[y], P);
                st_async(&pg.color[x], PG_NONE);
                if (P >= wbase) win[P - wbase] = y;
                if (match < 0) hash_patch(y, P);   // (a handle pushed in this step has no pop in it)
            } else if (in && push) {
                st_async(&pg.color[x], c);
                if (!s_cons[lane]) { st_async(&list[h], x); st_async(&pg.lpos[x], h); win[h - wbase] = x; }   // (a consumed push's handle already moved into a pop's hole)
            }
            L = (uint32_t)__shfl((int)h + delta, (int)(f - 1u));
            cur += f;
            __builtin_amdgcn_wave_barrier();
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) { pg.ctr[PGC_LEN + c] = L; pg.ctr[PGC_DBG + c] = dbg_iter; pg.ctr[PGC_DBG + 24 + c] = dbg_serial; pg.ctr[PGC_DBG + 48 + c] = dbg_reload; pg.ctr[PGC_DBG + 72 + c] = b1 - b0; }
}
// The same replay RW_B ops at a time by a whole workgroup (round 3; tools/experiments/replay_wide.py is its CPU model, checked against the
// serial semantics).  The wave version above spends ~4 us per 64 ops -- its stack matching walks the pushes of the batch one readlane at a
// time -- and a settled pile's largest colour has 4 600 ops per step (15 000 while the pile collapses): 0.3 - 1.6 ms with 23 of 24 waves
// long finished.  Per batch, with H_t the list length after op t and L the length at the batch start:
//   * a pop at t vacates level q = H_t; what sits there is the handle of the push s = 1 + (largest u < t with H_u <= q) -- lengths move by
//     +-1, so that op is the push that last went up through q --, or of op 0 if there is no such u and L <= q, else the list entry at q
//     as of the batch start.  One descent in a min-tree over H per pop instead of a walk over the pushes.
//   * at 1024 ops a filler taken from the list is popped LATER IN THE SAME BATCH a dozen times per batch (the wave version cut the batch
//     there): that pop finds it in the hole it was moved to -- pred[j] = the pop whose filler op j pops, P_j = P_pred through pointer
//     jumping --, and the placement of a filler that is popped later in the batch is never written.
//   * exact while every pop's resolved position is below the lowest level touched up to that op (checked with block-wide prefix min / max
//     when a quick whole-batch test fails); the first op that breaks it runs alone and the batch restarts behind it.
// Between batches the stores are drained and everything is re-read from memory (agent-scope loads): no staged window, no patching.
#define RW_B 1024u
#define RW_HASH 2048u
__global__ __launch_bounds__(RW_B) void k_pg_replay_wide(PG pg, const uint32_t* __restrict__ rx) {
    __shared__ int T[2 * RW_B];   // min-tree over H: leaves at [RW_B, 2 RW_B)
    __shared__ uint32_t s_x[RW_B], s_P[RW_B], s_flag[RW_B], hkey[RW_HASH], hval[RW_HASH];
    __shared__ int s_pred[RW_B];
    __shared__ int s_wsum[RW_B / 64];
    __shared__ uint32_t s_wmin[RW_B / 64], s_wmax[RW_B / 64];
    __shared__ uint32_t s_f, s_L;
    const uint32_t c = blockIdx.x, tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
    uint32_t b0 = 0;
    for (uint32_t i = 0; i < c; ++i) b0 += pg.ctr[PGC_BUCKET + i];
    const uint32_t b1 = b0 + pg.ctr[PGC_BUCKET + c];
    uint32_t* __restrict__ list = pg.lists + (size_t)c * pg.list_stride;
    uint32_t L = pg.ctr[PGC_LEN + c];
    uint32_t dbg_iter = 0, dbg_serial = 0, dbg_cut = 0;
    uint32_t nlim = 0;
    for (uint32_t cur = b0; cur < b1;) {
        uint32_t n = min(RW_B, b1 - cur);
        if (nlim) { n = nlim; nlim = 0; }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the previous batch's stores are performed ...
        __syncthreads();                                    // ... by every wave, before anything is read back
        const bool valid = tid < n;
        const uint32_t v = valid ? rx[cur + tid] : 0u;
        const uint32_t x = v & 0x7FFFFFFFu;
        const bool push = valid && (v >> 31), pop = valid && !(v >> 31);
        const uint32_t P0 = pop ? ld_agent(&pg.lpos[x]) : 0u;
        // lengths: prefix sum of +1 / -1 (ballots inside a wave, wave totals through LDS)
        const unsigned long long pushes_w = __ballot(push), pops_w = __ballot(pop);
        const unsigned long long le_mask = lane == 63 ? ~0ull : ((2ull << lane) - 1ull);
        int incl = __popcll(pushes_w & le_mask) - __popcll(pops_w & le_mask);
        if (lane == 0) s_wsum[wv] = __popcll(pushes_w) - __popcll(pops_w);
        s_x[tid] = x; s_P[tid] = P0; s_flag[tid] = 0u; s_pred[tid] = -1;
        hkey[tid] = 0xFFFFFFFFu; hkey[tid + RW_B] = 0xFFFFFFFFu; hval[tid] = 0xFFFFFFFFu; hval[tid + RW_B] = 0xFFFFFFFFu;
        __syncthreads();
        for (uint32_t k = 0; k < wv; ++k) incl += s_wsum[k];
        const int delta = valid ? (push ? 1 : -1) : 0;
        const int H = (int)L + incl;                       // length after this op
        const uint32_t h = (uint32_t)(H - delta);          // ... and before it
        T[RW_B + tid] = valid ? H : 0x7FFFFFFF;
        // the tree: every wave builds the six levels above its 64 leaves (LDS operations of a wave execute in order), one wave the top four
        for (uint32_t size = RW_B / 2; size >= RW_B / 64; size >>= 1) {
            const uint32_t per = size / (RW_B / 64);
            __builtin_amdgcn_wave_barrier();
            if (lane < per) { const uint32_t node = size + wv * per + lane; T[node] = min(T[2 * node], T[2 * node + 1]); }
        }
        __syncthreads();
        if (wv == 0)
            for (uint32_t size = RW_B / 128; size >= 1; size >>= 1) {
                __builtin_amdgcn_wave_barrier();
                if (lane < size) { const uint32_t node = size + lane; T[node] = min(T[2 * node], T[2 * node + 1]); }
            }
        __syncthreads();
        // fillers
        int match = -1;
        uint32_t y = 0;
        if (pop) {
            const int q = H;
            uint32_t node = RW_B + tid;
            int u = -1;
            while (node > 1u) {
                if ((node & 1u) && T[node - 1u] <= q) {   // the left sibling's range ends right in front of ours and holds a length <= q: its last one
                    node -= 1u;
                    while (node < RW_B) node = T[2u * node + 1u] <= q ? 2u * node + 1u : 2u * node;
                    u = (int)(node - RW_B);
                    break;
                }
                node >>= 1;
            }
            if (u >= 0) match = u + 1;
            else if ((int)L <= q) match = 0;
            if (match >= 0) { y = s_x[match]; atomicOr(&s_flag[match], 1u); }   // that push's handle goes straight into this pop's hole
            else {
                y = ld_agent(&list[q]);
                uint32_t hsl = (y * 2654435761u) >> 21;   // 11 bits
                for (;;) {
                    const uint32_t prev = atomicCAS(&hkey[hsl], 0xFFFFFFFFu, y);
                    if (prev == 0xFFFFFFFFu || prev == y) { atomicMin(&hval[hsl], tid); break; }   // (the FIRST pop that moves a handle: lanes behind a cut may hold stale duplicates)
                    hsl = (hsl + 1u) & (RW_HASH - 1u);
                }
            }
        }
        __syncthreads();
        // a pop of a handle that an earlier pop of this batch moved
        int pred0 = -1;
        if (pop) {
            uint32_t hsl = (x * 2654435761u) >> 21;
            for (;;) {
                const uint32_t k = hkey[hsl];
                if (k == x) { const uint32_t i = hval[hsl]; if (i < tid) pred0 = (int)i; break; }
                if (k == 0xFFFFFFFFu) break;
                hsl = (hsl + 1u) & (RW_HASH - 1u);
            }
            s_pred[tid] = pred0;
        }
        __syncthreads();
        if (pred0 >= 0) atomicOr(&s_flag[pred0], 2u);   // its placement is dead: this pop's filler takes the slot
        int root = pred0;
        for (;;) {   // pointer jumping to the pop whose position was read from memory
            int nxt = -1;
            if (root >= 0) nxt = s_pred[root];
            const int more = __syncthreads_or(nxt >= 0);
            if (!more) break;
            if (nxt >= 0) { root = nxt; s_pred[tid] = root; }
            __syncthreads();
        }
        const uint32_t P = root >= 0 ? s_P[root] : P0;
        // conflict-free prefix
        const uint32_t touch = valid ? (push ? h : h - 1u) : 0xFFFFFFFFu;
        const uint32_t pend = pop ? P + 1u : 0u;
        uint32_t tmin = touch, pmx = pend;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { tmin = min(tmin, (uint32_t)__shfl_xor((int)tmin, off)); pmx = max(pmx, (uint32_t)__shfl_xor((int)pmx, off)); }
        if (lane == 0) { s_wmin[wv] = tmin; s_wmax[wv] = pmx; }
        if (tid == 0) s_f = n;
        __syncthreads();
        uint32_t all_min = 0xFFFFFFFFu, all_max = 0u;
        for (uint32_t k = 0; k < RW_B / 64; ++k) { all_min = min(all_min, s_wmin[k]); all_max = max(all_max, s_wmax[k]); }
        if (all_max > all_min) {   // (block-uniform) rare: the exact rule, prefix min of the touched levels against prefix max of the pops' positions
            uint32_t lomin = wave_incl_min(touch, lane), pmax = wave_incl_max(pend, lane);
            for (uint32_t k = 0; k < wv; ++k) { lomin = min(lomin, s_wmin[k]); pmax = max(pmax, s_wmax[k]); }
            if (valid && pmax > lomin) atomicMin(&s_f, tid);
            __syncthreads();
        }
        const uint32_t f = s_f;
        ++dbg_iter;
        if (f == 0u) {   // the batch's first op alone: the serial statement of the reference
            ++dbg_serial;
            if (tid == 0) {
                if (push) { st_async(&list[L], x); st_async(&pg.lpos[x], L); st_async(&pg.color[x], c); s_L = L + 1u; }
                else {
                    if (P0 != L - 1u) { const uint32_t last = ld_agent(&list[L - 1u]); st_async(&list[P0], last); st_async(&pg.lpos[last], P0); }   // swap_remove + "fix moved manifold handle"
                    st_async(&pg.color[x], PG_NONE);
                    s_L = L - 1u;
                }
            }
            __syncthreads();
            L = s_L;
            cur += 1;
            continue;
        }
        if (f < n) { ++dbg_cut; nlim = f; continue; }   // the same ops again without the tail behind the cut (its lanes marked pushes consumed, fillers dead)
        if (pop) {
            if (!(s_flag[tid] & 2u)) { st_async(&list[P], y); st_async(&pg.lpos[y], P); }
            st_async(&pg.color[x], PG_NONE);
        } else if (push) {
            st_async(&pg.color[x], c);
            if (!(s_flag[tid] & 1u)) { st_async(&list[h], x); st_async(&pg.lpos[x], h); }
        }
        if (tid == n - 1u) s_L = (uint32_t)H;
        __syncthreads();
        L = s_L;
        cur += n;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (tid == 0) { pg.ctr[PGC_LEN + c] = L; pg.ctr[PGC_DBG + c] = dbg_iter; pg.ctr[PGC_DBG + 24 + c] = dbg_serial; pg.ctr[PGC_DBG + 48 + c] = dbg_cut; pg.ctr[PGC_DBG + 72 + c] = b1 - b0; }
}
// The ops of every colour, in op order, as one contiguous stream per colour (rx[b0 .. b1) with b0 = the bucket counts before the colour): a stable
// partition by colour in ONE launch (round 4: was k_pg_bucket_keys + a radix pass (histogram, scatter) + k_pg_replay_gather).  One workgroup
// per colour walks the ops 1 024 at a time and keeps its own: ballot ranks inside a wave, wave totals through LDS, a running count.
__global__ __launch_bounds__(1024) void k_pg_partition_colors(PG pg, uint32_t* __restrict__ rx, uint32_t n_ops) {
    __shared__ uint32_t s_w[16];
    const uint32_t c = blockIdx.x, tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
    uint32_t run = 0;
    for (uint32_t i = 0; i < c; ++i) run += pg.ctr[PGC_BUCKET + i];
    for (uint32_t k0 = 0; k0 < n_ops; k0 += 1024u) {
        const uint32_t k = k0 + tid;
        uint32_t info = 0;
        if (k < n_ops) info = pg.op_info[k];
        const uint32_t kind = info & 3u;
        const bool mine = kind != PG_KIND_NONE && ((info >> 8) & 0xFFu) == c;
        const unsigned long long m = __ballot(mine);
        if (lane == 0) s_w[wv] = (uint32_t)__popcll(m);
        __syncthreads();
        uint32_t before = 0, all = 0;
#pragma unroll
        for (uint32_t w = 0; w < 16u; ++w) { const uint32_t v = s_w[w]; if (w < wv) before += v; all += v; }
        if (mine) rx[run + before + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = pg.op_cid[k] | (kind == PG_KIND_PUSH ? 0x80000000u : 0u);
        run += all;
        __syncthreads();
    }
}
void launch_pg_replay(const PG& pg, uint32_t n_ops, hipStream_t s) {
    if (n_ops) hipLaunchKernelGGL(k_pg_partition_colors, dim3(AVN_GRAPH_COLOR_COUNT), dim3(1024), 0, s, pg, pg.ekey_a, n_ops);   // (ekey_a: the colouring's entry buffers are free again)
    static const bool wave_version = avn_env("AVN_PG_REPLAY_WAVE") && avn_env("AVN_PG_REPLAY_WAVE")[0] && avn_env("AVN_PG_REPLAY_WAVE")[0] != '0';   // (A/B runs)
    if (wave_version) hipLaunchKernelGGL(k_pg_replay, dim3(AVN_GRAPH_COLOR_COUNT), dim3(64), 0, s, pg, pg.ekey_a);
    else hipLaunchKernelGGL(k_pg_replay_wide, dim3(AVN_GRAPH_COLOR_COUNT), dim3(RW_B), 0, s, pg, pg.ekey_a);
}

// ---- removed pairs: ContactGraph::remove_edge_by_id + IdPool::free_id --------------------------------------------------------------
template <class T>
__global__ __launch_bounds__(256) void k_pg_remove(PG pg, CT<T> ct, BP<T> bp, uint32_t n_ops) {
    const uint32_t k = blockIdx.x * 256 + threadIdx.x;
    if (k >= n_ops || !pg.rem_flag[k]) return;
    const uint32_t c = pg.op_cid[k];
    pg.rem_ids[pg.rem_off[k]] = c;   // ascending: ops are in id order
    const uint4 meta = ct.meta[c];
    const uint32_t a = bp.col_info[meta.x].x, b = bp.col_info[meta.y].x;   // collider entities -> PairKey
    const uint64_t key = a < b ? ((uint64_t)a << 32) | b : ((uint64_t)b << 32) | a;
    ct.meta[c] = make_uint4(0u, 0u, 0u, 0u);
    ct.dcount[c] = 0;
    pg.color[c] = PG_NONE;
    if (bp.pair_set_cap) {
        const uint32_t mask = bp.pair_set_cap - 1u;
        uint32_t h = (uint32_t)pg_hs_mix(key) & mask;
        for (;;) {
            const uint64_t v = bp.pair_set[h];
            if (v == key) { bp.pair_set[h] = ~0ull - 1ull; break; }   // tombstone
            if (v == ~0ull) break;
            h = (h + 1u) & mask;
        }
    }
}
template <class T> void launch_pg_remove(const PG& pg, const CT<T>& ct, const BP<T>& bp, uint32_t n_ops, hipStream_t s) {
    if (n_ops) hipLaunchKernelGGL(k_pg_remove<T>, dim3((n_ops + 255) / 256), dim3(256), 0, s, pg, ct, bp, n_ops);
}
// free list (ascending) <- merge(live part of the free list, removed ids (ascending)); both inputs are duplicate-free and disjoint
__global__ __launch_bounds__(256) void k_pg_merge_free(PG pg, uint32_t head, uint32_t n_free, uint32_t n_rem) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    const uint32_t* __restrict__ A = pg.free_ids + head;
    const uint32_t* __restrict__ B = pg.rem_ids;
    if (i < n_free) {
        const uint32_t v = A[i];
        uint32_t lo = 0, hi = n_rem;
        while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (B[mid] < v) lo = mid + 1; else hi = mid; }
        pg.free_alt[i + lo] = v;
    } else if (i < n_free + n_rem) {
        const uint32_t j = i - n_free, v = B[j];
        uint32_t lo = 0, hi = n_free;
        while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (A[mid] < v) lo = mid + 1; else hi = mid; }
        pg.free_alt[j + lo] = v;
    }
    if (i == 0) { pg.ctr[PGC_FREE_HEAD] = 0; pg.ctr[PGC_N_FREE] = n_free + n_rem; }
}
void launch_pg_merge_free(const PG& pg, uint32_t head, uint32_t n_free, uint32_t n_rem, hipStream_t s) {
    hipLaunchKernelGGL(k_pg_merge_free, dim3((n_free + n_rem + 255) / 256 + 1), dim3(256), 0, s, pg, head, n_free, n_rem);
}

// ---- avn_despawn: the device half (world/despawn.hpp) -------------------------------------------------------------------------------
template <class T>
__global__ __launch_bounds__(256) void k_pg_collect_edges(PG pg, CT<T> ct, uint32_t n_rows, const uint32_t* __restrict__ rm_rank, PGEdgeRec* __restrict__ out, uint32_t cap) {
    const uint32_t c = blockIdx.x * 256 + threadIdx.x;
    if (c >= n_rows) return;
    const uint4 meta = ct.meta[c];
    if (!(meta.z & AVN_CP_ROW_USED)) return;
    if (rm_rank[meta.x] == PG_NONE && rm_rank[meta.y] == PG_NONE) return;
    const uint32_t k = atomicAdd(&pg.ctr[PGC_COLLECT], 1u);
    if (k >= cap) return;   // (the host sees the count and retries with room)
    const unsigned long long sq = pg.seq[c];
    out[k] = PGEdgeRec{c, meta.x, meta.y, meta.z, pg.color[c], (uint32_t)sq, (uint32_t)(sq >> 32), 0u};
}
template <class T> void launch_pg_collect_edges(const PG& pg, const CT<T>& ct, uint32_t n_rows, const uint32_t* rm_rank, PGEdgeRec* out, uint32_t cap, hipStream_t s) {
    (void)hipMemsetAsync(pg.ctr + PGC_COLLECT, 0, 4, s);
    if (n_rows) hipLaunchKernelGGL(k_pg_collect_edges<T>, dim3((n_rows + 255) / 256), dim3(256), 0, s, pg, ct, n_rows, rm_rank, out, cap);
}
template <class T>
__global__ __launch_bounds__(256) void k_pg_remove_list(PG pg, CT<T> ct, BP<T> bp, const uint32_t* __restrict__ ids, uint32_t n) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint32_t c = ids[i];
    pg.rem_ids[i] = c;
    const uint4 meta = ct.meta[c];
    const uint32_t a = bp.col_info[meta.x].x, b = bp.col_info[meta.y].x;   // collider entities -> PairKey
    const uint64_t key = a < b ? ((uint64_t)a << 32) | b : ((uint64_t)b << 32) | a;
    ct.meta[c] = make_uint4(0u, 0u, 0u, 0u);
    ct.dcount[c] = 0;
    pg.color[c] = PG_NONE;
    if (bp.pair_set_cap) {
        const uint32_t mask = bp.pair_set_cap - 1u;
        uint32_t h = (uint32_t)pg_hs_mix(key) & mask;
        for (;;) {
            const uint64_t v = bp.pair_set[h];
            if (v == key) { bp.pair_set[h] = ~0ull - 1ull; break; }   // tombstone
            if (v == ~0ull) break;
            h = (h + 1u) & mask;
        }
    }
}
template <class T> void launch_pg_remove_list(const PG& pg, const CT<T>& ct, const BP<T>& bp, const uint32_t* ids, uint32_t n, hipStream_t s) {
    if (n) hipLaunchKernelGGL(k_pg_remove_list<T>, dim3((n + 255) / 256), dim3(256), 0, s, pg, ct, bp, ids, n);
}
template <class T>
__global__ __launch_bounds__(256) void k_pg_renumber_rows(PG pg, CT<T> ct, uint32_t n_rows, const uint32_t* __restrict__ new_index) {
    const uint32_t c = blockIdx.x * 256 + threadIdx.x;
    if (c >= n_rows || !(ct.meta[c].z & AVN_CP_ROW_USED)) return;
    const int2 b = pg.bodies[c];
    pg.bodies[c] = make_int2((int)new_index[b.x], (int)new_index[b.y]);
}
template <class T> void launch_pg_renumber_rows(const PG& pg, const CT<T>& ct, uint32_t n_rows, const uint32_t* new_index, hipStream_t s) {
    if (n_rows) hipLaunchKernelGGL(k_pg_renumber_rows<T>, dim3((n_rows + 255) / 256), dim3(256), 0, s, pg, ct, n_rows, new_index);
}
__global__ __launch_bounds__(256) void k_renumber_int2(int2* __restrict__ v, uint32_t n, const uint32_t* __restrict__ new_index) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int2 b = v[i];
    v[i] = make_int2((int)new_index[b.x], (int)new_index[b.y]);
}
void launch_renumber_int2(int2* v, uint32_t n, const uint32_t* new_index, hipStream_t s) { if (n) hipLaunchKernelGGL(k_renumber_int2, dim3((n + 255) / 256), dim3(256), 0, s, v, n, new_index); }
template <class U> __global__ __launch_bounds__(256) void k_compact(const U* __restrict__ src, U* __restrict__ dst, const uint32_t* __restrict__ new_index, uint32_t n_old) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_old) return;
    const uint32_t j = new_index[i];
    if (j != PG_NONE) dst[j] = src[i];
}
void launch_compact_u32(const uint32_t* src, uint32_t* dst, const uint32_t* new_index, uint32_t n_old, hipStream_t s) { if (n_old) hipLaunchKernelGGL(k_compact<uint32_t>, dim3((n_old + 255) / 256), dim3(256), 0, s, src, dst, new_index, n_old); }
void launch_compact_u8(const uint8_t* src, uint8_t* dst, const uint32_t* new_index, uint32_t n_old, hipStream_t s) { if (n_old) hipLaunchKernelGGL(k_compact<uint8_t>, dim3((n_old + 255) / 256), dim3(256), 0, s, src, dst, new_index, n_old); }

// ---- GraphColor::manifold_handles of all colours, concatenated colour-major (what the solver's arrays are ordered by) ---------
// Round 5: the SOLVER's order inside colours 0..22 is a second, body-sorted one.  Inside such a colour no two manifolds share a non-static
// body (constraint_graph.rs:36-48), so every order gives the same bits; only the bookkeeping (PG::lists, swap_remove) needs the reference's
// history order, and it keeps it.  The handle lists of a pile that has been through 10^5 pops and pushes name rows in no order at all: every
// 16-byte record the warm start gathered cost a line (248.6 MB fetched per launch against 65.3 MB of records, r04_pmc_closed_loop_settled.json),
// the colour passes' body gathers and the constraint generation's likewise.  Sorted by KEY BODY (body1, or body2 next to a static body1) the
// neighbours of a manifold in its colour belong to neighbouring bodies, as in a freshly uploaded manifold set.
// No sort is needed: a colour holds at most one manifold per non-static body, so `tab[colour][key body] = ContactId` is a collision-free
// scatter and the sorted list is the table's non-empty entries in order -- k_pg_build_handles scatters (the overflow colour keeps its list
// order and is written directly), k_pg_sort_count counts the entries of every 2 048-body chunk of every colour, k_pg_sort_emit turns the
// counts in front of its chunk into its base and compacts (and leaves the table EMPTY again: no memset per step).  Should two manifolds of a
// colour ever name the same key body (the colouring's invariant broken, e.g. by a body that changed its type under a live contact), the
// loser of the compare-and-swap goes to the END of the colour's range (ctr[PGC_SORT_DUP + colour]): nothing is lost.
__global__ __launch_bounds__(256) void k_pg_build_handles(PG pg, uint32_t* __restrict__ handles, uint32_t* __restrict__ color_offsets, uint32_t total,
                                                          const uint4* __restrict__ ct_meta, uint32_t* __restrict__ tab, uint32_t tab_stride, uint32_t lens_at) {
    __shared__ uint32_t off[AVN_GRAPH_COLOR_COUNT + 1];
    if (threadIdx.x == 0) {
        uint32_t a = 0;
        for (uint32_t c = 0; c < AVN_GRAPH_COLOR_COUNT; ++c) { off[c] = a; a += pg.ctr[lens_at + c]; }
        off[AVN_GRAPH_COLOR_COUNT] = a;
    }
    __syncthreads();
    if (blockIdx.x == 0 && threadIdx.x <= AVN_GRAPH_COLOR_COUNT) { color_offsets[threadIdx.x] = off[threadIdx.x]; pg.ctr[PGC_OFFSETS + threadIdx.x] = off[threadIdx.x]; }
    // the op batch is over: its scoped counters start the next batch at zero without a memset launch (buckets of the replay, the dataflow
    // colouring's tile tickets, the narrow phase's removal count -- the host has read them)
    if (blockIdx.x == 0 && threadIdx.x < 32u) pg.ctr[PGC_BUCKET + threadIdx.x] = 0u;
    if (blockIdx.x == 0 && threadIdx.x == 32u) { pg.ctr[PGC_TILE] = 0u; pg.ctr[PGC_N_REM] = 0u; pg.ctr[PGC_N_SLEEP_OPS] = 0u; }
    const uint32_t m = blockIdx.x * 256 + threadIdx.x;
    if (m >= total || m >= off[AVN_GRAPH_COLOR_COUNT]) return;
    uint32_t lo = 0, hi = AVN_GRAPH_COLOR_COUNT;   // colour c: off[c] <= m < off[c + 1]
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (off[mid] <= m) lo = mid; else hi = mid; }
    const uint32_t cid = pg.lists[(size_t)lo * pg.list_stride + (m - off[lo])];
    if (!tab || lo == (uint32_t)AVN_COLOR_OVERFLOW_INDEX) { handles[m] = cid; return; }   // list order: the overflow colour is solved serially in it
    const int2 b = pg.bodies[cid];
    const uint32_t kb = (ct_meta[cid].z & AVN_CP_STATIC1) ? (uint32_t)b.y : (uint32_t)b.x;
    if (kb >= tab_stride || atomicCAS(&tab[(size_t)lo * tab_stride + kb], PG_NONE, cid) != PG_NONE) {
        const uint32_t k = atomicAdd(&pg.ctr[PGC_SORT_DUP + lo], 1u);
        handles[off[lo + 1] - 1u - k] = cid;
    }
}
#define PG_SORT_CHUNK 2048u
__global__ __launch_bounds__(256) void k_pg_sort_count(PG pg, const uint32_t* __restrict__ tab, uint32_t tab_stride, uint32_t n_chunks, uint32_t* __restrict__ cnt) {
    __shared__ uint32_t ws[4];
    const uint32_t chunk = blockIdx.x, c = blockIdx.y, t = threadIdx.x;
    const uint4* p = reinterpret_cast<const uint4*>(tab + (size_t)c * tab_stride + (size_t)chunk * PG_SORT_CHUNK) + 2u * t;
    const uint4 a = p[0], b = p[1];
    uint32_t s = (a.x != PG_NONE) + (a.y != PG_NONE) + (a.z != PG_NONE) + (a.w != PG_NONE) + (b.x != PG_NONE) + (b.y != PG_NONE) + (b.z != PG_NONE) + (b.w != PG_NONE);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += (uint32_t)__shfl_xor((int)s, o);
    if ((t & 63u) == 0u) ws[t >> 6] = s;
    __syncthreads();
    if (t == 0u) {
        cnt[c * n_chunks + chunk] = (ws[0] + ws[1]) + (ws[2] + ws[3]);
        if (chunk == 0u) pg.ctr[PGC_SORT_DUP + c] = 0u;   // (k_pg_build_handles of this batch is done with it; the next batch starts at zero)
    }
}
__global__ __launch_bounds__(256) void k_pg_sort_emit(PG pg, uint32_t* __restrict__ tab, uint32_t tab_stride, uint32_t n_chunks, const uint32_t* __restrict__ cnt, uint32_t* __restrict__ handles) {
    __shared__ uint32_t ws[4];
    __shared__ uint32_t s_base;
    const uint32_t chunk = blockIdx.x, c = blockIdx.y, t = threadIdx.x;
    if (cnt[c * n_chunks + chunk] == 0u) return;   // (block-uniform: an empty chunk has nothing to write and nothing to clean)
    uint32_t before = 0u;
    for (uint32_t j = t; j < chunk; j += 256u) before += cnt[c * n_chunks + j];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) before += (uint32_t)__shfl_xor((int)before, o);
    if ((t & 63u) == 0u) ws[t >> 6] = before;
    __syncthreads();
    if (t == 0u) s_base = pg.ctr[PGC_OFFSETS + c] + ((ws[0] + ws[1]) + (ws[2] + ws[3]));
    __syncthreads();   // (ws is reused by sc_block_excl, which starts with its own writes after this barrier)
    uint4* p = reinterpret_cast<uint4*>(tab + (size_t)c * tab_stride + (size_t)chunk * PG_SORT_CHUNK) + 2u * t;
    const uint4 a = p[0], b = p[1];
    const uint32_t v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    uint32_t s = 0u;
#pragma unroll
    for (uint32_t k = 0; k < 8u; ++k) s += v[k] != PG_NONE;
    uint32_t tile_sum;
    uint32_t pos = s_base + sc_block_excl(s, &tile_sum);
    if (s) {
#pragma unroll
        for (uint32_t k = 0; k < 8u; ++k) if (v[k] != PG_NONE) handles[pos++] = v[k];
        p[0] = make_uint4(PG_NONE, PG_NONE, PG_NONE, PG_NONE); p[1] = make_uint4(PG_NONE, PG_NONE, PG_NONE, PG_NONE);
    }
}
uint32_t pg_sort_stride(uint32_t n_bodies) { return ((n_bodies + PG_SORT_CHUNK - 1u) / PG_SORT_CHUNK) * PG_SORT_CHUNK; }
void launch_pg_build_handles(const PG& pg, uint32_t* handles, uint32_t* color_offsets, uint32_t total, const uint4* ct_meta, uint32_t* sort_tab, uint32_t* sort_cnt, uint32_t n_bodies, hipStream_t s, uint32_t lens_at) {
    const uint32_t stride = pg_sort_stride(n_bodies), n_chunks = stride / PG_SORT_CHUNK;
    hipLaunchKernelGGL(k_pg_build_handles, dim3((total + 255) / 256 + 1), dim3(256), 0, s, pg, handles, color_offsets, total, ct_meta, sort_tab, stride, lens_at);
    if (!sort_tab || !n_chunks) return;
    hipLaunchKernelGGL(k_pg_sort_count, dim3(n_chunks, AVN_COLOR_OVERFLOW_INDEX), dim3(256), 0, s, pg, sort_tab, stride, n_chunks, sort_cnt);
    hipLaunchKernelGGL(k_pg_sort_emit, dim3(n_chunks, AVN_COLOR_OVERFLOW_INDEX), dim3(256), 0, s, pg, sort_tab, stride, n_chunks, sort_cnt, handles);
}

// ---- ContactGraph::pair_set rebuilt from the live rows (after growth, or when tombstones pile up) ------------------------------
template <class T>
__global__ __launch_bounds__(256) void k_pg_rebuild_pair_set(CT<T> ct, BP<T> bp, uint32_t n_rows) {
    const uint32_t c = blockIdx.x * 256 + threadIdx.x;
    if (c >= n_rows) return;
    const uint4 meta = ct.meta[c];
    if (!(meta.z & AVN_CP_ROW_USED)) return;
    const uint32_t a = bp.col_info[meta.x].x, b = bp.col_info[meta.y].x;
    const uint64_t key = a < b ? ((uint64_t)a << 32) | b : ((uint64_t)b << 32) | a;
    const uint32_t mask = bp.pair_set_cap - 1u;
    uint32_t h = (uint32_t)pg_hs_mix(key) & mask;
    for (;;) {
        const unsigned long long prev = atomicCAS((unsigned long long*)&bp.pair_set[h], ~0ull, (unsigned long long)key);
        if (prev == ~0ull || prev == key) return;
        h = (h + 1u) & mask;
    }
}
template <class T> void launch_pg_rebuild_pair_set(const CT<T>& ct, const BP<T>& bp, uint32_t n_rows, hipStream_t s) {
    if (n_rows) hipLaunchKernelGGL(k_pg_rebuild_pair_set<T>, dim3((n_rows + 255) / 256), dim3(256), 0, s, ct, bp, n_rows);
}

// ---- the overflow colour on the device ------------------------------------------------------------------------------------------
// The reference solves colour 23 serially in list order (solver/plugin.rs:461-467).  Only the relative order of the manifolds
// that share a body matters, so the device needs, per body with a SolverBody, its overflow manifolds in list order: the CSR
// of the body-centric warm start (DW::inc_off / inc_ent), whose positions inside a body's segment are also the manifold's RANKS
// for the dataflow passes (k_overflow_flow, k_contacts.hip).
template <class T>
__global__ __launch_bounds__(256) void k_ovf_entries(DW<T> w, uint32_t o0, uint32_t n23, uint32_t* __restrict__ keys, uint32_t* __restrict__ vals) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n23) return;
    const int2 b = w.m_bodies[o0 + i];
    keys[2 * i] = meta_has_solver_body(w.bmeta[b.x]) ? (uint32_t)b.x : w.n_bodies;
    keys[2 * i + 1] = meta_has_solver_body(w.bmeta[b.y]) ? (uint32_t)b.y : w.n_bodies;
    vals[2 * i] = i; vals[2 * i + 1] = i | 0x80000000u;
}
template <class T> void launch_ovf_entries(const DW<T>& w, uint32_t o0, uint32_t n23, uint32_t* keys, uint32_t* vals, hipStream_t s) {
    if (n23) hipLaunchKernelGGL(k_ovf_entries<T>, dim3((n23 + 255) / 256), dim3(256), 0, s, w, o0, n23, keys, vals);
}
__global__ __launch_bounds__(256) void k_ovf_offsets(const uint32_t* __restrict__ keys, uint32_t n_e, uint32_t n_bodies, uint32_t* __restrict__ inc_off) {
    const uint32_t b = blockIdx.x * 256 + threadIdx.x;
    if (b > n_bodies) return;
    uint32_t lo = 0, hi = n_e;   // first sorted entry with key >= b
    while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (keys[mid] < b) lo = mid + 1; else hi = mid; }
    inc_off[b] = lo;
}
__global__ __launch_bounds__(256) void k_ovf_post(const uint32_t* __restrict__ keys, const uint32_t* __restrict__ vals, uint32_t n_e, uint32_t n_bodies, uint32_t o0,
                                                  const uint32_t* __restrict__ inc_off, uint32_t* __restrict__ inc_ent, uint32_t* __restrict__ rank) {
    const uint32_t e = blockIdx.x * 256 + threadIdx.x;
    if (e >= n_e) return;
    const uint32_t key = keys[e], val = vals[e];
    const uint32_t i = val & 0x7FFFFFFFu, side = val >> 31;
    if (key >= n_bodies) { rank[2 * i + side] = PG_NONE; return; }
    inc_ent[e] = (o0 + i) | (side << 31);
    rank[2 * i + side] = e - inc_off[key];
}
template <class T> void launch_ovf_csr(const DW<T>& w, uint32_t o0, uint32_t n23, const uint32_t* keys, const uint32_t* vals, uint32_t* inc_off, uint32_t* inc_ent, uint32_t* rank, hipStream_t s) {
    hipLaunchKernelGGL(k_ovf_offsets, dim3((w.n_bodies + 256) / 256), dim3(256), 0, s, keys, 2 * n23, w.n_bodies, inc_off);
    if (n23) hipLaunchKernelGGL(k_ovf_post, dim3((2 * n23 + 255) / 256), dim3(256), 0, s, keys, vals, 2 * n23, w.n_bodies, o0, inc_off, inc_ent, rank);
}

// ---- the sharded closed loop (avn_dshard_enable): this rank's share of the colour lists -----------------------------------------------------------------------
// Every rank replays every rank's status changes on the SAME ContactGraph / ConstraintGraph (ids, colours and list positions are global facts: DESIGN.md section 6), so
// PG::lists are the single world's lists on every rank.  The solver of a rank only takes the manifolds of the bodies it simulates: a stable compaction of every list by
// "the manifold's non-static body is mine" -- a restriction keeps the relative order, which is all the overflow colour's serial solve needs.  Two launches over (chunks of
// 2 048 entries) x colours: count, then place behind the counts in front (a first version walked every list with ONE workgroup, 256 entries and three barriers at a
// time: 0.2 ms of a 3.9 ms step at 4 * 10^5 manifolds).  A manifold between bodies of two ranks means the islands have met (the level-1 re-partition's business): error bit 16.
#define PG_LL_CHUNK 2048u
__device__ __forceinline__ uint32_t pg_ll_mine(const PG& pg, const uint4* __restrict__ ct_meta, const int32_t* __restrict__ owner, uint32_t rank, uint32_t cid) {
    const int2 b = pg.bodies[cid];
    const uint32_t fl = ct_meta[cid].z;
    const int o1 = (fl & AVN_CP_STATIC1) ? -1 : owner[b.x], o2 = (fl & AVN_CP_STATIC2) ? -1 : owner[b.y];
    if (o1 >= 0 && o2 >= 0 && o1 != o2) atomicOr(&pg.ctr[PGC_ERROR], 16u);
    return ((o1 >= 0 ? o1 : o2) == (int)rank) ? 1u : 0u;
}
// pass 1: this rank's manifolds in every 2 048-entry chunk of every colour list (flags kept as a bit per entry for pass 2)
__global__ __launch_bounds__(256) void k_pg_local_count(PG pg, const uint4* __restrict__ ct_meta, const int32_t* __restrict__ owner, uint32_t rank, uint32_t n_chunks, uint32_t* __restrict__ cnt) {
    __shared__ uint32_t ws[4];
    const uint32_t chunk = blockIdx.x, c = blockIdx.y, t = threadIdx.x;
    const uint32_t L = pg.ctr[PGC_LEN + c];
    if (chunk * PG_LL_CHUNK >= L) { if (t == 0) cnt[c * n_chunks + chunk] = 0u; return; }
    const uint32_t* __restrict__ list = pg.lists + (size_t)c * pg.list_stride;
    uint32_t s = 0;
#pragma unroll
    for (uint32_t k = 0; k < 8u; ++k) { const uint32_t i = chunk * PG_LL_CHUNK + k * 256u + t; if (i < L) s += pg_ll_mine(pg, ct_meta, owner, rank, list[i]); }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += (uint32_t)__shfl_xor((int)s, o);
    if ((t & 63u) == 0u) ws[t >> 6] = s;
    __syncthreads();
    if (t == 0) cnt[c * n_chunks + chunk] = (ws[0] + ws[1]) + (ws[2] + ws[3]);
}
// pass 2: a chunk's base = the counts in front of it; entries k * 256 + t of a chunk are taken in that order (k-major), so the order of the list is kept
__global__ __launch_bounds__(256) void k_pg_local_emit(PG pg, const uint4* __restrict__ ct_meta, const int32_t* __restrict__ owner, uint32_t rank, uint32_t n_chunks, const uint32_t* __restrict__ cnt,
                                                       uint32_t* __restrict__ local) {
    __shared__ uint32_t ws[4];
    __shared__ uint32_t s_base;
    const uint32_t chunk = blockIdx.x, c = blockIdx.y, t = threadIdx.x;
    const uint32_t L = pg.ctr[PGC_LEN + c];
    const uint32_t used = (L + PG_LL_CHUNK - 1u) / PG_LL_CHUNK;
    if (chunk >= used && !(chunk == 0u && L == 0u)) return;
    uint32_t before = 0u;
    for (uint32_t j = t; j < chunk; j += 256u) before += cnt[c * n_chunks + j];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) before += (uint32_t)__shfl_xor((int)before, o);
    if ((t & 63u) == 0u) ws[t >> 6] = before;
    __syncthreads();
    if (t == 0) s_base = (ws[0] + ws[1]) + (ws[2] + ws[3]);
    __syncthreads();
    if (L == 0u) { if (t == 0) pg.ctr[PGC_LLEN + c] = 0u; return; }
    const uint32_t* __restrict__ list = pg.lists + (size_t)c * pg.list_stride;
    uint32_t* __restrict__ out = local + (size_t)c * pg.list_stride;
    uint32_t base = s_base;
    for (uint32_t k = 0; k < 8u; ++k) {
        const uint32_t i = chunk * PG_LL_CHUNK + k * 256u + t;
        uint32_t cid = 0u, mine = 0u;
        if (i < L) { cid = list[i]; mine = pg_ll_mine(pg, ct_meta, owner, rank, cid); }
        const unsigned long long bal = __ballot(mine);
        const uint32_t lane = t & 63u, wv = t >> 6;
        __syncthreads();
        if (lane == 0) ws[wv] = (uint32_t)__popcll(bal);
        __syncthreads();
        uint32_t pos = base;
        for (uint32_t j = 0; j < wv; ++j) pos += ws[j];
        if (mine) out[pos + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull))] = cid;
        base += (ws[0] + ws[1]) + (ws[2] + ws[3]);
    }
    if (chunk == used - 1u && t == 0) pg.ctr[PGC_LLEN + c] = base;   // (the last chunk's end is the colour's local length)
}
void launch_pg_local_lists(const PG& pg, const uint4* ct_meta, const int32_t* owner, uint32_t rank, uint32_t* local_lists, uint32_t* cnt, uint32_t n_chunks, hipStream_t s) {
    hipLaunchKernelGGL(k_pg_local_count, dim3(n_chunks, AVN_GRAPH_COLOR_COUNT), dim3(256), 0, s, pg, ct_meta, owner, rank, n_chunks, cnt);
    hipLaunchKernelGGL(k_pg_local_emit, dim3(n_chunks, AVN_GRAPH_COLOR_COUNT), dim3(256), 0, s, pg, ct_meta, owner, rank, n_chunks, (const uint32_t*)cnt, local_lists);
}

// ---- the contact graph's adjacency for split_island (world/sleeping.hpp, round 6) -------------------------------------------------------
// split_island (islands/mod.rs:995-1280) walks, for every body it visits, the body's colliders in RigidBodyColliders order and per collider the
// ContactGraph's edge list -- outgoing edges newest first, then incoming edges newest first (stable_graph.rs:640-675) -- and follows the edges that hold
// constraint handles.  The host manager did that over its own edge lists: ~25 edges per body of a pile, one cache miss each into a 45 MB contact table,
// 50-70 ms for cfg2's one island every other settled step.  The rows know everything the walk reads: a row holds handles iff it has a colour
// (PG::color), its insertion stamp (PG::seq) is its place in both edge lists (newest first = descending stamp), its collider slots name the lists.
// So the device writes the walk's neighbour lists as a CSR over bodies: entries (side of a row) keyed by (rank of the collider in the body-major
// concatenation of RigidBodyColliders, direction, descending stamp) -- two stable radix sorts -- and the host's walk reads 4 bytes per edge it follows.
// Entries whose other body owns no island node (static ground) are dropped: the walk only tests them and moves on.
__device__ __forceinline__ bool pg_island_node(uint32_t bmeta) { return meta_rb_type(bmeta) != AVN_RB_STATIC && !(meta_flags(bmeta) & AVN_BODY_DISABLED); }
// one lane per constraint handle (the concatenated handle list the solver gathers through: exactly the rows that hold a colour): entries 2 m and 2 m + 1, no
// compaction -- a side whose other body owns no node gets the padding key and sorts behind every real entry.  (A first version scanned the 1.2 M rows and appended
// through one wave-aggregated counter: 226 us of same-address atomics next to the solver.)
__global__ __launch_bounds__(256) void k_adj_entries(PG pg, const uint4* __restrict__ ct_meta, const uint32_t* __restrict__ bmeta, const uint32_t* __restrict__ slot_rank,
                                                     const uint32_t* __restrict__ handles, uint32_t n_handles, uint32_t n_bodies, uint32_t seq_mask, uint32_t pad_key2, IslAdj a) {
    const uint32_t m = blockIdx.x * 256 + threadIdx.x;
    if (m >= n_handles) return;
    const uint32_t r = handles[m], e = 2u * m;
    bool take = false;
    int2 b = make_int2(-1, -1);
    uint4 mt = make_uint4(0, 0, 0, 0);
    if (r < pg.rows && pg.color[r] != PG_NONE) {
        mt = ct_meta[r];
        b = pg.bodies[r];
        take = (mt.z & AVN_CP_ROW_USED) && b.x >= 0 && b.y >= 0 && (uint32_t)b.x < n_bodies && (uint32_t)b.y < n_bodies && pg_island_node(bmeta[b.x]) && pg_island_node(bmeta[b.y]);
    }
    uint32_t k1 = seq_mask, key_out = pad_key2, key_in = pad_key2, body_out = PG_NONE, body_in = PG_NONE;
    if (take) {
        k1 = (~(uint32_t)pg.seq[r]) & seq_mask;   // descending stamp
        key_out = 2u * slot_rank[mt.x]; key_in = 2u * slot_rank[mt.y] + 1u;
        body_out = (uint32_t)b.x; body_in = (uint32_t)b.y;
    }
    a.k_a[e] = k1; a.v_a[e] = e; a.e_key2[e] = key_out; a.e_other[e] = body_in; a.e_body[e] = body_out;                         // outgoing edge of collider1
    a.k_a[e + 1u] = k1; a.v_a[e + 1u] = e + 1u; a.e_key2[e + 1u] = key_in; a.e_other[e + 1u] = body_out; a.e_body[e + 1u] = body_in;   // incoming edge of collider2
}
__global__ __launch_bounds__(256) void k_adj_key2(const uint32_t* __restrict__ v_sorted, const uint32_t* __restrict__ e_key2, uint32_t* __restrict__ k_out, uint32_t n) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) k_out[i] = e_key2[v_sorted[i]];
}
// adj[i] = the other body of the i-th entry in walk order; off[b] = the first entry of a body >= b (entries are body-major: the ranks are; padding entries carry
// body 0xFFFFFFFF and sort last, so off[n_bodies] = the number of real entries)
__global__ __launch_bounds__(256) void k_adj_finish(IslAdj a, const uint32_t* __restrict__ v_sorted, uint32_t n, uint32_t n_bodies, uint32_t* __restrict__ off, uint32_t* __restrict__ adj) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) adj[i] = a.e_other[v_sorted[i]];
    if (i <= n_bodies) {
        uint32_t lo = 0, hi = n;
        while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (a.e_body[v_sorted[mid]] < i) lo = mid + 1u; else hi = mid; }
        off[i] = lo;
        if (i == n_bodies) off[n_bodies + 1u] = lo;   // (the entry count, where the host looks for it)
    }
}
// n = 2 x the number of constraint handles (DW::n_manifolds; `handles` = the solver's concatenated handle list); the arrays of IslAdj hold n words each
void launch_isl_adjacency(const PG& pg, const uint4* ct_meta, const uint32_t* bmeta, const uint32_t* slot_rank, const uint32_t* handles, uint32_t n_handles, uint32_t n_bodies, uint32_t seq_bits, uint32_t rank_bits,
                          uint32_t pad_key2, const IslAdj& a, uint32_t* hist, uint32_t* block_sums, uint32_t* off, uint32_t* adj, hipStream_t s) {
    const uint32_t n = 2u * n_handles;
    const uint32_t seq_mask = seq_bits >= 32u ? 0xFFFFFFFFu : ((1u << seq_bits) - 1u);
    uint32_t* v = a.v_a;
    if (n) {
        hipLaunchKernelGGL(k_adj_entries, dim3((n_handles + 255) / 256), dim3(256), 0, s, pg, ct_meta, bmeta, slot_rank, handles, n_handles, n_bodies, seq_mask, pad_key2, a);
        uint32_t *k1, *v1;
        launch_radix_sort_bits(a.k_a, a.v_a, a.k_b, a.v_b, n, seq_bits, hist, block_sums, &k1, &v1, s);
        uint32_t* k2 = k1 == a.k_a ? a.k_b : a.k_a;          // the free key buffer takes the second key
        uint32_t* v2 = v1 == a.v_a ? a.v_b : a.v_a;
        hipLaunchKernelGGL(k_adj_key2, dim3((n + 255) / 256), dim3(256), 0, s, (const uint32_t*)v1, (const uint32_t*)a.e_key2, k2, n);
        uint32_t *k3, *v3;
        launch_radix_sort_bits(k2, v1, k1, v2, n, rank_bits, hist, block_sums, &k3, &v3, s);
        v = v3;
    }
    const uint32_t threads = (n > n_bodies + 1u ? n : n_bodies + 1u);
    hipLaunchKernelGGL(k_adj_finish, dim3((threads + 255) / 256), dim3(256), 0, s, a, (const uint32_t*)v, n, n_bodies, off, adj);
}

#define INST(T)                                                                                                         \
    template void launch_pg_add_pairs<T>(const PG&, const CT<T>&, const avn_pair*, uint32_t, uint64_t*, uint32_t, hipStream_t);              \
    template void launch_pg_remove<T>(const PG&, const CT<T>&, const BP<T>&, uint32_t, hipStream_t);                    \
    template void launch_pg_collect_edges<T>(const PG&, const CT<T>&, uint32_t, const uint32_t*, PGEdgeRec*, uint32_t, hipStream_t); \
    template void launch_pg_remove_list<T>(const PG&, const CT<T>&, const BP<T>&, const uint32_t*, uint32_t, hipStream_t); \
    template void launch_pg_renumber_rows<T>(const PG&, const CT<T>&, uint32_t, const uint32_t*, hipStream_t);           \
    template void launch_pg_rebuild_pair_set<T>(const CT<T>&, const BP<T>&, uint32_t, hipStream_t);                     \
    template void launch_ovf_entries<T>(const DW<T>&, uint32_t, uint32_t, uint32_t*, uint32_t*, hipStream_t);           \
    template void launch_ovf_csr<T>(const DW<T>&, uint32_t, uint32_t, const uint32_t*, const uint32_t*, uint32_t*, uint32_t*, uint32_t*, hipStream_t);
INST(float)
INST(double)
#undef INST

}  // namespace avn
