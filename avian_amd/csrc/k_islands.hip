// k_islands.hip — simulation islands and the sleeping decision on the device (include/avian_mi355x.h: avn_islands_get / avn_sleep_update).
//
// Islands: connected components of the constraint graph over non-static bodies (reference dynamics/solver/islands/mod.rs:1-10,
// merge_islands :814-990 -- a body without a BodyIslandNode, i.e. a static one, never links two islands).  One pass of a lock-free
// union-find over the edges (manifold body pairs + joints): find with path halving, hook the LARGER root under the SMALLER by compare-and-
// swap, retry from the value the failed swap returned.  Parents only ever decrease, so every tree's root is its lowest body index: the
// label is canonical without a renumbering pass.  Loads and stores of the parent array are agent-scope (the eight XCD L2s are not coherent
// with each other; the swap is the only operation correctness rests on, a stale parent merely costs a retry).
//
// Sleeping: update_sleeping_states (islands/sleeping.rs:184-241) with the island's awake bit as a store of 1 by any body that is not yet
// sleepy, and the resting islands / awake body counts by one more pass.
#include <algorithm>
#include "avn_kernels.h"

namespace avn {

__device__ __forceinline__ uint32_t cc_ld(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void cc_st(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ uint32_t cc_find(uint32_t* L, uint32_t x) {
    uint32_t cur = x, next = cc_ld(L + cur);
    while (next != cur) {
        const uint32_t nn = cc_ld(L + next);
        if (nn != next) cc_st(L + cur, nn);   // path halving: a benign race, parents only decrease
        cur = next; next = nn;
    }
    return cur;
}
__device__ __forceinline__ void cc_union(uint32_t* L, uint32_t a, uint32_t b) {
    uint32_t ra = cc_find(L, a), rb = cc_find(L, b);
    while (ra != rb) {
        if (ra < rb) { const uint32_t t = ra; ra = rb; rb = t; }   // ra > rb: ra goes under rb
        uint32_t expected = ra;
        if (__hip_atomic_compare_exchange_strong(L + ra, &expected, rb, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
        ra = cc_find(L, expected);   // ra had stopped being a root: `expected` is its parent now
        rb = cc_find(L, rb);
    }
}

// a body with a BodyIslandNode (islands/mod.rs:96-140): dynamic or kinematic, not disabled; sleeping bodies keep theirs
__device__ __forceinline__ bool island_node(uint32_t bmeta, uint32_t solver_nodes = 0u) {
    // solver_nodes: the island-BLOCK builder's notion (avn_world.hip rebuild_island_blocks) -- only bodies that own a SolverBody connect, a
    // sleeping body is as inert as a static one inside the solver
    if (solver_nodes) return meta_has_solver_body(bmeta);
    return meta_rb_type(bmeta) != AVN_RB_STATIC && !(meta_flags(bmeta) & AVN_BODY_DISABLED);
}

template <class T>
__global__ __launch_bounds__(256) void k_cc_init(DW<T> w, uint32_t* __restrict__ L) {
    const uint32_t b = blockIdx.x * 256 + threadIdx.x;
    if (b < w.n_bodies) L[b] = b;
}
template <class T>
__global__ __launch_bounds__(256) void k_cc_edges(DW<T> w, const int2* __restrict__ edges, uint32_t n, uint32_t* __restrict__ L, uint32_t solver_nodes) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int2 e = edges[i];
    if (e.x < 0 || e.y < 0 || (uint32_t)e.x >= w.n_bodies || (uint32_t)e.y >= w.n_bodies || e.x == e.y) return;
    if (!island_node(w.bmeta[e.x], solver_nodes) || !island_node(w.bmeta[e.y], solver_nodes)) return;
    cc_union(L, (uint32_t)e.x, (uint32_t)e.y);
}
// the same union over the contact table's rows that hold constraint handles (PG::color): the edges split_island's walk follows (world/sleeping.hpp, round 6)
// A FEW workgroups walk the handle list with a grid stride: the launch runs on a side stream next to the solver's latency-bound colour launches and has a
// millisecond of slack; 4 800 workgroups of compare-and-swaps slowed those launches six-fold while they ran (timeline of round 6).
template <class T>
__global__ __launch_bounds__(256) void k_cc_rows(DW<T> w, const int2* __restrict__ row_bodies, const uint32_t* __restrict__ row_color, const uint32_t* __restrict__ handles, uint32_t n_handles, uint32_t* __restrict__ L) {
    for (uint32_t m = blockIdx.x * 256 + threadIdx.x; m < n_handles; m += gridDim.x * 256) {
        const uint32_t r = handles[m];
        if (row_color[r] == 0xFFFFFFFFu) continue;
        const int2 e = row_bodies[r];
        if (e.x < 0 || e.y < 0 || (uint32_t)e.x >= w.n_bodies || (uint32_t)e.y >= w.n_bodies || e.x == e.y) continue;
        if (!island_node(w.bmeta[e.x], 0u) || !island_node(w.bmeta[e.y], 0u)) continue;
        cc_union(L, (uint32_t)e.x, (uint32_t)e.y);
    }
}
// labels out (lowest body index of the island, PG_NONE for static bodies) + ctr[0] = islands, ctr[1] = island bodies
template <class T>
__global__ __launch_bounds__(256) void k_cc_finish(DW<T> w, uint32_t* __restrict__ L, uint32_t* __restrict__ label, uint32_t* __restrict__ ctr, uint32_t solver_nodes) {
    const uint32_t b = blockIdx.x * 256 + threadIdx.x;
    bool node = false, root = false;
    if (b < w.n_bodies) {
        node = island_node(w.bmeta[b], solver_nodes);
        uint32_t r = 0xFFFFFFFFu;
        if (node) { r = cc_find(L, b); root = r == b; }
        label[b] = r;
    }
    const unsigned long long nb = __ballot(node), rb = __ballot(root);
    if ((threadIdx.x & 63) == 0) {
        if (rb) atomicAdd(ctr + 0, (uint32_t)__popcll(rb));
        if (nb) atomicAdd(ctr + 1, (uint32_t)__popcll(nb));
    }
}

// update_sleeping_states, the per-body half (sleeping.rs:200-229)
template <class T>
__global__ __launch_bounds__(256) void k_sleep_timers(DW<T> w, SleepParams<T> sp, const uint32_t* __restrict__ label, float* __restrict__ timer,
                                                      uint32_t* __restrict__ awake) {
    const uint32_t b = blockIdx.x * 256 + threadIdx.x;
    if (b >= w.n_bodies) return;
    const uint32_t l = label[b];
    if (l == 0xFFFFFFFFu) { timer[b] = 0.0f; return; }
    if (sp.body_disabled && sp.body_disabled[b]) {   // SleepingDisabled (sleeping.rs:164-182): timer reset, island kept awake -- and woken if it sleeps
        timer[b] = 0.0f;
        atomicOr(&awake[l], (meta_flags(w.bmeta[b]) & AVN_BODY_SLEEPING) ? 3u : 1u);
        return;
    }
    if (meta_flags(w.bmeta[b]) & AVN_BODY_SLEEPING) { atomicOr(&awake[l], 2u); return; }   // Without<Sleeping>: the island holds a sleeper
    const V3<T> v = xyz<T>(w.sb_lin[b]), om = xyz<T>(w.sb_ang[b]);
    const T v2 = length_squared(v), w2 = length_squared(om);
    T lin2 = sp.lin_threshold_squared, ang2 = sp.ang_threshold_squared;
    if (sp.body_lin) { const float l = sp.body_lin[b]; lin2 = (T)(l * fabsf(l)); }     // "Keep signs.": f32 product, then `as Scalar`
    if (sp.body_ang) { const float a = sp.body_ang[b]; ang2 = (T)(a * fabsf(a)); }
    float t = timer[b];
    if (v2 < sp.length_unit_squared * lin2 && w2 < ang2) t = t + sp.delta_secs;
    else t = 0.0f;
    timer[b] = t;
    if (t < sp.time_to_sleep) atomicOr(&awake[l], 1u);   // awake_island_bit_vec.set(island)
}
// sleep_islands' decision (sleeping.rs:256-266), per body.  awake[l]: bit 0 = some awake body keeps the island awake, bit 1 = it holds a sleeping
// body.  ctr[2] resting islands, [3] bodies in them, [4] waking islands, [5] sleeping bodies in them, [6] sleeping bodies
template <class T>
__global__ __launch_bounds__(256) void k_sleep_decide(DW<T> w, const uint32_t* __restrict__ label, const uint32_t* __restrict__ awake,
                                                      uint8_t* __restrict__ rests, uint8_t* __restrict__ wakes, uint32_t* __restrict__ ctr) {
    const uint32_t b = blockIdx.x * 256 + threadIdx.x;
    bool r = false, wk = false, root = false, sl = false;
    if (b < w.n_bodies) {
        const uint32_t l = label[b];
        if (l != 0xFFFFFFFFu) {
            const uint32_t a = awake[l];
            sl = (meta_flags(w.bmeta[b]) & AVN_BODY_SLEEPING) != 0;
            r = !(a & 1u) && !(a & 2u);
            wk = (a & 1u) && (a & 2u);
            root = l == b;
        }
        rests[b] = r ? 1 : 0;
        wakes[b] = wk ? 1 : 0;
    }
    const unsigned long long rb = __ballot(r), rr = __ballot(r && root), wr = __ballot(wk && root), ws = __ballot(wk && sl), sb = __ballot(sl);
    if ((threadIdx.x & 63) == 0) {
        if (rr) atomicAdd(ctr + 2, (uint32_t)__popcll(rr));
        if (rb) atomicAdd(ctr + 3, (uint32_t)__popcll(rb));
        if (wr) atomicAdd(ctr + 4, (uint32_t)__popcll(wr));
        if (ws) atomicAdd(ctr + 5, (uint32_t)__popcll(ws));
        if (sb) atomicAdd(ctr + 6, (uint32_t)__popcll(sb));
    }
}
// The closed loop with persistent islands (world/sleeping.hpp): the body half of update_sleeping_states (sleeping.rs:200-223) plus the timer
// reset of wake_islands_with_sleeping_disabled (:164-182); the island half runs in the host's island manager, which reads `flags`.
template <class T>
__global__ __launch_bounds__(256) void k_sleep_timers_flags(DW<T> w, SleepParams<T> sp, float* __restrict__ timer, uint8_t* __restrict__ flags) {
    const uint32_t b = blockIdx.x * 256 + threadIdx.x;
    if (b >= w.n_bodies) return;
    const uint32_t bm = w.bmeta[b];
    if (!island_node(bm, 0u)) { flags[b] = 0; return; }
    const bool has_sb = !(meta_flags(bm) & AVN_BODY_SLEEPING);   // a body with a node owns a SolverBody unless it sleeps
    if (sp.body_disabled && sp.body_disabled[b]) { flags[b] = (uint8_t)(2u | (has_sb ? 4u : 0u)); timer[b] = 0.0f; return; }
    if (!has_sb) { flags[b] = 0; return; }
    const V3<T> v = xyz<T>(w.sb_lin[b]), om = xyz<T>(w.sb_ang[b]);
    const T v2 = length_squared(v), w2 = length_squared(om);
    T lin2 = sp.lin_threshold_squared, ang2 = sp.ang_threshold_squared;
    if (sp.body_lin) { const float l = sp.body_lin[b]; lin2 = (T)(l * fabsf(l)); }
    if (sp.body_ang) { const float a = sp.body_ang[b]; ang2 = (T)(a * fabsf(a)); }
    float t = timer[b];
    if (v2 < sp.length_unit_squared * lin2 && w2 < ang2) t = t + sp.delta_secs;
    else t = 0.0f;
    timer[b] = t;
    flags[b] = 5;
}
template <class T>
__global__ __launch_bounds__(256) void k_bodies_set_sleeping(DW<T> w, const uint32_t* __restrict__ bodies, uint32_t n, uint32_t sleeping, float* __restrict__ timer) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint32_t b = bodies[i];
    if (b >= w.n_bodies) return;
    w.bmeta[b] = meta_with_flags(w.bmeta[b], sleeping ? (meta_flags(w.bmeta[b]) | AVN_BODY_SLEEPING) : (meta_flags(w.bmeta[b]) & ~(uint32_t)AVN_BODY_SLEEPING));
    if (!sleeping && timer) timer[b] = 0.0f;   // WakeIslands: sleep_timer.0 = 0.0 (sleeping.rs:492)
}
template <class T> void launch_sleep_timers_flags(const DW<T>& w, const SleepParams<T>& sp, float* timer, uint8_t* flags, hipStream_t s) {
    if (w.n_bodies) hipLaunchKernelGGL(k_sleep_timers_flags<T>, dim3((w.n_bodies + 255) / 256), dim3(256), 0, s, w, sp, timer, flags);
}
template <class T> void launch_bodies_set_sleeping(const DW<T>& w, const uint32_t* bodies, uint32_t n, uint32_t sleeping, float* timer, hipStream_t s) {
    if (n) hipLaunchKernelGGL(k_bodies_set_sleeping<T>, dim3((n + 255) / 256), dim3(256), 0, s, w, bodies, n, sleeping, timer);
}
__global__ __launch_bounds__(256) void k_sleep_reset(float* __restrict__ timer, const uint32_t* __restrict__ bodies, uint32_t n, uint32_t n_bodies) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint32_t b = bodies ? bodies[i] : i;
    if (b < n_bodies) timer[b] = 0.0f;
}

// Are last step's labels still a valid GROUPING for this step's manifolds?  Yes while no manifold joins two island nodes of different labels
// (and none touches a node that was not labelled): components that have split since are merely coarser than necessary, which the island
// blocks do not mind -- all they need is that no manifold crosses a block.  *invalid |= 1 otherwise.
template <class T>
__global__ __launch_bounds__(256) void k_cc_validate(DW<T> w, const int2* __restrict__ pairs, uint32_t n, const uint32_t* __restrict__ label, uint32_t* __restrict__ invalid, uint32_t solver_nodes) {
    const uint32_t e = blockIdx.x * 256 + threadIdx.x;
    if (e >= n) return;
    const int2 p = pairs[e];
    const bool na = p.x >= 0 && (uint32_t)p.x < w.n_bodies && island_node(w.bmeta[p.x], solver_nodes), nb = p.y >= 0 && (uint32_t)p.y < w.n_bodies && island_node(w.bmeta[p.y], solver_nodes);
    const uint32_t la = na ? label[p.x] : 0u, lb = nb ? label[p.y] : 0u;
    if ((na && la == 0xFFFFFFFFu) || (nb && lb == 0xFFFFFFFFu) || (na && nb && la != lb)) atomicOr(invalid, 1u);
}
template <class T> void launch_islands_validate(const DW<T>& w, const uint32_t* label, uint32_t* invalid, hipStream_t s, uint32_t solver_nodes) {
    if (w.n_manifolds) hipLaunchKernelGGL(k_cc_validate<T>, dim3((w.n_manifolds + 255) / 256), dim3(256), 0, s, w, (const int2*)w.m_bodies, w.n_manifolds, label, invalid, solver_nodes);
    if (w.n_joints) hipLaunchKernelGGL(k_cc_validate<T>, dim3((w.n_joints + 255) / 256), dim3(256), 0, s, w, (const int2*)w.j_bodies, w.n_joints, label, invalid, solver_nodes);
}
template <class T> void launch_islands(const DW<T>& w, uint32_t* parent, uint32_t* label, uint32_t* ctr, hipStream_t s, uint32_t solver_nodes) {
    if (!w.n_bodies) return;
    const uint32_t nb = (w.n_bodies + 255) / 256;
    hipLaunchKernelGGL(k_cc_init<T>, dim3(nb), dim3(256), 0, s, w, parent);
    if (w.n_manifolds) hipLaunchKernelGGL(k_cc_edges<T>, dim3((w.n_manifolds + 255) / 256), dim3(256), 0, s, w, (const int2*)w.m_bodies, w.n_manifolds, parent, solver_nodes);
    if (w.n_joints) hipLaunchKernelGGL(k_cc_edges<T>, dim3((w.n_joints + 255) / 256), dim3(256), 0, s, w, (const int2*)w.j_bodies, w.n_joints, parent, solver_nodes);
    hipLaunchKernelGGL(k_cc_finish<T>, dim3(nb), dim3(256), 0, s, w, parent, label, ctr, solver_nodes);
}
template <class T> void launch_islands_rows(const DW<T>& w, const int2* row_bodies, const uint32_t* row_color, const uint32_t* handles, uint32_t n_handles, uint32_t* parent, uint32_t* label, uint32_t* ctr, hipStream_t s) {
    if (!w.n_bodies) return;
    const uint32_t nb = (w.n_bodies + 255) / 256;
    hipLaunchKernelGGL(k_cc_init<T>, dim3(nb), dim3(256), 0, s, w, parent);
    if (n_handles) hipLaunchKernelGGL(k_cc_rows<T>, dim3(std::min<uint32_t>((n_handles + 255) / 256, 96u)), dim3(256), 0, s, w, row_bodies, row_color, handles, n_handles, parent);
    if (w.n_joints) hipLaunchKernelGGL(k_cc_edges<T>, dim3((w.n_joints + 255) / 256), dim3(256), 0, s, w, (const int2*)w.j_bodies, w.n_joints, parent, 0u);
    hipLaunchKernelGGL(k_cc_finish<T>, dim3(nb), dim3(256), 0, s, w, parent, label, ctr, 0u);
}
template <class T> void launch_sleep_update(const DW<T>& w, const SleepParams<T>& sp, const uint32_t* label, float* timer, uint32_t* awake, uint8_t* rests, uint8_t* wakes, uint32_t* ctr, hipStream_t s) {
    if (!w.n_bodies) return;
    const uint32_t nb = (w.n_bodies + 255) / 256;
    hipLaunchKernelGGL(k_sleep_timers<T>, dim3(nb), dim3(256), 0, s, w, sp, label, timer, awake);
    hipLaunchKernelGGL(k_sleep_decide<T>, dim3(nb), dim3(256), 0, s, w, label, (const uint32_t*)awake, rests, wakes, ctr);
}
void launch_sleep_reset(float* timer, const uint32_t* bodies, uint32_t n, uint32_t n_bodies, hipStream_t s) {
    if (n) hipLaunchKernelGGL(k_sleep_reset, dim3((n + 255) / 256), dim3(256), 0, s, timer, bodies, n, n_bodies);
}

#define INST(T)                                                                                              \
    template void launch_sleep_timers_flags<T>(const DW<T>&, const SleepParams<T>&, float*, uint8_t*, hipStream_t);   \
    template void launch_bodies_set_sleeping<T>(const DW<T>&, const uint32_t*, uint32_t, uint32_t, float*, hipStream_t); \
    template void launch_islands<T>(const DW<T>&, uint32_t*, uint32_t*, uint32_t*, hipStream_t, uint32_t);            \
    template void launch_islands_rows<T>(const DW<T>&, const int2*, const uint32_t*, const uint32_t*, uint32_t, uint32_t*, uint32_t*, uint32_t*, hipStream_t);   \
    template void launch_islands_validate<T>(const DW<T>&, const uint32_t*, uint32_t*, hipStream_t, uint32_t);       \
    template void launch_sleep_update<T>(const DW<T>&, const SleepParams<T>&, const uint32_t*, float*, uint32_t*, uint8_t*, uint8_t*, uint32_t*, hipStream_t);
INST(float)
INST(double)
#undef INST

}  // namespace avn
