// avn_scan.h -- device-side building blocks of the one-launch exclusive scan (k_broadphase.hip: k_scan_chained; k_graph.hip: the
// status-change scan fused with the classification of the changes).
//
// exclusive scan (uint32) in ONE launch: chained scan with decoupled look-back (round 4; rounds 1-3 used three kernels -- tile sums, a
// one-workgroup scan of the sums, apply -- and a closed-loop step makes four or five scans, each of them three launches on their
// latency floor in one serial chain).  A workgroup takes its tile from a ticket counter (tiles are numbered in START order, so a
// workgroup only ever waits for workgroups that already run), reduces it, publishes (AGGREGATE | sum), and one wave looks back over
// the predecessors' status words 64 at a time until it meets an INCLUSIVE PREFIX; then it publishes its own prefix.  Status words are
// 64-bit (flag << 32 | value) written and read with agent-scope atomics (the eight XCD L2s are not coherent with each other).
// The state is SELF-CLEANING: the last workgroup to finish its look-back zeroes the status words and both counters, so the next scan
// on the same stream finds them clean without a memset launch (the buffers are zeroed once when they are allocated).
// state layout (uint32 words of `block_sums`): [0] ticket, [1] done, [2 .. 2 + 2 nb) the status words (8-byte aligned).
#pragma once
#include "avn_device.h"

namespace avn {

#define SC_TILE 2048
#define SC_AGG 1ull
#define SC_PREFIX 2ull
__device__ __forceinline__ unsigned long long sc_ld(const unsigned long long* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void sc_st(unsigned long long* p, unsigned long long v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// sc_take_tile: the tile this workgroup owns (ticket).  sc_lookback: the exclusive prefix of that tile, given its own total (block-uniform).
// Both are called by all 256 threads and contain workgroup barriers.
__device__ __forceinline__ uint32_t sc_take_tile(uint32_t* __restrict__ st) {
    __shared__ uint32_t s_tile;
    if (threadIdx.x == 0) s_tile = atomicAdd(&st[0], 1u);
    __syncthreads();
    return s_tile;
}
__device__ __forceinline__ uint32_t sc_lookback(uint32_t* __restrict__ st, uint32_t tile, uint32_t nb, uint32_t tile_sum) {
    __shared__ uint32_t s_excl;
    __shared__ uint32_t s_last;
    unsigned long long* status = reinterpret_cast<unsigned long long*>(st + 2);
    const uint32_t t = threadIdx.x;
    if (t < 64u) {   // wave 0
        if (tile == 0u) {
            if (t == 0u) { sc_st(&status[0], (SC_PREFIX << 32) | tile_sum); s_excl = 0u; }
        } else {
            if (t == 0u) sc_st(&status[tile], (SC_AGG << 32) | tile_sum);
            uint32_t run = 0u;
            int p0 = (int)tile - 1;   // the window [p0 - 63, p0], lane l looks at p0 - l
            for (;;) {
                const int p = p0 - (int)t;
                unsigned long long v = p >= 0 ? 0ull : (SC_PREFIX << 32);   // in front of tile 0: an empty prefix
                if (p >= 0) { do { v = sc_ld(&status[p]); } while ((v >> 32) == 0ull); }
                const unsigned long long is_prefix = __ballot((v >> 32) == SC_PREFIX);
                const uint32_t first = is_prefix ? (uint32_t)__ffsll((long long)is_prefix) - 1u : 64u;   // nearest predecessor holding a prefix
                uint32_t x = t <= first ? (uint32_t)v : 0u;
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) x += (uint32_t)__shfl_xor((int)x, off);
                run += x;
                if (is_prefix) break;
                p0 -= 64;
            }
            if (t == 0u) { s_excl = run; sc_st(&status[tile], (SC_PREFIX << 32) | (run + tile_sum)); }
        }
    }
    __syncthreads();
    const uint32_t excl = s_excl;
    // self-cleaning: whoever finishes its look-back LAST (every other workgroup has read what it needed) resets the state
    if (t == 0u) { __threadfence(); s_last = atomicAdd(&st[1], 1u) == nb - 1u ? 1u : 0u; }
    __syncthreads();
    if (s_last) {
        for (uint32_t i = t; i < nb; i += 256u) status[i] = 0ull;
        if (t == 0u) { st[0] = 0u; st[1] = 0u; }
    }
    return excl;
}
// in-workgroup exclusive scan of one value per thread (256 threads); returns the thread's exclusive prefix, *total = the workgroup's sum
__device__ __forceinline__ uint32_t sc_block_excl(uint32_t s, uint32_t* total) {
    __shared__ uint32_t wsum[4];
    const uint32_t t = threadIdx.x, lane = t & 63u, wv = t >> 6;
    uint32_t incl = s;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const uint32_t u = (uint32_t)__shfl_up((int)incl, off); if ((int)lane >= off) incl += u; }
    if (lane == 63u) wsum[wv] = incl;
    __syncthreads();
    uint32_t before = 0u, all = 0u;
#pragma unroll
    for (uint32_t k = 0; k < 4u; ++k) { const uint32_t v = wsum[k]; if (k < wv) before += v; all += v; }
    __syncthreads();   // (wsum is reused by the next call)
    *total = all;
    return before + incl - s;
}

}  // namespace avn
