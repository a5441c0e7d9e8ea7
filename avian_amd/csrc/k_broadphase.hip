// k_broadphase.hip — AABB update + single-axis sweep-and-prune on device, bit-exact pair lists.
//
// The reference (collision/broad_phase.rs:373-474) keeps one Vec of intervals sorted by aabb.min.x with a
// STABLE insertion sort (swap only when prev.min.x > cur.min.x), then for every i scans j > i until
// min_x[j] > max_x[i], emitting pairs in (i ascending, j ascending) order.  Here:
//   1. k_update_aabb           one thread per collider (collider/backend.rs:498-624; parry cuboid/ball AABB)
//   2. k_interval_keys         order-preserving integer key of min.x (-0.0 == +0.0; non-finite -> dropped)
//   3. stable LSD radix sort   hand-written: per-tile digit histogram -> exclusive scan -> stable scatter
//                              (wave64 ballot match-any ranks).  A stable sort of the current order by the same
//                              comparison yields exactly the permutation of the stable insertion sort.
//   4. k_gather_sorted         interval records into sorted SoA
//                              (skipped on device when the persistent order is still sorted)
//   4. k_gather_sorted         interval records into sorted SoA (min.x | max.x | (min.y,max.y,min.z,max.z) | info | flags)
//   5. k_sweep_ranges          end(i) by binary search; scene-spanning intervals are cut into chunks (LongItems)
//   6. k_sweep<EMIT=false>     per-interval pair COUNT: wave = 64 consecutive i, candidate j wave-uniform and fetched by
//      k_sweep_long<false>     SCALAR loads, hits compacted into an LDS queue and filtered 64 at a time (see "sweep")
//   7. exclusive scan of the counts
//   8. k_sweep<EMIT=true>      every lane re-walks its candidates and writes its pairs at its own offset
//                              => output is already in the reference's emission order, no post-sort.
// Integer/compare work only; VALU-bound in the sweep for dense scenes (O(k) tests per interval), HBM/L2 traffic is the
// sorted records once per 64 intervals.
#include "avn_kernels.h"
#include "avn_scan.h"

namespace avn {

// ---------------------------------------------------------------------------------------------------------
// AABB
template <class T> __device__ __forceinline__ void shape_aabb(uint32_t shape, V3<T> h, V3<T> pos, Q4<T> q, V3<T>& mn, V3<T>& mx) {
    V3<T> he;
    if (shape == AVN_SHAPE_BALL) he = V3<T>{h.x, h.x, h.x};
    else {
        // parry3d Cuboid::aabb: centre +- |R| * half_extents, R = nalgebra UnitQuaternion::to_rotation_matrix
        T i = q.x, j = q.y, k = q.z, w = q.w;
        T ww = w * w, ii = i * i, jj = j * j, kk = k * k;
        T ij = i * j * T(2), wk = w * k * T(2), wj = w * j * T(2), ik = i * k * T(2), jk = j * k * T(2), wi = w * i * T(2);
        T m00 = fabs_t(ww + ii - jj - kk), m01 = fabs_t(ij - wk), m02 = fabs_t(wj + ik);
        T m10 = fabs_t(wk + ij), m11 = fabs_t(ww - ii + jj - kk), m12 = fabs_t(jk - wi);
        T m20 = fabs_t(ik - wj), m21 = fabs_t(wi + jk), m22 = fabs_t(ww - ii - jj + kk);
        he = V3<T>{(m00 * h.x + m01 * h.y) + m02 * h.z, (m10 * h.x + m11 * h.y) + m12 * h.z, (m20 * h.x + m21 * h.y) + m22 * h.z};
    }
    mn = pos - he;
    mx = pos + he;
}

template <class T>
__global__ __launch_bounds__(256) void k_update_aabb(DW<T> w, BP<T> bp, StepParams<T> p, uint32_t* __restrict__ zero_words, uint32_t n_zero) {
    uint32_t c = blockIdx.x * 256 + threadIdx.x;
    // the first kernel of a step also clears the step-scoped counters of the kernels behind it (constraint count, dropped / unsorted,
    // long-interval chunks): one memset launch less per counter on the step's serial chain
    if (c < n_zero) zero_words[c] = 0u;
    if (c >= bp.n_colliders) return;
    uint4 ci = bp.col_info[c];  // entity, body, shape | cflags << 8, -
    Vec4<T> he4 = bp.col_he[c];  // (half_extents.xyz, collision_margin)
    T spec = bp.col_spec[c];
    uint32_t shape = ci.z & 0xFFu, cflags = (ci.z >> 8) & 0xFFu;
    if (shape == AVN_SHAPE_HOST) return;   // AnyCollider::aabb_with_context is the host's: k_host_aabb_queries / k_host_aabb_apply wrote this collider's box
    int body = (int)ci.y;
    V3<T> pos = xyz<T>(w.pos[body]);
    Q4<T> rot = quat<T>(w.rot[body]);
    V3<T> lv = xyz<T>(w.lvel[body]), av = xyz<T>(w.avel[body]);
    if (bp.col_lpos) {   // a child collider: its own pose, and the body's velocity at its offset from the centre of mass (backend.rs:569-586)
        const V3<T> bpos = pos; const Q4<T> brot = rot;
        bool child;
        collider_pose<T>(bp, c, bpos, brot, pos, rot, &child);
        if (child) { const V3<T> offset = (pos - bpos) - qrot(brot, xyz<T>(w.com[body])); lv = lv + cross(av, offset); }
    }
    T delta_secs = p.dt_adj;
    T speculative_margin = (cflags & AVN_COLLIDER_SWEPT_CCD) ? Limits<T>::max : (spec >= T(0) ? spec : p.default_speculative_margin);
    T g = p.contact_tolerance + he4.w;
    V3<T> mn, mx;
    V3<T> h = xyz<T>(he4);
    if (speculative_margin <= T(0)) {
        shape_aabb<T>(shape, h, pos, rot, mn, mx);
    } else {
        Q4<T> end_rot = fast_renormalize(qmul(from_scaled_axis(av * delta_secs), rot));
        V3<T> end_pos = pos + clamp_length_max(lv * delta_secs, smax(speculative_margin, p.contact_tolerance));
        V3<T> mn0, mx0, mn1, mx1;
        shape_aabb<T>(shape, h, pos, rot, mn0, mx0);
        shape_aabb<T>(shape, h, end_pos, end_rot, mn1, mx1);
        mn = vmin(mn0, mn1);
        mx = vmax(mx0, mx1);
    }
    V3<T> gg{g, g, g};
    bp.aabb_min[c] = make4<T>(mn - gg, 0);
    bp.aabb_max[c] = make4<T>(mx + gg, 0);
}

// Host shapes (include/avian_mi355x.h "host shapes"): update_aabb for the colliders whose AnyCollider::aabb lives on the host.  k_host_aabb_queries writes what
// aabb_with_context / swept_aabb_with_context are called with (backend.rs:556-620: the start pose and, with a positive speculative margin, the predicted end pose --
// the same expressions as k_update_aabb); the host answers one box per collider; k_host_aabb_apply grows it by contact_tolerance + collision margin (backend.rs:560,618).
template <class T> struct HostAabbQ { uint32_t collider, swept; T start_position[3], start_rotation[4], end_position[3], end_rotation[4]; };   // == avn_host_aabb_query_fNN
template <class T> struct HostAabb { T min[3], max[3]; };                                                                                    // == avn_host_aabb_fNN
static_assert(sizeof(HostAabbQ<float>) == sizeof(avn_host_aabb_query_f32) && sizeof(HostAabbQ<double>) == sizeof(avn_host_aabb_query_f64), "host aabb query layout");
static_assert(sizeof(HostAabb<float>) == sizeof(avn_host_aabb_f32) && sizeof(HostAabb<double>) == sizeof(avn_host_aabb_f64), "host aabb layout");
template <class T>
__global__ __launch_bounds__(256) void k_host_aabb_queries(DW<T> w, BP<T> bp, StepParams<T> p, const uint32_t* __restrict__ slots, uint32_t n, HostAabbQ<T>* __restrict__ out) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint32_t c = slots[i];
    const uint4 ci = bp.col_info[c];
    const T spec = bp.col_spec[c];
    const uint32_t cflags = (ci.z >> 8) & 0xFFu;
    const int body = (int)ci.y;
    V3<T> pos = xyz<T>(w.pos[body]);
    Q4<T> rot = quat<T>(w.rot[body]);
    V3<T> lv = xyz<T>(w.lvel[body]);
    const V3<T> av = xyz<T>(w.avel[body]);
    if (bp.col_lpos) {   // a child collider (as in k_update_aabb)
        const V3<T> bpos = pos; const Q4<T> brot = rot;
        bool child;
        collider_pose<T>(bp, c, bpos, brot, pos, rot, &child);
        if (child) { const V3<T> offset = (pos - bpos) - qrot(brot, xyz<T>(w.com[body])); lv = lv + cross(av, offset); }
    }
    const T delta_secs = p.dt_adj;
    const T speculative_margin = (cflags & AVN_COLLIDER_SWEPT_CCD) ? Limits<T>::max : (spec >= T(0) ? spec : p.default_speculative_margin);
    HostAabbQ<T> q;
    q.collider = ci.x;
    q.swept = speculative_margin <= T(0) ? 0u : 1u;
    Q4<T> end_rot = rot; V3<T> end_pos = pos;
    if (q.swept) {
        end_rot = fast_renormalize(qmul(from_scaled_axis(av * delta_secs), rot));
        end_pos = pos + clamp_length_max(lv * delta_secs, smax(speculative_margin, p.contact_tolerance));
    }
    q.start_position[0] = pos.x; q.start_position[1] = pos.y; q.start_position[2] = pos.z;
    q.start_rotation[0] = rot.x; q.start_rotation[1] = rot.y; q.start_rotation[2] = rot.z; q.start_rotation[3] = rot.w;
    q.end_position[0] = end_pos.x; q.end_position[1] = end_pos.y; q.end_position[2] = end_pos.z;
    q.end_rotation[0] = end_rot.x; q.end_rotation[1] = end_rot.y; q.end_rotation[2] = end_rot.z; q.end_rotation[3] = end_rot.w;
    out[i] = q;
}
template <class T>
__global__ __launch_bounds__(256) void k_host_aabb_apply(BP<T> bp, StepParams<T> p, const uint32_t* __restrict__ slots, uint32_t n, const HostAabb<T>* __restrict__ in) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint32_t c = slots[i];
    const T g = p.contact_tolerance + bp.col_he[c].w;
    const HostAabb<T> a = in[i];
    const V3<T> gg{g, g, g};
    bp.aabb_min[c] = make4<T>(V3<T>{a.min[0], a.min[1], a.min[2]} - gg, 0);
    bp.aabb_max[c] = make4<T>(V3<T>{a.max[0], a.max[1], a.max[2]} + gg, 0);
}
template <class T> void launch_host_aabb_queries(const DW<T>& w, const BP<T>& bp, const StepParams<T>& p, const uint32_t* slots, uint32_t n, void* out, hipStream_t s) {
    if (n) hipLaunchKernelGGL(k_host_aabb_queries<T>, dim3((n + 255) / 256), dim3(256), 0, s, w, bp, p, slots, n, (HostAabbQ<T>*)out);
}
template <class T> void launch_host_aabb_apply(const BP<T>& bp, const StepParams<T>& p, const uint32_t* slots, uint32_t n, const void* in, hipStream_t s) {
    if (n) hipLaunchKernelGGL(k_host_aabb_apply<T>, dim3((n + 255) / 256), dim3(256), 0, s, bp, p, slots, n, (const HostAabb<T>*)in);
}
template void launch_host_aabb_queries<float>(const DW<float>&, const BP<float>&, const StepParams<float>&, const uint32_t*, uint32_t, void*, hipStream_t);
template void launch_host_aabb_queries<double>(const DW<double>&, const BP<double>&, const StepParams<double>&, const uint32_t*, uint32_t, void*, hipStream_t);
template void launch_host_aabb_apply<float>(const BP<float>&, const StepParams<float>&, const uint32_t*, uint32_t, const void*, hipStream_t);
template void launch_host_aabb_apply<double>(const BP<double>&, const StepParams<double>&, const uint32_t*, uint32_t, const void*, hipStream_t);

// Per-workgroup partial union of the AABBs of colliders on non-static bodies (multi-GPU proximity bound, header:
// avn_dynamic_bounds).  partial[b] = (min.xyz, max.xyz) as T; the host reduces the few hundred partials.
template <class T>
__global__ __launch_bounds__(256) void k_dynamic_bounds(DW<T> w, BP<T> bp, T* __restrict__ partial) {
    __shared__ T red[6][256];
    uint32_t c = blockIdx.x * 256 + threadIdx.x, t = threadIdx.x;
    T inf = Limits<T>::max * T(2);
    T v[6] = {inf, inf, inf, -inf, -inf, -inf};
    if (c < bp.n_colliders) {
        uint4 ci = bp.col_info[c];
        if (meta_rb_type(w.bmeta[ci.y]) != AVN_RB_STATIC) {
            Vec4<T> mn = bp.aabb_min[c], mx = bp.aabb_max[c];
            v[0] = mn.x; v[1] = mn.y; v[2] = mn.z; v[3] = mx.x; v[4] = mx.y; v[5] = mx.z;
        }
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) red[k][t] = v[k];
    __syncthreads();
    for (uint32_t st = 128; st > 0; st >>= 1) {
        if (t < st) {
#pragma unroll
            for (int k = 0; k < 3; ++k) red[k][t] = smin(red[k][t], red[k][t + st]);
#pragma unroll
            for (int k = 3; k < 6; ++k) red[k][t] = smax(red[k][t], red[k][t + st]);
        }
        __syncthreads();
    }
    if (t < 6) partial[blockIdx.x * 6 + t] = red[t][0];
}
template <class T> uint32_t launch_dynamic_bounds(const DW<T>& w, const BP<T>& bp, T* partial, hipStream_t s) {
    uint32_t nb = (bp.n_colliders + 255) / 256;
    if (nb) hipLaunchKernelGGL(k_dynamic_bounds<T>, dim3(nb), dim3(256), 0, s, w, bp, partial);
    return nb;
}

// the per-workgroup partials of k_dynamic_bounds -> (min.xyz, max.xyz) as doubles: what a rank sends into the bounds all-gather
template <class T>
__global__ __launch_bounds__(64) void k_bounds_reduce(const T* __restrict__ partial, uint32_t nb, double* __restrict__ out) {
    const uint32_t t = threadIdx.x;
    const double inf = __longlong_as_double(0x7FF0000000000000ll);
    double v[6] = {inf, inf, inf, -inf, -inf, -inf};
    for (uint32_t b = t; b < nb; b += 64)
        for (int k = 0; k < 6; ++k) { const double x = (double)partial[b * 6 + k]; v[k] = k < 3 ? (x < v[k] ? x : v[k]) : (x > v[k] ? x : v[k]); }
    for (int off = 32; off > 0; off >>= 1)
        for (int k = 0; k < 6; ++k) { const double o = __shfl_xor(v[k], off); v[k] = k < 3 ? (o < v[k] ? o : v[k]) : (o > v[k] ? o : v[k]); }
    if (t < 6) out[t] = v[t];
}
template <class T> void launch_bounds_reduce(const T* partial, uint32_t nb, double* out, hipStream_t s) {
    hipLaunchKernelGGL(k_bounds_reduce<T>, dim3(1), dim3(64), 0, s, partial, nb, out);
}
template void launch_bounds_reduce<float>(const float*, uint32_t, double*, hipStream_t);
template void launch_bounds_reduce<double>(const double*, uint32_t, double*, hipStream_t);

// ---------------------------------------------------------------------------------------------------------
// keys
__device__ __forceinline__ uint32_t order_key(float x) {
    uint32_t b = __float_as_uint(x);
    if (b == 0x80000000u) b = 0;  // -0.0 == +0.0 under the reference's `>` comparison
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ uint64_t order_key(double x) {
    uint64_t b = (uint64_t)__double_as_longlong(x);
    if (b == 0x8000000000000000ull) b = 0;
    return (b & 0x8000000000000000ull) ? ~b : (b | 0x8000000000000000ull);
}
template <class K> struct KeyMax;
template <> struct KeyMax<uint32_t> { static constexpr uint32_t v = 0xFFFFFFFFu; };
template <> struct KeyMax<uint64_t> { static constexpr uint64_t v = 0xFFFFFFFFFFFFFFFFull; };

// update_aabb_intervals (broad_phase.rs:214-280): refresh flags, drop non-finite AABBs (key = MAX sorts them last,
// the host truncates the interval list by *n_dropped after the sort).
template <class T>
__global__ __launch_bounds__(256) void k_interval_keys(DW<T> w, BP<T> bp, typename BP<T>::Key* keys, uint32_t* vals, uint32_t* n_dropped) {
    uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= bp.n_intervals) return;
    uint32_t c = bp.iv_collider[i];
    Vec4<T> mn = bp.aabb_min[c], mx = bp.aabb_max[c];
    bool finite = is_finite(xyz<T>(mn)) && is_finite(xyz<T>(mx));
    typename BP<T>::Key k = finite ? order_key(mn.x) : KeyMax<typename BP<T>::Key>::v;
    if (finite && k == KeyMax<typename BP<T>::Key>::v) k -= 1;  // keep MAX reserved for dropped entries
    keys[i] = k;
    vals[i] = c;
    if (!finite) atomicAdd(n_dropped, 1u);
    // is the persistent order still sorted by this frame's keys?  (n_dropped[1] = "unsorted" flag; a dropped entry that is
    // not already last also counts as unsorted because its key is MAX)
    if (i > 0) {
        uint32_t cp = bp.iv_collider[i - 1];
        Vec4<T> pmn = bp.aabb_min[cp], pmx = bp.aabb_max[cp];
        bool pfinite = is_finite(xyz<T>(pmn)) && is_finite(xyz<T>(pmx));
        typename BP<T>::Key kp = pfinite ? order_key(pmn.x) : KeyMax<typename BP<T>::Key>::v;
        if (pfinite && kp == KeyMax<typename BP<T>::Key>::v) kp -= 1;
        if (kp > k) n_dropped[1] = 1u;
    }
}

// ---------------------------------------------------------------------------------------------------------
// stable LSD radix sort, 8 bits per pass, one wave per 1024-item tile
#define RS_TILE 1024
#define RS_ROUNDS 16

template <class K, bool BLOCK_MAJOR>
__global__ __launch_bounds__(64) void k_radix_hist(const K* __restrict__ keys, uint32_t n, uint32_t shift, uint32_t* __restrict__ hist, uint32_t nblocks,
                                                   const uint32_t* __restrict__ enabled) {
    __shared__ uint32_t cnt[256];
    if (enabled && *enabled == 0u) return;
    uint32_t lane = threadIdx.x, b = blockIdx.x;
    for (uint32_t d = lane; d < 256; d += 64) cnt[d] = 0;
    __syncthreads();
    uint32_t base = b * RS_TILE;
    for (uint32_t r = 0; r < RS_ROUNDS; ++r) {
        uint32_t idx = base + r * 64 + lane;
        if (idx < n) atomicAdd(&cnt[(uint32_t)(keys[idx] >> shift) & 255u], 1u);
    }
    __syncthreads();
    for (uint32_t d = lane; d < 256; d += 64) hist[BLOCK_MAJOR ? b * 256u + d : d * nblocks + b] = cnt[d];
}

// SELF_OFFSETS = false: `hist` is the exclusive scan of the digit-major [256][nblocks] histogram (three scan launches
// between hist and scatter).  SELF_OFFSETS = true (small sorts, nblocks <= RS_FUSED_MAX_BLOCKS): `hist` is the raw
// block-major [nblocks][256] histogram and every tile derives its own 256 start offsets from it -- per digit the total
// over all tiles (-> exclusive scan over the digits inside the wave) plus the counts of the tiles before it -- so a pass
// is two launches instead of five: at 100k keys the sort is launch-latency bound, not bandwidth bound.
#define RS_FUSED_MAX_BLOCKS 128u
template <class K, bool SELF_OFFSETS>
__global__ __launch_bounds__(64) void k_radix_scatter(const K* __restrict__ keys_in, const uint32_t* __restrict__ vals_in, K* __restrict__ keys_out,
                                                      uint32_t* __restrict__ vals_out, uint32_t n, uint32_t shift,
                                                      const uint32_t* __restrict__ hist, uint32_t nblocks, const uint32_t* __restrict__ enabled) {
    __shared__ uint32_t run[256];
    if (enabled && *enabled == 0u) return;
    uint32_t lane = threadIdx.x, b = blockIdx.x;
    if (!SELF_OFFSETS) {
        for (uint32_t d = lane; d < 256; d += 64) run[d] = hist[d * nblocks + b];
    } else {
        // lane owns digits 4 * lane .. 4 * lane + 3: one coalesced 16-byte load per tile row
        const uint4* __restrict__ h4 = reinterpret_cast<const uint4*>(hist);
        uint4 pre = make_uint4(0, 0, 0, 0), tot = make_uint4(0, 0, 0, 0);
#pragma unroll 8
        for (uint32_t bb = 0; bb < nblocks; ++bb) {
            uint4 h = h4[bb * 64u + lane];
            tot.x += h.x; tot.y += h.y; tot.z += h.z; tot.w += h.w;
            if (bb < b) { pre.x += h.x; pre.y += h.y; pre.z += h.z; pre.w += h.w; }
        }
        uint32_t mine = tot.x + tot.y + tot.z + tot.w, incl = mine;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { uint32_t v = (uint32_t)__shfl_up((int)incl, off); if ((int)lane >= off) incl += v; }
        uint32_t excl = incl - mine;
        run[4 * lane + 0] = excl + pre.x;
        run[4 * lane + 1] = excl + tot.x + pre.y;
        run[4 * lane + 2] = excl + tot.x + tot.y + pre.z;
        run[4 * lane + 3] = excl + tot.x + tot.y + tot.z + pre.w;
    }
    __syncthreads();
    uint32_t base = b * RS_TILE;
    unsigned long long lt_mask = (1ull << lane) - 1ull;
    for (uint32_t r = 0; r < RS_ROUNDS; ++r) {
        uint32_t idx = base + r * 64 + lane;
        bool valid = idx < n;
        K key = valid ? keys_in[idx] : K(0);
        uint32_t val = valid ? vals_in[idx] : 0u;
        uint32_t digit = (uint32_t)(key >> shift) & 255u;
        // match-any over the 8 digit bits: lanes holding the same digit (among valid lanes)
        unsigned long long peers = __ballot(valid);
#pragma unroll
        for (uint32_t bit = 0; bit < 8; ++bit) {
            bool set = (digit >> bit) & 1u;
            unsigned long long m = __ballot(set);
            peers &= set ? m : ~m;
        }
        uint32_t rank = (uint32_t)__popcll(peers & lt_mask);
        uint32_t pos = 0;
        if (valid) pos = run[digit] + rank;  // all peers read the running offset before the leader bumps it
        __syncthreads();
        if (valid && rank == 0) run[digit] += (uint32_t)__popcll(peers);
        __syncthreads();
        if (valid) { keys_out[pos] = key; vals_out[pos] = val; }
    }
}

// ---------------------------------------------------------------------------------------------------------
// exclusive scan (uint32) in ONE launch: avn_scan.h (chained scan, decoupled look-back, self-cleaning state)
__global__ __launch_bounds__(256) void k_scan_chained(const uint32_t* in, uint32_t* out /* may alias `in` */, uint32_t n, uint32_t* __restrict__ st, uint32_t* __restrict__ total,
                                                      const uint32_t* __restrict__ enabled) {
    if (enabled && *enabled == 0u) return;
    const uint32_t nb = gridDim.x, t = threadIdx.x;
    const uint32_t tile = sc_take_tile(st);
    const uint32_t base = tile * SC_TILE;
    constexpr uint32_t per = SC_TILE / 256;
    uint32_t v[per], s = 0;
#pragma unroll
    for (uint32_t k = 0; k < per; ++k) { const uint32_t i = base + t * per + k; v[k] = i < n ? in[i] : 0u; s += v[k]; }
    uint32_t tile_sum;
    uint32_t excl = sc_block_excl(s, &tile_sum);
    excl += sc_lookback(st, tile, nb, tile_sum);
#pragma unroll
    for (uint32_t k = 0; k < per; ++k) { const uint32_t i = base + t * per + k; if (i < n) out[i] = excl; excl += v[k]; }
    if (total && tile == nb - 1u && t == 255u) *total = excl;   // (thread 255 of the last tile ends on the grand total)
}
void launch_exclusive_scan(const uint32_t* in, uint32_t* out, uint32_t n, uint32_t* block_sums, uint32_t* total, hipStream_t s, const uint32_t* enabled) {
    if (n == 0) { if (total) (void)hipMemsetAsync(total, 0, sizeof(uint32_t), s); return; }
    uint32_t nb = (n + SC_TILE - 1) / SC_TILE;
    hipLaunchKernelGGL(k_scan_chained, dim3(nb), dim3(256), 0, s, in, out, n, block_sums, total, enabled);
}
uint32_t exclusive_scan_launches(uint32_t n) { return n == 0 ? 0u : 1u; }
uint32_t scan_block_sums_needed(uint32_t n) { return 2u * ((n + SC_TILE - 1) / SC_TILE) + 8u; }   // uint32 words of scan state: ticket, done, 64-bit status per tile (zeroed at allocation, self-cleaning after)

// ---------------------------------------------------------------------------------------------------------
// gather the sorted interval records
template <class T>
__global__ __launch_bounds__(256) void k_gather_sorted(DW<T> w, BP<T> bp, const uint32_t* __restrict__ sorted_collider, uint32_t n) {
    uint32_t i = blockIdx.x * 256 + threadIdx.x;
    const bool live = i < n;
    // (no early return: every lane takes part in the shuffles of the batch-cull bounds at the end; lanes past n contribute neutral boxes)
    uint32_t c = live ? sorted_collider[i] : 0u;
    T by_lo = Limits<T>::max * T(2), by_hi = -(Limits<T>::max * T(2)), bz_lo = by_lo, bz_hi = by_hi;
    bool bnan = false;
    if (live) {
    bp.iv_collider[i] = c;  // the persistent interval order for the next frame
    uint4 ci = bp.col_info[c];
    uint2 layers = bp.col_layers[c];
    uint32_t cflags = (ci.z >> 8) & 0xFFu;
    uint32_t meta = w.bmeta[ci.y];
    bool is_static = meta_rb_type(meta) == AVN_RB_STATIC;
    bool is_sleeping = meta_flags(meta) & AVN_BODY_SLEEPING;
    bool is_disabled = meta_flags(meta) & AVN_BODY_DISABLED;
    uint32_t f = 0;
    if (is_static || is_sleeping) f |= AVN_AABB_IS_INACTIVE;
    if (cflags & AVN_COLLIDER_EVENTS) f |= AVN_AABB_CONTACT_EVENTS;
    if (!(cflags & AVN_COLLIDER_SENSOR) && !is_disabled) f |= AVN_AABB_GENERATE_CONSTRAINTS;
    if (cflags & AVN_COLLIDER_FILTER_PAIRS) f |= AVN_AABB_CUSTOM_FILTER;
    if (cflags & AVN_COLLIDER_MODIFY_CONTACTS) f |= AVN_AABB_MODIFY_CONTACTS;
    Vec4<T> mn = bp.aabb_min[c], mx = bp.aabb_max[c];
    if (!(is_finite(xyz<T>(mn)) && is_finite(xyz<T>(mx)))) {
        // dropped by update_aabb_intervals (broad_phase.rs:243-245): sorted last (key = MAX), never overlaps, and the
        // host truncates it off the interval list after this frame's read-back
        T inf = Limits<T>::max * T(2);
        mn = make4<T>(inf, inf, inf, 0);
        mx = make4<T>(-inf, -inf, -inf, 0);
        f = AVN_IV_DROPPED;
    }
    bp.s_minx[i] = mn.x;
    bp.s_maxx[i] = mx.x;
    bp.s_yz[i] = make4<T>(mn.y, mx.y, mn.z, mx.z);
    bp.s_info[i] = make_uint4(ci.x, ci.y, layers.x, layers.y);
    bp.s_flags[i] = f;
    by_lo = mn.y; by_hi = mx.y; bz_lo = mn.z; bz_hi = mx.z;
    bnan = (mn.y != mn.y) | (mx.y != mx.y) | (mn.z != mn.z) | (mx.z != mx.z);
    }
    // the sweep's batch cull (see "Batch cull" below; round 4: was its own kernel, k_batch_bounds, reading s_yz back): the y / z bounds of every 8
    // and of every 64 consecutive sorted records, reduced across the lanes that just wrote them.  A workgroup starts on a multiple of 256
    // records, so octets and waves are aligned with the groups.  Conservative as before: a NaN anywhere in a group disables the cull for it.
    const T inf = Limits<T>::max * T(2);
    uint32_t nn = bnan ? 1u : 0u;
#pragma unroll
    for (int off = 1; off < 8; off <<= 1) {
        const T a = __shfl_xor(by_lo, off), b = __shfl_xor(by_hi, off), cc = __shfl_xor(bz_lo, off), d = __shfl_xor(bz_hi, off);
        nn |= (uint32_t)__shfl_xor((int)nn, off);
        by_lo = a < by_lo ? a : by_lo; by_hi = b > by_hi ? b : by_hi; bz_lo = cc < bz_lo ? cc : bz_lo; bz_hi = d > bz_hi ? d : bz_hi;
    }
    if (live && (i & 7u) == 0u) bp.s_bb[i >> 3] = nn ? make4<T>(-inf, inf, -inf, inf) : make4<T>(by_lo, by_hi, bz_lo, bz_hi);
#pragma unroll
    for (int off = 8; off < 64; off <<= 1) {
        const T a = __shfl_xor(by_lo, off), b = __shfl_xor(by_hi, off), cc = __shfl_xor(bz_lo, off), d = __shfl_xor(bz_hi, off);
        nn |= (uint32_t)__shfl_xor((int)nn, off);
        by_lo = a < by_lo ? a : by_lo; by_hi = b > by_hi ? b : by_hi; bz_lo = cc < bz_lo ? cc : bz_lo; bz_hi = d > bz_hi ? d : bz_hi;
    }
    if (live && (i & 63u) == 0u) bp.s_bb2[i >> 6] = nn ? make4<T>(-inf, inf, -inf, inf) : make4<T>(by_lo, by_hi, bz_lo, bz_hi);
}

// ---------------------------------------------------------------------------------------------------------
// u64 hash set (open addressing, linear probing); EMPTY = ~0
__device__ __forceinline__ uint64_t hs_mix(uint64_t k) {
    k ^= k >> 33; k *= 0xff51afd7ed558ccdull; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ull; k ^= k >> 33;
    return k;
}
__device__ __forceinline__ bool hs_contains(const uint64_t* __restrict__ tab, uint32_t cap_mask, uint64_t key) {
    uint32_t h = (uint32_t)hs_mix(key) & cap_mask;
    for (;;) {
        uint64_t v = tab[h];
        if (v == key) return true;
        if (v == ~0ull) return false;
        h = (h + 1) & cap_mask;
    }
}
__global__ __launch_bounds__(256) void k_hs_insert(uint64_t* tab, uint32_t cap_mask, const uint64_t* __restrict__ keys, uint32_t n) {
    uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    uint64_t key = keys[i];
    uint32_t h = (uint32_t)hs_mix(key) & cap_mask;
    for (;;) {
        unsigned long long prev = atomicCAS((unsigned long long*)&tab[h], ~0ull, (unsigned long long)key);
        if (prev == ~0ull || prev == key) return;
        h = (h + 1) & cap_mask;
    }
}
__global__ __launch_bounds__(256) void k_hs_insert_pairs(uint64_t* tab, uint32_t cap_mask, const avn_pair* __restrict__ pairs, uint32_t n) {
    uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    uint32_t a = pairs[i].collider1, b = pairs[i].collider2;
    uint64_t key = a < b ? ((uint64_t)a << 32) | b : ((uint64_t)b << 32) | a;
    uint32_t h = (uint32_t)hs_mix(key) & cap_mask;
    for (;;) {
        unsigned long long prev = atomicCAS((unsigned long long*)&tab[h], ~0ull, (unsigned long long)key);
        if (prev == ~0ull || prev == key) return;
        h = (h + 1) & cap_mask;
    }
}
// removal leaves a tombstone (~0 - 1): probes walk over it, inserts do not reuse it (the host rebuilds the set when it fills)
__global__ __launch_bounds__(256) void k_hs_remove(uint64_t* tab, uint32_t cap_mask, const uint64_t* __restrict__ keys, uint32_t n) {
    uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    uint64_t key = keys[i];
    uint32_t h = (uint32_t)hs_mix(key) & cap_mask;
    for (;;) {
        uint64_t v = tab[h];
        if (v == key) { tab[h] = ~0ull - 1ull; return; }
        if (v == ~0ull) return;
        h = (h + 1) & cap_mask;
    }
}
void launch_hs_remove(uint64_t* tab, uint32_t cap, const uint64_t* keys, uint32_t n, hipStream_t s) {
    if (n && cap) hipLaunchKernelGGL(k_hs_remove, dim3((n + 255) / 256), dim3(256), 0, s, tab, cap - 1, keys, n);
}
void launch_hs_insert(uint64_t* tab, uint32_t cap, const uint64_t* keys, uint32_t n, hipStream_t s) {
    if (n) hipLaunchKernelGGL(k_hs_insert, dim3((n + 255) / 256), dim3(256), 0, s, tab, cap - 1, keys, n);
}
void launch_hs_insert_pairs(uint64_t* tab, uint32_t cap, const avn_pair* pairs, uint32_t n, hipStream_t s) {
    if (n) hipLaunchKernelGGL(k_hs_insert_pairs, dim3((n + 255) / 256), dim3(256), 0, s, tab, cap - 1, pairs, n);
}

// ---------------------------------------------------------------------------------------------------------
// sweep
//
// For interval i (sorted by min.x) the reference walks j = i+1.. until min_x[j] > max_x[i] (broad_phase.rs:387-392).
// Because the list is sorted, that stop index end(i) is a pure function of the sorted keys: k_sweep_ranges finds it by
// binary search, so the sweep itself has NO data-dependent loop exit.
//
// k_sweep: one WAVE owns 64 consecutive intervals (lane = i).  The candidate index j is wave-uniform, so the candidate's
// (min.y, max.y, min.z, max.z) record is fetched with SCALAR loads (s_load_dwordx4/x8 through the scalar cache) eight at
// a time and compared against the lane's own box held in VGPRs: ~7 VALU instructions per 64 tests, no LDS traffic, no
// barriers.  Geometric hits are rare (a few per thousand tests); they are compacted with ballot/popcount into a
// per-wave LDS queue of (lane, j) entries and filtered 64 at a time (layers, same body, inactive, ContactGraph pair set,
// joint-disabled set: random 8-byte probes, latency amortised over the wave).  Entries of one lane stay in ascending j
// order, so every lane emits at its own running offset => the output is already in the reference's (i asc, j asc) order.
//
// Intervals with more than SW_CAP candidates (a ground slab that spans the scene) would serialise a wave: they are cut
// into chunks of SW_LCHUNK candidates by k_sweep_ranges and swept by k_sweep_long with a whole workgroup per chunk
// (lane = j, coalesced vector loads); k_long_finish turns the chunk counts into in-interval offsets.
#define SW_WAVES 4
#define SW_THREADS (64 * SW_WAVES)
#define SW_Q 1024   // ring slots per wave: < 64 carried + 8 x 64 appended per batch
#define SW_HIT_WORDS 16u     // per wave of k_sweep, one 64-byte line: [0] pairs the count pass found, [1 ..] the first SW_HIT_RECORDS of them
#define SW_HIT_RECORDS 15u
#define SW_CAP 65536u   // absolute long-interval rule; the two-level batch cull below keeps lattice layers of ~10^4 x-overlapping candidates on the k_sweep path
#define SW_LONG_MIN 256u   // relative long-interval rule (k_sweep_ranges): never below this many candidates
#define SW_LCHUNK 512u    // candidates per LongItem: two iterations of a workgroup (4096 made the ground of cfg2 25 workgroups x 16 dependent iterations: 29 us)

// wave-wide compare masks (LLVM fcmp predicates: UGE = 11, ULE = 13); inactive lanes read 0
__device__ __forceinline__ unsigned long long lane_mask_ule(float a, float b) { return __builtin_amdgcn_fcmpf(a, b, 13); }
__device__ __forceinline__ unsigned long long lane_mask_uge(float a, float b) { return __builtin_amdgcn_fcmpf(a, b, 11); }
__device__ __forceinline__ unsigned long long lane_mask_ule(double a, double b) { return __builtin_amdgcn_fcmp(a, b, 13); }
__device__ __forceinline__ unsigned long long lane_mask_uge(double a, double b) { return __builtin_amdgcn_fcmp(a, b, 11); }

struct LongItem { uint32_t i, j_start, j_end, n_chunks; };  // n_chunks != 0 only on the first chunk of an interval

struct PairSets { const uint64_t* pair_set; uint32_t pair_set_cap; const uint64_t* disabled_set; uint32_t disabled_cap; };

// pair filters of broad_phase.rs:407-439 (after the x / y / z tests); true when (1, 2) becomes a new pair
__device__ __forceinline__ bool pair_filters(const PairSets& hs, uint4 in1, uint32_t f1, uint4 in2, uint32_t f2) {
    bool interacts = (in1.z & in2.w) != 0 && (in2.z & in1.w) != 0;                         // CollisionLayers::interacts_with
    if ((f1 & f2 & AVN_AABB_IS_INACTIVE) || !interacts || in1.y == in2.y) return false;
    uint64_t key = in1.x < in2.x ? ((uint64_t)in1.x << 32) | in2.x : ((uint64_t)in2.x << 32) | in1.x;
    if (hs.pair_set_cap && hs_contains(hs.pair_set, hs.pair_set_cap - 1, key)) return false;
    if (hs.disabled_cap) {
        uint64_t bk = in1.y < in2.y ? ((uint64_t)in1.y << 32) | in2.y : ((uint64_t)in2.y << 32) | in1.y;
        if (hs_contains(hs.disabled_set, hs.disabled_cap - 1, bk)) return false;
    }
    return true;
}
__device__ __forceinline__ avn_pair make_pair(uint4 in1, uint32_t f1, uint4 in2, uint32_t f2) {
    uint32_t u = f1 | f2;
    avn_pair pr;
    pr.collider1 = in1.x; pr.collider2 = in2.x; pr.body1 = (int)in1.y; pr.body2 = (int)in2.y;
    pr.flags = ((u & AVN_AABB_CONTACT_EVENTS) ? AVN_PAIR_CONTACT_EVENTS : 0u) | ((u & AVN_AABB_MODIFY_CONTACTS) ? AVN_PAIR_MODIFY_CONTACTS : 0u) |
               ((u & AVN_AABB_GENERATE_CONSTRAINTS) ? AVN_PAIR_GENERATE_CONSTRAINTS : 0u) | ((u & AVN_AABB_CUSTOM_FILTER) ? AVN_PAIR_NEEDS_CUSTOM_FILTER : 0u);
    pr.reserved = 0;
    return pr;
}

#define SW_BB_GROUP 8u   // sorted records per bounds group (written by k_gather_sorted; the sweep's `j / SW_BB_GROUP`, the host's s_bb sizing)
// Batch cull of the sweep: the y/z bounds of every group of 8 consecutive sorted records.  A wave tests its 64 boxes against a
// group's bounds before it loads the group's records; on a lattice consecutive records share a y row, so most groups of the
// ~6 000 x-overlapping candidates of an interval are rejected with 4 compares instead of 32.  Conservative by construction
// (a record that overlaps a lane overlaps the union of its group; a NaN anywhere in a group disables the cull for it, because the
// reference's negated compares let NaN boxes through): the emitted pairs and their order do not change.
uint32_t sweep_bounds_group() { return SW_BB_GROUP; }
// Second level: the bounds of every 64 consecutive sorted records (8 groups).  cfg5's lattice layers hold 5 000 boxes with equal min.x, so an
// interval has ~10^4 x-overlapping candidates of which the ~5 rows that overlap it in y are ~500: a wave that tests 64 records with four
// compares skips whole rows (records of one layer are consecutive in (y, z) upload order) and only descends into the few groups of its
// own and the neighbouring rows.  Same conservativeness rules as level one (NaN anywhere in the 64: never culled).
#define SW_BB2_GROUP 64u
uint32_t sweep_bounds_words(uint32_t n_records) { return n_records / SW_BB_GROUP + 2u + n_records / SW_BB2_GROUP + 2u; }   // Vec4 records of both levels
uint32_t sweep_bounds_level2_offset(uint32_t n_records) { return n_records / SW_BB_GROUP + 2u; }
static_assert(SW_BB_GROUP == 8u && SW_BB2_GROUP == 64u, "k_gather_sorted reduces the cull bounds over octets and whole waves");

// end(i) = first j > i with min_x[j] > max_x[i]; long intervals are cut into LongItems.
// "Long" is absolute (more than SW_CAP candidates) or RELATIVE: a wave of k_sweep walks the union of its 64 lanes' candidate
// ranges, so one interval that reaches much further than its 63 neighbours (a ground plate among boxes: 5 500 candidates
// against ~200 in the reference's pyramid scenes) drags the whole wave through its range on the slow per-lane-bounds path --
// 160 us of a 5 k-box step.  An interval with more than SW_LONG_MIN candidates and more than 4x the mean of the other lanes
// of its wave goes to the long path too (at most 16 lanes of a wave can satisfy that, so the item capacity holds).  The
// split only moves work between two order-exact paths: the pair list does not change.
template <class T>
__global__ __launch_bounds__(256) void k_sweep_ranges(uint32_t n, const T* __restrict__ s_minx, const T* __restrict__ s_maxx, uint32_t* __restrict__ s_end,
                                                       uint32_t* __restrict__ s_flags, LongItem* __restrict__ items, uint32_t* __restrict__ n_long,
                                                       uint32_t long_cap, uint32_t* __restrict__ overflow) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;   // wave w of the block = the 64 intervals of one k_sweep wave
    const bool valid = i < n;
    const uint32_t f = valid ? s_flags[i] : 0u;
    uint32_t end = i + 1;
    if (valid && !(f & AVN_IV_DROPPED)) {
        T mx = s_maxx[i];
        uint32_t lo = i + 1, hi = n;  // first index in [lo, hi] whose min.x > mx  (x test: `min_x[j] > max_x[i]` ends the sweep)
        while (lo < hi) {
            uint32_t mid = lo + ((hi - lo) >> 1);
            if (s_minx[mid] > mx) hi = mid; else lo = mid + 1;
        }
        end = lo;
    }
    const uint32_t len = valid ? end - (i + 1) : 0u;
    uint32_t sum = len;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) sum += (uint32_t)__shfl_xor((int)sum, off);
    const uint32_t others = (sum - len) / 63u;
    if (!valid) return;
    if (len > SW_CAP || (len > SW_LONG_MIN && len > 4u * others)) {
        uint32_t nch = (len + SW_LCHUNK - 1) / SW_LCHUNK;
        uint32_t k = atomicAdd(n_long, nch);
        if (k + nch > long_cap) { *overflow = 1u; }
        else
            for (uint32_t c = 0; c < nch; ++c) {
                uint32_t js = i + 1 + c * SW_LCHUNK;
                items[k + c] = LongItem{i, js, min(js + SW_LCHUNK, end), c == 0 ? nch : 0u};
            }
        s_flags[i] = f | AVN_IV_LONG;
        end = i + 1;  // nothing left for k_sweep
    }
    s_end[i] = end;
}

template <class T, bool EMIT>
__global__ __launch_bounds__(SW_THREADS) void k_sweep(uint32_t n, const Vec4<T>* __restrict__ s_yz, const Vec4<T>* __restrict__ s_bb, const Vec4<T>* __restrict__ s_bb2, const uint32_t* __restrict__ s_end,
                                                       const uint4* __restrict__ s_info, const uint32_t* __restrict__ s_flags, PairSets hs,
                                                       uint32_t* __restrict__ counts, const uint32_t* __restrict__ offsets, avn_pair* __restrict__ out, uint32_t* __restrict__ hits) {
    __shared__ uint32_t l_q[SW_WAVES][SW_Q];
    __shared__ uint32_t l_hits[SW_WAVES];
    __shared__ uint4 l_info[SW_WAVES][64];
    __shared__ uint32_t l_flags[SW_WAVES][64];
    __shared__ uint32_t l_cnt[SW_WAVES][64];  // count pass: pairs found; emit pass: running output position
    const uint32_t lane = threadIdx.x & 63u;
    // threadIdx.x >> 6 is wave-uniform, but the compiler cannot know: readfirstlane makes it (and i0, j) provably
    // uniform so that the candidate records are fetched with scalar loads
    const uint32_t wv = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    // a workgroup owns 64 consecutive intervals; its SW_WAVES waves each sweep one contiguous quarter of the candidate
    // range (more waves in flight to hide the scalar-load latency, and 4x shorter serial loops).  Output stays ordered:
    // counts/offsets are kept per (interval, quarter), quarter-major inside the interval.
    const uint32_t i0 = xcd_block(blockIdx.x, gridDim.x) * 64u;
    if (i0 >= n) return;  // workgroup-uniform; the kernel has no workgroup barrier
    const uint32_t i = i0 + lane;
    const bool valid = i < n;
    Vec4<T> me = valid ? s_yz[i] : make4<T>(0, 0, 0, 0);  // (min.y, max.y, min.z, max.z)
    const uint32_t end_i = valid ? s_end[i] : 0u;
    // lanes without candidates (padding, dropped, long intervals: end = i + 1) get a box nothing overlaps, so that the
    // fast path below can leave out the per-lane index tests
    const bool has_candidates = valid && end_i > i + 1u;
    if (!has_candidates) { const T inf = Limits<T>::max * T(2); me = make4<T>(inf, -inf, inf, -inf); }
    const uint32_t my_flags = valid ? s_flags[i] : 0u;
    if (EMIT) {
        // the emit pass repeats the sweep only where the count pass found something: a wave whose 64 (interval, quarter) slots are all
        // empty has nothing to write (a settled scene finds a few hundred new pairs among 4 x 10^5 slots)
        const uint32_t mine = (valid && !(my_flags & AVN_IV_LONG)) ? counts[i * SW_WAVES + wv] : 0u;
        if (!__any(mine != 0u)) return;   // wave-uniform; the kernel has no workgroup barrier
        // ... and where it found only a few, it left them as (lane, j) records: no second sweep, the pairs are written from the records
        // (a settled pile's few thousand new pairs per step are spread over a third of the waves, one or two each).  A lane's pairs go
        // out in ascending j: the position of a record is its rank among the records of its lane.
        const uint32_t* __restrict__ rec = hits + (size_t)((i0 >> 6) * SW_WAVES + wv) * SW_HIT_WORDS;
        const uint32_t nh = (uint32_t)__builtin_amdgcn_readfirstlane((int)rec[0]);
        if (nh <= SW_HIT_RECORDS) {
            const uint32_t e = lane < nh ? rec[1u + lane] : 0xFFFFFFFFu;
            uint32_t rank = 0;
            for (uint32_t k = 0; k < nh; ++k) {
                const uint32_t ek = (uint32_t)__shfl((int)e, (int)k);
                rank += ((ek >> 26) == (e >> 26) && ek < e) ? 1u : 0u;
            }
            if (lane < nh) {
                const uint32_t src = e >> 26, jj = i0 + (e & 0x3FFFFFFu);
                out[offsets[(i0 + src) * SW_WAVES + wv] + rank] = make_pair(s_info[i0 + src], s_flags[i0 + src], s_info[jj], s_flags[jj]);
            }
            return;
        }
    } else {
        if (lane == 0) l_hits[wv] = 0u;
        __builtin_amdgcn_wave_barrier();
    }
    l_info[wv][lane] = valid ? s_info[i] : make_uint4(0, 0, 0, 0);
    l_flags[wv][lane] = my_flags;
    l_cnt[wv][lane] = (EMIT && valid) ? offsets[i * SW_WAVES + wv] : 0u;
    uint32_t je = end_i;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) je = max(je, (uint32_t)__shfl_xor((int)je, off));
    je = (uint32_t)__builtin_amdgcn_readfirstlane((int)je);
    uint32_t min_end = has_candidates ? end_i : 0xFFFFFFFFu;  // every candidate below it is inside EVERY active lane's range
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) min_end = min(min_end, (uint32_t)__shfl_xor((int)min_end, off));
    min_end = (uint32_t)__builtin_amdgcn_readfirstlane((int)min_end);
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    uint32_t qn = 0;  // wave-uniform queue fill

    // filter up to 64 queued (lane, j) entries at once; the queue is a ring of SW_Q slots starting at `head`
    uint32_t head = 0;
    auto drain = [&](uint32_t nq) {
        bool act = lane < nq;
        uint32_t e = act ? l_q[wv][(head + lane) & (SW_Q - 1u)] : 0u;
        uint32_t src = e >> 26, jj = i0 + (e & 0x3FFFFFFu);
        bool pass = false;
        uint4 in1 = make_uint4(0, 0, 0, 0), in2 = in1;
        uint32_t f1 = 0, f2 = 0;
        if (act) {
            in1 = l_info[wv][src]; f1 = l_flags[wv][src];
            in2 = s_info[jj]; f2 = s_flags[jj];
            pass = pair_filters(hs, in1, f1, in2, f2);
        }
        if (!EMIT) {
            if (pass) {
                atomicAdd(&l_cnt[wv][src], 1u);
                const uint32_t h = atomicAdd(&l_hits[wv], 1u);
                if (h < SW_HIT_RECORDS) hits[(size_t)((i0 >> 6) * SW_WAVES + wv) * SW_HIT_WORDS + 1u + h] = e;
            }
        } else {
            // deterministic in-order positions: entries of one source lane are in ascending j (= ascending queue slot)
            unsigned long long rem = __ballot(pass);
            uint32_t pos = 0;
            while (rem) {
                int leader = __ffsll((long long)rem) - 1;
                uint32_t s = (uint32_t)__shfl((int)src, leader);
                unsigned long long m = __ballot(pass && src == s);
                uint32_t base = l_cnt[wv][s];
                if (pass && src == s) pos = base + (uint32_t)__popcll(m & lt_mask);
                __builtin_amdgcn_wave_barrier();
                if ((int)lane == leader) l_cnt[wv][s] = base + (uint32_t)__popcll(m);
                __builtin_amdgcn_wave_barrier();
                rem &= ~m;
            }
            if (pass) out[pos] = make_pair(in1, f1, in2, f2);
        }
        __builtin_amdgcn_wave_barrier();
    };

    constexpr uint32_t SW_BATCH = sizeof(T) == 4 ? 8u : 4u;  // candidates per scalar-load batch (SGPR budget)
    // the batch cull below tests a whole batch against ONE group's bounds: a batch must never straddle two groups (else true pairs of the
    // second group could be culled), i.e. batches start on multiples of SW_BATCH (j_first) and SW_BATCH divides the group size
    static_assert(SW_BB_GROUP % SW_BATCH == 0 && (SW_BB_GROUP & (SW_BB_GROUP - 1u)) == 0, "sweep batch cull: SW_BATCH must divide the bounds group size");
    // this wave's quarter [jb, jq) of the workgroup's candidate range [i0 + 1, je), cut at positions aligned to the LEVEL-TWO groups (64
    // records) so that every batch lies inside one group of each level of k_batch_bounds (i0 is a multiple of 64: the range starts at the
    // workgroup's own records, which fail every lane's `jj > i` test or are the lane's genuine later neighbours)
    static_assert(SW_BB2_GROUP % SW_BB_GROUP == 0, "sweep batch cull: level two must be made of whole level-one groups");
    const uint32_t j_first = i0;
    uint32_t qlen = (je - j_first + SW_WAVES - 1u) / SW_WAVES;
    qlen = (qlen + SW_BB2_GROUP - 1u) / SW_BB2_GROUP * SW_BB2_GROUP;
    const uint32_t jb = j_first + wv * qlen;
    const uint32_t jq = min(je, jb + qlen);
    for (uint32_t j2 = jb; j2 < jq; j2 += SW_BB2_GROUP) {
      const Vec4<T> b2 = s_bb2[j2 / SW_BB2_GROUP];   // wave-uniform: one scalar load per 64 candidates
      if (!(lane_mask_ule(me.x, b2.y) & lane_mask_uge(me.y, b2.x) & lane_mask_ule(me.z, b2.w) & lane_mask_uge(me.w, b2.z))) continue;
      const uint32_t j2e = min(jq, j2 + SW_BB2_GROUP);
      for (uint32_t j = j2; j < j2e; j += SW_BATCH) {
        const Vec4<T> bb = s_bb[j / SW_BB_GROUP];   // wave-uniform: one scalar load
        if (!(lane_mask_ule(me.x, bb.y) & lane_mask_uge(me.y, bb.x) & lane_mask_ule(me.z, bb.w) & lane_mask_uge(me.w, bb.z))) continue;
        Vec4<T> c[SW_BATCH];
#pragma unroll
        for (uint32_t k = 0; k < SW_BATCH; ++k) c[k] = s_yz[j + k];  // wave-uniform, contiguous: wide scalar loads (s_yz is padded by sweep_pad_records())
        unsigned long long hm[SW_BATCH], any = 0ull;
        // y / z rejection exactly as broad_phase.rs:394-403 (strict compares: touching counts as overlapping);
        // bitwise & keeps the test branch-free: v_cmp whose SGPR masks are and-ed on the scalar unit
        if (j >= i0 + 64u && j + SW_BATCH <= min_end && j + SW_BATCH <= jq) {
            // interior batch (the common case): every candidate is after every lane's own index and inside every active
            // lane's [i + 1, end) range -> only the four box compares remain
#pragma unroll
            for (uint32_t k = 0; k < SW_BATCH; ++k) {
                // v_cmp straight into SGPR lane masks (no bool round trip through a VGPR): !(a > b) == ULE, !(a < b) == UGE
                hm[k] = lane_mask_ule(me.x, c[k].y) & lane_mask_uge(me.y, c[k].x) & lane_mask_ule(me.z, c[k].w) & lane_mask_uge(me.w, c[k].z);
                any |= hm[k];
            }
        } else {
#pragma unroll
            for (uint32_t k = 0; k < SW_BATCH; ++k) {
                uint32_t jj = j + k;
                bool hit = (jj > i) & (jj < end_i) & (jj < jq) & !(me.x > c[k].y) & !(me.y < c[k].x) & !(me.z > c[k].w) & !(me.w < c[k].z);
                hm[k] = __ballot(hit);
                any |= hm[k];
            }
        }
        if (any) {  // rare: compact the hits into the queue (j ascending, then lane ascending)
#pragma unroll
            for (uint32_t k = 0; k < SW_BATCH; ++k) {
                if (hm[k]) {
                    if ((hm[k] >> lane) & 1ull) l_q[wv][(head + qn + (uint32_t)__popcll(hm[k] & lt_mask)) & (SW_Q - 1u)] = (lane << 26) | (j + k - i0);
                    qn += (uint32_t)__popcll(hm[k]);
                }
            }
            __builtin_amdgcn_wave_barrier();
            while (qn >= 64u) { drain(64u); head = (head + 64u) & (SW_Q - 1u); qn -= 64u; }
        }
      }
    }
    if (qn) drain(qn);
    __builtin_amdgcn_wave_barrier();
    if (!EMIT && lane == 0) hits[(size_t)((i0 >> 6) * SW_WAVES + wv) * SW_HIT_WORDS] = l_hits[wv];
    if (!EMIT && valid) {
        if (!(my_flags & AVN_IV_LONG)) counts[i * SW_WAVES + wv] = l_cnt[wv][lane];
        else if (wv != 0) counts[i * SW_WAVES + wv] = 0u;  // slot 0 of a long interval is written by k_long_finish
    }
}

// One workgroup per LongItem chunk (grid-stride): lane = candidate j.
template <class T, bool EMIT>
__global__ __launch_bounds__(SW_THREADS) void k_sweep_long(const Vec4<T>* __restrict__ s_yz, const uint4* __restrict__ s_info, const uint32_t* __restrict__ s_flags,
                                                            PairSets hs, const LongItem* __restrict__ items, const uint32_t* __restrict__ n_long,
                                                            uint32_t* __restrict__ long_counts, const uint32_t* __restrict__ long_off,
                                                            const uint32_t* __restrict__ offsets, avn_pair* __restrict__ out) {
    __shared__ uint32_t wave_tot[SW_WAVES];
    uint32_t t = threadIdx.x, lane = t & 63u, wv = t >> 6;
    // (n_long[1] = k_sweep_ranges ran out of chunk slots: the reserved items were never written -- nothing here may touch them;
    //  the host grows the chunk arrays to the requested count and runs the count pass again)
    uint32_t nl = n_long[1] ? 0u : *n_long;
    for (uint32_t it = blockIdx.x; it < nl; it += gridDim.x) {
        LongItem item = items[it];
        uint32_t i = item.i;
        Vec4<T> me = s_yz[i];
        uint4 in1 = s_info[i];
        uint32_t f1 = s_flags[i];
        uint32_t running = 0;
        uint32_t base = EMIT ? offsets[i * SW_WAVES] + long_off[it] : 0u;
        for (uint32_t j0 = item.j_start; j0 < item.j_end; j0 += SW_THREADS) {
            uint32_t j = j0 + t;
            bool ok = false;
            uint4 in2 = make_uint4(0, 0, 0, 0);
            uint32_t f2 = 0;
            if (j < item.j_end) {
                Vec4<T> c = s_yz[j];
                if (!(me.x > c.y || me.y < c.x) && !(me.z > c.w || me.w < c.z)) {
                    in2 = s_info[j]; f2 = s_flags[j];
                    ok = pair_filters(hs, in1, f1, in2, f2);
                }
            }
            unsigned long long bal = __ballot(ok);
            if (lane == 0) wave_tot[wv] = (uint32_t)__popcll(bal);
            __syncthreads();
            uint32_t before = 0, total = 0;
#pragma unroll
            for (uint32_t k = 0; k < SW_WAVES; ++k) { uint32_t v = wave_tot[k]; if (k < wv) before += v; total += v; }
            if (EMIT && ok) out[base + running + before + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull))] = make_pair(in1, f1, in2, f2);
            running += total;
            __syncthreads();
        }
        if (!EMIT && t == 0) long_counts[it] = running;
    }
}

// per long interval: chunk counts -> in-interval chunk offsets, and the interval's total into counts[i].  One WAVE per interval (round 4: one
// lane walked cfg2's ground -- 196 chunks of 512 candidates -- load by dependent load, 30 us on the broad phase's chain): 64 chunks per
// round, a wave scan, the running total carried in a register.
__global__ __launch_bounds__(256) void k_long_finish(const LongItem* __restrict__ items, const uint32_t* __restrict__ n_long, const uint32_t* __restrict__ long_counts,
                                                      uint32_t* __restrict__ long_off, uint32_t* __restrict__ counts) {
    const uint32_t nl = n_long[1] ? 0u : *n_long;
    const uint32_t lane = threadIdx.x & 63u, wave = (blockIdx.x * 256 + threadIdx.x) >> 6, n_waves = (gridDim.x * 256) >> 6;
    for (uint32_t k = wave; k < nl; k += n_waves) {   // (wave-uniform)
        const LongItem it = items[k];
        if (!it.n_chunks) continue;
        uint32_t run = 0;
        for (uint32_t c0 = 0; c0 < it.n_chunks; c0 += 64u) {
            const uint32_t c = c0 + lane;
            const uint32_t v = c < it.n_chunks ? long_counts[k + c] : 0u;
            uint32_t incl = v;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) { const uint32_t u = (uint32_t)__shfl_up((int)incl, off); if ((int)lane >= off) incl += u; }
            if (c < it.n_chunks) long_off[k + c] = run + incl - v;
            run += (uint32_t)__shfl((int)incl, 63);
        }
        if (lane == 0) counts[it.i * SW_WAVES] = run;
    }
}

// ---------------------------------------------------------------------------------------------------------
// launchers
template <class T> bool launch_update_aabb(const DW<T>& w, const BP<T>& bp, const StepParams<T>& p, hipStream_t s, uint32_t* zero_words, uint32_t n_zero) {
    if (bp.n_colliders) hipLaunchKernelGGL(k_update_aabb<T>, dim3((bp.n_colliders + 255) / 256), dim3(256), 0, s, w, bp, p, zero_words, n_zero < 256u ? n_zero : 256u);
    return bp.n_colliders != 0;   // false: nothing was launched, nothing was cleared
}
template <class T> void launch_interval_keys(const DW<T>& w, const BP<T>& bp, typename BP<T>::Key* keys, uint32_t* vals, uint32_t* n_dropped, hipStream_t s, bool counters_clean) {
    if (!counters_clean) (void)hipMemsetAsync(n_dropped, 0, 2 * sizeof(uint32_t), s);  // [n_dropped, unsorted]
    if (bp.n_intervals) hipLaunchKernelGGL(k_interval_keys<T>, dim3((bp.n_intervals + 255) / 256), dim3(256), 0, s, w, bp, keys, vals, n_dropped);
}
uint32_t radix_blocks(uint32_t n) { return (n + RS_TILE - 1) / RS_TILE; }
uint32_t radix_pass_launches(uint32_t n) { uint32_t nb = radix_blocks(n); return nb <= RS_FUSED_MAX_BLOCKS ? 2u : 2u + exclusive_scan_launches(256 * nb); }
// Small scenes (the reference's own 5 k-box benches): the whole sort in ONE launch.  Four radix passes of two launches each
// cost ~75 us at 5.6 k keys -- every launch on its latency floor -- so up to SS_MAX 32-bit keys are sorted by one workgroup
// whose 8 waves play the tiles of the multi-launch sort: per pass a per-wave digit histogram in LDS, the (digit-major,
// wave-minor) exclusive scan, the same ballot-ranked stable scatter as k_radix_scatter, a workgroup barrier where the
// launches had a kernel boundary.  (A one-workgroup bitonic network over key << 32 | position composites was tried first:
// 61-68 us -- 373 k 64-bit compare-exchanges are ~50 us of VALU time on ONE CU however they are scheduled.)
#define SS_MAX 8192u
#define SS_WAVES 8
#define SS_THREADS (64 * SS_WAVES)
// the sort of one tile of <= SS_MAX (key, value) pairs by ONE workgroup: `bits` / 8 passes, ping-pong between (keys_a, vals_a) and (keys_b, vals_b)
__device__ __forceinline__ void sort_small_body(uint32_t* keys_a, uint32_t* vals_a, uint32_t* keys_b, uint32_t* vals_b, uint32_t n, uint32_t bits) {
    __shared__ uint32_t cnt[SS_WAVES][256];    // per-wave digit counts, then the wave's running output offsets
    __shared__ uint32_t tot[256];
    const uint32_t t = threadIdx.x, lane = t & 63u;
    const uint32_t wv = (uint32_t)__builtin_amdgcn_readfirstlane((int)(t >> 6));
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    // the keys are dealt to ALL eight waves in equal tiles of `rounds` x 64 (a settled pile's overflow colour sorts ~600 entries: with fixed
    // 1 024-key tiles one wave walked ten rounds while seven watched), and only the passes the key width needs are made
    const uint32_t rounds = max(1u, (n + SS_THREADS - 1u) / SS_THREADS);   // <= RS_ROUNDS for n <= SS_MAX
    const uint32_t base = wv * rounds * 64u;
    uint32_t* ki = keys_a; uint32_t* vi = vals_a; uint32_t* ko = keys_b; uint32_t* vo = vals_b;
    for (uint32_t shift = 0; shift < bits; shift += 8u) {
        uint32_t key[RS_ROUNDS], val[RS_ROUNDS];
#pragma unroll
        for (uint32_t r = 0; r < RS_ROUNDS; ++r) {   // the wave's tile, all loads in flight at once
            const uint32_t idx = base + r * 64u + lane;
            const bool in = r < rounds && idx < n;   // (r < rounds is wave-uniform)
            key[r] = in ? ki[idx] : 0u;
            val[r] = in ? vi[idx] : 0u;
        }
        for (uint32_t d = lane; d < 256u; d += 64u) cnt[wv][d] = 0u;
        __syncthreads();
#pragma unroll
        for (uint32_t r = 0; r < RS_ROUNDS; ++r)
            if (r < rounds && base + r * 64u + lane < n) atomicAdd(&cnt[wv][(key[r] >> shift) & 255u], 1u);
        __syncthreads();
        // thread d < 256: digit total, and the exclusive prefix over the waves (the tiles before this one)
        if (t < 256u) {
            uint32_t run = 0;
#pragma unroll
            for (uint32_t w = 0; w < SS_WAVES; ++w) { const uint32_t c = cnt[w][t]; cnt[w][t] = run; run += c; }
            tot[t] = run;
        }
        __syncthreads();
        if (t < 64u) {   // exclusive scan over the 256 digits: lane owns digits 4 lane .. 4 lane + 3
            const uint32_t a0 = tot[4 * t], a1 = tot[4 * t + 1], a2 = tot[4 * t + 2], a3 = tot[4 * t + 3];
            const uint32_t mine = a0 + a1 + a2 + a3;
            uint32_t incl = mine;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) { const uint32_t v = (uint32_t)__shfl_up((int)incl, off); if ((int)t >= off) incl += v; }
            const uint32_t excl = incl - mine;
            tot[4 * t] = excl; tot[4 * t + 1] = excl + a0; tot[4 * t + 2] = excl + a0 + a1; tot[4 * t + 3] = excl + a0 + a1 + a2;
        }
        __syncthreads();
        for (uint32_t d = lane; d < 256u; d += 64u) cnt[wv][d] += tot[d];   // the wave's start offset per digit
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (uint32_t r = 0; r < RS_ROUNDS; ++r) {
            if (r < rounds) {   // (wave-uniform)
            const bool valid = base + r * 64u + lane < n;
            const uint32_t digit = (key[r] >> shift) & 255u;
            unsigned long long peers = __ballot(valid);   // match-any over the 8 digit bits
#pragma unroll
            for (uint32_t bit = 0; bit < 8; ++bit) {
                const bool set = (digit >> bit) & 1u;
                const unsigned long long m = __ballot(set);
                peers &= set ? m : ~m;
            }
            const uint32_t rank = (uint32_t)__popcll(peers & lt_mask);
            uint32_t pos = 0;
            if (valid) pos = cnt[wv][digit] + rank;   // all peers read the running offset before the leader bumps it
            __builtin_amdgcn_wave_barrier();
            if (valid && rank == 0) cnt[wv][digit] += (uint32_t)__popcll(peers);
            __builtin_amdgcn_wave_barrier();
            if (valid) { ko[pos] = key[r]; vo[pos] = val[r]; }
            }
        }
        __syncthreads();   // (workgroup-scope release/acquire: the next pass reads what other waves wrote)
        uint32_t* tk = ki; ki = ko; ko = tk;
        uint32_t* tv = vi; vi = vo; vo = tv;
    }
}
__global__ __launch_bounds__(SS_THREADS) void k_sort_small(uint32_t* keys_a, uint32_t* vals_a, uint32_t* keys_b, uint32_t* vals_b, uint32_t n, const uint32_t* __restrict__ unsorted, uint32_t bits) {
    if (unsorted && *unsorted == 0u) return;   // the persistent order is still sorted
    sort_small_body(keys_a, vals_a, keys_b, vals_b, n, bits);
}
// Medium sorts (SS_MAX < n <= SM_TILE x SM_MAX_TILES keys; round 4: cfg2's 100 k intervals took four radix passes of two launches each,
// 155 us on the closed loop's critical path): TWO launches.  k_sort_tiles: every workgroup sorts one tile of SM_TILE pairs as k_sort_small
// does (tile-local ping-pong inside the two buffers).  k_merge_tiles: an element's final position is its position in its own tile plus,
// for every EARLIER tile, the number of keys <= its key and, for every LATER tile, the number of keys < its key -- exactly the stable
// order (equal keys keep their input order: earlier tile first, tile order inside a tile).  The tiles' first / last keys sit in LDS and
// decide most (element, tile) pairs without a search: the broad phase sorts LAST frame's order by this frame's keys, so a tile's key range
// overlaps its neighbours' at most; only there an element binary-searches.  Worst case (unrelated order: a scene's first frame) every
// element searches every tile: T x log2(SM_TILE) dependent loads from L2 -- slower than the radix passes, once.
#define SM_TILE 2048u
#define SM_MAX_TILES 128u
__global__ __launch_bounds__(SS_THREADS) void k_sort_tiles(uint32_t* keys_a, uint32_t* vals_a, uint32_t* keys_b, uint32_t* vals_b, uint32_t n, const uint32_t* __restrict__ unsorted, uint32_t bits) {
    if (unsorted && *unsorted == 0u) return;
    const uint32_t t0 = blockIdx.x * SM_TILE;
    sort_small_body(keys_a + t0, vals_a + t0, keys_b + t0, vals_b + t0, min(SM_TILE, n - t0), bits);
}
__global__ __launch_bounds__(256) void k_merge_tiles(const uint32_t* __restrict__ kin, const uint32_t* __restrict__ vin, uint32_t* __restrict__ kout, uint32_t* __restrict__ vout, uint32_t n,
                                                     const uint32_t* __restrict__ unsorted) {
    __shared__ uint32_t s_first[SM_MAX_TILES], s_last[SM_MAX_TILES];
    const uint32_t e = blockIdx.x * 256 + threadIdx.x;
    if (unsorted && *unsorted == 0u) {   // (uniform) nothing was sorted: the result is the input, in the output buffers
        if (e < n) { kout[e] = kin[e]; vout[e] = vin[e]; }
        return;
    }
    const uint32_t T = (n + SM_TILE - 1u) / SM_TILE;
    if (threadIdx.x < T) {
        const uint32_t t0 = threadIdx.x * SM_TILE, nt = min(SM_TILE, n - t0);
        s_first[threadIdx.x] = kin[t0]; s_last[threadIdx.x] = kin[t0 + nt - 1u];
    }
    __syncthreads();
    if (e >= n) return;
    const uint32_t key = kin[e], val = vin[e];
    const uint32_t t = e / SM_TILE;
    uint32_t pos = e - t * SM_TILE;
    for (uint32_t u = 0; u < T; ++u) {
        if (u == t) continue;
        const uint32_t t0 = u * SM_TILE, nt = min(SM_TILE, n - t0);
        const uint32_t first = s_first[u], last = s_last[u];
        if (u < t) {   // keys <= key of an earlier tile
            if (key >= last) pos += nt;
            else if (key >= first) {
                uint32_t lo = 0, hi = nt;
                while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (kin[t0 + mid] <= key) lo = mid + 1u; else hi = mid; }
                pos += lo;
            }
        } else {       // keys < key of a later tile
            if (key > last) pos += nt;
            else if (key > first) {
                uint32_t lo = 0, hi = nt;
                while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (kin[t0 + mid] < key) lo = mid + 1u; else hi = mid; }
                pos += lo;
            }
        }
    }
    kout[pos] = key; vout[pos] = val;
}
template <class K> static bool sort_small(K*, uint32_t*, K*, uint32_t*, uint32_t, const uint32_t*, K**, uint32_t**, hipStream_t) { return false; }
template <> bool sort_small<uint32_t>(uint32_t* keys_a, uint32_t* vals_a, uint32_t* keys_b, uint32_t* vals_b, uint32_t n, const uint32_t* unsorted, uint32_t** keys_out, uint32_t** vals_out, hipStream_t s) {
    if (n <= SS_MAX) {
        hipLaunchKernelGGL(k_sort_small, dim3(1), dim3(SS_THREADS), 0, s, keys_a, vals_a, keys_b, vals_b, n, unsorted, 32u);   // four passes: ends in (keys_a, vals_a)
        *keys_out = keys_a; *vals_out = vals_a;
        return true;
    }
    if (n <= SM_TILE * SM_MAX_TILES) {   // tiles (four passes: sorted tiles in (keys_a, vals_a)), then the merge into (keys_b, vals_b)
        hipLaunchKernelGGL(k_sort_tiles, dim3((n + SM_TILE - 1u) / SM_TILE), dim3(SS_THREADS), 0, s, keys_a, vals_a, keys_b, vals_b, n, unsorted, 32u);
        hipLaunchKernelGGL(k_merge_tiles, dim3((n + 255u) / 256u), dim3(256), 0, s, keys_a, vals_a, keys_b, vals_b, n, unsorted);
        *keys_out = keys_b; *vals_out = vals_b;
        return true;
    }
    return false;
}
uint32_t radix_sort_launches(uint32_t n, uint32_t key_bytes) { return key_bytes == 4 && n <= SS_MAX ? 1u : key_bytes == 4 && n <= SM_TILE * SM_MAX_TILES ? 2u : key_bytes * radix_pass_launches(n); }
template <class K> void launch_radix_sort(K* keys_a, uint32_t* vals_a, K* keys_b, uint32_t* vals_b, uint32_t n, uint32_t* hist, uint32_t* block_sums,
                                          const uint32_t* unsorted, K** keys_out, uint32_t** vals_out, hipStream_t s) {
    // Every kernel returns at once (or copies through) when *unsorted == 0 (the persistent interval order is still sorted: the reference's
    // insertion sort is O(n) then, ours is O(launch)); the result is in (*keys_out, *vals_out) either way.
    *keys_out = keys_a; *vals_out = vals_a;
    if (n == 0) return;
    if (sort_small<K>(keys_a, vals_a, keys_b, vals_b, n, unsorted, keys_out, vals_out, s)) return;   // one workgroup / tiles + merge: one or two launches
    // sizeof(K) passes of 8 bits: the result ends in (keys_a, vals_a) because the pass count is even
    uint32_t nb = radix_blocks(n);
    K* ki = keys_a; uint32_t* vi = vals_a; K* ko = keys_b; uint32_t* vo = vals_b;
    for (uint32_t pass = 0; pass < sizeof(K); ++pass) {
        uint32_t shift = pass * 8;
        if (nb <= RS_FUSED_MAX_BLOCKS) {
            hipLaunchKernelGGL((k_radix_hist<K, true>), dim3(nb), dim3(64), 0, s, ki, n, shift, hist, nb, unsorted);
            hipLaunchKernelGGL((k_radix_scatter<K, true>), dim3(nb), dim3(64), 0, s, ki, vi, ko, vo, n, shift, hist, nb, unsorted);
        } else {
            hipLaunchKernelGGL((k_radix_hist<K, false>), dim3(nb), dim3(64), 0, s, ki, n, shift, hist, nb, unsorted);
            launch_exclusive_scan(hist, hist, 256 * nb, block_sums, nullptr, s, unsorted);
            hipLaunchKernelGGL((k_radix_scatter<K, false>), dim3(nb), dim3(64), 0, s, ki, vi, ko, vo, n, shift, hist, nb, unsorted);
        }
        K* tk = ki; ki = ko; ko = tk;
        uint32_t* tv = vi; vi = vo; vo = tv;
    }
}
// the same sort on the low `bits` bits only (device ConstraintGraph: keys are body indices / colours); any n, no single-launch form
void launch_radix_sort_bits(uint32_t* keys_a, uint32_t* vals_a, uint32_t* keys_b, uint32_t* vals_b, uint32_t n, uint32_t bits, uint32_t* hist, uint32_t* block_sums,
                            uint32_t** keys_out, uint32_t** vals_out, hipStream_t s) {
    uint32_t* ki = keys_a; uint32_t* vi = vals_a; uint32_t* ko = keys_b; uint32_t* vo = vals_b;
    // a settled pile's overflow colour (a few hundred manifolds -> < 1 000 entries) and small scenes: the one-workgroup sort, 1 launch instead
    // of 2 per 8 bits (it sorts all 32 bits: the same permutation, the keys have no bits above `bits`)
    if (n && n <= SS_MAX && bits > 8) {
        const uint32_t passes = (bits + 7u) / 8u;   // an odd number of passes ends in the second pair of buffers
        hipLaunchKernelGGL(k_sort_small, dim3(1), dim3(SS_THREADS), 0, s, keys_a, vals_a, keys_b, vals_b, n, (const uint32_t*)nullptr, 8u * passes);
        *keys_out = (passes & 1u) ? keys_b : keys_a; *vals_out = (passes & 1u) ? vals_b : vals_a;
        return;
    }
    const uint32_t nb = radix_blocks(n);
    for (uint32_t shift = 0; n && shift < bits; shift += 8) {
        if (nb <= RS_FUSED_MAX_BLOCKS) {
            hipLaunchKernelGGL((k_radix_hist<uint32_t, true>), dim3(nb), dim3(64), 0, s, ki, n, shift, hist, nb, nullptr);
            hipLaunchKernelGGL((k_radix_scatter<uint32_t, true>), dim3(nb), dim3(64), 0, s, ki, vi, ko, vo, n, shift, hist, nb, nullptr);
        } else {
            hipLaunchKernelGGL((k_radix_hist<uint32_t, false>), dim3(nb), dim3(64), 0, s, ki, n, shift, hist, nb, nullptr);
            launch_exclusive_scan(hist, hist, 256 * nb, block_sums, nullptr, s, nullptr);
            hipLaunchKernelGGL((k_radix_scatter<uint32_t, false>), dim3(nb), dim3(64), 0, s, ki, vi, ko, vo, n, shift, hist, nb, nullptr);
        }
        uint32_t* tk = ki; ki = ko; ko = tk;
        uint32_t* tv = vi; vi = vo; vo = tv;
    }
    *keys_out = ki; *vals_out = vi;
}
template <class T> void launch_gather_sorted(const DW<T>& w, const BP<T>& bp, const uint32_t* sorted_collider, uint32_t n, hipStream_t s) {
    if (n) hipLaunchKernelGGL(k_gather_sorted<T>, dim3((n + 255) / 256), dim3(256), 0, s, w, bp, sorted_collider, n);
}
template <class T> void launch_sweep_ranges(const BP<T>& bp, uint32_t n, const SweepScratch& sc, hipStream_t s, bool counters_clean) {
    if (!n) return;
    if (!counters_clean) (void)hipMemsetAsync(sc.n_long, 0, 2 * sizeof(uint32_t), s);  // [n_long, overflow]
    hipLaunchKernelGGL(k_sweep_ranges<T>, dim3((n + 255) / 256), dim3(256), 0, s, n, bp.s_minx, bp.s_maxx, bp.s_end, bp.s_flags, (LongItem*)sc.long_items,
                       sc.n_long, sc.long_cap, sc.n_long + 1);
}
template <class T> void launch_sweep(const BP<T>& bp, uint32_t n, bool emit, const SweepScratch& sc, uint32_t* counts, const uint32_t* offsets, avn_pair* out,
                                     hipStream_t s) {
    if (!n) return;
    uint32_t nb = (n + 63u) / 64u;
    nb = ((nb + 7) / 8) * 8;
    LongItem* li = (LongItem*)sc.long_items;
    PairSets hs{bp.pair_set, bp.pair_set_cap, bp.disabled_set, bp.disabled_cap};
    if (emit) {
        hipLaunchKernelGGL((k_sweep<T, true>), dim3(nb), dim3(SW_THREADS), 0, s, n, bp.s_yz, bp.s_bb, bp.s_bb2, bp.s_end, bp.s_info, bp.s_flags, hs, counts, offsets, out, sc.hits);
        hipLaunchKernelGGL((k_sweep_long<T, true>), dim3(2048), dim3(SW_THREADS), 0, s, bp.s_yz, bp.s_info, bp.s_flags, hs, li, sc.n_long, sc.long_counts,
                           sc.long_off, offsets, out);
    } else {
        hipLaunchKernelGGL((k_sweep<T, false>), dim3(nb), dim3(SW_THREADS), 0, s, n, bp.s_yz, bp.s_bb, bp.s_bb2, bp.s_end, bp.s_info, bp.s_flags, hs, counts, offsets, out, sc.hits);
        hipLaunchKernelGGL((k_sweep_long<T, false>), dim3(2048), dim3(SW_THREADS), 0, s, bp.s_yz, bp.s_info, bp.s_flags, hs, li, sc.n_long, sc.long_counts,
                           sc.long_off, offsets, out);
        hipLaunchKernelGGL(k_long_finish, dim3(64), dim3(256), 0, s, li, sc.n_long, sc.long_counts, sc.long_off, counts);
    }
}
size_t sweep_long_item_bytes() { return sizeof(LongItem); }
size_t sweep_hit_words(uint32_t n) { return ((size_t)(n + 63u) / 64u + 8u) * SW_WAVES * SW_HIT_WORDS; }
uint32_t sweep_count_slots() { return SW_WAVES; }
uint32_t sweep_pad_records() { return 8u; }  // k_sweep reads whole candidate batches: s_yz needs this many records past n

#define INST(T)                                                                                          \
    template bool launch_update_aabb<T>(const DW<T>&, const BP<T>&, const StepParams<T>&, hipStream_t, uint32_t*, uint32_t);  \
    template void launch_interval_keys<T>(const DW<T>&, const BP<T>&, typename BP<T>::Key*, uint32_t*, uint32_t*, hipStream_t, bool); \
    template void launch_gather_sorted<T>(const DW<T>&, const BP<T>&, const uint32_t*, uint32_t, hipStream_t); \
    template void launch_sweep_ranges<T>(const BP<T>&, uint32_t, const SweepScratch&, hipStream_t, bool);     \
    template uint32_t launch_dynamic_bounds<T>(const DW<T>&, const BP<T>&, T*, hipStream_t);            \
    template void launch_sweep<T>(const BP<T>&, uint32_t, bool, const SweepScratch&, uint32_t*, const uint32_t*, avn_pair*, hipStream_t);
INST(float)
INST(double)
#undef INST
template void launch_radix_sort<uint32_t>(uint32_t*, uint32_t*, uint32_t*, uint32_t*, uint32_t, uint32_t*, uint32_t*, const uint32_t*, uint32_t**, uint32_t**, hipStream_t);
template void launch_radix_sort<uint64_t>(uint64_t*, uint32_t*, uint64_t*, uint32_t*, uint32_t, uint32_t*, uint32_t*, const uint32_t*, uint64_t**, uint32_t**, hipStream_t);

}  // namespace avn
