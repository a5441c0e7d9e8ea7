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
//   5. k_sweep<EMIT=false>     per-interval pair COUNT; block = 256 consecutive i, candidate j staged through
//                              LDS in tiles of 256 and broadcast-read (conflict-free), early-out per lane
//   6. exclusive scan of the counts
//   7. k_sweep<EMIT=true>      every lane re-walks its candidates and writes its pairs at its own offset
//                              => output is already in the reference's emission order, no post-sort.
// Integer/compare work only; HBM-bound on the sorted interval records, ALU-bound in the sweep for dense
// scenes (every lane tests O(k) candidates out of LDS).
#include "avn_kernels.h"

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
__global__ __launch_bounds__(256) void k_update_aabb(DW<T> w, BP<T> bp, StepParams<T> p) {
    uint32_t c = blockIdx.x * 256 + threadIdx.x;
    if (c >= bp.n_colliders) return;
    uint4 ci = bp.col_info[c];  // entity, body, shape | cflags << 8, -
    Vec4<T> he4 = bp.col_he[c];  // (half_extents.xyz, collision_margin)
    T spec = bp.col_spec[c];
    uint32_t shape = ci.z & 0xFFu, cflags = (ci.z >> 8) & 0xFFu;
    int body = (int)ci.y;
    V3<T> pos = xyz<T>(w.pos[body]);
    Q4<T> rot = quat<T>(w.rot[body]);
    V3<T> lv = xyz<T>(w.lvel[body]), av = xyz<T>(w.avel[body]);
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
}

// ---------------------------------------------------------------------------------------------------------
// stable LSD radix sort, 8 bits per pass, one wave per 1024-item tile
#define RS_TILE 1024
#define RS_ROUNDS 16

template <class K>
__global__ __launch_bounds__(64) void k_radix_hist(const K* __restrict__ keys, uint32_t n, uint32_t shift, uint32_t* __restrict__ hist, uint32_t nblocks) {
    __shared__ uint32_t cnt[256];
    uint32_t lane = threadIdx.x, b = blockIdx.x;
    for (uint32_t d = lane; d < 256; d += 64) cnt[d] = 0;
    __syncthreads();
    uint32_t base = b * RS_TILE;
    for (uint32_t r = 0; r < RS_ROUNDS; ++r) {
        uint32_t idx = base + r * 64 + lane;
        if (idx < n) atomicAdd(&cnt[(uint32_t)(keys[idx] >> shift) & 255u], 1u);
    }
    __syncthreads();
    for (uint32_t d = lane; d < 256; d += 64) hist[d * nblocks + b] = cnt[d];
}

template <class K>
__global__ __launch_bounds__(64) void k_radix_scatter(const K* __restrict__ keys_in, const uint32_t* __restrict__ vals_in, K* __restrict__ keys_out,
                                                      uint32_t* __restrict__ vals_out, uint32_t n, uint32_t shift,
                                                      const uint32_t* __restrict__ hist_scanned, uint32_t nblocks) {
    __shared__ uint32_t run[256];
    uint32_t lane = threadIdx.x, b = blockIdx.x;
    for (uint32_t d = lane; d < 256; d += 64) run[d] = hist_scanned[d * nblocks + b];
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
// exclusive scan (uint32), three kernels; tile = 2048
#define SC_TILE 2048
__global__ __launch_bounds__(256) void k_scan_sums(const uint32_t* __restrict__ in, uint32_t n, uint32_t* __restrict__ sums) {
    __shared__ uint32_t red[256];
    uint32_t base = blockIdx.x * SC_TILE, t = threadIdx.x, s = 0;
    for (uint32_t k = 0; k < SC_TILE / 256; ++k) { uint32_t i = base + t * (SC_TILE / 256) + k; if (i < n) s += in[i]; }
    red[t] = s;
    __syncthreads();
    for (uint32_t st = 128; st > 0; st >>= 1) { if (t < st) red[t] += red[t + st]; __syncthreads(); }
    if (t == 0) sums[blockIdx.x] = red[0];
}
__global__ __launch_bounds__(256) void k_scan_top(uint32_t* sums, uint32_t nb, uint32_t* total) {
    // single block: sequential over chunks of 256 with a running carry
    __shared__ uint32_t buf[256];
    __shared__ uint32_t carry;
    uint32_t t = threadIdx.x;
    if (t == 0) carry = 0;
    __syncthreads();
    for (uint32_t c0 = 0; c0 < nb; c0 += 256) {
        uint32_t i = c0 + t;
        uint32_t v = i < nb ? sums[i] : 0u;
        buf[t] = v;
        __syncthreads();
        for (uint32_t off = 1; off < 256; off <<= 1) {
            uint32_t add = t >= off ? buf[t - off] : 0u;
            __syncthreads();
            buf[t] += add;
            __syncthreads();
        }
        uint32_t incl = buf[t];
        if (i < nb) sums[i] = carry + incl - v;
        __syncthreads();
        if (t == 255) carry += incl;
        __syncthreads();
    }
    if (t == 0 && total) *total = carry;
}
__global__ __launch_bounds__(256) void k_scan_apply(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, uint32_t n, const uint32_t* __restrict__ sums) {
    __shared__ uint32_t buf[256];
    uint32_t base = blockIdx.x * SC_TILE, t = threadIdx.x;
    const uint32_t per = SC_TILE / 256;
    uint32_t v[per];
    uint32_t s = 0;
    for (uint32_t k = 0; k < per; ++k) { uint32_t i = base + t * per + k; v[k] = i < n ? in[i] : 0u; s += v[k]; }
    buf[t] = s;
    __syncthreads();
    for (uint32_t off = 1; off < 256; off <<= 1) {
        uint32_t add = t >= off ? buf[t - off] : 0u;
        __syncthreads();
        buf[t] += add;
        __syncthreads();
    }
    uint32_t excl = buf[t] - s + sums[blockIdx.x];
    for (uint32_t k = 0; k < per; ++k) { uint32_t i = base + t * per + k; if (i < n) out[i] = excl; excl += v[k]; }
}
void launch_exclusive_scan(const uint32_t* in, uint32_t* out, uint32_t n, uint32_t* block_sums, uint32_t* total, hipStream_t s) {
    if (n == 0) { if (total) (void)hipMemsetAsync(total, 0, sizeof(uint32_t), s); return; }
    uint32_t nb = (n + SC_TILE - 1) / SC_TILE;
    hipLaunchKernelGGL(k_scan_sums, dim3(nb), dim3(256), 0, s, in, n, block_sums);
    hipLaunchKernelGGL(k_scan_top, dim3(1), dim3(256), 0, s, block_sums, nb, total);
    hipLaunchKernelGGL(k_scan_apply, dim3(nb), dim3(256), 0, s, in, out, n, block_sums);
}
uint32_t scan_block_sums_needed(uint32_t n) { return (n + SC_TILE - 1) / SC_TILE + 1; }

// ---------------------------------------------------------------------------------------------------------
// gather the sorted interval records
template <class T>
__global__ __launch_bounds__(256) void k_gather_sorted(DW<T> w, BP<T> bp, const uint32_t* __restrict__ sorted_collider, uint32_t n) {
    uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    uint32_t c = sorted_collider[i];
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
    bp.s_min[i] = mn;
    bp.s_max[i] = mx;
    bp.s_info[i] = make_uint4(ci.x, ci.y, layers.x, layers.y);
    bp.s_flags[i] = f;
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
void launch_hs_insert(uint64_t* tab, uint32_t cap, const uint64_t* keys, uint32_t n, hipStream_t s) {
    if (n) hipLaunchKernelGGL(k_hs_insert, dim3((n + 255) / 256), dim3(256), 0, s, tab, cap - 1, keys, n);
}
void launch_hs_insert_pairs(uint64_t* tab, uint32_t cap, const avn_pair* pairs, uint32_t n, hipStream_t s) {
    if (n) hipLaunchKernelGGL(k_hs_insert_pairs, dim3((n + 255) / 256), dim3(256), 0, s, tab, cap - 1, pairs, n);
}

// ---------------------------------------------------------------------------------------------------------
// sweep
//
// Load balance: with one lane per interval, an interval whose AABB spans the scene (a ground plane) would make a single
// lane walk every later interval (100k serial iterations ~ 18 ms on cfg2).  Each lane therefore walks at most SW_CAP
// candidates; the remainder of such a "long" interval is recorded as a LongItem and swept by k_sweep_long, where a whole
// workgroup strides over the candidates of ONE interval (256 at a time, ballot + LDS scan for in-order positions).
// Emission order per interval stays j ascending: [short part][long part], i.e. exactly the reference's order.
#define SW_THREADS 256
#define SW_CAP 8192u

struct LongItem { uint32_t i, j_start, short_count, pad; };

template <class T> struct SweepSelf { V3<T> mn, mx; uint4 info; uint32_t flags; };

// all pair filters of broad_phase.rs:390-439 after the x test; returns true when (self, other) becomes a new pair
template <class T>
__device__ __forceinline__ bool pair_passes(const BP<T>& bp, const SweepSelf<T>& a, T miny, T minz, T maxy, T maxz, uint4 in2, uint32_t f2) {
    if (a.mn.y > maxy || a.mx.y < miny) return false;                                        // y disjoint
    if (a.mn.z > maxz || a.mx.z < minz) return false;                                        // z disjoint
    bool interacts = (a.info.z & in2.w) != 0 && (in2.z & a.info.w) != 0;                     // CollisionLayers::interacts_with
    if ((a.flags & f2 & AVN_AABB_IS_INACTIVE) || !interacts || a.info.y == in2.y) return false;
    uint64_t key = a.info.x < in2.x ? ((uint64_t)a.info.x << 32) | in2.x : ((uint64_t)in2.x << 32) | a.info.x;
    if (bp.pair_set_cap && hs_contains(bp.pair_set, bp.pair_set_cap - 1, key)) return false;
    if (bp.disabled_cap) {
        uint64_t bk = a.info.y < in2.y ? ((uint64_t)a.info.y << 32) | in2.y : ((uint64_t)in2.y << 32) | a.info.y;
        if (hs_contains(bp.disabled_set, bp.disabled_cap - 1, bk)) return false;
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

template <class T, bool EMIT>
__global__ __launch_bounds__(SW_THREADS) void k_sweep(BP<T> bp, uint32_t n, uint32_t* __restrict__ counts, const uint32_t* __restrict__ offsets,
                                                       avn_pair* __restrict__ out, LongItem* __restrict__ long_items, uint32_t* __restrict__ n_long) {
    __shared__ Vec4<T> l_a[SW_THREADS];   // (min.x, min.y, min.z, max.y)
    __shared__ T l_maxz[SW_THREADS];
    __shared__ uint4 l_info[SW_THREADS];
    __shared__ uint32_t l_flags[SW_THREADS];
    uint32_t t = threadIdx.x;
    uint32_t tile = xcd_block(blockIdx.x, gridDim.x);
    uint32_t i0 = tile * SW_THREADS;
    if (i0 >= n) return;  // uniform per block
    uint32_t i = i0 + t;
    bool valid = i < n;
    SweepSelf<T> self;
    self.mn = vzero<T>(); self.mx = vzero<T>(); self.info = make_uint4(0, 0, 0, 0); self.flags = 0;
    if (valid) { self.mn = xyz<T>(bp.s_min[i]); self.mx = xyz<T>(bp.s_max[i]); self.info = bp.s_info[i]; self.flags = bp.s_flags[i]; }
    bool done = !valid || (self.flags & AVN_IV_DROPPED);
    uint32_t count = 0;
    uint32_t pos = (EMIT && valid) ? offsets[i] : 0u;
    uint32_t j_cap = i + 1u + SW_CAP;  // first candidate NOT handled by this lane
    for (uint32_t j0 = i0; j0 < n; j0 += SW_THREADS) {
        uint32_t jl = j0 + t;
        if (jl < n) {
            Vec4<T> a = bp.s_min[jl], b = bp.s_max[jl];
            l_a[t] = make4<T>(a.x, a.y, a.z, b.y);
            l_maxz[t] = b.z;
            l_info[t] = bp.s_info[jl];
            l_flags[t] = bp.s_flags[jl];
        }
        __syncthreads();
        if (!done) {
            uint32_t lim = min((uint32_t)SW_THREADS, n - j0);
            uint32_t jj = (j0 == i0) ? t + 1 : 0u;  // j > i
            for (; jj < lim; ++jj) {
                if (j0 + jj >= j_cap) {  // hand the rest of a long interval to k_sweep_long
                    if (!EMIT) { uint32_t k = atomicAdd(n_long, 1u); long_items[k] = LongItem{i, j_cap, count, 0u}; }
                    done = true;
                    break;
                }
                Vec4<T> a = l_a[jj];
                if (a.x > self.mx.x) { done = true; break; }  // x: sweep ends (broad_phase.rs:390-392)
                if (self.mn.y > a.w || self.mx.y < a.y) continue;  // cheap y reject before touching the other LDS arrays
                if (!pair_passes<T>(bp, self, a.y, a.z, a.w, l_maxz[jj], l_info[jj], l_flags[jj])) continue;
                if (EMIT) out[pos] = make_pair(self.info, self.flags, l_info[jj], l_flags[jj]);
                ++pos;
                ++count;
            }
        }
        if (__syncthreads_and(done ? 1 : 0)) break;
    }
    if (!EMIT && valid) counts[i] = count;
}

// One workgroup per long interval (grid-stride over the LongItem list).
template <class T, bool EMIT>
__global__ __launch_bounds__(SW_THREADS) void k_sweep_long(BP<T> bp, uint32_t n, const LongItem* __restrict__ items, const uint32_t* __restrict__ n_long,
                                                            uint32_t* __restrict__ counts, const uint32_t* __restrict__ offsets, avn_pair* __restrict__ out) {
    __shared__ uint32_t wave_tot[SW_THREADS / 64];
    uint32_t t = threadIdx.x, lane = t & 63u, wv = t >> 6;
    uint32_t nl = *n_long;
    for (uint32_t it = blockIdx.x; it < nl; it += gridDim.x) {
        LongItem item = items[it];
        uint32_t i = item.i;
        SweepSelf<T> self;
        self.mn = xyz<T>(bp.s_min[i]); self.mx = xyz<T>(bp.s_max[i]); self.info = bp.s_info[i]; self.flags = bp.s_flags[i];
        uint32_t running = 0;
        uint32_t base = EMIT ? offsets[i] + item.short_count : 0u;
        for (uint32_t j0 = item.j_start; j0 < n; j0 += SW_THREADS) {
            uint32_t j = j0 + t;
            bool beyond = true, ok = false;
            uint4 in2 = make_uint4(0, 0, 0, 0);
            uint32_t f2 = 0;
            if (j < n) {
                Vec4<T> a = bp.s_min[j];
                beyond = a.x > self.mx.x;
                if (!beyond) {
                    Vec4<T> b = bp.s_max[j];
                    in2 = bp.s_info[j]; f2 = bp.s_flags[j];
                    ok = pair_passes<T>(bp, self, a.y, a.z, b.y, b.z, in2, f2);
                }
            }
            unsigned long long bal = __ballot(ok);
            if (lane == 0) wave_tot[wv] = (uint32_t)__popcll(bal);
            __syncthreads();
            uint32_t before = 0, total = 0;
#pragma unroll
            for (uint32_t k = 0; k < SW_THREADS / 64; ++k) { uint32_t v = wave_tot[k]; if (k < wv) before += v; total += v; }
            if (EMIT && ok) out[base + running + before + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull))] = make_pair(self.info, self.flags, in2, f2);
            running += total;
            // sorted by min.x: once any candidate of this stride is beyond max.x, every later one is too
            if (__syncthreads_or(beyond ? 1 : 0)) break;
        }
        if (!EMIT && t == 0) counts[i] += running;  // single writer per interval, after k_sweep<false> completed
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------------
// launchers
template <class T> void launch_update_aabb(const DW<T>& w, const BP<T>& bp, const StepParams<T>& p, hipStream_t s) {
    if (bp.n_colliders) hipLaunchKernelGGL(k_update_aabb<T>, dim3((bp.n_colliders + 255) / 256), dim3(256), 0, s, w, bp, p);
}
template <class T> void launch_interval_keys(const DW<T>& w, const BP<T>& bp, typename BP<T>::Key* keys, uint32_t* vals, uint32_t* n_dropped, hipStream_t s) {
    (void)hipMemsetAsync(n_dropped, 0, sizeof(uint32_t), s);
    if (bp.n_intervals) hipLaunchKernelGGL(k_interval_keys<T>, dim3((bp.n_intervals + 255) / 256), dim3(256), 0, s, w, bp, keys, vals, n_dropped);
}
uint32_t radix_blocks(uint32_t n) { return (n + RS_TILE - 1) / RS_TILE; }
template <class K> void launch_radix_sort(K* keys_a, uint32_t* vals_a, K* keys_b, uint32_t* vals_b, uint32_t n, uint32_t* hist, uint32_t* block_sums, hipStream_t s) {
    // sizeof(K) passes of 8 bits: the result ends in (keys_a, vals_a) because the pass count is even
    if (n == 0) return;
    uint32_t nb = radix_blocks(n);
    K* ki = keys_a; uint32_t* vi = vals_a; K* ko = keys_b; uint32_t* vo = vals_b;
    for (uint32_t pass = 0; pass < sizeof(K); ++pass) {
        uint32_t shift = pass * 8;
        hipLaunchKernelGGL(k_radix_hist<K>, dim3(nb), dim3(64), 0, s, ki, n, shift, hist, nb);
        launch_exclusive_scan(hist, hist, 256 * nb, block_sums, nullptr, s);
        hipLaunchKernelGGL(k_radix_scatter<K>, dim3(nb), dim3(64), 0, s, ki, vi, ko, vo, n, shift, hist, nb);
        K* tk = ki; ki = ko; ko = tk;
        uint32_t* tv = vi; vi = vo; vo = tv;
    }
}
template <class T> void launch_gather_sorted(const DW<T>& w, const BP<T>& bp, const uint32_t* sorted_collider, uint32_t n, hipStream_t s) {
    if (n) hipLaunchKernelGGL(k_gather_sorted<T>, dim3((n + 255) / 256), dim3(256), 0, s, w, bp, sorted_collider, n);
}
template <class T> void launch_sweep(const BP<T>& bp, uint32_t n, bool emit, uint32_t* counts, const uint32_t* offsets, avn_pair* out, void* long_items,
                                     uint32_t* n_long, hipStream_t s) {
    if (!n) return;
    uint32_t nb = (n + SW_THREADS - 1) / SW_THREADS;
    nb = ((nb + 7) / 8) * 8;
    LongItem* li = (LongItem*)long_items;
    if (emit) {
        hipLaunchKernelGGL((k_sweep<T, true>), dim3(nb), dim3(SW_THREADS), 0, s, bp, n, counts, offsets, out, li, n_long);
        hipLaunchKernelGGL((k_sweep_long<T, true>), dim3(1024), dim3(SW_THREADS), 0, s, bp, n, li, n_long, counts, offsets, out);
    } else {
        (void)hipMemsetAsync(n_long, 0, sizeof(uint32_t), s);
        hipLaunchKernelGGL((k_sweep<T, false>), dim3(nb), dim3(SW_THREADS), 0, s, bp, n, counts, offsets, out, li, n_long);
        hipLaunchKernelGGL((k_sweep_long<T, false>), dim3(1024), dim3(SW_THREADS), 0, s, bp, n, li, n_long, counts, offsets, out);
    }
}
size_t sweep_long_item_bytes() { return sizeof(LongItem); }

#define INST(T)                                                                                          \
    template void launch_update_aabb<T>(const DW<T>&, const BP<T>&, const StepParams<T>&, hipStream_t);  \
    template void launch_interval_keys<T>(const DW<T>&, const BP<T>&, typename BP<T>::Key*, uint32_t*, uint32_t*, hipStream_t); \
    template void launch_gather_sorted<T>(const DW<T>&, const BP<T>&, const uint32_t*, uint32_t, hipStream_t); \
    template void launch_sweep<T>(const BP<T>&, uint32_t, bool, uint32_t*, const uint32_t*, avn_pair*, void*, uint32_t*, hipStream_t);
INST(float)
INST(double)
#undef INST
template void launch_radix_sort<uint32_t>(uint32_t*, uint32_t*, uint32_t*, uint32_t*, uint32_t, uint32_t*, uint32_t*, hipStream_t);
template void launch_radix_sort<uint64_t>(uint64_t*, uint32_t*, uint64_t*, uint32_t*, uint32_t, uint32_t*, uint32_t*, hipStream_t);

}  // namespace avn
