// k_narrow.hip — narrow-phase kernels (one lane per collider pair).  Device functions: avn_narrow.h.
//
//   k_contact_manifolds_query   batch form of contact_query::contact_manifolds
//                               (reference collision/collider/parry/contact_query.rs:156-261)
#include "avn_kernels.h"
#include "avn_narrow.h"

namespace avn {

template <class T> __device__ __forceinline__ V3<T> ld3(const T* p, size_t i) { return {p[3 * i], p[3 * i + 1], p[3 * i + 2]}; }
template <class T> __device__ __forceinline__ void st3(T* p, size_t i, V3<T> v) { if (p) { p[3 * i] = v.x; p[3 * i + 1] = v.y; p[3 * i + 2] = v.z; } }

template <class T>
__global__ __launch_bounds__(64) void k_contact_manifolds_query(QueryStage<T> s, uint32_t n) {
    uint32_t i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    V3<T> p1 = ld3(s.position1, i), p2 = ld3(s.position2, i);
    Q4<T> r1{s.rotation1[4 * i], s.rotation1[4 * i + 1], s.rotation1[4 * i + 2], s.rotation1[4 * i + 3]};
    Q4<T> r2{s.rotation2[4 * i], s.rotation2[4 * i + 1], s.rotation2[4 * i + 2], s.rotation2[4 * i + 3]};
    NpManifold<T> m;
    bool has = contact_manifolds_pair<T>(s.shape1[i], ld3(s.half_extents1, i), p1, r1, s.shape2[i], ld3(s.half_extents2, i), p2, r2, s.prediction[i], m);
    int cnt = has ? m.n : 0;
    if (s.point_count) s.point_count[i] = (uint8_t)cnt;
    st3(s.normal, i, has ? m.normal : vzero<T>());
    for (int k = 0; k < AVN_MAX_QUERY_POINTS; ++k) {
        size_t slot = (size_t)AVN_MAX_QUERY_POINTS * i + k;
        bool live = k < cnt;
        V3<T> a1 = live ? m.pts[k].anchor1 : vzero<T>();
        st3(s.anchor1, slot, a1);
        st3(s.anchor2, slot, live ? m.pts[k].anchor2 : vzero<T>());
        st3(s.point, slot, live ? p1 + a1 : vzero<T>());   // world_point = position1 + anchor1
        if (s.penetration) s.penetration[slot] = live ? m.pts[k].penetration : T(0);
        if (s.feature_id1) s.feature_id1[slot] = live ? m.pts[k].fid1 : 0u;
        if (s.feature_id2) s.feature_id2[slot] = live ? m.pts[k].fid2 : 0u;
    }
}
template <class T> void launch_contact_manifolds_query(const QueryStage<T>& s, uint32_t n, hipStream_t st) {
    if (n) hipLaunchKernelGGL(k_contact_manifolds_query<T>, dim3((n + 63) / 64), dim3(64), 0, st, s, n);
}
template void launch_contact_manifolds_query<float>(const QueryStage<float>&, uint32_t, hipStream_t);
template void launch_contact_manifolds_query<double>(const QueryStage<double>&, uint32_t, hipStream_t);

}  // namespace avn
