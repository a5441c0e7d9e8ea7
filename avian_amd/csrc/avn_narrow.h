// avn_narrow.h — device functions of the narrow phase for Ball / Cuboid collider pairs.
//
// What the reference computes here (paths relative to /root/reference/src):
//   contact_query::contact_manifolds        collision/collider/parry/contact_query.rs:156-261   (Avian)
//   make_isometry (3D)                      math/mod.rs:608-620                                  (Avian)
//   parry3d 0.25 DefaultQueryDispatcher::contact_manifolds for (Ball|Cuboid) x (Ball|Cuboid)     (third party, un-vendored:
//     contact_manifold_ball_ball, contact_manifold_convex_ball, contact_manifold_cuboid_cuboid = SAT over the 3 + 3 face
//     normals and the 9 edge cross products, Cuboid::support_face, PolygonalFeature::contacts face/face clipping),
//     on nalgebra Isometry3 / UnitQuaternion arithmetic.
// The third-party part is written from the published algorithm (the crates are not readable in this build environment):
// results are checked bit for bit against the CPU oracle's independent restatement and through geometric invariants, not
// against the crates themselves (DESIGN.md "narrow phase": parity unpinned).
//
// One lane generates one pair's manifold.  Bound: per-pair latency (a few hundred dependent flops and at most two
// square roots per SAT axis); the kernel that calls this streams 2 x 80 B of collider / pose data per pair.
#pragma once
#include "avn_device.h"

namespace avn {

#define AVN_NP_MAX_RAW 16  // 4 + 4 vertex contacts + up to 8 edge/edge crossings of two quads

__device__ __forceinline__ float copysign_t(float a, float b) { return __builtin_copysignf(a, b); }
__device__ __forceinline__ double copysign_t(double a, double b) { return __builtin_copysign(a, b); }

// deterministic atan for x >= 0 (fdlibm reduction + polynomial in T; the "libm" of this build, shared with sin_cos_t)
template <class T> __device__ __forceinline__ T atan_pos_t(T x) {
    int id;
    if (x < T(0.4375)) id = -1;
    else if (x < T(0.6875)) { id = 0; x = (T(2) * x - T(1)) / (T(2) + x); }
    else if (x < T(1.1875)) { id = 1; x = (x - T(1)) / (x + T(1)); }
    else if (x < T(2.4375)) { id = 2; x = (x - T(1.5)) / (T(1) + T(1.5) * x); }
    else { id = 3; x = T(-1) / x; }
    T z = x * x, w = z * z;
    T s1 = z * (T(3.33333333333329318027e-01) + w * (T(1.42857142725034663711e-01) + w * (T(9.09088713343650656196e-02) +
           w * (T(6.66107313738753120669e-02) + w * (T(4.97687799461593236017e-02) + w * T(1.62858201153657823623e-02))))));
    T s2 = w * (T(-1.99999999998764832476e-01) + w * (T(-1.11111104054623557880e-01) + w * (T(-7.69187620504482999495e-02) +
           w * (T(-5.83357013379057348645e-02) + w * T(-3.65315727442169155270e-02)))));
    if (id < 0) return x - x * (s1 + s2);
    T hi = id == 0 ? T(4.63647609000806093515e-01) : id == 1 ? T(7.85398163397448278999e-01) : id == 2 ? T(9.82793723247329054082e-01) : T(1.57079632679489655800e+00);
    return hi - (x * (s1 + s2) - x);
}
template <class T> __device__ __forceinline__ T atan2_ypos_t(T y, T x) {  // y > 0
    if (x == T(0)) return T(1.57079632679489661923);
    T a = atan_pos_t<T>(y / fabs_t(x));
    return x > T(0) ? a : T(3.14159265358979323846) - a;
}

// ---- nalgebra arithmetic ---------------------------------------------------------------------------------------------
template <class T> __device__ __forceinline__ V3<T> na_cross(V3<T> a, V3<T> b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
template <class T> __device__ __forceinline__ T na_dot(V3<T> a, V3<T> b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
template <class T> __device__ __forceinline__ T na_norm(V3<T> a) { return sqrt_t(na_dot(a, a)); }
template <class T> __device__ __forceinline__ V3<T> na_qrot(Q4<T> q, V3<T> v) {
    V3<T> qv{q.x, q.y, q.z};
    V3<T> t = na_cross(qv, v) * T(2);
    V3<T> c = na_cross(qv, t);
    return (t * q.w + c) + v;
}
template <class T> __device__ __forceinline__ Q4<T> na_qmul(Q4<T> a, Q4<T> b) {
    return {a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y, a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x,
            a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w, a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z};
}
template <class T> struct Iso { Q4<T> r; V3<T> t; };
template <class T> __device__ __forceinline__ Iso<T> iso_inverse(const Iso<T>& a) { Q4<T> ri = qinverse(a.r); return {ri, na_qrot(ri, -a.t)}; }
template <class T> __device__ __forceinline__ Iso<T> iso_inv_mul(const Iso<T>& a, const Iso<T>& b) { Q4<T> ri = qinverse(a.r); return {na_qmul(ri, b.r), na_qrot(ri, b.t - a.t)}; }
template <class T> __device__ __forceinline__ V3<T> iso_point(const Iso<T>& a, V3<T> p) { return na_qrot(a.r, p) + a.t; }
template <class T> __device__ __forceinline__ V3<T> iso_vec(const Iso<T>& a, V3<T> v) { return na_qrot(a.r, v); }
template <class T> __device__ __forceinline__ V3<T> iso_inv_vec(const Iso<T>& a, V3<T> v) { return na_qrot(qinverse(a.r), v); }
template <class T> __device__ __forceinline__ V3<T> iso_inv_point(const Iso<T>& a, V3<T> p) { return na_qrot(qinverse(a.r), p - a.t); }

// make_isometry: glam Quat::to_scaled_axis -> nalgebra UnitQuaternion::from_scaled_axis
template <class T> __device__ __forceinline__ Iso<T> make_isometry(V3<T> position, Q4<T> rotation) {
    V3<T> v{rotation.x, rotation.y, rotation.z};
    T len = length(v);
    V3<T> scaled = vzero<T>();
    if (len >= T(1.0e-8)) {
        T angle = T(2) * atan2_ypos_t<T>(len, rotation.w);
        scaled = (v / len) * angle;
    }
    T angle = na_norm(scaled);
    Q4<T> q{T(0), T(0), T(0), T(1)};
    if (angle != T(0)) {
        V3<T> axis = scaled / angle;
        T s, c;
        sin_cos_t(angle * T(0.5), s, c);
        q = {axis.x * s, axis.y * s, axis.z * s, c};
    }
    return {q, position};
}

// PackedFeatureId
__device__ __forceinline__ uint32_t fid_vertex(uint32_t c) { return (1u << 30) | c; }
__device__ __forceinline__ uint32_t fid_edge(uint32_t c) { return (2u << 30) | c; }
__device__ __forceinline__ uint32_t fid_face(uint32_t c) { return (3u << 30) | c; }

// The manifold as contact_query::contact_manifolds returns it (anchors relative to the collider origins); points are
// converted as they are generated, so no local-frame copy of the parry manifold is kept.
template <class T> struct NpPoint { V3<T> anchor1, anchor2; T penetration; uint32_t fid1, fid2; };
template <class T> struct NpManifold { V3<T> normal; int n; NpPoint<T> pts[AVN_NP_MAX_RAW]; };

// Where the raw points of a manifold go.  NpArraySink: the NpManifold of the batch query (per-lane array).  k_narrow.hip has an LDS sink.
template <class T> struct NpArraySink {
    NpManifold<T>* out;
    V3<T> d12;  // position1 - position2
    __device__ __forceinline__ int n() const { return out->n; }
    __device__ __forceinline__ void put(V3<T> anchor1, T penetration, uint32_t f1, uint32_t f2) { out->pts[out->n++] = NpPoint<T>{anchor1, anchor1 + d12, penetration, f1, f2}; }
};
template <class T, class Sink> struct NpEmit {  // conversion context of contact_query.rs:233-252
    Q4<T> rotation1;
    V3<T> normal;  // world normal
    Sink* sink;
    __device__ __forceinline__ void push(V3<T> local_p1, T dist, uint32_t f1, uint32_t f2) {
        if (sink->n() >= AVN_NP_MAX_RAW) return;
        V3<T> point1 = qrot(rotation1, local_p1);
        V3<T> anchor1 = point1 + (normal * dist) * T(0.5);
        sink->put(anchor1, -dist, f1, f2);   // (anchor2 = anchor1 + (position1 - position2): the sink's business)
    }
};
// local_n1 -> world normal (normalise, rotate, is_normalized check); false = the manifold is dropped
template <class T> __device__ __forceinline__ bool np_world_normal(Q4<T> rotation1, V3<T> local_n1, V3<T>& normal) {
    V3<T> local_normal = local_n1 / na_norm(local_n1);
    normal = qrot(rotation1, local_normal);
    return fabs_t(length_squared(normal) - T(1)) <= T(2e-4);
}

template <class T> __device__ __forceinline__ V3<T> cuboid_support_point(V3<T> he, V3<T> dir) {
    return {copysign_t(he.x, dir.x), copysign_t(he.y, dir.y), copysign_t(he.z, dir.z)};
}
template <class T> __device__ __forceinline__ T vget(V3<T> v, int i) { return i == 0 ? v.x : (i == 1 ? v.y : v.z); }

template <class T> __device__ __forceinline__ T sat_normal_oneway(V3<T> he1, V3<T> he2, const Iso<T>& pos12, V3<T>& best_dir) {
    T best = -Limits<T>::max;
    best_dir = vzero<T>();
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        T sign = copysign_t(T(1), vget(pos12.t, i));
        V3<T> axis1{i == 0 ? sign : T(0), i == 1 ? sign : T(0), i == 2 ? sign : T(0)};
        V3<T> axis2 = iso_inv_vec(pos12, -axis1);
        V3<T> pt2 = iso_point(pos12, cuboid_support_point(he2, axis2));
        T separation = vget(pt2, i) * sign - vget(he1, i);
        if (separation > best) { best = separation; best_dir = axis1; }
    }
    return best;
}
template <class T> __device__ __forceinline__ T sat_line_separation(V3<T> he1, V3<T> he2, const Iso<T>& pos12, V3<T> axis1, V3<T>& out_axis) {
    V3<T> axis1_2 = iso_inv_vec(pos12, axis1);
    V3<T> pa = cuboid_support_point(he1, axis1);
    V3<T> pb = iso_point(pos12, cuboid_support_point(he2, -axis1_2));
    T separation1 = na_dot(pb - pa, axis1);
    V3<T> pc = cuboid_support_point(he1, -axis1);
    V3<T> pd = iso_point(pos12, cuboid_support_point(he2, axis1_2));
    T separation2 = na_dot(pd - pc, -axis1);
    if (separation1 > separation2) { out_axis = axis1; return separation1; }
    out_axis = -axis1;
    return separation2;
}
template <class T> __device__ __forceinline__ T sat_edge_twoway(V3<T> he1, V3<T> he2, const Iso<T>& pos12, V3<T>& best_dir) {
    V3<T> e2[3] = {iso_vec(pos12, V3<T>{T(1), T(0), T(0)}), iso_vec(pos12, V3<T>{T(0), T(1), T(0)}), iso_vec(pos12, V3<T>{T(0), T(0), T(1)})};
    T best = -Limits<T>::max;
    best_dir = vzero<T>();
    for (int b = 0; b < 3; ++b)
        for (int a = 0; a < 3; ++a) {
            V3<T> u = e2[b];
            V3<T> axis = a == 0 ? V3<T>{T(0), -u.z, u.y} : (a == 1 ? V3<T>{u.z, T(0), -u.x} : V3<T>{-u.y, u.x, T(0)});
            T norm1 = na_norm(axis);
            if (norm1 > Limits<T>::eps) {
                V3<T> ax;
                T sep = sat_line_separation(he1, he2, pos12, axis / norm1, ax);
                if (sep > best) { best = sep; best_dir = ax; }
            }
        }
    return best;
}
template <class T> struct Face { V3<T> v[4]; uint32_t vid[4], eid[4], fid; };
template <class T> __device__ __forceinline__ void cuboid_support_face(V3<T> he, V3<T> dir, Face<T>& f) {
    T ax = fabs_t(dir.x), ay = fabs_t(dir.y), az = fabs_t(dir.z);
    int iamax = 0;
    T mx = ax;
    if (ay > mx) { mx = ay; iamax = 1; }
    if (az > mx) { mx = az; iamax = 2; }
    T sign = copysign_t(T(1), vget(dir, iamax));
    if (iamax == 0) { f.v[0] = {he.x * sign, he.y, he.z}; f.v[1] = {he.x * sign, -he.y, he.z}; f.v[2] = {he.x * sign, -he.y, -he.z}; f.v[3] = {he.x * sign, he.y, -he.z}; }
    else if (iamax == 1) { f.v[0] = {he.x, he.y * sign, he.z}; f.v[1] = {-he.x, he.y * sign, he.z}; f.v[2] = {-he.x, he.y * sign, -he.z}; f.v[3] = {he.x, he.y * sign, -he.z}; }
    else { f.v[0] = {he.x, he.y, he.z * sign}; f.v[1] = {he.x, -he.y, he.z * sign}; f.v[2] = {-he.x, -he.y, he.z * sign}; f.v[3] = {-he.x, he.y, he.z * sign}; }
#pragma unroll
    for (int k = 0; k < 4; ++k) f.vid[k] = fid_vertex((f.v[k].x < T(0) ? 1u : 0u) | (f.v[k].y < T(0) ? 2u : 0u) | (f.v[k].z < T(0) ? 4u : 0u));
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        uint32_t a = f.vid[k] & 7u, b = f.vid[(k + 1) & 3] & 7u;
        uint32_t hi = a > b ? a : b, lo = a > b ? b : a;
        f.eid[k] = fid_edge((hi << 3) | lo | 0xC0u);
    }
    f.fid = fid_face((uint32_t)iamax + (sign > T(0) ? 3u : 0u) + 10u);
}
template <class T> __device__ __forceinline__ void na_orthonormal_basis(V3<T> n, V3<T>& b0, V3<T>& b1) {
    T sign = copysign_t(T(1), n.z);
    T a = T(-1) / (sign + n.z);
    T b = n.x * n.y * a;
    b0 = {T(1) + sign * n.x * n.x * a, sign * b, -sign * n.x};
    b1 = {b, sign + n.y * n.y * a, -n.y};
}
template <class T> __device__ __forceinline__ T perp2(V2<T> a, V2<T> b) { return a.x * b.y - a.y * b.x; }
__device__ __forceinline__ bool ulps_apart_le4(float a, float b) {
    if (signbit_t(a) != signbit_t(b)) return false;
    long long d = (long long)(int)__float_as_uint(a) - (long long)(int)__float_as_uint(b);
    return (d < 0 ? -d : d) <= 4;
}
__device__ __forceinline__ bool ulps_apart_le4(double a, double b) {
    if (signbit_t(a) != signbit_t(b)) return false;
    long long d = __double_as_longlong(a) - __double_as_longlong(b);
    return (d < 0 ? -d : d) <= 4;
}
template <class T> __device__ __forceinline__ bool closest_points_line2d(V2<T> e1a, V2<T> e1b, V2<T> e2a, V2<T> e2b, T& s_out, T& t_out) {
    V2<T> dir1{e1b.x - e1a.x, e1b.y - e1a.y}, dir2{e2b.x - e2a.x, e2b.y - e2a.y}, r{e1a.x - e2a.x, e1a.y - e2a.y};
    T a = dir1.x * dir1.x + dir1.y * dir1.y, e = dir2.x * dir2.x + dir2.y * dir2.y, f = dir2.x * r.x + dir2.y * r.y;
    const T eps = Limits<T>::eps;
    if (a <= eps && e <= eps) { s_out = T(0); t_out = T(0); return true; }
    if (a <= eps) { s_out = T(0); t_out = f / e; return true; }
    T c = dir1.x * r.x + dir1.y * r.y;
    if (e <= eps) { s_out = -c / a; t_out = T(0); return true; }
    T b = dir1.x * dir2.x + dir1.y * dir2.y;
    T ae = a * e, bb = b * b, denom = ae - bb;
    if (denom <= eps || fabs_t(ae - bb) <= eps || ulps_apart_le4(ae, bb)) return false;
    T s = (b * f - c * e) / denom;
    s_out = s;
    t_out = (b * s + f) / e;
    return true;
}
template <class T> __device__ __forceinline__ bool np_inside(const V2<T>* poly, V2<T> p) {
    T sign = perp2(V2<T>{poly[0].x - poly[3].x, poly[0].y - poly[3].y}, V2<T>{p.x - poly[3].x, p.y - poly[3].y});
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        T ns = perp2(V2<T>{poly[j + 1].x - poly[j].x, poly[j + 1].y - poly[j].y}, V2<T>{p.x - poly[j].x, p.y - poly[j].y});
        if (sign == T(0)) sign = ns;
        else if (sign * ns < T(0)) return false;
    }
    return true;
}
// (every loop has a constant trip count and is unrolled: the faces and their projections are indexed by compile-time constants only and
//  stay in registers -- as dynamically indexed arrays they lived in scratch memory, which made this function the narrow phase's cost)
template <class T, class Em> __device__ void face_face_contacts(const Iso<T>& pos12, const Face<T>& face1, V3<T> sep_axis1, const Face<T>& face2, Em& em) {
    V3<T> b0, b1;
    na_orthonormal_basis(sep_axis1, b0, b1);
    V2<T> p1[4], p2[4];
    V3<T> v2_1[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        p1[k] = {na_dot(face1.v[k], b0), na_dot(face1.v[k], b1)};
        v2_1[k] = iso_point(pos12, face2.v[k]);
        p2[k] = {na_dot(v2_1[k], b0), na_dot(v2_1[k], b1)};
    }
    {
        V3<T> normal2_1 = na_cross(v2_1[2] - v2_1[1], v2_1[0] - v2_1[1]);
        T denom = na_dot(normal2_1, sep_axis1);
        if (!(fabs_t(denom) <= Limits<T>::eps)) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (np_inside(p2, p1[i])) {
                    T dist = na_dot(v2_1[0] - face1.v[i], normal2_1) / denom;
                    em.push(face1.v[i], dist, face1.vid[i], face2.fid);
                }
        }
    }
    {
        V3<T> normal1 = na_cross(face1.v[2] - face1.v[1], face1.v[0] - face1.v[1]);
        T denom = -na_dot(normal1, sep_axis1);
        if (!(fabs_t(denom) <= Limits<T>::eps)) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (np_inside(p1, p2[i])) {
                    T dist = na_dot(face1.v[0] - v2_1[i], normal1) / denom;
                    em.push(v2_1[i] - sep_axis1 * dist, dist, face1.fid, face2.vid[i]);
                }
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            T s, t;
            if (!closest_points_line2d(p1[i], p1[(i + 1) & 3], p2[j], p2[(j + 1) & 3], s, t)) continue;
            if (s > T(0) && s < T(1) && t > T(0) && t < T(1)) {
                V3<T> local_p1 = face1.v[i] * (T(1) - s) + face1.v[(i + 1) & 3] * s;
                V3<T> local_p2_1 = v2_1[j] * (T(1) - t) + v2_1[(j + 1) & 3] * t;
                em.push(local_p1, na_dot(local_p2_1 - local_p1, sep_axis1), face1.eid[i], face2.eid[j]);
            }
        }
}

// contact_query::contact_manifolds for one pair (he: cuboid half extents, ball radius in x); false = no manifold
// `defer` (optional): the cuboid-cuboid path has a cheap half -- the three SAT sweeps, after which most AABB-overlapping pairs are known to be
// apart -- and a heavy half (support faces, clipping in the plane, conversion of up to 16 raw points).  With defer != nullptr and
// *defer == false on entry the function stops after the SAT of a pair that survives it, leaves the separating direction in *axis and sets
// *defer = true (nothing of `out` is valid); called again with *defer == true it skips the SAT and continues from *axis.  A caller that
// gathers the survivors of a workgroup into dense waves (k_narrow_phase) runs the heavy half at full lane occupancy; the result is the
// one-call result bit for bit.
template <class T, class Sink, int MODE = 0>   // MODE 1: `defer` given and false on entry (stop after the SAT), 2: given and true (continue from *axis), 0: decided at run time
__device__ bool contact_manifolds_pair_sink(uint32_t shape1, V3<T> he1, V3<T> position1, Q4<T> rotation1, uint32_t shape2, V3<T> he2, V3<T> position2, Q4<T> rotation2,
                                            T prediction, Sink& sink, V3<T>& normal_out, bool* defer = nullptr, V3<T>* axis = nullptr) {
    Iso<T> isometry1 = make_isometry(position1, rotation1), isometry2 = make_isometry(position2, rotation2);
    Iso<T> pos12 = iso_inv_mul(isometry1, isometry2);
    NpEmit<T, Sink> em;
    em.rotation1 = rotation1; em.sink = &sink;
    const bool ball1 = shape1 == AVN_SHAPE_BALL, ball2 = shape2 == AVN_SHAPE_BALL;
    if (ball1 && ball2) {  // contact_manifold_ball_ball
        T r1 = he1.x, r2 = he2.x;
        V3<T> dcenter = pos12.t;
        T center_dist = na_norm(dcenter);
        T dist = center_dist - r1 - r2;
        if (!(dist < prediction)) return false;
        V3<T> local_n1 = center_dist != T(0) ? dcenter / center_dist : V3<T>{T(0), T(1), T(0)};
        if (!np_world_normal(rotation1, local_n1, em.normal)) return false;
        em.push(local_n1 * r1, dist, fid_face(0), fid_face(0));
    } else if (ball1 || ball2) {  // contact_manifold_convex_ball (flipped when the ball is collider 1)
        const bool flipped = ball1;
        Iso<T> p = flipped ? iso_inverse(pos12) : pos12;  // the ball in the cuboid's frame
        V3<T> he = flipped ? he2 : he1;
        T r = flipped ? he1.x : he2.x;
        V3<T> c = p.t;
        V3<T> shift{smax(-he.x - c.x, T(0)) - smax(c.x - he.x, T(0)), smax(-he.y - c.y, T(0)) - smax(c.y - he.y, T(0)), smax(-he.z - c.z, T(0)) - smax(c.z - he.z, T(0))};
        bool inside = shift.x == T(0) && shift.y == T(0) && shift.z == T(0);
        V3<T> proj = inside ? c : c + shift;
        V3<T> dpos = c - proj;
        T dist = na_norm(dpos);
        if (!(dist > T(0))) return false;
        V3<T> n_cub = dpos / dist;  // normal in the cuboid's frame
        if (inside) { n_cub = -n_cub; dist = -dist; }
        if (!(dist <= r + prediction)) return false;
        V3<T> n_ball = iso_inv_vec(p, -n_cub);  // the same direction, negated, in the ball's frame
        V3<T> p_ball = n_ball * r;
        if (flipped) {   // collider 1 is the ball: local_n1 = n_ball, local_p1 = the point on the ball
            if (!np_world_normal(rotation1, n_ball, em.normal)) return false;
            em.push(p_ball, dist - r, fid_face(0), 0u);
        } else {
            if (!np_world_normal(rotation1, n_cub, em.normal)) return false;
            em.push(proj, dist - r, 0u, fid_face(0));
        }
    } else {  // contact_manifold_cuboid_cuboid
        Iso<T> pos21 = iso_inverse(pos12);
        V3<T> best;
        if (MODE == 2 || (MODE == 0 && defer && *defer)) best = *axis;
        else {
            V3<T> d1, d2, d3;
            T sep1 = sat_normal_oneway(he1, he2, pos12, d1);
            if (sep1 > prediction) return false;
            T sep2 = sat_normal_oneway(he2, he1, pos21, d2);
            if (sep2 > prediction) return false;
            T sep3 = sat_edge_twoway(he1, he2, pos12, d3);
            if (sep3 > prediction) return false;
            best = d1;
            if (sep2 > sep1 && sep2 > sep3) best = iso_vec(pos12, -d2);
            else if (sep3 > sep1) best = d3;
            if (MODE == 1 || (MODE == 0 && defer)) { *defer = true; *axis = best; return false; }
        }
        V3<T> local_n2 = iso_vec(pos21, -best);
        if (!np_world_normal(rotation1, best, em.normal)) return false;
        Face<T> f1, f2;
        cuboid_support_face(he1, best, f1);
        cuboid_support_face(he2, local_n2, f2);
        face_face_contacts(pos12, f1, best, f2, em);
    }
    normal_out = em.normal;
    return sink.n() > 0;
}
template <class T>
__device__ bool contact_manifolds_pair(uint32_t shape1, V3<T> he1, V3<T> position1, Q4<T> rotation1, uint32_t shape2, V3<T> he2, V3<T> position2, Q4<T> rotation2,
                                       T prediction, NpManifold<T>& out) {
    out.n = 0;
    NpArraySink<T> sink{&out, position1 - position2};
    return contact_manifolds_pair_sink<T, NpArraySink<T>>(shape1, he1, position1, rotation1, shape2, he2, position2, rotation2, prediction, sink, out.normal);
}

}  // namespace avn
