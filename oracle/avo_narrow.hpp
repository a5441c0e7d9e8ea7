// ORACLE — TEST INFRASTRUCTURE ONLY.  Never linked, imported or called by the product path.
//
// avo_narrow.hpp: CPU restatement of the contact-manifold generation the reference's narrow phase performs for
// Ball / Cuboid collider pairs:
//
//   * Avian's own code, restated line by line:
//       contact_query::contact_manifolds            src/collision/collider/parry/contact_query.rs:156-261
//       make_isometry (3D)                          src/math/mod.rs:608-620
//       ContactManifold::prune_points               src/collision/contact_types/mod.rs:477-566
//       ContactManifold::match_contacts             src/collision/contact_types/mod.rs:425-475
//       PackedFeatureId                             src/collision/contact_types/feature_id.rs:11-45
//   * THIRD-PARTY arithmetic that is NOT vendored under /root/reference (Cargo.lock pins parry3d 0.25.0 :3564-3565,
//     nalgebra through it, glam 0.30.8 :2472-2473): parry3d's `DefaultQueryDispatcher::contact_manifolds` for
//     (Ball, Ball), (Ball, Cuboid), (Cuboid, Ball), (Cuboid, Cuboid) — `contact_manifold_ball_ball`,
//     `contact_manifold_convex_ball`, `contact_manifold_cuboid_cuboid` (SAT: two one-way face searches + the two-way
//     edge-edge search; `Cuboid::support_face`; `PolygonalFeature::contacts` face/face clipping in the plane orthogonal
//     to the separating axis), nalgebra's `Isometry3` / `UnitQuaternion` arithmetic and glam's `Quat::to_scaled_axis`.
//     Their PUBLISHED ALGORITHMS are restated here from the crate documentation / source as published; the crates cannot
//     be built or read in this image, so this part is **PARITY UNPINNED**: the reference holds no golden manifold values
//      and bit-level agreement with the real crates is not claimed.  What IS checked: the HIP product
//     against this restatement bit for bit, and geometric invariants (tests/test_narrow_*.py).
#pragma once
#include <cstring>

#include "avo_math.hpp"

namespace avo {

// ---- deterministic atan (fdlibm s_atan.c argument reduction + polynomial, evaluated in S with plain IEEE ops and
//      without the hi/lo split of the reduction constants; like sin_cos_det it is "the platform libm" of this build) ----
template <class S> inline S atan_pos_det(S x) {  // x >= 0
    static const double aT[11] = {3.33333333333329318027e-01, -1.99999999998764832476e-01, 1.42857142725034663711e-01, -1.11111104054623557880e-01,
                                  9.09088713343650656196e-02, -7.69187620504482999495e-02, 6.66107313738753120669e-02, -5.83357013379057348645e-02,
                                  4.97687799461593236017e-02, -3.65315727442169155270e-02, 1.62858201153657823623e-02};
    static const double atanhi[4] = {4.63647609000806093515e-01, 7.85398163397448278999e-01, 9.82793723247329054082e-01, 1.57079632679489655800e+00};
    int id;
    if (x < S(0.4375)) id = -1;
    else if (x < S(0.6875)) { id = 0; x = (S(2) * x - S(1)) / (S(2) + x); }
    else if (x < S(1.1875)) { id = 1; x = (x - S(1)) / (x + S(1)); }
    else if (x < S(2.4375)) { id = 2; x = (x - S(1.5)) / (S(1) + S(1.5) * x); }
    else { id = 3; x = S(-1) / x; }
    S z = x * x, w = z * z;
    S s1 = z * (S(aT[0]) + w * (S(aT[2]) + w * (S(aT[4]) + w * (S(aT[6]) + w * (S(aT[8]) + w * S(aT[10]))))));
    S s2 = w * (S(aT[1]) + w * (S(aT[3]) + w * (S(aT[5]) + w * (S(aT[7]) + w * S(aT[9])))));
    if (id < 0) return x - x * (s1 + s2);
    return S(atanhi[id]) - (x * (s1 + s2) - x);
}
// atan2(y, x) for y > 0 (the only case Quat::to_axis_angle produces: y = |q.xyz| >= 1e-8)
template <class S> inline S atan2_ypos_det(S y, S x) {
    const S PI = S(3.14159265358979323846), HALF_PI = S(1.57079632679489661923);
    if (x == S(0)) return HALF_PI;
    S a = atan_pos_det<S>(y / std::fabs(x));
    return x > S(0) ? a : PI - a;
}

// ---- nalgebra (parry3d's math backend), restated ---------------------------------------------------------------
// Vector3::cross: [a.y b.z - a.z b.y, a.z b.x - a.x b.z, a.x b.y - a.y b.x]; dot / norm_squared: left-to-right sums.
template <class S> inline V3<S> na_cross(V3<S> a, V3<S> b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
template <class S> inline S na_dot(V3<S> a, V3<S> b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
template <class S> inline S na_norm(V3<S> a) { return std::sqrt(na_dot(a, a)); }
// UnitQuaternion * Vector3:  t = 2 (q.v x v);  q.w t + q.v x t + v
template <class S> inline V3<S> na_qrot(Q4<S> q, V3<S> v) {
    V3<S> qv{q.x, q.y, q.z};
    V3<S> t = na_cross(qv, v) * S(2);
    V3<S> c = na_cross(qv, t);
    return (t * q.w + c) + v;
}
template <class S> inline Q4<S> na_qmul(Q4<S> a, Q4<S> b) {
    return {a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
            a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x,
            a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w,
            a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z};
}
template <class S> inline Q4<S> na_qconj(Q4<S> q) { return {-q.x, -q.y, -q.z, q.w}; }
template <class S> struct Iso { Q4<S> r; V3<S> t; };  // Isometry3 { rotation, translation }
template <class S> inline Iso<S> iso_inverse(const Iso<S>& a) { Q4<S> ri = na_qconj(a.r); return {ri, na_qrot(ri, -a.t)}; }
template <class S> inline Iso<S> iso_inv_mul(const Iso<S>& a, const Iso<S>& b) {  // a^-1 * b
    Q4<S> ri = na_qconj(a.r);
    return {na_qmul(ri, b.r), na_qrot(ri, b.t - a.t)};
}
template <class S> inline V3<S> iso_point(const Iso<S>& a, V3<S> p) { return na_qrot(a.r, p) + a.t; }
template <class S> inline V3<S> iso_vec(const Iso<S>& a, V3<S> v) { return na_qrot(a.r, v); }
template <class S> inline V3<S> iso_inv_vec(const Iso<S>& a, V3<S> v) { return na_qrot(na_qconj(a.r), v); }
template <class S> inline V3<S> iso_inv_point(const Iso<S>& a, V3<S> p) { return na_qrot(na_qconj(a.r), p - a.t); }

// make_isometry (math/mod.rs:608-620): Isometry::new(position, rotation.to_scaled_axis()) — glam Quat::to_axis_angle
// (|v| >= 1e-8 ? (v / |v|, 2 atan2(|v|, w)) : (X, 0)) then nalgebra UnitQuaternion::from_scaled_axis (angle = |a|;
// (sin(angle / 2) a / angle, cos(angle / 2)); zero vector -> identity).
template <class S> inline Iso<S> make_isometry(V3<S> position, Q4<S> rotation) {
    V3<S> v{rotation.x, rotation.y, rotation.z};
    S len = length(v);
    V3<S> scaled = vzero<S>();
    if (len >= S(1.0e-8)) {
        S angle = S(2) * atan2_ypos_det<S>(len, rotation.w);
        scaled = (v / len) * angle;
    }
    S angle = na_norm(scaled);
    Q4<S> q{S(0), S(0), S(0), S(1)};
    if (angle != S(0)) {
        V3<S> axis = scaled / angle;
        S s, c;
        sin_cos_det(angle * S(0.5), s, c);
        q = {axis.x * s, axis.y * s, axis.z * s, c};
    }
    return {q, position};
}

// ---- PackedFeatureId (feature_id.rs:11-45 = parry's) ----------------------------------------------------------------
static const uint32_t FID_UNKNOWN = 0u;
inline uint32_t fid_vertex(uint32_t code) { return (1u << 30) | code; }
inline uint32_t fid_edge(uint32_t code) { return (2u << 30) | code; }
inline uint32_t fid_face(uint32_t code) { return (3u << 30) | code; }

// ---- parry3d ContactManifold (local frames) ------------------------------------------------------------------------
#define AVO_MAX_RAW_POINTS 16  /* 4 + 4 vertex contacts + up to 8 edge/edge crossings of two quads */
template <class S> struct TrackedContact { V3<S> local_p1, local_p2; S dist; uint32_t fid1, fid2; };
template <class S> struct RawManifold {
    V3<S> local_n1, local_n2;
    int n;
    TrackedContact<S> pts[AVO_MAX_RAW_POINTS];
    void push(V3<S> p1, V3<S> p2, uint32_t f1, uint32_t f2, S dist, bool flipped) {
        if (n >= AVO_MAX_RAW_POINTS) return;
        pts[n++] = flipped ? TrackedContact<S>{p2, p1, dist, f2, f1} : TrackedContact<S>{p1, p2, dist, f1, f2};  // TrackedContact::flipped
    }
};

// contact_manifold_ball_ball
template <class S> inline void manifold_ball_ball(const Iso<S>& pos12, S r1, S r2, S prediction, RawManifold<S>& m) {
    m.n = 0;
    V3<S> dcenter = pos12.t;
    S center_dist = na_norm(dcenter);
    S dist = center_dist - r1 - r2;
    if (dist < prediction) {
        V3<S> local_n1 = center_dist != S(0) ? dcenter / center_dist : V3<S>{S(0), S(1), S(0)};
        V3<S> local_n2 = iso_inv_vec(pos12, -local_n1);
        m.push(local_n1 * r1, local_n2 * r2, fid_face(0), fid_face(0), dist, false);
        m.local_n1 = local_n1; m.local_n2 = local_n2;
    }
}

// Cuboid::project_local_point(pt, solid = true) through Aabb::do_project_local_point
template <class S> inline V3<S> cuboid_project(V3<S> he, V3<S> pt, bool& inside) {
    auto sup0 = [](S a) { return a > S(0) ? a : S(0); };
    V3<S> shift{sup0(-he.x - pt.x) - sup0(pt.x - he.x), sup0(-he.y - pt.y) - sup0(pt.y - he.y), sup0(-he.z - pt.z) - sup0(pt.z - he.z)};
    inside = shift.x == S(0) && shift.y == S(0) && shift.z == S(0);
    return inside ? pt : pt + shift;
}
// contact_manifold_convex_ball: shape1 = the cuboid (in whose frame pos12 places the ball), `flipped` when the ball is
// collider 1 (parry then calls it with pos12.inverse()).
template <class S> inline void manifold_cuboid_ball(const Iso<S>& pos12, V3<S> he1, S r2, S prediction, bool flipped, RawManifold<S>& m) {
    m.n = 0;
    V3<S> local_p2_1 = pos12.t;
    bool inside;
    V3<S> proj = cuboid_project(he1, local_p2_1, inside);
    V3<S> dpos = local_p2_1 - proj;
    S dist = na_norm(dpos);
    if (!(dist > S(0))) return;  // Unit::try_new_and_get(dpos, 0.0) fails (centre inside a solid cuboid): no contact
    V3<S> local_n1 = dpos / dist;
    if (inside) { local_n1 = -local_n1; dist = -dist; }
    if (dist <= r2 + prediction) {
        V3<S> local_n2 = iso_inv_vec(pos12, -local_n1);
        V3<S> local_p2 = local_n2 * r2;
        m.push(proj, local_p2, FID_UNKNOWN, fid_face(0), dist - r2, flipped);
        if (flipped) { m.local_n1 = local_n2; m.local_n2 = local_n1; }
        else { m.local_n1 = local_n1; m.local_n2 = local_n2; }
    }
}

// ---- cuboid / cuboid -------------------------------------------------------------------------------------------------
template <class S> inline S vget(V3<S> v, int i) { return i == 0 ? v.x : (i == 1 ? v.y : v.z); }
template <class S> inline V3<S> cuboid_support_point(V3<S> he, V3<S> dir) {  // local_support_point: he_i.copysign(dir_i)
    return {std::copysign(he.x, dir.x), std::copysign(he.y, dir.y), std::copysign(he.z, dir.z)};
}
// sat::cuboid_cuboid_find_local_separating_normal_oneway
template <class S> inline S sat_normal_oneway(V3<S> he1, V3<S> he2, const Iso<S>& pos12, V3<S>& best_dir) {
    S best = -std::numeric_limits<S>::max();
    best_dir = vzero<S>();
    for (int i = 0; i < 3; ++i) {
        S sign = std::copysign(S(1), vget(pos12.t, i));
        V3<S> axis1{i == 0 ? sign : S(0), i == 1 ? sign : S(0), i == 2 ? sign : S(0)};
        V3<S> axis2 = iso_inv_vec(pos12, -axis1);
        V3<S> pt2 = iso_point(pos12, cuboid_support_point(he2, axis2));
        S separation = vget(pt2, i) * sign - vget(he1, i);
        if (separation > best) { best = separation; best_dir = axis1; }
    }
    return best;
}
// sat::cuboid_support_map_compute_separation_wrt_local_line
template <class S> inline S sat_line_separation(V3<S> he1, V3<S> he2, const Iso<S>& pos12, V3<S> axis1, V3<S>& out_axis) {
    V3<S> axis1_2 = iso_inv_vec(pos12, axis1);
    S separation1, separation2;
    {
        V3<S> pt1 = cuboid_support_point(he1, axis1);
        V3<S> pt2 = iso_point(pos12, cuboid_support_point(he2, -axis1_2));
        separation1 = na_dot(pt2 - pt1, axis1);
    }
    {
        V3<S> pt1 = cuboid_support_point(he1, -axis1);
        V3<S> pt2 = iso_point(pos12, cuboid_support_point(he2, axis1_2));
        separation2 = na_dot(pt2 - pt1, -axis1);
    }
    if (separation1 > separation2) { out_axis = axis1; return separation1; }
    out_axis = -axis1;
    return separation2;
}
// sat::cuboid_cuboid_find_local_separating_edge_twoway
template <class S> inline S sat_edge_twoway(V3<S> he1, V3<S> he2, const Iso<S>& pos12, V3<S>& best_dir) {
    V3<S> x2 = iso_vec(pos12, V3<S>{S(1), S(0), S(0)}), y2 = iso_vec(pos12, V3<S>{S(0), S(1), S(0)}), z2 = iso_vec(pos12, V3<S>{S(0), S(0), S(1)});
    const V3<S> axes[9] = {{S(0), -x2.z, x2.y}, {x2.z, S(0), -x2.x}, {-x2.y, x2.x, S(0)},
                           {S(0), -y2.z, y2.y}, {y2.z, S(0), -y2.x}, {-y2.y, y2.x, S(0)},
                           {S(0), -z2.z, z2.y}, {z2.z, S(0), -z2.x}, {-z2.y, z2.x, S(0)}};
    S best = -std::numeric_limits<S>::max();
    best_dir = vzero<S>();
    for (int k = 0; k < 9; ++k) {
        S norm1 = na_norm(axes[k]);
        if (norm1 > std::numeric_limits<S>::epsilon()) {
            V3<S> ax;
            S sep = sat_line_separation(he1, he2, pos12, axes[k] / norm1, ax);
            if (sep > best) { best = sep; best_dir = ax; }
        }
    }
    return best;
}
// Cuboid::support_face -> PolygonalFeature { vertices[4], vids, eids, fid }
template <class S> struct Face { V3<S> v[4]; uint32_t vid[4], eid[4], fid; };
template <class S> inline Face<S> cuboid_support_face(V3<S> he, V3<S> dir) {
    S ax = std::fabs(dir.x), ay = std::fabs(dir.y), az = std::fabs(dir.z);
    int iamax = 0;  // nalgebra iamax: first maximum
    S mx = ax;
    if (ay > mx) { mx = ay; iamax = 1; }
    if (az > mx) { mx = az; iamax = 2; }
    S sign = std::copysign(S(1), vget(dir, iamax));
    Face<S> f;
    if (iamax == 0) { f.v[0] = {he.x * sign, he.y, he.z}; f.v[1] = {he.x * sign, -he.y, he.z}; f.v[2] = {he.x * sign, -he.y, -he.z}; f.v[3] = {he.x * sign, he.y, -he.z}; }
    else if (iamax == 1) { f.v[0] = {he.x, he.y * sign, he.z}; f.v[1] = {-he.x, he.y * sign, he.z}; f.v[2] = {-he.x, he.y * sign, -he.z}; f.v[3] = {he.x, he.y * sign, -he.z}; }
    else { f.v[0] = {he.x, he.y, he.z * sign}; f.v[1] = {he.x, -he.y, he.z * sign}; f.v[2] = {-he.x, -he.y, he.z * sign}; f.v[3] = {-he.x, he.y, he.z * sign}; }
    // vertex id: bit 0 / 1 / 2 set when the x / y / z component is negative
    for (int k = 0; k < 4; ++k) {
        uint32_t id = (f.v[k].x < S(0) ? 1u : 0u) | (f.v[k].y < S(0) ? 2u : 0u) | (f.v[k].z < S(0) ? 4u : 0u);
        f.vid[k] = fid_vertex(id);
    }
    for (int k = 0; k < 4; ++k) {  // edge k joins vertices k and k + 1: (larger id << 3) | smaller id | 0b11000000
        uint32_t a = f.vid[k] & 7u, b = f.vid[(k + 1) & 3] & 7u;
        uint32_t hi = a > b ? a : b, lo = a > b ? b : a;
        f.eid[k] = fid_edge((hi << 3) | lo | 0xC0u);
    }
    uint32_t sign_index = sign > S(0) ? 1u : 0u;
    f.fid = fid_face((uint32_t)iamax + sign_index * 3u + 10u);
    return f;
}
// utils::WBasis::orthonormal_basis (Duff et al., "Building an Orthonormal Basis, Revisited")
template <class S> inline void na_orthonormal_basis(V3<S> n, V3<S>& b0, V3<S>& b1) {
    S sign = std::copysign(S(1), n.z);
    S a = S(-1) / (sign + n.z);
    S b = n.x * n.y * a;
    b0 = {S(1) + sign * n.x * n.x * a, sign * b, -sign * n.x};
    b1 = {b, sign + n.y * n.y * a, -n.y};
}
template <class S> inline S perp2(V2<S> a, V2<S> b) { return a.x * b.y - a.y * b.x; }
inline bool ulps_apart_le4(float a, float b) {
    if (std::signbit(a) != std::signbit(b)) return false;
    int32_t x, y; std::memcpy(&x, &a, 4); std::memcpy(&y, &b, 4);
    int64_t d = (int64_t)x - (int64_t)y;
    return (d < 0 ? -d : d) <= 4;
}
inline bool ulps_apart_le4(double a, double b) {
    if (std::signbit(a) != std::signbit(b)) return false;
    int64_t x, y; std::memcpy(&x, &a, 8); std::memcpy(&y, &b, 8);
    // same sign: the difference of the bit patterns cannot overflow
    int64_t d = x - y;
    return (d < 0 ? -d : d) <= 4;
}
// query::closest_points::closest_points_line2d: barycentric parameters of the crossing of two 2-D lines
template <class S> inline bool closest_points_line2d(V2<S> e1a, V2<S> e1b, V2<S> e2a, V2<S> e2b, S& s_out, S& t_out) {
    V2<S> dir1{e1b.x - e1a.x, e1b.y - e1a.y}, dir2{e2b.x - e2a.x, e2b.y - e2a.y}, r{e1a.x - e2a.x, e1a.y - e2a.y};
    S a = dir1.x * dir1.x + dir1.y * dir1.y, e = dir2.x * dir2.x + dir2.y * dir2.y, f = dir2.x * r.x + dir2.y * r.y;
    const S eps = std::numeric_limits<S>::epsilon();
    if (a <= eps && e <= eps) { s_out = S(0); t_out = S(0); return true; }
    if (a <= eps) { s_out = S(0); t_out = f / e; return true; }
    S c = dir1.x * r.x + dir1.y * r.y;
    if (e <= eps) { s_out = -c / a; t_out = S(0); return true; }
    S b = dir1.x * dir2.x + dir1.y * dir2.y;
    S ae = a * e, bb = b * b, denom = ae - bb;
    // `denom <= eps || ulps_eq!(ae, bb)`; approx::ulps_eq (epsilon = EPSILON, max_ulps = 4): |a - b| <= eps, or same sign and the
    // bit patterns are at most 4 apart
    bool parallel = denom <= eps || std::fabs(ae - bb) <= eps || ulps_apart_le4(ae, bb);
    if (parallel) return false;
    S s = (b * f - c * e) / denom;
    s_out = s;
    t_out = (b * s + f) / e;
    return true;
}
// PolygonalFeature::contacts, face / face branch
template <class S> inline void face_face_contacts(const Iso<S>& pos12, const Face<S>& face1, V3<S> sep_axis1, const Face<S>& face2, RawManifold<S>& m) {
    V3<S> b0, b1;
    na_orthonormal_basis(sep_axis1, b0, b1);
    V2<S> p1[4], p2[4];
    V3<S> v2_1[4];
    for (int k = 0; k < 4; ++k) {
        p1[k] = {na_dot(face1.v[k], b0), na_dot(face1.v[k], b1)};
        v2_1[k] = iso_point(pos12, face2.v[k]);
        p2[k] = {na_dot(v2_1[k], b0), na_dot(v2_1[k], b1)};
    }
    auto inside = [](const V2<S>* poly, V2<S> p) {
        S sign = perp2(V2<S>{poly[0].x - poly[3].x, poly[0].y - poly[3].y}, V2<S>{p.x - poly[3].x, p.y - poly[3].y});
        for (int j = 0; j < 3; ++j) {
            S ns = perp2(V2<S>{poly[j + 1].x - poly[j].x, poly[j + 1].y - poly[j].y}, V2<S>{p.x - poly[j].x, p.y - poly[j].y});
            if (sign == S(0)) sign = ns;
            else if (sign * ns < S(0)) return false;
        }
        return true;
    };
    auto approx_zero = [](S d) { return std::fabs(d) <= std::numeric_limits<S>::epsilon(); };  // relative_eq!(denom, 0.0)
    {   // vertices of face1 inside the projection of face2
        V3<S> normal2_1 = na_cross(v2_1[2] - v2_1[1], v2_1[0] - v2_1[1]);
        S denom = na_dot(normal2_1, sep_axis1);
        if (!approx_zero(denom))
            for (int i = 0; i < 4; ++i)
                if (inside(p2, p1[i])) {
                    S dist = na_dot(v2_1[0] - face1.v[i], normal2_1) / denom;
                    V3<S> local_p1 = face1.v[i];
                    V3<S> local_p2_1 = face1.v[i] + sep_axis1 * dist;
                    m.push(local_p1, iso_inv_point(pos12, local_p2_1), face1.vid[i], face2.fid, dist, false);
                }
    }
    {   // vertices of face2 inside the projection of face1
        V3<S> normal1 = na_cross(face1.v[2] - face1.v[1], face1.v[0] - face1.v[1]);
        S denom = -na_dot(normal1, sep_axis1);
        if (!approx_zero(denom))
            for (int i = 0; i < 4; ++i)
                if (inside(p1, p2[i])) {
                    S dist = na_dot(face1.v[0] - v2_1[i], normal1) / denom;
                    V3<S> local_p2_1 = v2_1[i];
                    V3<S> local_p1 = v2_1[i] - sep_axis1 * dist;
                    m.push(local_p1, iso_inv_point(pos12, local_p2_1), face1.fid, face2.vid[i], dist, false);
                }
    }
    // edge / edge crossings
    for (int j = 0; j < 4; ++j)
        for (int i = 0; i < 4; ++i) {
            S s, t;
            if (!closest_points_line2d(p1[i], p1[(i + 1) & 3], p2[j], p2[(j + 1) & 3], s, t)) continue;
            if (s > S(0) && s < S(1) && t > S(0) && t < S(1)) {
                V3<S> local_p1 = face1.v[i] * (S(1) - s) + face1.v[(i + 1) & 3] * s;
                V3<S> local_p2_1 = v2_1[j] * (S(1) - t) + v2_1[(j + 1) & 3] * t;
                S dist = na_dot(local_p2_1 - local_p1, sep_axis1);
                m.push(local_p1, iso_inv_point(pos12, local_p2_1), face1.eid[i], face2.eid[j], dist, false);
            }
        }
}
// contact_manifold_cuboid_cuboid
template <class S> inline void manifold_cuboid_cuboid(const Iso<S>& pos12, V3<S> he1, V3<S> he2, S prediction, RawManifold<S>& m) {
    m.n = 0;
    Iso<S> pos21 = iso_inverse(pos12);
    V3<S> d1, d2, d3;
    S sep1 = sat_normal_oneway(he1, he2, pos12, d1);
    if (sep1 > prediction) return;
    S sep2 = sat_normal_oneway(he2, he1, pos21, d2);
    if (sep2 > prediction) return;
    S sep3 = sat_edge_twoway(he1, he2, pos12, d3);
    if (sep3 > prediction) return;
    V3<S> best = d1;
    if (sep2 > sep1 && sep2 > sep3) best = iso_vec(pos12, -d2);
    else if (sep3 > sep1) best = d3;
    V3<S> local_n2 = iso_vec(pos21, -best);
    Face<S> f1 = cuboid_support_face(he1, best), f2 = cuboid_support_face(he2, local_n2);
    face_face_contacts(pos12, f1, best, f2, m);
    m.local_n1 = best; m.local_n2 = local_n2;
}

// ---- Avian's ContactManifold as produced by contact_query::contact_manifolds (anchors relative to the collider origins) ----
template <class S> struct QueryPoint { V3<S> anchor1, anchor2, point; S penetration; uint32_t fid1, fid2; };
template <class S> struct QueryManifold { V3<S> normal; int n; QueryPoint<S> pts[AVO_MAX_RAW_POINTS]; };

// contact_query.rs:156-261 for one pair; returns false when there is no manifold
template <class S>
inline bool contact_manifolds_pair(uint8_t shape1, V3<S> he1, V3<S> position1, Q4<S> rotation1, uint8_t shape2, V3<S> he2, V3<S> position2, Q4<S> rotation2,
                                   S prediction_distance, QueryManifold<S>& out) {
    out.n = 0;
    Iso<S> isometry1 = make_isometry(position1, rotation1), isometry2 = make_isometry(position2, rotation2);
    Iso<S> pos12 = iso_inv_mul(isometry1, isometry2);
    RawManifold<S> m;
    m.n = 0;
    const bool ball1 = shape1 == AVN_SHAPE_BALL, ball2 = shape2 == AVN_SHAPE_BALL;
    if (ball1 && ball2) manifold_ball_ball(pos12, he1.x, he2.x, prediction_distance, m);
    else if (ball1) manifold_cuboid_ball(iso_inverse(pos12), he2, he1.x, prediction_distance, true, m);
    else if (ball2) manifold_cuboid_ball(pos12, he1, he2.x, prediction_distance, false, m);
    else manifold_cuboid_cuboid(pos12, he1, he2, prediction_distance, m);
    if (m.n == 0) return false;  // "Skip empty manifolds."
    // local_normal = subpos1.rotation * local_n1 (identity), normalised (nalgebra: v / |v|); normal = rotation1 * local_normal
    V3<S> local_normal = m.local_n1 / na_norm(m.local_n1);
    V3<S> normal = qrot(rotation1, local_normal);
    if (!(std::fabs(length_squared(normal) - S(1)) <= S(2e-4))) return false;  // glam is_normalized
    out.normal = normal;
    for (int k = 0; k < m.n; ++k) {
        const TrackedContact<S>& c = m.pts[k];
        V3<S> point1 = qrot(rotation1, c.local_p1);
        V3<S> anchor1 = point1 + (normal * c.dist) * S(0.5);
        V3<S> anchor2 = anchor1 + (position1 - position2);
        out.pts[out.n++] = {anchor1, anchor2, position1 + anchor1, -c.dist, c.fid1, c.fid2};
    }
    return true;
}

}  // namespace avo
