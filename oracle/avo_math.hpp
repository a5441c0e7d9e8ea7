// ORACLE — TEST INFRASTRUCTURE ONLY.  Never linked, imported or called by the product path.
//
// avo_math.hpp: CPU restatement of the glam 0.30.8 / glam_matrix_extras 0.1.0 arithmetic the
// reference relies on.  Those crates are third-party dependencies that
// are NOT vendored under /root/reference (Cargo.lock pins glam 0.30.8, glam_matrix_extras 0.1.0,
// parry3d 0.25.0); their published algorithms are re-stated here, operation order included, and
// are "parity unpinned" against the real crates (no Rust toolchain in this image).
//
// Build with -ffp-contract=off: every expression below is meant to round exactly as written.
#pragma once
#include <cmath>
#include <cstdint>
#include <limits>

namespace avo {

template <class S> struct V3 { S x, y, z; };
template <class S> struct V2 { S x, y; };
template <class S> struct Q4 { S x, y, z, w; };
// glam_matrix_extras::SymmetricMat3 field order
template <class S> struct Sym3 { S m00, m01, m02, m11, m12, m22; };
// glam::Mat3 (column major: x_axis, y_axis, z_axis)
template <class S> struct M3 { V3<S> c0, c1, c2; };

template <class S> inline V3<S> v3(S x, S y, S z) { return V3<S>{x, y, z}; }
template <class S> inline V3<S> vzero() { return V3<S>{S(0), S(0), S(0)}; }
template <class S> inline V3<S> operator+(V3<S> a, V3<S> b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
template <class S> inline V3<S> operator-(V3<S> a, V3<S> b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
template <class S> inline V3<S> operator-(V3<S> a) { return {-a.x, -a.y, -a.z}; }
template <class S> inline V3<S> operator*(V3<S> a, S s) { return {a.x * s, a.y * s, a.z * s}; }
template <class S> inline V3<S> operator*(S s, V3<S> a) { return {s * a.x, s * a.y, s * a.z}; }
template <class S> inline V3<S> operator/(V3<S> a, S s) { return {a.x / s, a.y / s, a.z / s}; }
// component-wise product (Vec3 * Vec3)
template <class S> inline V3<S> cmul(V3<S> a, V3<S> b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
// glam Vec3::dot: (x*x) + (y*y) + (z*z), left to right
template <class S> inline S dot(V3<S> a, V3<S> b) { return (a.x * b.x) + (a.y * b.y) + (a.z * b.z); }
// glam Vec3::cross
template <class S> inline V3<S> cross(V3<S> a, V3<S> b) {
    return {a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y};
}
template <class S> inline S length_squared(V3<S> a) { return dot(a, a); }
template <class S> inline S length(V3<S> a) { return std::sqrt(dot(a, a)); }
template <class S> inline S max_element(V3<S> a) { S m = a.x > a.y ? a.x : a.y; return m > a.z ? m : a.z; }
template <class S> inline bool is_finite(V3<S> a) { return std::isfinite(a.x) && std::isfinite(a.y) && std::isfinite(a.z); }
template <class S> inline V3<S> vmin(V3<S> a, V3<S> b) { return {a.x < b.x ? a.x : b.x, a.y < b.y ? a.y : b.y, a.z < b.z ? a.z : b.z}; }
template <class S> inline V3<S> vmax(V3<S> a, V3<S> b) { return {a.x > b.x ? a.x : b.x, a.y > b.y ? a.y : b.y, a.z > b.z ? a.z : b.z}; }

// Rust f32::max / f32::min (NaN-ignoring); inputs here are never NaN in valid runs.
template <class S> inline S smax(S a, S b) { return (a > b || b != b) ? a : b; }
template <class S> inline S smin(S a, S b) { return (a < b || b != b) ? a : b; }

// math/mod.rs:248-257 RecipOrZero
template <class S> inline S recip_or_zero(S x) { return (x != S(0) && std::isfinite(x)) ? S(1) / x : S(0); }
template <class S> inline V3<S> recip_or_zero(V3<S> v) { return {recip_or_zero(v.x), recip_or_zero(v.y), recip_or_zero(v.z)}; }

// glam Vec3::try_normalize: rcp = 1/length; Some(v*rcp) iff rcp finite and > 0
template <class S> inline bool try_normalize(V3<S> v, V3<S>& out) {
    S rcp = S(1) / length(v);
    if (std::isfinite(rcp) && rcp > S(0)) { out = v * rcp; return true; }
    return false;
}
// glam Vec3::clamp_length_max: if len_sq > max*max { max * (self / sqrt(len_sq)) }
template <class S> inline V3<S> clamp_length_max(V3<S> v, S max) {
    S len_sq = length_squared(v);
    if (len_sq > max * max) return max * (v / std::sqrt(len_sq));
    return v;
}
template <class S> inline V2<S> clamp_length_max(V2<S> v, S max) {
    S len_sq = (v.x * v.x) + (v.y * v.y);
    if (len_sq > max * max) { S l = std::sqrt(len_sq); return {max * (v.x / l), max * (v.y / l)}; }
    return v;
}
// glam Vec3::any_orthonormal_vector (Duff et al. branchless ONB); signum(+0)=1, signum(-0)=-1
template <class S> inline V3<S> any_orthonormal_vector(V3<S> v) {
    S sign = std::signbit(v.z) ? S(-1) : S(1);
    S a = S(-1) / (sign + v.z);
    S b = v.x * v.y * a;
    return {b, sign + v.y * v.y * a, -v.y};
}

// ---- deterministic sin/cos ------------------------------------------------------------------
// The reference calls Rust's f32::sin_cos, i.e. "the platform libm" (the
// reference is not bit-reproducible across platforms in its default configuration).  The oracle
// and the HIP product both use the SAME published algorithm (Cody-Waite 3-term reduction by pi/2
// + Cephes minimax polynomials on [-pi/4, pi/4]), written with plain IEEE ops, so that the two
// can be compared bit-for-bit.  avo::g_use_libm_trig switches to the host libm to measure the
// deviation (tests/test_oracle_math.py pins it to <= 2 ulp for |x| <= 100).
inline bool& use_libm_trig() { static bool v = false; return v; }

inline void sin_cos_det(float a, float& s, float& c) {
    const float TWO_OVER_PI = 0.63661977236758134308f;
    const float P1 = 1.5703125f, P2 = 4.837512969970703125e-4f, P3 = 7.54978995489188216e-8f;
    float kf = std::nearbyint(a * TWO_OVER_PI);
    float r = ((a - kf * P1) - kf * P2) - kf * P3;
    float z = r * r;
    float sp = ((-1.9515295891e-4f * z + 8.3321608736e-3f) * z - 1.6666654611e-1f) * z * r + r;
    float cp = ((2.443315711809948e-5f * z - 1.388731625493765e-3f) * z + 4.166664568298827e-2f) * z * z
               - 0.5f * z + 1.0f;
    float qf = kf - 4.0f * std::floor(kf * 0.25f);  // kf mod 4, exact for every finite kf (no float->int conversion: UB-free for huge arguments)
    int q = qf == 1.0f ? 1 : qf == 2.0f ? 2 : qf == 3.0f ? 3 : 0;
    switch (q) {
        case 0: s = sp; c = cp; break;
        case 1: s = cp; c = -sp; break;
        case 2: s = -sp; c = -cp; break;
        default: s = -cp; c = sp; break;
    }
}
inline void sin_cos_det(double a, double& s, double& c) {
    const double TWO_OVER_PI = 6.36619772367581382433e-01;
    const double P1 = 1.57079632673412561417e+00, P2 = 6.07710050630396597660e-11, P3 = 2.02226624871116645580e-21;
    double kf = std::nearbyint(a * TWO_OVER_PI);
    double r = ((a - kf * P1) - kf * P2) - kf * P3;
    double z = r * r;
    double sp = (((((1.58962301576546568060e-10 * z - 2.50507477628578072866e-8) * z + 2.75573136213857245213e-6) * z
                   - 1.98412698295895385996e-4) * z + 8.33333333332211858878e-3) * z - 1.66666666666666307295e-1) * z * r + r;
    double cp = (((((-1.13585365213876817300e-11 * z + 2.08757008419747316778e-9) * z - 2.75573141792967388112e-7) * z
                   + 2.48015872888517045348e-5) * z - 1.38888888888730564116e-3) * z + 4.16666666666665929218e-2) * z * z
                - 0.5 * z + 1.0;
    double qf = kf - 4.0 * std::floor(kf * 0.25);
    int q = qf == 1.0 ? 1 : qf == 2.0 ? 2 : qf == 3.0 ? 3 : 0;
    switch (q) {
        case 0: s = sp; c = cp; break;
        case 1: s = cp; c = -sp; break;
        case 2: s = -sp; c = -cp; break;
        default: s = -cp; c = sp; break;
    }
}
template <class S> inline void sin_cos(S a, S& s, S& c) {
    if (use_libm_trig()) { s = std::sin(a); c = std::cos(a); return; }
    sin_cos_det(a, s, c);
}

// ---- quaternions -----------------------------------------------------------------------------
template <class S> inline Q4<S> qidentity() { return {S(0), S(0), S(0), S(1)}; }
// glam Quat::from_axis_angle / from_scaled_axis
template <class S> inline Q4<S> from_scaled_axis(V3<S> v) {
    S len = length(v);
    if (len == S(0)) return qidentity<S>();
    V3<S> axis = v / len;
    S s, c;
    sin_cos(len * S(0.5), s, c);
    V3<S> a = axis * s;
    return {a.x, a.y, a.z, c};
}
// Deterministic asin for AngleLimit::compute_correction (dynamics/joints/mod.rs:437): Rust's f32/f64::asin is the platform
// libm; this restatement fixes ONE algorithm (fdlibm e_asin.c rational R = p/q, no high/low sqrt split) so that results
// do not depend on the host libm.  |x| > 1 -> NaN (as libm).  With use_libm_trig() the host libm is used instead.
template <class S> inline S asin_rational(S z) {
    S p = z * (S(1.66666666666666657415e-01) + z * (S(-3.25565818622400915405e-01) + z * (S(2.01212532134862925881e-01) +
          z * (S(-4.00555345006794114027e-02) + z * (S(7.91534994289814532176e-04) + z * S(3.47933107596021167570e-05))))));
    S q = S(1) + z * (S(-2.40339491173441421878e+00) + z * (S(2.02094576023350569471e+00) + z * (S(-6.88283971605453293030e-01) +
          z * S(7.70381505559019352791e-02))));
    return p / q;
}
template <class S> inline S asin_det(S x) {
    S ax = std::fabs(x);
    if (!(ax <= S(1))) return (x - x) / (x - x);
    if (ax < S(0.5)) return x + x * asin_rational<S>(x * x);
    S z = (S(1) - ax) * S(0.5);
    S s = std::sqrt(z);
    S r = S(1.57079632679489661923) - S(2) * (s + s * asin_rational<S>(z));
    return x < S(0) ? -r : r;
}
template <class S> inline S asin_s(S x) { return use_libm_trig() ? std::asin(x) : asin_det(x); }
// glam Quat::from_axis_angle
template <class S> inline Q4<S> from_axis_angle(V3<S> axis, S angle) {
    S s, c;
    sin_cos(angle * S(0.5), s, c);
    V3<S> v = axis * s;
    return {v.x, v.y, v.z, c};
}
// glam Vec3::any_orthogonal_vector: |x| > |y| ? (-z, 0, x) : (0, z, -y)
template <class S> inline V3<S> any_orthogonal_vector(V3<S> v) {
    if (std::fabs(v.x) > std::fabs(v.y)) return {-v.z, S(0), v.x};
    return {S(0), v.z, -v.y};
}
template <class S> inline S clamp_s(S x, S lo, S hi) { S r = x; if (r < lo) r = lo; if (r > hi) r = hi; return r; }
// glam Quat * Quat.  f32 `Quat` is SSE2-backed on x86_64 (rtm::quat_mul association):
//   (w_l*rhs + x_l*rhs.wzyx*[+,-,+,-]) + (y_l*rhs.zwxy*[+,+,-,-] + z_l*rhs.yxwz*[-,+,+,-])
// f64 `DQuat` is the scalar implementation (left-to-right sums).
// use_scalar_quat_mul(): glam's scalar-math f32 `Quat` product (targets without SSE2 / with the `scalar-math` feature) sums left to right
// like DQuat; switched on only to MEASURE how far the two associations drift apart (tests/test_oracle_tolerance.py).
inline bool& use_scalar_quat_mul() { static bool v = false; return v; }
inline Q4<float> qmul(Q4<float> l, Q4<float> r) {
    if (use_scalar_quat_mul())
        return {l.w * r.x + l.x * r.w + l.y * r.z - l.z * r.y, l.w * r.y - l.x * r.z + l.y * r.w + l.z * r.x, l.w * r.z + l.x * r.y - l.y * r.x + l.z * r.w,
                l.w * r.w - l.x * r.x - l.y * r.y - l.z * r.z};
    return {(l.w * r.x + l.x * r.w) + (l.y * r.z + -(l.z * r.y)),
            (l.w * r.y + -(l.x * r.z)) + (l.y * r.w + l.z * r.x),
            (l.w * r.z + l.x * r.y) + (-(l.y * r.x) + l.z * r.w),
            (l.w * r.w + -(l.x * r.x)) + (-(l.y * r.y) + -(l.z * r.z))};
}
inline Q4<double> qmul(Q4<double> l, Q4<double> r) {
    return {l.w * r.x + l.x * r.w + l.y * r.z - l.z * r.y,
            l.w * r.y - l.x * r.z + l.y * r.w + l.z * r.x,
            l.w * r.z + l.x * r.y - l.y * r.x + l.z * r.w,
            l.w * r.w - l.x * r.x - l.y * r.y - l.z * r.z};
}
// glam Quat::inverse == conjugate
template <class S> inline Q4<S> qinverse(Q4<S> q) { return {-q.x, -q.y, -q.z, q.w}; }
// glam Quat * Vec3:  v*(w*w - b.b) + b*(2*(v.b)) + (b x v)*(2*w)
template <class S> inline V3<S> qrot(Q4<S> q, V3<S> v) {
    S w = q.w;
    V3<S> b{q.x, q.y, q.z};
    S b2 = dot(b, b);
    return (v * (w * w - b2) + b * (dot(v, b) * S(2))) + cross(b, v) * (w * S(2));
}
// glam Quat::length_squared: SSE2 dot4 = (x2+z2)+(y2+w2) for f32, scalar for f64
inline float qlength_squared(Q4<float> q) { return (q.x * q.x + q.z * q.z) + (q.y * q.y + q.w * q.w); }
inline double qlength_squared(Q4<double> q) { return q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w; }
// glam Quat::normalize = Vec4::normalize: SSE2 (f32) divides by the length, the scalar path (f64) multiplies by its reciprocal
inline Q4<float> qnormalize(Q4<float> q) { const float l = std::sqrt(qlength_squared(q)); return {q.x / l, q.y / l, q.z / l, q.w / l}; }
inline Q4<double> qnormalize(Q4<double> q) { const double r = 1.0 / std::sqrt(qlength_squared(q)); return {q.x * r, q.y * r, q.z * r, q.w * r}; }
// physics_transform/transform.rs:811-817 Rotation::fast_renormalize
template <class S> inline Q4<S> fast_renormalize(Q4<S> q) {
    S l2 = qlength_squared(q);
    S k = S(0.5) * (S(3) - l2);
    return {q.x * k, q.y * k, q.z * k, q.w * k};
}

// ---- matrices ---------------------------------------------------------------------------------
// glam Mat3::from_quat
template <class S> inline M3<S> mat3_from_quat(Q4<S> r) {
    S x2 = r.x + r.x, y2 = r.y + r.y, z2 = r.z + r.z;
    S xx = r.x * x2, xy = r.x * y2, xz = r.x * z2;
    S yy = r.y * y2, yz = r.y * z2, zz = r.z * z2;
    S wx = r.w * x2, wy = r.w * y2, wz = r.w * z2;
    return {{S(1) - (yy + zz), xy + wz, xz - wy},
            {xy - wz, S(1) - (xx + zz), yz + wx},
            {xz + wy, yz - wx, S(1) - (xx + yy)}};
}
// glam Mat3 * Vec3: x_axis*v.x + y_axis*v.y + z_axis*v.z (left to right)
template <class S> inline V3<S> mmul(const M3<S>& m, V3<S> v) { return (m.c0 * v.x + m.c1 * v.y) + m.c2 * v.z; }
template <class S> inline M3<S> mmul(const M3<S>& a, const M3<S>& b) { return {mmul(a, b.c0), mmul(a, b.c1), mmul(a, b.c2)}; }
template <class S> inline M3<S> transpose(const M3<S>& m) {
    return {{m.c0.x, m.c1.x, m.c2.x}, {m.c0.y, m.c1.y, m.c2.y}, {m.c0.z, m.c1.z, m.c2.z}};
}
template <class S> inline M3<S> to_mat3(const Sym3<S>& s) {
    return {{s.m00, s.m01, s.m02}, {s.m01, s.m11, s.m12}, {s.m02, s.m12, s.m22}};
}
// SymmetricMat3::from_mat3_unchecked: takes the upper triangle (column j >= row i => m.cj[i])
template <class S> inline Sym3<S> sym_from_mat3_unchecked(const M3<S>& m) {
    return {m.c0.x, m.c1.x, m.c2.x, m.c1.y, m.c2.y, m.c2.z};
}
// SymmetricMat3 * Vec3 (mirrors Mat3::mul_vec3 over the symmetric columns)
template <class S> inline V3<S> smul(const Sym3<S>& s, V3<S> v) {
    V3<S> c0{s.m00, s.m01, s.m02}, c1{s.m01, s.m11, s.m12}, c2{s.m02, s.m12, s.m22};
    return (c0 * v.x + c1 * v.y) + c2 * v.z;
}
template <class S> inline Sym3<S> sym_zero() { return {S(0), S(0), S(0), S(0), S(0), S(0)}; }
template <class S> inline bool sym_is_zero(const Sym3<S>& s) {
    return s.m00 == 0 && s.m01 == 0 && s.m02 == 0 && s.m11 == 0 && s.m12 == 0 && s.m22 == 0;
}
// SymmetricMat3::determinant / inverse (cofactor form)
template <class S> inline S sym_determinant(const Sym3<S>& s) {
    S a = s.m11 * s.m22 - s.m12 * s.m12;
    S b = s.m12 * s.m02 - s.m22 * s.m01;
    S c = s.m01 * s.m12 - s.m02 * s.m11;
    return s.m00 * a + s.m01 * b + s.m02 * c;
}
template <class S> inline Sym3<S> sym_inverse(const Sym3<S>& s) {
    S a = s.m11 * s.m22 - s.m12 * s.m12;
    S b = s.m12 * s.m02 - s.m22 * s.m01;
    S c = s.m01 * s.m12 - s.m02 * s.m11;
    S inv_det = S(1) / (s.m00 * a + s.m01 * b + s.m02 * c);
    S m11 = s.m22 * s.m00 - s.m02 * s.m02;
    S m12 = s.m02 * s.m01 - s.m00 * s.m12;
    S m22 = s.m00 * s.m11 - s.m01 * s.m01;
    return {a * inv_det, b * inv_det, c * inv_det, m11 * inv_det, m12 * inv_det, m22 * inv_det};
}
// math/mod.rs:515-525 MatExt::inverse_or_zero
template <class S> inline Sym3<S> sym_inverse_or_zero(const Sym3<S>& s) {
    if (sym_determinant(s) == S(0)) return sym_zero<S>();
    return sym_inverse(s);
}
// math/mod.rs:527-544 MatExt::is_isotropic (approx abs_diff_ne!: |a-b| > eps)
template <class S> inline bool sym_is_isotropic(const Sym3<S>& s, S eps) {
    if (std::fabs(s.m00 - s.m11) > eps || std::fabs(s.m11 - s.m22) > eps) return false;
    return std::fabs(s.m01) < eps && std::fabs(s.m02) < eps && std::fabs(s.m12) < eps;
}
// mass_properties/components/computed.rs:663-668 ComputedAngularInertia::rotated (on the INVERSE tensor):
//   from_mat3_unchecked((R * inv) * R^T)
template <class S> inline Sym3<S> rotated_inverse_inertia(const Sym3<S>& inv_local, Q4<S> rotation) {
    M3<S> R = mat3_from_quat(rotation);
    return sym_from_mat3_unchecked(mmul(mmul(R, to_mat3(inv_local)), transpose(R)));
}

}  // namespace avo
