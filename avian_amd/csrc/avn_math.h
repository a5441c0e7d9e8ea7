// avn_math.h — device/host math for the MI355X physics step (gfx950).
//
// The arithmetic follows glam 0.30.8 / glam_matrix_extras 0.1.0 semantics that the reference relies on
// (SURVEY.md Appendix B), operation order included: the translation units that include this header are
// compiled with -ffp-contract=off so that results are IEEE-exact as written and comparable bit-for-bit
// with a scalar CPU evaluation of the same formulas.  f32 division and sqrt are hipcc's default
// correctly-rounded forms (no -ffast-math, no -fno-hip-fp32-correctly-rounded-divide-sqrt).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define AVN_HD __host__ __device__ __forceinline__

namespace avn {

template <class T> struct Vec4T;
template <> struct Vec4T<float> { using type = float4; };
template <> struct Vec4T<double> { using type = double4; };
template <class T> using Vec4 = typename Vec4T<T>::type;

template <class T> struct V3 { T x, y, z; };
template <class T> struct V2 { T x, y; };
template <class T> struct Q4 { T x, y, z, w; };
template <class T> struct Sym3 { T m00, m01, m02, m11, m12, m22; };
template <class T> struct M3 { V3<T> c0, c1, c2; };

template <class T> AVN_HD Vec4<T> make4(T x, T y, T z, T w) { Vec4<T> r; r.x = x; r.y = y; r.z = z; r.w = w; return r; }
template <class T> AVN_HD Vec4<T> make4(V3<T> v, T w) { return make4<T>(v.x, v.y, v.z, w); }
template <class T> AVN_HD V3<T> xyz(const Vec4<T>& v) { return V3<T>{v.x, v.y, v.z}; }
template <class T> AVN_HD Q4<T> quat(const Vec4<T>& v) { return Q4<T>{v.x, v.y, v.z, v.w}; }
template <class T> AVN_HD Vec4<T> make4(Q4<T> q) { return make4<T>(q.x, q.y, q.z, q.w); }

template <class T> AVN_HD V3<T> vzero() { return V3<T>{T(0), T(0), T(0)}; }
template <class T> AVN_HD V3<T> operator+(V3<T> a, V3<T> b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
template <class T> AVN_HD V3<T> operator-(V3<T> a, V3<T> b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
template <class T> AVN_HD V3<T> operator-(V3<T> a) { return {-a.x, -a.y, -a.z}; }
template <class T> AVN_HD V3<T> operator*(V3<T> a, T s) { return {a.x * s, a.y * s, a.z * s}; }
template <class T> AVN_HD V3<T> operator*(T s, V3<T> a) { return {s * a.x, s * a.y, s * a.z}; }
template <class T> AVN_HD V3<T> operator/(V3<T> a, T s) { return {a.x / s, a.y / s, a.z / s}; }
template <class T> AVN_HD V3<T> cmul(V3<T> a, V3<T> b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
template <class T> AVN_HD T dot(V3<T> a, V3<T> b) { return (a.x * b.x) + (a.y * b.y) + (a.z * b.z); }
template <class T> AVN_HD V3<T> cross(V3<T> a, V3<T> b) {
    return {a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y};
}
template <class T> AVN_HD T length_squared(V3<T> a) { return dot(a, a); }
AVN_HD float sqrt_t(float x) { return sqrtf(x); }
AVN_HD double sqrt_t(double x) { return sqrt(x); }
AVN_HD float fabs_t(float x) { return fabsf(x); }
AVN_HD double fabs_t(double x) { return fabs(x); }
AVN_HD bool finite_t(float x) { return fabsf(x) <= 3.402823466e+38f; }  // false for NaN and inf
AVN_HD bool finite_t(double x) { return fabs(x) <= 1.7976931348623157e+308; }
template <class T> AVN_HD T length(V3<T> a) { return sqrt_t(dot(a, a)); }
template <class T> AVN_HD T max_element(V3<T> a) { T m = a.x > a.y ? a.x : a.y; return m > a.z ? m : a.z; }
template <class T> AVN_HD bool is_finite(V3<T> a) { return finite_t(a.x) && finite_t(a.y) && finite_t(a.z); }
template <class T> AVN_HD V3<T> vmin(V3<T> a, V3<T> b) { return {a.x < b.x ? a.x : b.x, a.y < b.y ? a.y : b.y, a.z < b.z ? a.z : b.z}; }
template <class T> AVN_HD V3<T> vmax(V3<T> a, V3<T> b) { return {a.x > b.x ? a.x : b.x, a.y > b.y ? a.y : b.y, a.z > b.z ? a.z : b.z}; }
// Rust f32::max/min semantics (NaN-ignoring)
template <class T> AVN_HD T smax(T a, T b) { return (a > b || b != b) ? a : b; }
template <class T> AVN_HD T smin(T a, T b) { return (a < b || b != b) ? a : b; }
template <class T> struct Limits;
template <> struct Limits<float> { static constexpr float eps = 1.1920929e-7f; static constexpr float max = 3.402823466e+38f; };
template <> struct Limits<double> { static constexpr double eps = 2.220446049250313e-16; static constexpr double max = 1.7976931348623157e+308; };

// reference src/math/mod.rs:248-257
template <class T> AVN_HD T recip_or_zero(T x) { return (x != T(0) && finite_t(x)) ? T(1) / x : T(0); }
template <class T> AVN_HD V3<T> recip_or_zero(V3<T> v) { return {recip_or_zero(v.x), recip_or_zero(v.y), recip_or_zero(v.z)}; }

template <class T> AVN_HD bool try_normalize(V3<T> v, V3<T>& out) {
    T rcp = T(1) / length(v);
    if (finite_t(rcp) && rcp > T(0)) { out = v * rcp; return true; }
    return false;
}
template <class T> AVN_HD V3<T> clamp_length_max(V3<T> v, T mx) {
    T len_sq = length_squared(v);
    if (len_sq > mx * mx) return mx * (v / sqrt_t(len_sq));
    return v;
}
template <class T> AVN_HD V2<T> clamp_length_max(V2<T> v, T mx) {
    T len_sq = (v.x * v.x) + (v.y * v.y);
    if (len_sq > mx * mx) { T l = sqrt_t(len_sq); return {mx * (v.x / l), mx * (v.y / l)}; }
    return v;
}
AVN_HD bool signbit_t(float x) { return __builtin_signbit(x); }
AVN_HD bool signbit_t(double x) { return __builtin_signbit(x); }
template <class T> AVN_HD V3<T> any_orthonormal_vector(V3<T> v) {
    T sign = signbit_t(v.z) ? T(-1) : T(1);
    T a = T(-1) / (sign + v.z);
    T b = v.x * v.y * a;
    return {b, sign + v.y * v.y * a, -v.y};
}

// Deterministic sin/cos: Cody-Waite 3-term reduction by pi/2 + Cephes minimax polynomials on
// [-pi/4, pi/4], plain IEEE ops only (no libm / ocml call, no FMA): one result on every platform.
// The reference uses Rust's f32::sin_cos = "the platform libm"; this is that libm for this build.
AVN_HD void sin_cos_t(float a, float& s, float& c) {
    const float TWO_OVER_PI = 0.63661977236758134308f;
    const float P1 = 1.5703125f, P2 = 4.837512969970703125e-4f, P3 = 7.54978995489188216e-8f;
    float kf = __builtin_rintf(a * TWO_OVER_PI);
    float r = ((a - kf * P1) - kf * P2) - kf * P3;
    float z = r * r;
    float sp = ((-1.9515295891e-4f * z + 8.3321608736e-3f) * z - 1.6666654611e-1f) * z * r + r;
    float cp = ((2.443315711809948e-5f * z - 1.388731625493765e-3f) * z + 4.166664568298827e-2f) * z * z - 0.5f * z + 1.0f;
    float qf = kf - 4.0f * __builtin_floorf(kf * 0.25f);  // kf mod 4, exact for every finite kf (no float->int conversion: UB-free for huge arguments)
    int q = qf == 1.0f ? 1 : qf == 2.0f ? 2 : qf == 3.0f ? 3 : 0;
    if (q == 0) { s = sp; c = cp; }
    else if (q == 1) { s = cp; c = -sp; }
    else if (q == 2) { s = -sp; c = -cp; }
    else { s = -cp; c = sp; }
}
AVN_HD void sin_cos_t(double a, double& s, double& c) {
    const double TWO_OVER_PI = 6.36619772367581382433e-01;
    const double P1 = 1.57079632673412561417e+00, P2 = 6.07710050630396597660e-11, P3 = 2.02226624871116645580e-21;
    double kf = __builtin_rint(a * TWO_OVER_PI);
    double r = ((a - kf * P1) - kf * P2) - kf * P3;
    double z = r * r;
    double sp = (((((1.58962301576546568060e-10 * z - 2.50507477628578072866e-8) * z + 2.75573136213857245213e-6) * z
                   - 1.98412698295895385996e-4) * z + 8.33333333332211858878e-3) * z - 1.66666666666666307295e-1) * z * r + r;
    double cp = (((((-1.13585365213876817300e-11 * z + 2.08757008419747316778e-9) * z - 2.75573141792967388112e-7) * z
                   + 2.48015872888517045348e-5) * z - 1.38888888888730564116e-3) * z + 4.16666666666665929218e-2) * z * z
                - 0.5 * z + 1.0;
    double qf = kf - 4.0 * __builtin_floor(kf * 0.25);
    int q = qf == 1.0 ? 1 : qf == 2.0 ? 2 : qf == 3.0 ? 3 : 0;
    if (q == 0) { s = sp; c = cp; }
    else if (q == 1) { s = cp; c = -sp; }
    else if (q == 2) { s = -sp; c = -cp; }
    else { s = -cp; c = sp; }
}

template <class T> AVN_HD Q4<T> qidentity() { return {T(0), T(0), T(0), T(1)}; }
template <class T> AVN_HD Q4<T> from_scaled_axis(V3<T> v) {
    T len = length(v);
    if (len == T(0)) return qidentity<T>();
    V3<T> axis = v / len;
    T s, c;
    sin_cos_t(len * T(0.5), s, c);
    V3<T> a = axis * s;
    return {a.x, a.y, a.z, c};
}
// Deterministic asin (used by AngleLimit::compute_correction, dynamics/joints/mod.rs:427-472): the fdlibm e_asin.c
// rational approximation R(z) = p(z)/q(z) evaluated in T with plain IEEE ops, without fdlibm's high/low split of the
// square root: asin(x) = x + x R(x^2) for |x| < 0.5, pi/2 - 2 (s + s R(z)) with z = (1 - |x|)/2, s = sqrt(z) otherwise;
// |x| > 1 -> NaN like Rust's asin.  Like sin_cos_t it is "the platform libm" of this build (<= 2 ulp f32, tests).
template <class T> AVN_HD T asin_rational(T z) {
    T p = z * (T(1.66666666666666657415e-01) + z * (T(-3.25565818622400915405e-01) + z * (T(2.01212532134862925881e-01) +
          z * (T(-4.00555345006794114027e-02) + z * (T(7.91534994289814532176e-04) + z * T(3.47933107596021167570e-05))))));
    T q = T(1) + z * (T(-2.40339491173441421878e+00) + z * (T(2.02094576023350569471e+00) + z * (T(-6.88283971605453293030e-01) +
          z * T(7.70381505559019352791e-02))));
    return p / q;
}
template <class T> AVN_HD T asin_t(T x) {
    T ax = fabs_t(x);
    if (!(ax <= T(1))) return (x - x) / (x - x);  // NaN for |x| > 1 and for NaN
    if (ax < T(0.5)) return x + x * asin_rational<T>(x * x);
    T z = (T(1) - ax) * T(0.5);
    T s = sqrt_t(z);
    T r = T(1.57079632679489661923) - T(2) * (s + s * asin_rational<T>(z));
    return x < T(0) ? -r : r;
}
// glam Quat::from_axis_angle (axis must be unit)
template <class T> AVN_HD Q4<T> from_axis_angle(V3<T> axis, T angle) {
    T s, c;
    sin_cos_t(angle * T(0.5), s, c);
    V3<T> v = axis * s;
    return {v.x, v.y, v.z, c};
}
// glam Vec3::any_orthogonal_vector
template <class T> AVN_HD V3<T> any_orthogonal_vector(V3<T> v) {
    if (fabs_t(v.x) > fabs_t(v.y)) return {-v.z, T(0), v.x};
    return {T(0), v.z, -v.y};
}
// f32::clamp (min <= max assumed, like the reference's AngleLimit)
template <class T> AVN_HD T clamp_t(T x, T lo, T hi) { T r = x; if (r < lo) r = lo; if (r > hi) r = hi; return r; }
// glam Quat*Quat: f32 follows the SSE2 (rtm::quat_mul) association, f64 the scalar one.
AVN_HD Q4<float> qmul(Q4<float> l, Q4<float> r) {
    return {(l.w * r.x + l.x * r.w) + (l.y * r.z + -(l.z * r.y)),
            (l.w * r.y + -(l.x * r.z)) + (l.y * r.w + l.z * r.x),
            (l.w * r.z + l.x * r.y) + (-(l.y * r.x) + l.z * r.w),
            (l.w * r.w + -(l.x * r.x)) + (-(l.y * r.y) + -(l.z * r.z))};
}
AVN_HD Q4<double> qmul(Q4<double> l, Q4<double> r) {
    return {l.w * r.x + l.x * r.w + l.y * r.z - l.z * r.y,
            l.w * r.y - l.x * r.z + l.y * r.w + l.z * r.x,
            l.w * r.z + l.x * r.y - l.y * r.x + l.z * r.w,
            l.w * r.w - l.x * r.x - l.y * r.y - l.z * r.z};
}
template <class T> AVN_HD Q4<T> qinverse(Q4<T> q) { return {-q.x, -q.y, -q.z, q.w}; }
template <class T> AVN_HD V3<T> qrot(Q4<T> q, V3<T> v) {
    T w = q.w;
    V3<T> b{q.x, q.y, q.z};
    T b2 = dot(b, b);
    return (v * (w * w - b2) + b * (dot(v, b) * T(2))) + cross(b, v) * (w * T(2));
}
AVN_HD float qlength_squared(Q4<float> q) { return (q.x * q.x + q.z * q.z) + (q.y * q.y + q.w * q.w); }
AVN_HD double qlength_squared(Q4<double> q) { return q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w; }
// glam Quat::normalize = Vec4::normalize: SSE2 (f32) divides by the length, the scalar path (f64) multiplies by its reciprocal
AVN_HD Q4<float> qnormalize(Q4<float> q) { const float l = sqrt_t(qlength_squared(q)); return Q4<float>{q.x / l, q.y / l, q.z / l, q.w / l}; }
AVN_HD Q4<double> qnormalize(Q4<double> q) { const double r = 1.0 / sqrt_t(qlength_squared(q)); return Q4<double>{q.x * r, q.y * r, q.z * r, q.w * r}; }
// reference physics_transform/transform.rs:811-817
template <class T> AVN_HD Q4<T> fast_renormalize(Q4<T> q) {
    T k = T(0.5) * (T(3) - qlength_squared(q));
    return {q.x * k, q.y * k, q.z * k, q.w * k};
}

template <class T> AVN_HD M3<T> mat3_from_quat(Q4<T> r) {
    T x2 = r.x + r.x, y2 = r.y + r.y, z2 = r.z + r.z;
    T xx = r.x * x2, xy = r.x * y2, xz = r.x * z2;
    T yy = r.y * y2, yz = r.y * z2, zz = r.z * z2;
    T wx = r.w * x2, wy = r.w * y2, wz = r.w * z2;
    return {{T(1) - (yy + zz), xy + wz, xz - wy}, {xy - wz, T(1) - (xx + zz), yz + wx}, {xz + wy, yz - wx, T(1) - (xx + yy)}};
}
template <class T> AVN_HD V3<T> mmul(const M3<T>& m, V3<T> v) { return (m.c0 * v.x + m.c1 * v.y) + m.c2 * v.z; }
template <class T> AVN_HD V3<T> smul(const Sym3<T>& s, V3<T> v) {
    V3<T> c0{s.m00, s.m01, s.m02}, c1{s.m01, s.m11, s.m12}, c2{s.m02, s.m12, s.m22};
    return (c0 * v.x + c1 * v.y) + c2 * v.z;
}
template <class T> AVN_HD Sym3<T> sym_zero() { return {T(0), T(0), T(0), T(0), T(0), T(0)}; }
template <class T> AVN_HD bool sym_is_zero(const Sym3<T>& s) { return s.m00 == 0 && s.m01 == 0 && s.m02 == 0 && s.m11 == 0 && s.m12 == 0 && s.m22 == 0; }
template <class T> AVN_HD T sym_determinant(const Sym3<T>& s) {
    T a = s.m11 * s.m22 - s.m12 * s.m12;
    T b = s.m12 * s.m02 - s.m22 * s.m01;
    T c = s.m01 * s.m12 - s.m02 * s.m11;
    return s.m00 * a + s.m01 * b + s.m02 * c;
}
template <class T> AVN_HD Sym3<T> sym_inverse_or_zero(const Sym3<T>& s) {  // reference math/mod.rs:515-525
    if (sym_determinant(s) == T(0)) return sym_zero<T>();
    T a = s.m11 * s.m22 - s.m12 * s.m12;
    T b = s.m12 * s.m02 - s.m22 * s.m01;
    T c = s.m01 * s.m12 - s.m02 * s.m11;
    T inv_det = T(1) / (s.m00 * a + s.m01 * b + s.m02 * c);
    T m11 = s.m22 * s.m00 - s.m02 * s.m02;
    T m12 = s.m02 * s.m01 - s.m00 * s.m12;
    T m22 = s.m00 * s.m11 - s.m01 * s.m01;
    return {a * inv_det, b * inv_det, c * inv_det, m11 * inv_det, m12 * inv_det, m22 * inv_det};
}
template <class T> AVN_HD bool sym_is_isotropic(const Sym3<T>& s, T eps) {  // reference math/mod.rs:527-544
    if (fabs_t(s.m00 - s.m11) > eps || fabs_t(s.m11 - s.m22) > eps) return false;
    return fabs_t(s.m01) < eps && fabs_t(s.m02) < eps && fabs_t(s.m12) < eps;
}
// reference mass_properties/components/computed.rs:663-668: from_mat3_unchecked((R * inv) * R^T), upper triangle
template <class T> AVN_HD Sym3<T> rotated_inverse_inertia(const Sym3<T>& s, Q4<T> rotation) {
    M3<T> R = mat3_from_quat(rotation);
    // A = R * S (columns of S are (m00,m01,m02), (m01,m11,m12), (m02,m12,m22))
    V3<T> a0 = mmul(R, V3<T>{s.m00, s.m01, s.m02});
    V3<T> a1 = mmul(R, V3<T>{s.m01, s.m11, s.m12});
    V3<T> a2 = mmul(R, V3<T>{s.m02, s.m12, s.m22});
    M3<T> A{a0, a1, a2};
    // B = A * R^T ; column j of R^T is row j of R
    V3<T> b0 = mmul(A, V3<T>{R.c0.x, R.c1.x, R.c2.x});
    V3<T> b1 = mmul(A, V3<T>{R.c0.y, R.c1.y, R.c2.y});
    V3<T> b2 = mmul(A, V3<T>{R.c0.z, R.c1.z, R.c2.z});
    return {b0.x, b1.x, b2.x, b1.y, b2.y, b2.z};
}
template <class T> AVN_HD void lock_rotation_axes(Sym3<T>& t, uint32_t locked) {  // reference solver_body/mod.rs:400-414
    if (locked & 4u) { t.m00 = 0; t.m01 = 0; t.m02 = 0; }
    if (locked & 2u) { t.m01 = 0; t.m11 = 0; t.m12 = 0; }
    if (locked & 1u) { t.m02 = 0; t.m12 = 0; t.m22 = 0; }
}
template <class T> AVN_HD V3<T> effective_inv_mass(T inv_mass, uint32_t flags) {  // reference solver_body/mod.rs:437-451
    V3<T> m{inv_mass, inv_mass, inv_mass};
    if (flags & 0x20u) m.x = 0;
    if (flags & 0x10u) m.y = 0;
    if (flags & 0x08u) m.z = 0;
    return m;
}

// reinterpret helpers for packing flag words into the w lane of a Vec4
AVN_HD float bits_to_scalar(uint32_t u, float) { return __builtin_bit_cast(float, u); }
AVN_HD double bits_to_scalar(uint32_t u, double) { return __builtin_bit_cast(double, (uint64_t)u); }
AVN_HD uint32_t scalar_to_bits(float f) { return __builtin_bit_cast(uint32_t, f); }
AVN_HD uint32_t scalar_to_bits(double d) { return (uint32_t)__builtin_bit_cast(uint64_t, d); }

}  // namespace avn
