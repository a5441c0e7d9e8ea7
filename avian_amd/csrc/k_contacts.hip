// k_contacts.hip — TGS-Soft contact solver kernels (one thread per contact manifold).
//
// Parallel unit: a graph colour (reference constraint_graph.rs:36-48): within colours 0..22 no two
// manifolds share a non-static body, so one launch per colour solves all its manifolds concurrently with
// plain loads/stores (no atomics).  The overflow colour (23) is solved first and serially, in list order,
// exactly like the reference (solver/plugin.rs:461-467).  The points of one manifold are solved
// sequentially by the owning thread (Gauss-Seidel order inside ContactConstraint::solve).
//
// Memory: constraint records are [point][manifold] Vec4 arrays, so the 64 lanes of a wave (64 consecutive
// manifolds of one colour) read 1 KiB contiguous per load instruction; body state is gathered with six
// aligned Vec4 loads per body (L2 / Infinity-Cache resident) and scattered back with two.
// Bound: HBM stream of the constraint records (248+88P B read/written per manifold pass).
//
// Reference functions replaced (paths relative to /root/reference/src/dynamics/solver):
//   k_prepare_contact_constraints  plugin.rs:363-448, contact/mod.rs:110-220,427-449,
//                                  contact/normal_part.rs:39-112, contact/tangent_part.rs:35-151
//   k_warm_start                   plugin.rs:453-515, contact/mod.rs:223-264
//   k_solve_contacts<USE_BIAS>     plugin.rs:531-619, contact/mod.rs:267-354,
//                                  normal_part.rs:116-166, tangent_part.rs:155-244
//   k_solve_restitution            plugin.rs:630-718, contact/mod.rs:358-407
//   k_store_contact_impulses       plugin.rs:722-755
#include <cstdlib>

#include "avn_kernels.h"
#include "avn_body_ops.h"

namespace avn {

#define CONTACT_THREADS 64

template <class T> struct BodyRef {
    V3<T> v, om, dp;
    Q4<T> dq;
    V3<T> inv_mass;  // effective_inv_mass()
    Sym3<T> I;       // effective_inv_angular_inertia()
    T lin_w, ang_w;  // pass-through w lanes
};

// Where the SolverBody / SolverBodyInertia records of a pass live: the world's HBM arrays (paired slots: record i at
// base[2 i], STRIDE = 2) or an island's block staged in LDS (STRIDE = 1, indices local to the block; k_island_substeps).
template <class T> struct BodyView { Vec4<T> *lin, *ang, *dp, *dq, *sia, *sib; };
template <class T> __device__ __forceinline__ BodyView<T> global_bodies(const DW<T>& w) { return {w.sb_lin.p, w.sb_ang.p, w.sb_dp.p, w.sb_dq.p, w.si_a.p, w.si_b.p}; }

// Fetch SolverBody + SolverBodyInertia, substituting DUMMY for a missing body and a DUMMY inertia for a
// dominant one (solver/plugin.rs:491-512).
// Agent-scope accesses of one record (k_overflow_flow: lanes of different workgroups hand a body's velocities to each other inside
// ONE launch; the XCDs' L2s are not coherent with each other, so these go to the coherence point component by component).
// 16-byte agent-scope (sc1: write-through / L1-bypassing) accesses of one record through a buffer descriptor -- one instruction per
// 16 bytes instead of four dword atomics (a dword sc1 store is one fabric write each), and, unlike inline asm, visible to the compiler's
// vmcnt bookkeeping.  `base` is wave-uniform (the world's array), `byte_off` the record's offset (< 2^31: 32 B x 2^26 bodies).
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define AVN_BUF_AUX_SC1 16
template <class V> __device__ __forceinline__ __amdgpu_buffer_rsrc_t rec_rsrc(V* base) { return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0x7FFFFFFF, 0x00020000); }
__device__ __forceinline__ Vec4<float> ld_rec_agent(Vec4<float>* base, size_t idx) {
    const u32x4 r = __builtin_amdgcn_raw_buffer_load_b128(rec_rsrc(base), (int)(idx * sizeof(Vec4<float>)), 0, AVN_BUF_AUX_SC1);
    return make4<float>(__uint_as_float(r.x), __uint_as_float(r.y), __uint_as_float(r.z), __uint_as_float(r.w));
}
__device__ __forceinline__ Vec4<double> ld_rec_agent(Vec4<double>* base, size_t idx) {
    const __amdgpu_buffer_rsrc_t rs = rec_rsrc(base);
    const u32x4 a = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(idx * sizeof(Vec4<double>)), 0, AVN_BUF_AUX_SC1);
    const u32x4 b = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(idx * sizeof(Vec4<double>) + 16), 0, AVN_BUF_AUX_SC1);
    return make4<double>(__hiloint2double((int)a.y, (int)a.x), __hiloint2double((int)a.w, (int)a.z), __hiloint2double((int)b.y, (int)b.x), __hiloint2double((int)b.w, (int)b.z));
}
__device__ __forceinline__ void st_rec_agent(Vec4<float>* base, size_t idx, Vec4<float> v) {
    __builtin_amdgcn_raw_buffer_store_b128(u32x4{__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)}, rec_rsrc(base), (int)(idx * sizeof(Vec4<float>)), 0,
                                           AVN_BUF_AUX_SC1);
}
__device__ __forceinline__ void st_rec_agent(Vec4<double>* base, size_t idx, Vec4<double> v) {
    const __amdgpu_buffer_rsrc_t rs = rec_rsrc(base);
    __builtin_amdgcn_raw_buffer_store_b128(u32x4{(uint32_t)__double2loint(v.x), (uint32_t)__double2hiint(v.x), (uint32_t)__double2loint(v.y), (uint32_t)__double2hiint(v.y)}, rs,
                                           (int)(idx * sizeof(Vec4<double>)), 0, AVN_BUF_AUX_SC1);
    __builtin_amdgcn_raw_buffer_store_b128(u32x4{(uint32_t)__double2loint(v.z), (uint32_t)__double2hiint(v.z), (uint32_t)__double2loint(v.w), (uint32_t)__double2hiint(v.w)}, rs,
                                           (int)(idx * sizeof(Vec4<double>) + 16), 0, AVN_BUF_AUX_SC1);
}
// the DUMMY substitutions on the six records of a body as they lie in memory (by value: plain registers for the compiler)
template <class T, bool WITH_DELTA> __device__ __forceinline__ void body_from_recs(Vec4<T> l, Vec4<T> a, Vec4<T> dp, Vec4<T> dq, Vec4<T> sa, Vec4<T> sb, bool no_body, bool dummy_inertia, BodyRef<T>& b) {
    // branch-free selects (v_cndmask): nothing for the compiler to sink the loads into
    const T z = T(0);
    b.v = V3<T>{no_body ? z : l.x, no_body ? z : l.y, no_body ? z : l.z};
    b.om = V3<T>{no_body ? z : a.x, no_body ? z : a.y, no_body ? z : a.z};
    b.lin_w = no_body ? z : l.w; b.ang_w = no_body ? z : a.w;
    const bool nd = no_body || !WITH_DELTA;
    b.dp = V3<T>{nd ? z : dp.x, nd ? z : dp.y, nd ? z : dp.z};
    b.dq = Q4<T>{nd ? z : dq.x, nd ? z : dq.y, nd ? z : dq.z, nd ? T(1) : dq.w};
    const bool ni = no_body || dummy_inertia;
    V3<T> em = effective_inv_mass<T>(sa.x, scalar_to_bits(sb.w));
    b.inv_mass = V3<T>{ni ? z : em.x, ni ? z : em.y, ni ? z : em.z};
    b.I = Sym3<T>{ni ? z : sa.y, ni ? z : sa.z, ni ? z : sa.w, ni ? z : sb.x, ni ? z : sb.y, ni ? z : sb.z};
}
template <class T, bool WITH_DELTA, int STRIDE, bool COH = false>
__device__ __forceinline__ void load_body(const BodyView<T>& bv, int idx, bool no_body, bool dummy_inertia, BodyRef<T>& b) {
    // The six records are fetched UNCONDITIONALLY (every body index is a valid row; rows of bodies without a SolverBody
    // hold DUMMY values) and the DUMMY substitution is a select afterwards: no load waits on the constraint's flag word,
    // so the kernel has two dependent memory levels (headers + point records | body gathers) instead of three.
    const size_t o = (size_t)idx * STRIDE;
    Vec4<T> l, a;
    if (COH) { l = ld_rec_agent(bv.lin, o); a = ld_rec_agent(bv.ang, o); }   // (the velocities are the only records a contact pass writes)
    else { l = bv.lin[o]; a = bv.ang[o]; }
    Vec4<T> dp = make4<T>(0, 0, 0, 0), dq = make4<T>(0, 0, 0, 1);
    if (WITH_DELTA) { dp = bv.dp[o]; dq = bv.dq[o]; }
    Vec4<T> sa = bv.sia[o], sb = bv.sib[o];
    body_from_recs<T, WITH_DELTA>(l, a, dp, dq, sa, sb, no_body, dummy_inertia, b);
}
template <class T, int STRIDE, bool COH = false> __device__ __forceinline__ void store_body(const BodyView<T>& bv, int idx, bool no_body, const BodyRef<T>& b) {
    if (no_body) return;  // writes to a DUMMY body are discarded
    if (COH) { st_rec_agent(bv.lin, (size_t)idx * STRIDE, make4<T>(b.v, b.lin_w)); st_rec_agent(bv.ang, (size_t)idx * STRIDE, make4<T>(b.om, b.ang_w)); return; }
    bv.lin[(size_t)idx * STRIDE] = make4<T>(b.v, b.lin_w);
    bv.ang[(size_t)idx * STRIDE] = make4<T>(b.om, b.ang_w);
}
template <class T> __device__ __forceinline__ void apply_impulse(BodyRef<T>& b1, BodyRef<T>& b2, V3<T> imp, V3<T> r1, V3<T> r2) {
    b1.v = b1.v - cmul(imp, b1.inv_mass);
    b1.om = b1.om - smul(b1.I, cross(r1, imp));
    b2.v = b2.v + cmul(imp, b2.inv_mass);
    b2.om = b2.om + smul(b2.I, cross(r2, imp));
}
template <class T> __device__ __forceinline__ V3<T> velocity_at_point(const BodyRef<T>& b, V3<T> p) { return b.v + cross(b.om, p); }

// ------------------------------------------------------------------------------------------------------
// ROWS (handle mode, round 4): the manifold's ContactGraph side is read straight from the contact table through GraphColor::manifold_handles
// (plugin.rs:389-398) -- what k_gather_manifolds used to copy into the colour-major arrays first (64 us + 26 MB written and read back per
// settled cfg2 step).  The headers the solve passes read every substep (m_bodies, m_n, m_tv, m_meta) are still laid out colour-major here;
// the anchors / penetrations / warm-start impulses are only ever read by this kernel and are no longer copied.
template <class T, bool ROWS>
__global__ __launch_bounds__(256) void k_prepare_contact_constraints(DW<T> w, StepParams<T> p, RowsView<T> rv) {
    uint32_t m = blockIdx.x * 256 + threadIdx.x;
    bool generated = false;
    // handle mode: the live manifold count is the device's (the colour offsets k_pg_build_handles / the upload left); DW::n_manifolds may be the
    // host's UPPER BOUND of it -- the device closed loop launches this kernel before it has read the step's counts back
    uint32_t n_live = w.n_manifolds;
    if (ROWS) n_live = min(n_live, w.color_offsets[AVN_GRAPH_COLOR_COUNT]);
    if (m < n_live) {
        uint32_t row = 0;
        if (ROWS) {
            row = rv.handles[m];
            const uint4 meta = rv.meta[row];
            const uint32_t npr = (meta.w & 0xFFu) ? ((meta.w >> 8) & 0xFFu) : 0u;
            w.m_bodies[m] = make_int2((int)rv.col_info[meta.x].y, (int)rv.col_info[meta.y].y);
            w.m_n[m] = rv.rows[(size_t)row * AVN_CT_ROW_V4];
            w.m_tv[m] = rv.rows[(size_t)row * AVN_CT_ROW_V4 + 1];
            w.m_meta[m] = npr | (((meta.z & AVN_CP_GENERATE_CONSTRAINTS) ? (uint32_t)AVN_MANIFOLD_GENERATES_CONSTRAINTS : 0u) << 8);
        }
        uint32_t mm = w.m_meta[m];
        uint32_t np = mm & 7u, mflags = mm >> 8;
        int2 b = w.m_bodies[m];
        uint32_t meta1 = w.bmeta[b.x], meta2 = w.bmeta[b.y];
        bool skip = !(mflags & AVN_MANIFOLD_GENERATES_CONSTRAINTS) || !meta_active(meta1) || !meta_active(meta2) ||
                    (meta_rb_type(meta1) != AVN_RB_DYNAMIC && meta_rb_type(meta2) != AVN_RB_DYNAMIC) || np == 0;
        if (skip) {
            w.c_h1[m] = make4<T>(0, 0, 0, bits_to_scalar(0u, T(0)));
            w.c_reldom[m] = 0;
        } else {
            bool nobody1 = !meta_has_solver_body(meta1), nobody2 = !meta_has_solver_body(meta2);
            // SolverBodyInertia (DUMMY rows were written by k_prepare_solver_bodies for bodies without a SolverBody)
            Vec4<T> sa1 = w.si_a[b.x], sb1 = w.si_b[b.x], sa2 = w.si_a[b.y], sb2 = w.si_b[b.y];
            uint32_t if1 = scalar_to_bits(sb1.w), if2 = scalar_to_bits(sb2.w);
            int dom1 = (int)(int16_t)(if1 >> 16), dom2 = (int)(int16_t)(if2 >> 16);
            int relative_dominance = dom1 - dom2;
            V3<T> inv_mass1 = effective_inv_mass<T>(sa1.x, if1), inv_mass2 = effective_inv_mass<T>(sa2.x, if2);
            Sym3<T> i1{sa1.y, sa1.z, sa1.w, sb1.x, sb1.y, sb1.z}, i2{sa2.y, sa2.z, sa2.w, sb2.x, sb2.y, sb2.z};
            if (relative_dominance > 0) { inv_mass1 = vzero<T>(); i1 = sym_zero<T>(); }
            else if (relative_dominance < 0) { inv_mass2 = vzero<T>(); i2 = sym_zero<T>(); }
            SoftCoef<T> soft = relative_dominance != 0 ? p.soft_non_dynamic : p.soft_dynamic;
            (void)soft;  // coefficients are step constants: only the choice is stored per manifold
            V3<T> w_sum = inv_mass1 + inv_mass2;
            Vec4<T> n4 = w.m_n[m];
            V3<T> normal = xyz<T>(n4);
            T friction = n4.w;
            // compute_tangent_directions (contact/mod.rs:427-449) from the LinearVelocity COMPONENTS
            V3<T> force_direction = -normal;
            V3<T> relative_velocity = xyz<T>(w.lvel[b.x]) - xyz<T>(w.lvel[b.y]);
            V3<T> tangent_velocity = relative_velocity - force_direction * dot(force_direction, relative_velocity);
            V3<T> t0;
            if (!try_normalize(tangent_velocity, t0)) t0 = any_orthonormal_vector(force_direction);
            V3<T> t1 = cross(force_direction, t0);
            bool warm = p.match_contacts != 0;
            bool has_tangent = friction > T(0);
            uint32_t S = w.m_stride;
            // all twelve point records up front (slots >= np are allocated, merely unused): one memory level for the
            // whole manifold instead of one per point iteration
            Vec4<T> pa1[AVN_MAX_MANIFOLD_POINTS], pa2[AVN_MAX_MANIFOLD_POINTS], pww[AVN_MAX_MANIFOLD_POINTS];
#pragma unroll
            for (uint32_t k = 0; k < AVN_MAX_MANIFOLD_POINTS; ++k) {
                if (ROWS) { const Vec4<T>* __restrict__ r = rv.rows + (size_t)row * AVN_CT_ROW_V4; pa1[k] = r[2 + k]; pa2[k] = r[6 + k]; pww[k] = r[10 + k]; }
                else { uint32_t s = k * S + m; pa1[k] = w.mp_a1[s]; pa2[k] = w.mp_a2[s]; pww[k] = w.mp_w[s]; }
            }
#pragma unroll
            for (uint32_t k = 0; k < AVN_MAX_MANIFOLD_POINTS; ++k) {
                if (k >= np) break;
                uint32_t s = k * S + m;
                Vec4<T> a1 = pa1[k], a2 = pa2[k], ww = pww[k];
                V3<T> r1 = xyz<T>(a1), r2 = xyz<T>(a2);
                T penetration = a1.w, normal_speed = a2.w;
                // ContactNormalPart::generate
                V3<T> r1_cross_n = cross(r1, normal), r2_cross_n = cross(r2, normal);
                T k_linear = dot(normal, cmul(w_sum, normal));
                T kk = k_linear + dot(r1_cross_n, smul(i1, r1_cross_n)) + dot(r2_cross_n, smul(i2, r2_cross_n));
                T eff_mass = recip_or_zero(kk);
                T ni = warm ? ww.x : T(0);
                T k0 = 0, k1 = 0, k2 = 0, tx = 0, ty = 0;
                if (has_tangent) {  // ContactTangentPart::generate
                    V3<T> rt11 = cross(r1, t0), rt12 = cross(r2, t0), rt21 = cross(r1, t1), rt22 = cross(r2, t1);
                    V3<T> i1_rt11 = smul(i1, rt11), i2_rt12 = smul(i2, rt12), i1_rt21 = smul(i1, rt21), i2_rt22 = smul(i2, rt22);
                    T k_linear1 = dot(t0, cmul(w_sum, t0));
                    T k_linear2 = dot(t1, cmul(w_sum, t1));
                    k0 = k_linear1 + dot(rt11, i1_rt11) + dot(rt12, i2_rt12);
                    k1 = k_linear2 + dot(rt21, i1_rt21) + dot(rt22, i2_rt22);
                    k2 = T(2) * (dot(rt11, i1_rt21) + dot(rt12, i2_rt22));
                    if (warm) { tx = ww.y; ty = ww.z; }
                }
                T initial_separation = -penetration - dot(r2 - r1, normal);
                w.c_pa[s] = make4<T>(r1, initial_separation);
                w.c_pb[s] = make4<T>(r2, eff_mass);
                w.c_pc[s] = make4<T>(k0, k1, k2, normal_speed);
                w.c_pd[s] = make4<T>(ni, T(0), tx, ty);
            }
            uint32_t cm = np;
            if (relative_dominance > 0) cm |= AVN_CM_DOM1;
            if (relative_dominance < 0) cm |= AVN_CM_DOM2;
            if (relative_dominance != 0) cm |= AVN_CM_SOFT_ND;
            if (has_tangent) cm |= AVN_CM_TANGENT;
            if (nobody1) cm |= AVN_CM_NOBODY1;
            if (nobody2) cm |= AVN_CM_NOBODY2;
            w.c_h1[m] = make4<T>(t0, bits_to_scalar(cm, T(0)));
            w.c_reldom[m] = (int16_t)relative_dominance;
            generated = true;
        }
    }
    unsigned long long bal = __ballot(generated);
    if ((threadIdx.x & 63) == 0 && bal) atomicAdd(w.constraint_count, (uint32_t)__popcll(bal));
}

// ------------------------------------------------------------------------------------------------------
// Warm start, BODY-centric (reference plugin.rs:453-515, contact/mod.rs:223-264), fused with integrate_velocities.
//
// A warm-start impulse does not depend on the velocities: p = c (l_n n + l_t.x t1 + l_t.y t2) per point, applied as
// v1 -= p * w1, om1 -= I1 (r1 x p), v2 += p * w2, om2 += I2 (r2 x p).  The only order-dependent part is the sequence of
// floating-point additions on each body, and that sequence is the body's incident manifolds in solve order (overflow
// colour first, then colours 0..22) with the points of a manifold in order.  So instead of one launch per colour
// (15 launches whose ~5 us latency floors dominate at 45k manifolds each), ONE launch does it for all colours: one lane
// per body walks the body's incidence list (CSR, solve order) and applies the SAME expressions as the manifold-centric
// form, in the same order -- bit-identical to the colour-by-colour result.  Neighbouring lanes are neighbouring bodies,
// whose j-th incident manifolds are neighbours in the colour-major arrays, so the record gathers of a wave coalesce; the
// next entry's ten records are in flight while the current one is applied.  The lane first runs integrate_velocities
// (the system that precedes warm start in the SubstepSchedule) on its body.
// (Round 4, tried on this form: a ring of 2 / 3 / 4 / 6 record sets in flight (all slots loaded up front, loads made unconditional so that the
//  vmcnt bookkeeping stays exact -- checked in the ISA): cfg2 2 481 / 2 458 / 2 492 / 2 434 substeps/s against 2 494 on the same box, the closed
//  loop's launch unchanged at 61-67 us.  A per-manifold block of exactly the records this kernel reads (closed loop: 191 -> 78 MB fetched per
//  launch, PMC): 61 -> 58 us, paid back by the extra writes of the constraint generation.  So it is neither round trips nor bytes: frozen cfg2
//  moves its 217 MB at 6 TB/s already, and in the closed loop a wave issues 5 500 VALU instructions -- the whole `apply` for each of the 21
//  populated colours, for whichever of its 64 bodies has a manifold there -- at 1.5 waves per SIMD.  What helped there is the four-lanes-per-body
//  form below; neither of the two was kept.)
#define WS_THREADS 64
template <class T> struct WarmRecords { Vec4<T> h1, h0, pr[AVN_MAX_MANIFOLD_POINTS], pd[AVN_MAX_MANIFOLD_POINTS]; };
template <class T> __device__ __forceinline__ void warm_fetch(const DW<T>& w, uint32_t ent, WarmRecords<T>& r) {
    const uint32_t m = ent & 0x7FFFFFFFu;
    const Vec4<T>* __restrict__ anchors = (ent >> 31) ? w.c_pb : w.c_pa;  // this body's side of the manifold
    r.h1 = w.c_h1[m];
    r.h0 = w.m_n[m];
#pragma unroll
    for (uint32_t k = 0; k < AVN_MAX_MANIFOLD_POINTS; ++k) { uint32_t s = k * w.m_stride + m; r.pr[k] = anchors[s]; r.pd[k] = w.c_pd[s]; }
}
template <class T, bool FUSE_INTEGRATE>
__device__ __forceinline__ void body_warm_start_one(const DW<T>& w, const StepParams<T>& p, uint32_t body) {
    const uint32_t sbf = w.sb_flags[body];
    if (sbf & AVN_SBF_NO_SOLVER_BODY) return;
    // the 23 colour slots of this body: coalesced across the wave (neighbouring lanes = neighbouring bodies), issued up front
    uint32_t slot0 = w.inc_slot[body];
    uint32_t e = w.inc_off[body];
    const uint32_t end = w.inc_off[body + 1];
    Vec4<T> l4 = w.sb_lin[body], a4 = w.sb_ang[body];
    Vec4<T> sa = w.si_a[body], sb = w.si_b[body];
    V3<T> v = xyz<T>(l4), om = xyz<T>(a4);
    bool touched = false;
    if (FUSE_INTEGRATE) touched = integrate_velocities_one<T>(w, p, body, sbf, v, om, &w.sb_dq[body]);
    const V3<T> em = effective_inv_mass<T>(sa.x, scalar_to_bits(sb.w));
    const T coeff = p.warm_start_coefficient;
    const T z = T(0);
    auto apply = [&](const WarmRecords<T>& r, uint32_t ent) {
        const uint32_t side = ent >> 31;
        const uint32_t cm = scalar_to_bits(r.h1.w);
        uint32_t np = cm & 7u;
        if (cm & (side ? AVN_CM_NOBODY2 : AVN_CM_NOBODY1)) np = 0;  // (stale incidence: the body lost its SolverBody)
        if (!np) return;
        // SolverBodyInertia, or DUMMY for the dominant body (plugin.rs:508-512)
        const bool ni = cm & (side ? AVN_CM_DOM2 : AVN_CM_DOM1);
        const V3<T> inv_mass{ni ? z : em.x, ni ? z : em.y, ni ? z : em.z};
        const Sym3<T> I{ni ? z : sa.y, ni ? z : sa.z, ni ? z : sa.w, ni ? z : sb.x, ni ? z : sb.y, ni ? z : sb.z};
        const V3<T> normal = xyz<T>(r.h0);
        const V3<T> t0 = xyz<T>(r.h1), t1 = cross(t0, normal);  // tangent_directions(), contact/mod.rs:411-421
#pragma unroll
        for (uint32_t k = 0; k < AVN_MAX_MANIFOLD_POINTS; ++k) {
            if (k < np) {
                const V3<T> rr = xyz<T>(r.pr[k]);
                const Vec4<T> d = r.pd[k];
                const T tx = (cm & AVN_CM_TANGENT) ? d.z : T(0), ty = (cm & AVN_CM_TANGENT) ? d.w : T(0);
                const V3<T> imp = coeff * ((d.x * normal + tx * t0) + ty * t1);
                const V3<T> dv = cmul(imp, inv_mass);
                const V3<T> dw = smul(I, cross(rr, imp));
                if (side) { v = v + dv; om = om + dw; }   // body2: v += p * w2, om += I2 (r2 x p)
                else { v = v - dv; om = om - dw; }         // body1: v -= p * w1, om -= I1 (r1 x p)
            }
        }
        touched = true;
    };
    WarmRecords<T> cur, nxt;
    // (1) the overflow colour's entries, in list order (solved FIRST, plugin.rs:461-467); usually none
    for (; e < end; ++e) {
        uint32_t ent = w.inc_ent[e];
        warm_fetch<T>(w, ent, cur);
        apply(cur, ent);
    }
    // (2) colours 0..22 through the slot table; the next colour's ten records are in flight while the current one is applied
    bool cur_valid = false;
    uint32_t ent_cur = 0, slot_next = slot0;
    for (uint32_t c = 0; c < AVN_COLOR_OVERFLOW_INDEX; ++c) {
        const uint32_t sc = slot_next;
        slot_next = c + 1 < AVN_COLOR_OVERFLOW_INDEX ? w.inc_slot[(size_t)(c + 1) * w.inc_stride + body] : 0xFFFFFFFFu;
        const bool valid = sc != 0xFFFFFFFFu;
        if (valid) warm_fetch<T>(w, sc, nxt);
        if (cur_valid) apply(cur, ent_cur);
        cur_valid = valid;
        if (valid) { cur = nxt; ent_cur = sc; }
    }
    if (cur_valid) apply(cur, ent_cur);
    if (touched) {
        w.sb_lin[body] = make4<T>(v, l4.w);
        w.sb_ang[body] = make4<T>(om, a4.w);
    }
}
template <class T, bool FUSE_INTEGRATE>
__global__ __launch_bounds__(WS_THREADS) void k_body_warm_start(DW<T> w, StepParams<T> p) {
    const uint32_t body = xcd_block(blockIdx.x, gridDim.x) * WS_THREADS + threadIdx.x;
    if (body >= w.n_bodies) return;
    if (!body_in_group(w, body)) return;
    body_warm_start_one<T, FUSE_INTEGRATE>(w, p, body);
}

// The same launch with FOUR LANES PER BODY (round 4).  PMC on the lane-per-body form (settled cfg2 closed loop): 5 500 VALU instructions per wave -- a wave
// walks all 21 populated colours and runs the whole `apply` for every one of them, for whichever of its 64 bodies has a manifold there -- on 1 568 waves,
// 1.5 per SIMD, which spend 37 % of their cycles in s_waitcnt and 35 % in issue stalls; neither fewer bytes (a block layout: 191 -> 78 MB fetched, 61 -> 58 us)
// nor more loads in flight (a ring of record sets) moved it.  Here lane q of a body's quad owns the colours c = q (mod 4): it fetches the records and computes the
// point contributions (-p w1 | +p w2, -I1 (r1 x p) | +I2 (r2 x p): they do not depend on the velocities) of ITS colours, six iterations instead of 23; then the four
// lanes add the contributions of colours 4i, 4i+1, 4i+2, 4i+3 to their (identical) copies of v and omega IN THAT ORDER, points in order, through quad broadcasts.
// x - y is x + (-y) bit for bit, and an absent point contributes -0.0 (x + -0.0 == x for every x, -0.0 and NaN included), so the additions on a body are the
// lane-per-body form's sequence: same bits.  Four times the waves, each a quarter as long: the stalls of one hide behind the others.
template <int J> __device__ __forceinline__ float quad_bcast(float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), J * 0x55, 0xF, 0xF, false));   // quad_perm:[J,J,J,J]
}
template <int J> __device__ __forceinline__ double quad_bcast(double x) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), J * 0x55, 0xF, 0xF, false), hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), J * 0x55, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}
template <class T> struct QuadDelta { T d[AVN_MAX_MANIFOLD_POINTS][6]; };   // per point: the additions to (v.xyz, om.xyz)
template <int J, class T> __device__ __forceinline__ void quad_accumulate(const QuadDelta<T>& q, V3<T>& v, V3<T>& om) {
#pragma unroll
    for (uint32_t k = 0; k < AVN_MAX_MANIFOLD_POINTS; ++k) {
        v.x = v.x + quad_bcast<J>(q.d[k][0]); v.y = v.y + quad_bcast<J>(q.d[k][1]); v.z = v.z + quad_bcast<J>(q.d[k][2]);
        om.x = om.x + quad_bcast<J>(q.d[k][3]); om.y = om.y + quad_bcast<J>(q.d[k][4]); om.z = om.z + quad_bcast<J>(q.d[k][5]);
    }
}
#define WSQ_THREADS 256
template <class T, bool FUSE_INTEGRATE>
__global__ __launch_bounds__(WSQ_THREADS) void k_body_warm_start_quad(DW<T> w, StepParams<T> p) {
    const uint32_t gid = xcd_block(blockIdx.x, gridDim.x) * WSQ_THREADS + threadIdx.x;
    const uint32_t body = gid >> 2, q = gid & 3u;
    // (every exit below depends on the body only: the four lanes of a quad leave together, so the quad broadcasts always see all four)
    if (body >= w.n_bodies) return;
    if (!body_in_group(w, body)) return;
    const uint32_t sbf = w.sb_flags[body];
    if (sbf & AVN_SBF_NO_SOLVER_BODY) return;
    constexpr uint32_t NONE = 0xFFFFFFFFu, NC = AVN_COLOR_OVERFLOW_INDEX;
    // which colours hold a manifold of this body: lane q looks at the slots of c = q (mod 4), the quad ORs the four partial masks
    uint32_t mask = 0u;
#pragma unroll
    for (uint32_t i = 0; i < (NC + 3u) / 4u; ++i) {
        const uint32_t c = 4u * i + q;
        if (c < NC && w.inc_slot[(size_t)c * w.inc_stride + body] != NONE) mask |= 1u << c;
    }
    mask |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)mask, 0xB1, 0xF, 0xF, false);   // quad_perm:[1,0,3,2]
    mask |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)mask, 0x4E, 0xF, 0xF, false);   // quad_perm:[2,3,0,1]
    uint32_t e = w.inc_off[body];
    const uint32_t end = w.inc_off[body + 1];
    Vec4<T> l4 = w.sb_lin[body], a4 = w.sb_ang[body];
    const Vec4<T> sa = w.si_a[body], sb = w.si_b[body];
    V3<T> v = xyz<T>(l4), om = xyz<T>(a4);
    if (FUSE_INTEGRATE) (void)integrate_velocities_one<T>(w, p, body, sbf, v, om, &w.sb_dq[body]);   // (the same arithmetic on all four lanes)
    const V3<T> em = effective_inv_mass<T>(sa.x, scalar_to_bits(sb.w));
    const T coeff = p.warm_start_coefficient;
    const T z = T(0), nz = -T(0);
    // the contributions of one entry: what the lane-per-body form adds, as signed additions; absent points are -0.0
    auto contributions = [&](const WarmRecords<T>& r, uint32_t ent, bool present, QuadDelta<T>& out) {
        const uint32_t side = ent >> 31;
        const uint32_t cm = scalar_to_bits(r.h1.w);
        uint32_t np = present ? (cm & 7u) : 0u;
        if (cm & (side ? AVN_CM_NOBODY2 : AVN_CM_NOBODY1)) np = 0;  // (stale incidence: the body lost its SolverBody)
        const bool ni = cm & (side ? AVN_CM_DOM2 : AVN_CM_DOM1);
        const V3<T> inv_mass{ni ? z : em.x, ni ? z : em.y, ni ? z : em.z};
        const Sym3<T> I{ni ? z : sa.y, ni ? z : sa.z, ni ? z : sa.w, ni ? z : sb.x, ni ? z : sb.y, ni ? z : sb.z};
        const V3<T> normal = xyz<T>(r.h0);
        const V3<T> t0 = xyz<T>(r.h1), t1 = cross(t0, normal);
#pragma unroll
        for (uint32_t k = 0; k < AVN_MAX_MANIFOLD_POINTS; ++k) {
            const V3<T> rr = xyz<T>(r.pr[k]);
            const Vec4<T> d = r.pd[k];
            const T tx = (cm & AVN_CM_TANGENT) ? d.z : T(0), ty = (cm & AVN_CM_TANGENT) ? d.w : T(0);
            const V3<T> imp = coeff * ((d.x * normal + tx * t0) + ty * t1);
            const V3<T> dv = cmul(imp, inv_mass);
            const V3<T> dw = smul(I, cross(rr, imp));
            const bool on = k < np;
            out.d[k][0] = on ? (side ? dv.x : -dv.x) : nz; out.d[k][1] = on ? (side ? dv.y : -dv.y) : nz; out.d[k][2] = on ? (side ? dv.z : -dv.z) : nz;
            out.d[k][3] = on ? (side ? dw.x : -dw.x) : nz; out.d[k][4] = on ? (side ? dw.y : -dw.y) : nz; out.d[k][5] = on ? (side ? dw.z : -dw.z) : nz;
        }
    };
    // (1) the overflow colour's entries, in list order (solved FIRST); usually none.  Every lane of the quad walks the same list: same additions.
    for (; e < end; ++e) {
        WarmRecords<T> r; QuadDelta<T> dl;
        const uint32_t ent = w.inc_ent[e];
        warm_fetch<T>(w, ent, r);
        contributions(r, ent, true, dl);
#pragma unroll
        for (uint32_t k = 0; k < AVN_MAX_MANIFOLD_POINTS; ++k) { v.x = v.x + dl.d[k][0]; v.y = v.y + dl.d[k][1]; v.z = v.z + dl.d[k][2]; om.x = om.x + dl.d[k][3]; om.y = om.y + dl.d[k][4]; om.z = om.z + dl.d[k][5]; }
    }
    // (2) colours 0..22 in order: the body's j-th manifold (j-th set bit of the mask) belongs to lane j mod 4 in round j / 4 -- a body of a settled pile has four
    // or five, so a quad makes one or two rounds where a lane of the lane-per-body form walks 23 colours
    const uint32_t n_ent = (uint32_t)__popc(mask);
    uint32_t rest = mask;
    for (uint32_t j0 = 0; j0 < n_ent; j0 += 4u) {   // (trip count per quad: the four lanes of a quad stay together)
        // this lane's colour of the round: skip q set bits of `rest`; then all four lanes drop the round's four bits
        uint32_t mine = rest;
        for (uint32_t t = 0; t < q; ++t) mine &= mine - 1u;
        const bool present = mine != 0u;
        const uint32_t c = present ? (uint32_t)__ffs((int)mine) - 1u : 0u;
        rest &= rest - 1u; rest &= rest - 1u; rest &= rest - 1u; rest &= rest - 1u;
        const uint32_t sc = present ? w.inc_slot[(size_t)c * w.inc_stride + body] : 0u;   // (read again: it is in the L1 the mask pass left it in)
        WarmRecords<T> r; QuadDelta<T> dl;
        warm_fetch<T>(w, sc, r);   // (an absent entry fetches manifold 0's records and contributes -0.0: no branch around the loads)
        contributions(r, sc, present, dl);
        quad_accumulate<0, T>(dl, v, om); quad_accumulate<1, T>(dl, v, om); quad_accumulate<2, T>(dl, v, om); quad_accumulate<3, T>(dl, v, om);
    }
    if (q == 0u) {
        w.sb_lin[body] = make4<T>(v, l4.w);
        w.sb_ang[body] = make4<T>(om, a4.w);
    }
}

// Warm start of ONE manifold against a BodyView (the manifold-centric form of the same arithmetic: applied colour by colour in
// solve order every body sees the additions of k_body_warm_start in the same sequence).  Used by the island blocks, whose
// bodies live in LDS where a colour sweep costs a workgroup barrier instead of a launch.
template <class T, int STRIDE>
__device__ __forceinline__ void warm_core(const DW<T>& w, const StepParams<T>& p, uint32_t m, const BodyView<T>& bv, int i1, int i2) {
    Vec4<T> h1 = w.c_h1[m], h0 = w.m_n[m];
    const uint32_t S = w.m_stride;
    Vec4<T> pa[AVN_MAX_MANIFOLD_POINTS], pb[AVN_MAX_MANIFOLD_POINTS], pd[AVN_MAX_MANIFOLD_POINTS];
#pragma unroll
    for (uint32_t k = 0; k < AVN_MAX_MANIFOLD_POINTS; ++k) { uint32_t s = k * S + m; pa[k] = w.c_pa[s]; pb[k] = w.c_pb[s]; pd[k] = w.c_pd[s]; }
    const uint32_t cm = scalar_to_bits(h1.w);
    const uint32_t np = cm & 7u;
    BodyRef<T> b1, b2;
    load_body<T, false, STRIDE>(bv, i1, cm & AVN_CM_NOBODY1, cm & AVN_CM_DOM1, b1);
    load_body<T, false, STRIDE>(bv, i2, cm & AVN_CM_NOBODY2, cm & AVN_CM_DOM2, b2);
    if (np == 0) return;
    const V3<T> normal = xyz<T>(h0);
    const V3<T> t0 = xyz<T>(h1), t1 = cross(t0, normal);  // tangent_directions(), contact/mod.rs:411-421
    const T coeff = p.warm_start_coefficient;
#pragma unroll
    for (uint32_t k = 0; k < AVN_MAX_MANIFOLD_POINTS; ++k) {
        if (k < np) {
            const Vec4<T> d = pd[k];
            const T tx = (cm & AVN_CM_TANGENT) ? d.z : T(0), ty = (cm & AVN_CM_TANGENT) ? d.w : T(0);
            const V3<T> imp = coeff * ((d.x * normal + tx * t0) + ty * t1);
            apply_impulse(b1, b2, imp, xyz<T>(pa[k]), xyz<T>(pb[k]));
        }
    }
    store_body<T, STRIDE>(bv, i1, cm & AVN_CM_NOBODY1, b1);
    store_body<T, STRIDE>(bv, i2, cm & AVN_CM_NOBODY2, b2);
}

template <class T, bool USE_BIAS, int STRIDE, bool COH = false>
__device__ __forceinline__ void solve_core(const DW<T>& w, const StepParams<T>& p, uint32_t m, const BodyView<T>& bv, int i1, int i2) {
    // level 1: every load that only depends on m, issued up front and unconditionally (memory-level parallelism: the
    // kernel is latency-bound at one manifold per lane; the point planes hold 4 slots per manifold, so unused points
    // are in bounds and ignored); level 2: the body gathers; then the sequential impulse iteration out of registers
    Vec4<T> h1 = w.c_h1[m];
    Vec4<T> h0 = w.m_n[m];
    Vec4<T> h2 = w.m_tv[m];
    uint32_t S = w.m_stride;
    Vec4<T> pa[AVN_MAX_MANIFOLD_POINTS], pb[AVN_MAX_MANIFOLD_POINTS], pc[AVN_MAX_MANIFOLD_POINTS], pd[AVN_MAX_MANIFOLD_POINTS];
#pragma unroll
    for (uint32_t k = 0; k < AVN_MAX_MANIFOLD_POINTS; ++k) {
        uint32_t s = k * S + m;
        pa[k] = w.c_pa[s]; pb[k] = w.c_pb[s]; pc[k] = w.c_pc[s]; pd[k] = w.c_pd[s];
    }
    uint32_t cm = scalar_to_bits(h1.w);
    uint32_t np = cm & 7u;
    BodyRef<T> b1, b2;
    load_body<T, true, STRIDE, COH>(bv, i1, cm & AVN_CM_NOBODY1, cm & AVN_CM_DOM1, b1);
    load_body<T, true, STRIDE, COH>(bv, i2, cm & AVN_CM_NOBODY2, cm & AVN_CM_DOM2, b2);
    if (np == 0) return;
    V3<T> normal = xyz<T>(h0);
    T friction = h0.w;
    SoftCoef<T> soft = (cm & AVN_CM_SOFT_ND) ? p.soft_non_dynamic : p.soft_dynamic;
    T delta_secs = p.h_adj;
    V3<T> delta_translation = b2.dp - b1.dp;
    // normal impulses
#pragma unroll
    for (uint32_t k = 0; k < AVN_MAX_MANIFOLD_POINTS; ++k) {
        if (k < np) {
            V3<T> a1 = xyz<T>(pa[k]), a2 = xyz<T>(pb[k]);
            V3<T> r1w = qrot(b1.dq, a1), r2w = qrot(b2.dq, a2);
            V3<T> delta_separation = delta_translation + (r2w - r1w);
            T separation = dot(delta_separation, normal) + pa[k].w;
            V3<T> relative_velocity = velocity_at_point(b2, a2) - velocity_at_point(b1, a1);
            // ContactNormalPart::solve_impulse
            T normal_speed = dot(relative_velocity, normal);
            T eff_mass = pb[k].w, acc = pd[k].x;
            T impulse;
            if (separation > T(0)) {
                impulse = -eff_mass * (normal_speed + separation / delta_secs);
            } else if (USE_BIAS) {
                T bias = smax(soft.bias * separation, -p.max_overlap_solve_speed);
                T scaled_mass = soft.mass_scale * eff_mass;
                T scaled_impulse = soft.impulse_scale * acc;
                impulse = -scaled_mass * (normal_speed + bias) - scaled_impulse;
            } else {
                impulse = -eff_mass * normal_speed;
            }
            T new_impulse = smax(acc + impulse, T(0));
            impulse = new_impulse - acc;
            pd[k].x = new_impulse;
            pd[k].y = pd[k].y + new_impulse;  // total_impulse += new accumulated value (normal_part.rs:162)
            apply_impulse(b1, b2, impulse * normal, a1, a2);
        }
    }
    // friction
    if (cm & AVN_CM_TANGENT) {
        V3<T> t0 = xyz<T>(h1), t1 = cross(t0, normal);
        V3<T> surface_velocity = xyz<T>(h2);
#pragma unroll
        for (uint32_t k = 0; k < AVN_MAX_MANIFOLD_POINTS; ++k) {
            if (k < np) {
                V3<T> a1 = xyz<T>(pa[k]), a2 = xyz<T>(pb[k]);
                V3<T> relative_velocity = velocity_at_point(b2, a2) - velocity_at_point(b1, a1);
                // ContactTangentPart::solve_impulse
                T impulse_limit = friction * pd[k].x;
                V3<T> rv = relative_velocity + surface_velocity;
                T ts1 = dot(rv, t0), ts2 = dot(rv, t1);
                T t11 = ts1 * ts1, t22 = ts2 * ts2, t12 = ts1 * ts2;
                T inv = t11 * pc[k].x + t22 * pc[k].y + t12 * pc[k].z;
                T effective_mass = (t11 + t22) * (T(1) / inv);
                if (finite_t(effective_mass)) {
                    V2<T> delta{effective_mass * ts1, effective_mass * ts2};
                    V2<T> ni = clamp_length_max(V2<T>{pd[k].z - delta.x, pd[k].w - delta.y}, impulse_limit);
                    V2<T> di{ni.x - pd[k].z, ni.y - pd[k].w};
                    pd[k].z = ni.x; pd[k].w = ni.y;
                    apply_impulse(b1, b2, di.x * t0 + di.y * t1, a1, a2);
                } else {
                    // returns Vector::ZERO; the reference still applies the zero impulse (v -= 0)
                    apply_impulse(b1, b2, vzero<T>(), a1, a2);
                }
            }
        }
    }
#pragma unroll
    for (uint32_t k = 0; k < AVN_MAX_MANIFOLD_POINTS; ++k)
        if (k < np) w.c_pd[k * S + m] = pd[k];
    store_body<T, STRIDE, COH>(bv, i1, cm & AVN_CM_NOBODY1, b1);
    store_body<T, STRIDE, COH>(bv, i2, cm & AVN_CM_NOBODY2, b2);
}

// ---- f32: the two bodies of a manifold as the two halves of PACKED-f32 registers -----------------------------------------
// qrot of the anchors, velocity_at_point and apply_impulse evaluate the SAME expression once for body1 and once for
// body2; with (body1, body2) held in the (lo, hi) halves of 64-bit VGPR pairs every such operation is ONE v_pk_mul_f32 /
// v_pk_add_f32 for both bodies.  Each half goes through exactly the expression tree of the scalar form (the math
// templates of avn_math.h are instantiated with T = F2), so the results are bit-identical; the kernel's in-lane
// dependent chain -- the latency floor of a colour launch at one wave per SIMD -- loses about 40 % of its VALU issues.
typedef float f2raw __attribute__((ext_vector_type(2)));
struct F2 {
    f2raw v;
    F2() = default;
    __device__ __forceinline__ F2(float a) : v{a, a} {}
    __device__ __forceinline__ F2(float lo, float hi) : v{lo, hi} {}
    __device__ __forceinline__ explicit F2(f2raw x) : v(x) {}
};
__device__ __forceinline__ F2 operator+(F2 a, F2 b) { return F2(a.v + b.v); }
__device__ __forceinline__ F2 operator-(F2 a, F2 b) { return F2(a.v - b.v); }
__device__ __forceinline__ F2 operator*(F2 a, F2 b) { return F2(a.v * b.v); }
__device__ __forceinline__ F2 operator-(F2 a) { return F2(-a.v); }
// body1 (lo): a - b; body2 (hi): a + b   (apply_impulse's  v1 -= ..., v2 += ...)
// a - b == a + (-b) bit for bit: ONE v_pk_add_f32 with neg_lo on the second operand.  Written as inline asm: from `a.v + f2raw{-b.v.x, b.v.y}`
// hipcc builds the half-negated vector first (v_pk_add_f32 t, b, 0 neg_lo neg_hi; v_mov_b32 t.hi, b.hi) -- three instructions per component,
// 2 x 6 components x 8 impulse applications = 96 of the ~1 500 VALU issues of a lane's solve (round 3: the "v_mov pack / unpack" of the ISA).
__device__ __forceinline__ F2 sub_lo_add_hi(F2 a, F2 b) {
    f2raw r;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1]" : "=v"(r) : "v"(a.v), "v"(b.v));
    return F2(r);
}
__device__ __forceinline__ V3<F2> sub_lo_add_hi(V3<F2> a, V3<F2> b) { return {sub_lo_add_hi(a.x, b.x), sub_lo_add_hi(a.y, b.y), sub_lo_add_hi(a.z, b.z)}; }
__device__ __forceinline__ V3<F2> pair3(V3<float> lo, V3<float> hi) { return {F2(lo.x, hi.x), F2(lo.y, hi.y), F2(lo.z, hi.z)}; }
__device__ __forceinline__ V3<F2> splat3(V3<float> a) { return {F2(a.x), F2(a.y), F2(a.z)}; }
__device__ __forceinline__ V3<float> lo3(V3<F2> a) { return {a.x.v.x, a.y.v.x, a.z.v.x}; }
__device__ __forceinline__ V3<float> hi3(V3<F2> a) { return {a.x.v.y, a.y.v.y, a.z.v.y}; }

struct BodyPair {
    V3<F2> v, om, inv_mass;
    Sym3<F2> I;
};
__device__ __forceinline__ void apply_impulse(BodyPair& b, V3<float> imp, V3<F2> anchors) {
    V3<F2> p = splat3(imp);
    b.v = sub_lo_add_hi(b.v, cmul(p, b.inv_mass));
    b.om = sub_lo_add_hi(b.om, smul(b.I, cross(anchors, p)));
}

// (Tried and rejected, round 2 -- DESIGN.md section 4.1: compiler fences that put every level-1 load in flight before the first wait, and
//  an L2 prefetch of the NEXT colour launch's records issued during the impulse chain.  cfg2, two runs each on one box: neither 2 468
//  substeps/s | fences only 2 420 | prefetch only 2 359 | both 2 487 -- and the PMC pass showed the prefetch as 13 MB of EXTRA fetch traffic
//  per launch (42.3 MB against 29.2 MB): the lines do not survive in L2 until the next launch.  +0.8 % for +45 % traffic: removed.
//  Non-temporal (`nt`) loads / stores of the records, meant to keep the 9.6 MB of body records in L2 across launches: the PMC traffic
//  stays at 29.3 MB per launch (the bodies are not retained either way) and the launch takes 9.26 us instead of 8.34: removed as well.
//  Occupancy hints, `__launch_bounds__(64, 3)` for f32 (170 -> 168 VGPRs, 2 -> 3 waves per SIMD, 16 B of scratch per lane) and `(64, 2)`
//  for f64 (256 + 24 AGPRs -> 256, 1 -> 2 waves, 100 B of scratch): measured where occupancy could matter, at 230-250 k manifolds per launch
//  (cfg5 and its f32 twin, two runs each on one box): f64 23.6-24.0 -> 24.9-25.0 ms per step, f32 13.1-13.2 -> 13.3-13.4.  The spills cost
//  more than the extra wave hides; at that size a launch already moves its bytes at ~5.8 TB/s behind the launch-to-launch floor.
//  A whole pass as ONE persistent launch again, this time without cache maintenance (round 1's version paid ~15 us of buffer_wbl2 / buffer_inv
//  per barrier; tools/neighbour_sync_probe.hip: with agent-scope accesses only, a device-wide hand-over costs ~3.6 us): 256 workgroups of 256
//  lanes, one per CU (96 KB of dynamic LDS requested to keep it so: two on a CU double the chain), velocities through the COH accesses,
//  per colour "s_waitcnt vmcnt(0) -> s_barrier -> one arrival atomic -> the last arriver stores a release word -> one lane polls it", and
//  the NEXT colour's 600 B of records per lane requested into a second register set before the impulse chain.  Bit-identical (49 parity /
//  closed-loop tests).  cfg2, same box: 2 170 substeps/s without the prefetch wait, 2 040 with the prefetch behind the gathers, against
//  2 430-2 465 for the launches.  Per-workgroup timestamps of one launch: released -> gathers back 1.4 us, impulse chain + stores 6.0-6.7 us,
//  stores performed + arrival 1.2 us, release 0.45 us = ~9.5-10 us per colour, the launches' 9.7: the kernel boundary (1.6 us) was never the
//  expensive part, the in-lane chain is, and it does not overlap with anything when all waves of a colour run in lockstep.  Removed.
//  Hoisting what does not depend on the velocities (rotated anchors, separation and its speculative / biased terms for all four points) in
//  front of the chain as one branch-free block, to give the in-order VALU stream independent work: bit-identical, 8.9 us per launch instead
//  of 8.4 (longer live ranges, nothing gained: the compiler already interleaves what the basic blocks allow).  Removed.
//  Round 3, the anchors delivered as (body1, body2) pairs by the loads -- records laid out (a1.x, a2.x, a1.y, a2.y) (a1.z, a2.z, ..) instead of
//  one record per side -- timed with the existing records reinterpreted that way (wrong values, same instruction stream): 120 -> 84 v_mov,
//  1 386 -> 1 353 VALU instructions, 8.006 -> 7.926 us per isolated launch (A/B on one box, twice).  1 % for a second copy of the anchors
//  (the body-centric warm start reads ONE side per entry and would otherwise fetch both): not built.)
// What k_overflow_flow_tag hands to solve_core_packed: the records of the manifold and of its two bodies already in registers (loaded before
// / by the tag wait) and the tags the velocity records leave with.  NoPre: the pass fetches everything itself (every other caller).
struct NoPre { static constexpr bool on = false; };
struct FlowPre {
    static constexpr bool on = true;
    Vec4<float> h0, h1, h2, pa[AVN_MAX_MANIFOLD_POINTS], pb[AVN_MAX_MANIFOLD_POINTS], pc[AVN_MAX_MANIFOLD_POINTS], pd[AVN_MAX_MANIFOLD_POINTS];
    Vec4<float> l1, a1, dp1, dq1, sa1, sb1, l2, a2, dp2, dq2, sa2, sb2;
    bool no1, no2;          // no SolverBody on that side (rank PG_NONE)
    uint32_t tag1, tag2;    // next tags
};
template <bool USE_BIAS, int STRIDE, bool COH = false, class PRE = NoPre>
__device__ __forceinline__ void solve_core_packed(const DW<float>& w, const StepParams<float>& p, uint32_t m, const BodyView<float>& bv, int i1, int i2, const PRE& pre = PRE()) {
    typedef float T;
    // memory levels exactly as solve_one: (headers + point records) | body gathers
    Vec4<T> h1, h0, h2;
    uint32_t S = w.m_stride;
    Vec4<T> pa[AVN_MAX_MANIFOLD_POINTS], pb[AVN_MAX_MANIFOLD_POINTS], pc[AVN_MAX_MANIFOLD_POINTS], pd[AVN_MAX_MANIFOLD_POINTS];
    if constexpr (PRE::on) {
        h1 = pre.h1; h0 = pre.h0; h2 = pre.h2;
#pragma unroll
        for (uint32_t k = 0; k < AVN_MAX_MANIFOLD_POINTS; ++k) { pa[k] = pre.pa[k]; pb[k] = pre.pb[k]; pc[k] = pre.pc[k]; pd[k] = pre.pd[k]; }
    } else {
        h1 = w.c_h1[m];
        h0 = w.m_n[m];
        h2 = w.m_tv[m];
#pragma unroll
        for (uint32_t k = 0; k < AVN_MAX_MANIFOLD_POINTS; ++k) {
            uint32_t s = k * S + m;
            pa[k] = w.c_pa[s]; pb[k] = w.c_pb[s]; pc[k] = w.c_pc[s]; pd[k] = w.c_pd[s];
        }
    }
    uint32_t cm = scalar_to_bits(h1.w);
    uint32_t np = cm & 7u;
    BodyRef<T> b1, b2;
    bool no1 = cm & AVN_CM_NOBODY1, no2 = cm & AVN_CM_NOBODY2;
    if constexpr (PRE::on) {
        no1 = pre.no1; no2 = pre.no2;
        body_from_recs<T, true>(pre.l1, pre.a1, pre.dp1, pre.dq1, pre.sa1, pre.sb1, no1, cm & AVN_CM_DOM1, b1);
        body_from_recs<T, true>(pre.l2, pre.a2, pre.dp2, pre.dq2, pre.sa2, pre.sb2, no2, cm & AVN_CM_DOM2, b2);
        b1.lin_w = b1.ang_w = __uint_as_float(pre.tag1); b2.lin_w = b2.ang_w = __uint_as_float(pre.tag2);
        if (np == 0) {   // (nothing to solve; the tags still move: they are the hand-over)
            store_body<T, STRIDE, COH>(bv, i1, no1, b1);
            store_body<T, STRIDE, COH>(bv, i2, no2, b2);
            return;
        }
    } else {
        load_body<T, true, STRIDE, COH>(bv, i1, no1, cm & AVN_CM_DOM1, b1);
        load_body<T, true, STRIDE, COH>(bv, i2, no2, cm & AVN_CM_DOM2, b2);
        if (np == 0) return;
    }
    BodyPair bp;
    bp.v = pair3(b1.v, b2.v); bp.om = pair3(b1.om, b2.om); bp.inv_mass = pair3(b1.inv_mass, b2.inv_mass);
    bp.I = Sym3<F2>{F2(b1.I.m00, b2.I.m00), F2(b1.I.m01, b2.I.m01), F2(b1.I.m02, b2.I.m02), F2(b1.I.m11, b2.I.m11), F2(b1.I.m12, b2.I.m12), F2(b1.I.m22, b2.I.m22)};
    Q4<F2> dq{F2(b1.dq.x, b2.dq.x), F2(b1.dq.y, b2.dq.y), F2(b1.dq.z, b2.dq.z), F2(b1.dq.w, b2.dq.w)};
    V3<T> normal = xyz<T>(h0);
    T friction = h0.w;
    SoftCoef<T> soft = (cm & AVN_CM_SOFT_ND) ? p.soft_non_dynamic : p.soft_dynamic;
    T delta_secs = p.h_adj;
    V3<T> delta_translation = b2.dp - b1.dp;
    // normal impulses
#pragma unroll
    for (uint32_t k = 0; k < AVN_MAX_MANIFOLD_POINTS; ++k) {
        if (k < np) {
            V3<F2> anchors = pair3(xyz<T>(pa[k]), xyz<T>(pb[k]));
            V3<F2> rw = qrot(dq, anchors);
            V3<T> delta_separation = delta_translation + (hi3(rw) - lo3(rw));
            T separation = dot(delta_separation, normal) + pa[k].w;
            V3<F2> vap = bp.v + cross(bp.om, anchors);  // velocity_at_point of both bodies
            V3<T> relative_velocity = hi3(vap) - lo3(vap);
            // ContactNormalPart::solve_impulse
            T normal_speed = dot(relative_velocity, normal);
            T eff_mass = pb[k].w, acc = pd[k].x;
            T impulse;
            if (separation > T(0)) {
                impulse = -eff_mass * (normal_speed + separation / delta_secs);
            } else if (USE_BIAS) {
                T bias = smax(soft.bias * separation, -p.max_overlap_solve_speed);
                T scaled_mass = soft.mass_scale * eff_mass;
                T scaled_impulse = soft.impulse_scale * acc;
                impulse = -scaled_mass * (normal_speed + bias) - scaled_impulse;
            } else {
                impulse = -eff_mass * normal_speed;
            }
            T new_impulse = smax(acc + impulse, T(0));
            impulse = new_impulse - acc;
            pd[k].x = new_impulse;
            pd[k].y = pd[k].y + new_impulse;  // total_impulse += new accumulated value (normal_part.rs:162)
            apply_impulse(bp, impulse * normal, anchors);
        }
    }
    // friction
    if (cm & AVN_CM_TANGENT) {
        V3<T> t0 = xyz<T>(h1), t1 = cross(t0, normal);
        V3<T> surface_velocity = xyz<T>(h2);
#pragma unroll
        for (uint32_t k = 0; k < AVN_MAX_MANIFOLD_POINTS; ++k) {
            if (k < np) {
                V3<F2> anchors = pair3(xyz<T>(pa[k]), xyz<T>(pb[k]));
                V3<F2> vap = bp.v + cross(bp.om, anchors);
                V3<T> relative_velocity = hi3(vap) - lo3(vap);
                // ContactTangentPart::solve_impulse
                T impulse_limit = friction * pd[k].x;
                V3<T> rv = relative_velocity + surface_velocity;
                T ts1 = dot(rv, t0), ts2 = dot(rv, t1);
                T t11 = ts1 * ts1, t22 = ts2 * ts2, t12 = ts1 * ts2;
                T inv = t11 * pc[k].x + t22 * pc[k].y + t12 * pc[k].z;
                T effective_mass = (t11 + t22) * (T(1) / inv);
                if (finite_t(effective_mass)) {
                    V2<T> delta{effective_mass * ts1, effective_mass * ts2};
                    V2<T> ni = clamp_length_max(V2<T>{pd[k].z - delta.x, pd[k].w - delta.y}, impulse_limit);
                    V2<T> di{ni.x - pd[k].z, ni.y - pd[k].w};
                    pd[k].z = ni.x; pd[k].w = ni.y;
                    apply_impulse(bp, di.x * t0 + di.y * t1, anchors);
                } else {
                    // returns Vector::ZERO; the reference still applies the zero impulse (v -= 0)
                    apply_impulse(bp, vzero<T>(), anchors);
                }
            }
        }
    }
#pragma unroll
    for (uint32_t k = 0; k < AVN_MAX_MANIFOLD_POINTS; ++k)
        if (k < np) w.c_pd[k * S + m] = pd[k];
    b1.v = lo3(bp.v); b1.om = lo3(bp.om); b2.v = hi3(bp.v); b2.om = hi3(bp.om);
    store_body<T, STRIDE, COH>(bv, i1, no1, b1);
    store_body<T, STRIDE, COH>(bv, i2, no2, b2);
}
template <class T, bool USE_BIAS, int STRIDE, bool COH = false> struct SolveDispatch {
    static __device__ __forceinline__ void run(const DW<T>& w, const StepParams<T>& p, uint32_t m, const BodyView<T>& bv, int i1, int i2) { solve_core<T, USE_BIAS, STRIDE, COH>(w, p, m, bv, i1, i2); }
};
template <bool USE_BIAS, int STRIDE, bool COH> struct SolveDispatch<float, USE_BIAS, STRIDE, COH> {
    static __device__ __forceinline__ void run(const DW<float>& w, const StepParams<float>& p, uint32_t m, const BodyView<float>& bv, int i1, int i2) { solve_core_packed<USE_BIAS, STRIDE, COH>(w, p, m, bv, i1, i2); }
};

// ---- f32, EIGHT LANES PER MANIFOLD (round 5) ---------------------------------------------------------------------------------
// The lane-per-manifold solve above is bound by its own instruction stream: 1 398 VALU issues on one wave per SIMD = 2.6-2.8 us of a launch whose
// fixed cost is ~4 us, and a settled closed-loop colour holds a few thousand manifolds -- under 150 waves for 1 024 SIMDs
// (profiles/r05_color_pass_issue_model.txt).  Here a manifold is EIGHT lanes: quad 0 = body1, quad 1 = body2, lanes 0..2 of a quad = the x / y / z
// components of every vector of that body (lane 3 idles).  What was one v_pk_* per component for both bodies becomes ONE plain instruction for both
// bodies and all three components: cross products read the two other components through quad_perm operands, a dot is the lane's product summed as
// (x + y) + z through quad broadcasts, the two bodies meet through row_shl:4 / row_shr:4.  Every lane evaluates, for its own component, exactly the
// expression of solve_core<float> in its order (subtraction from body1 as addition of the exactly negated term: x - y == x + (-y) for every x, y), and the
// scalar impulse arithmetic is replicated in all eight lanes from bit-identical inputs: the result is the lane-per-manifold form's, bit for bit
// (the closed-loop parity suites run on this kernel).  Eight times the waves, each ~0.6 of the instruction stream: it pays while a colour's waves
// still find idle SIMDs (below ~16 k manifolds: launch_pass chooses per colour at capture time).
namespace oct {
template <int CTRL> __device__ __forceinline__ float dpp(float x) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xF, 0xF, true)); }
__device__ __forceinline__ float rot1(float x) { return dpp<0xC9>(x); }   // quad_perm:[1,2,0,3]: lane j reads component j + 1 (mod 3)
__device__ __forceinline__ float rot2(float x) { return dpp<0xD2>(x); }   // quad_perm:[2,0,1,3]: component j + 2
__device__ __forceinline__ float bx(float x) { return dpp<0x00>(x); }     // quad broadcasts of the x / y / z lane
__device__ __forceinline__ float by(float x) { return dpp<0x55>(x); }
__device__ __forceinline__ float bz(float x) { return dpp<0xAA>(x); }
// the other body's lane of the same component: quads 0 and 2 of a row read four lanes up, quads 1 and 3 four lanes down
__device__ __forceinline__ float partner(float x) {
    int t = __builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x104 /* row_shl:4 */, 0xF, 0x5, false);
    t = __builtin_amdgcn_update_dpp(t, __float_as_int(x), 0x114 /* row_shr:4 */, 0xF, 0xA, false);
    return __int_as_float(t);
}
// dot(a, b) = (a.x b.x + a.y b.y) + a.z b.z, the same value in all four lanes of the quad
__device__ __forceinline__ float dot3(float a, float b) { const float p = a * b; return (bx(p) + by(p)) + bz(p); }
}  // namespace oct

#define OCT_MANIFOLDS_PER_WAVE 8
template <bool USE_BIAS>
__device__ __forceinline__ void solve_core_oct(const DW<float>& w, const StepParams<float>& p, uint32_t m, uint32_t lane8) {
    using namespace oct;
    const uint32_t j = lane8 & 3u;
    const bool q = (lane8 & 4u) != 0u;   // this lane's body: false = body1, true = body2
    auto sel = [&](float x, float y, float z) { return j == 0u ? x : (j == 1u ? y : z); };
    // level 1: headers, the four points' records, the body indices
    const Vec4<float> h1 = w.c_h1[m], h0 = w.m_n[m], h2 = w.m_tv[m];
    const uint32_t S = w.m_stride;
    Vec4<float> pa[AVN_MAX_MANIFOLD_POINTS], pb[AVN_MAX_MANIFOLD_POINTS], pc[AVN_MAX_MANIFOLD_POINTS], pd[AVN_MAX_MANIFOLD_POINTS];
#pragma unroll
    for (uint32_t k = 0; k < AVN_MAX_MANIFOLD_POINTS; ++k) {
        const uint32_t s = k * S + m;
        pa[k] = w.c_pa[s]; pb[k] = w.c_pb[s]; pc[k] = w.c_pc[s]; pd[k] = w.c_pd[s];
    }
    const int2 bi = w.m_bodies[m];
    const uint32_t cm = scalar_to_bits(h1.w);
    const uint32_t np = cm & 7u;
    // level 2: this quad's body (Pair2::operator[] addresses the paired slots: record i at base[2 i]); DUMMY by selects, as load_body
    const size_t o = (size_t)(q ? bi.y : bi.x);
    const Vec4<float> l4 = w.sb_lin[o], a4 = w.sb_ang[o], dp4 = w.sb_dp[o], dq4 = w.sb_dq[o], sa = w.si_a[o], sb = w.si_b[o];
    if (np == 0) return;
    const bool no_body = cm & (q ? AVN_CM_NOBODY2 : AVN_CM_NOBODY1);
    const bool ni = no_body || (cm & (q ? AVN_CM_DOM2 : AVN_CM_DOM1));
    float v0 = no_body ? 0.0f : sel(l4.x, l4.y, l4.z);
    float om0 = no_body ? 0.0f : sel(a4.x, a4.y, a4.z);
    const float dpj = no_body ? 0.0f : sel(dp4.x, dp4.y, dp4.z);
    const float b0 = no_body ? 0.0f : sel(dq4.x, dq4.y, dq4.z);
    const float qw = no_body ? 1.0f : dq4.w;
    const V3<float> em = effective_inv_mass<float>(sa.x, scalar_to_bits(sb.w));
    const float im = ni ? 0.0f : sel(em.x, em.y, em.z);
    // row j of the symmetric tensor {m00 m01 m02 m11 m12 m22} = {sa.y sa.z sa.w sb.x sb.y sb.z}
    const float I0 = ni ? 0.0f : sel(sa.y, sa.z, sa.w), I1 = ni ? 0.0f : sel(sa.z, sb.x, sb.y), I2 = ni ? 0.0f : sel(sa.w, sb.y, sb.z);
    const uint32_t neg = q ? 0u : 0x80000000u;   // body1's updates are subtractions
    auto signed_for_body = [&](float x) { return __uint_as_float(__float_as_uint(x) ^ neg); };
    // body2's value minus body1's, the same bits in both quads: body1's lanes (quads 0 and 2 of a row) compute partner - own, body2's own - partner, as
    // two DPP subtractions (s_nop: a VALU result read through DPP needs two wait states, and the assembler sees no hazards inside inline asm)
    auto d21 = [&](float x) {
        float r;
        asm("s_nop 1\n\tv_sub_f32_dpp %0, %1, %1 row_shl:4 row_mask:0xf bank_mask:0x5\n\tv_subrev_f32_dpp %0, %1, %1 row_shr:4 row_mask:0xf bank_mask:0xa" : "=&v"(r) : "v"(x));
        return r;
    };
    // apply_impulse: v -/+= imp * inv_mass, om -/+= I (r x imp); `a1, a2` = components j + 1, j + 2 of this body's anchor
    auto apply = [&](float imp0, float a1, float a2) {
        v0 = v0 + signed_for_body(imp0 * im);
        const float cx = a1 * rot2(imp0) - rot1(imp0) * a2;                 // cross(r, imp)_j = r[j+1] imp[j+2] - imp[j+1] r[j+2]
        const float dw_ = (I0 * bx(cx) + I1 * by(cx)) + I2 * bz(cx);       // smul(I, c)_j = (c0[j] c.x + c1[j] c.y) + c2[j] c.z
        om0 = om0 + signed_for_body(dw_);
    };
    const float n0 = sel(h0.x, h0.y, h0.z);
    const float friction = h0.w;
    const SoftCoef<float> soft = (cm & AVN_CM_SOFT_ND) ? p.soft_non_dynamic : p.soft_dynamic;
    const float delta_secs = p.h_adj;
    const float dtr = d21(dpj);                              // delta_translation = b2.dp - b1.dp
    // qrot(dq, v) = (v (w w - b.b) + b ((v.b) 2)) + (b x v) (w 2)
    const float qb1 = rot1(b0), qb2 = rot2(b0);
    const float kA = qw * qw - dot3(b0, b0), kC = qw * 2.0f;
    float ax0[AVN_MAX_MANIFOLD_POINTS], ax1[AVN_MAX_MANIFOLD_POINTS], ax2[AVN_MAX_MANIFOLD_POINTS];   // this body's anchor: components j, j + 1, j + 2
#pragma unroll
    for (uint32_t k = 0; k < AVN_MAX_MANIFOLD_POINTS; ++k) {
        const Vec4<float> an = q ? pb[k] : pa[k];
        ax0[k] = sel(an.x, an.y, an.z); ax1[k] = rot1(ax0[k]); ax2[k] = rot2(ax0[k]);
    }
    // normal impulses
#pragma unroll
    for (uint32_t k = 0; k < AVN_MAX_MANIFOLD_POINTS; ++k) {
        if (k < np) {
            const float a0 = ax0[k], a1 = ax1[k], a2 = ax2[k];
            const float vb = dot3(a0, b0);
            const float rw = (a0 * kA + b0 * (vb * 2.0f)) + (qb1 * a2 - a1 * qb2) * kC;
            const float dsep = dtr + d21(rw);                              // delta_translation + (r2w - r1w)
            const float separation = dot3(dsep, n0) + pa[k].w;
            const float vap = v0 + (rot1(om0) * a2 - a1 * rot2(om0));      // velocity_at_point: v + om x anchor
            const float rel = d21(vap);
            const float normal_speed = dot3(rel, n0);
            const float eff_mass = pb[k].w, acc = pd[k].x;
            float impulse;
            if (separation > 0.0f) {
                impulse = -eff_mass * (normal_speed + separation / delta_secs);
            } else if (USE_BIAS) {
                const float bias = smax(soft.bias * separation, -p.max_overlap_solve_speed);
                const float scaled_mass = soft.mass_scale * eff_mass;
                const float scaled_impulse = soft.impulse_scale * acc;
                impulse = -scaled_mass * (normal_speed + bias) - scaled_impulse;
            } else {
                impulse = -eff_mass * normal_speed;
            }
            const float new_impulse = smax(acc + impulse, 0.0f);
            impulse = new_impulse - acc;
            pd[k].x = new_impulse;
            pd[k].y = pd[k].y + new_impulse;
            apply(impulse * n0, a1, a2);
        }
    }
    // friction
    if (cm & AVN_CM_TANGENT) {
        const float t0 = sel(h1.x, h1.y, h1.z);
        const float t1 = rot1(t0) * rot2(n0) - rot1(n0) * rot2(t0);       // cross(t0, normal)_j
        const float tv0 = sel(h2.x, h2.y, h2.z);
#pragma unroll
        for (uint32_t k = 0; k < AVN_MAX_MANIFOLD_POINTS; ++k) {
            if (k < np) {
                const float a1 = ax1[k], a2 = ax2[k];
                const float vap = v0 + (rot1(om0) * a2 - a1 * rot2(om0));
                const float rel = d21(vap);
                const float impulse_limit = friction * pd[k].x;
                const float rv = rel + tv0;
                const float ts1 = dot3(rv, t0), ts2 = dot3(rv, t1);
                const float t11 = ts1 * ts1, t22 = ts2 * ts2, t12 = ts1 * ts2;
                const float inv = t11 * pc[k].x + t22 * pc[k].y + t12 * pc[k].z;
                const float effective_mass = (t11 + t22) * (1.0f / inv);
                float out0 = 0.0f;                                        // a non-finite effective mass: Vector::ZERO, still applied
                if (finite_t(effective_mass)) {
                    const V2<float> delta{effective_mass * ts1, effective_mass * ts2};
                    const V2<float> nw = clamp_length_max(V2<float>{pd[k].z - delta.x, pd[k].w - delta.y}, impulse_limit);
                    const V2<float> di{nw.x - pd[k].z, nw.y - pd[k].w};
                    pd[k].z = nw.x; pd[k].w = nw.y;
                    out0 = di.x * t0 + di.y * t1;
                }
                apply(out0, a1, a2);
            }
        }
    }
    if (lane8 == 0u) {
#pragma unroll
        for (uint32_t k = 0; k < AVN_MAX_MANIFOLD_POINTS; ++k)
            if (k < np) w.c_pd[k * S + m] = pd[k];
    }
    // the x lane of each quad collects its body's components and writes the two velocity records
    const float vy = rot1(v0), vz = rot2(v0), oy = rot1(om0), oz = rot2(om0);
    if (j == 0u && !no_body) {
        w.sb_lin[o] = make4<float>(v0, vy, vz, l4.w);
        w.sb_ang[o] = make4<float>(om0, oy, oz, a4.w);
    }
}
template <int PASS>
__global__ __launch_bounds__(CONTACT_THREADS) void k_color_pass_oct(DW<float> w, StepParams<float> p, uint32_t color, uint32_t arg_base, uint32_t arg_end) {
    uint32_t base = arg_base, end = arg_end;
    if (arg_end == 0u) { base = w.color_offsets[color]; end = w.color_offsets[color + 1]; }
    // as k_color_pass: block b on XCD b % 8, every XCD a contiguous eighth of the colour's REAL tiles (here 8 manifolds per tile)
    const uint32_t tiles = (end - base + OCT_MANIFOLDS_PER_WAVE - 1u) / OCT_MANIFOLDS_PER_WAVE, per = (tiles + 7u) >> 3, row = blockIdx.x >> 3;
    if (row >= per) return;
    const uint32_t blk = (blockIdx.x & 7u) * per + row;
    const uint32_t m = base + blk * OCT_MANIFOLDS_PER_WAVE + (threadIdx.x >> 3);
    if (m >= end) return;   // (whole manifolds leave: the eight lanes of a group stay together)
    solve_core_oct<PASS == 1 /* PASS_BIAS */>(w, p, m, threadIdx.x & 7u);
}

// the restitution impulses of one manifold with np > 0 points and restitution != 0 on (b1, b2)
template <class T>
__device__ __forceinline__ void restitution_chain(const DW<T>& w, const StepParams<T>& p, uint32_t m, uint32_t np, T restitution, V3<T> normal, BodyRef<T>& b1, BodyRef<T>& b2) {
    uint32_t S = w.m_stride;
    uint32_t iterations = np > 1 ? p.restitution_iterations : 1u;
    T threshold = p.restitution_threshold;
    for (uint32_t it = 0; it < iterations; ++it)
        for (uint32_t k = 0; k < np; ++k) {
            uint32_t s = k * S + m;
            Vec4<T> pa = w.c_pa[s], pb = w.c_pb[s], pc = w.c_pc[s], pd = w.c_pd[s];
            T pre_normal_speed = pc.w;
            if (pre_normal_speed > -threshold || pd.y == T(0)) continue;
            V3<T> a1 = xyz<T>(pa), a2 = xyz<T>(pb);
            V3<T> relative_velocity = velocity_at_point(b2, a2) - velocity_at_point(b1, a1);
            T normal_speed = dot(relative_velocity, normal);
            T impulse = -pb.w * (normal_speed + restitution * pre_normal_speed);
            T new_impulse = smax(pd.x + impulse, T(0));
            impulse = new_impulse - pd.x;
            pd.x = new_impulse;
            pd.y = pd.y + impulse;
            w.c_pd[s] = pd;
            apply_impulse(b1, b2, impulse * normal, a1, a2);
        }
}
template <class T, int STRIDE, bool COH = false>
__device__ __forceinline__ void restitution_core(const DW<T>& w, const StepParams<T>& p, uint32_t m, const BodyView<T>& bv, int i1, int i2) {
    Vec4<T> h1 = w.c_h1[m];
    uint32_t cm = scalar_to_bits(h1.w);
    uint32_t np = cm & 7u;
    if (np == 0) return;
    T restitution = w.m_tv[m].w;
    if (restitution == T(0)) return;
    V3<T> normal = xyz<T>(w.m_n[m]);
    BodyRef<T> b1, b2;
    load_body<T, false, STRIDE, COH>(bv, i1, cm & AVN_CM_NOBODY1, cm & AVN_CM_DOM1, b1);
    load_body<T, false, STRIDE, COH>(bv, i2, cm & AVN_CM_NOBODY2, cm & AVN_CM_DOM2, b2);
    restitution_chain<T>(w, p, m, np, restitution, normal, b1, b2);
    store_body<T, STRIDE, COH>(bv, i1, cm & AVN_CM_NOBODY1, b1);
    store_body<T, STRIDE, COH>(bv, i2, cm & AVN_CM_NOBODY2, b2);
}

// Measurement aid (AVN_BIAS_SKELETON=1, tools/measure_floor.py): the memory skeleton of the solve pass -- every load of solve_core
// (records, body gathers), every store (accumulated impulses, both bodies), values written back unchanged, no arithmetic beyond one add
// per loaded word that keeps the loads alive.  Its duration is what the launch costs when the solve itself is free.
template <class T, int STRIDE>
__device__ __forceinline__ void skeleton_core(const DW<T>& w, uint32_t m, const BodyView<T>& bv, int i1, int i2) {
    Vec4<T> h1 = w.c_h1[m];
    Vec4<T> h0 = w.m_n[m];
    Vec4<T> h2 = w.m_tv[m];
    uint32_t S = w.m_stride;
    Vec4<T> pa[AVN_MAX_MANIFOLD_POINTS], pb[AVN_MAX_MANIFOLD_POINTS], pc[AVN_MAX_MANIFOLD_POINTS], pd[AVN_MAX_MANIFOLD_POINTS];
#pragma unroll
    for (uint32_t k = 0; k < AVN_MAX_MANIFOLD_POINTS; ++k) {
        uint32_t s = k * S + m;
        pa[k] = w.c_pa[s]; pb[k] = w.c_pb[s]; pc[k] = w.c_pc[s]; pd[k] = w.c_pd[s];
    }
    uint32_t cm = scalar_to_bits(h1.w);
    uint32_t np = cm & 7u;
    BodyRef<T> b1, b2;
    load_body<T, true, STRIDE>(bv, i1, cm & AVN_CM_NOBODY1, cm & AVN_CM_DOM1, b1);
    load_body<T, true, STRIDE>(bv, i2, cm & AVN_CM_NOBODY2, cm & AVN_CM_DOM2, b2);
    if (np == 0) return;
    T acc = ((h0.x + h0.y) + (h0.z + h0.w)) + ((h2.x + h2.y) + (h2.z + h2.w)) + ((h1.x + h1.y) + h1.z);
#pragma unroll
    for (uint32_t k = 0; k < AVN_MAX_MANIFOLD_POINTS; ++k)
        acc = acc + (((pa[k].x + pa[k].y) + (pa[k].z + pa[k].w)) + ((pb[k].x + pb[k].y) + (pb[k].z + pb[k].w)) + ((pc[k].x + pc[k].y) + (pc[k].z + pc[k].w)));
    acc = acc + (dot(b1.dp, b2.dp) + dot(b1.inv_mass, b2.inv_mass)) + ((b1.dq.x + b1.dq.y) + (b1.dq.z + b1.dq.w)) + ((b2.dq.x + b2.dq.y) + (b2.dq.z + b2.dq.w));
    acc = acc + ((b1.I.m00 + b1.I.m01) + (b1.I.m02 + b1.I.m11) + (b1.I.m12 + b1.I.m22)) + ((b2.I.m00 + b2.I.m01) + (b2.I.m02 + b2.I.m11) + (b2.I.m12 + b2.I.m22));
    const bool never = scalar_to_bits(acc) == 0x7FC12345u;   // (not a value the sum can take: the stores below write back what was read)
#pragma unroll
    for (uint32_t k = 0; k < AVN_MAX_MANIFOLD_POINTS; ++k)
        if (k < np) { if (never) pd[k].y = acc; w.c_pd[k * S + m] = pd[k]; }
    if (never) { b1.v.x = acc; b2.om.x = acc; }
    store_body<T, STRIDE>(bv, i1, cm & AVN_CM_NOBODY1, b1);
    store_body<T, STRIDE>(bv, i2, cm & AVN_CM_NOBODY2, b2);
}

enum { PASS_WARM = 0, PASS_BIAS = 1, PASS_RELAX = 2, PASS_RESTITUTION = 3, PASS_SKELETON = 5 };
// one manifold of a pass against the world's HBM body arrays
template <class T, int PASS, bool COH = false> __device__ __forceinline__ void pass_one(const DW<T>& w, const StepParams<T>& p, uint32_t m) {
    const int2 b = w.m_bodies[m];   // (a level-1 load like the constraint records: the body gathers depend on it)
    const BodyView<T> bv = global_bodies(w);
    if (PASS == PASS_WARM) warm_core<T, 2>(w, p, m, bv, b.x, b.y);   // (manifold-centric warm start of one colour: level-2 sharding exchanges after every colour)
    else if (PASS == PASS_SKELETON) skeleton_core<T, 2>(w, m, bv, b.x, b.y);
    else if (PASS == PASS_BIAS) SolveDispatch<T, true, 2, COH>::run(w, p, m, bv, b.x, b.y);
    else if (PASS == PASS_RELAX) SolveDispatch<T, false, 2, COH>::run(w, p, m, bv, b.x, b.y);
    else restitution_core<T, 2, COH>(w, p, m, bv, b.x, b.y);
}

// (Tried and rejected: the whole substep loop as ONE persistent launch, <= 1 workgroup per CU, with a device-wide barrier --
//  agent-scope release (buffer_wbl2 sc1), one arrival atomic per workgroup, bounded spin, acquire (buffer_inv sc1) -- where
//  the kernel boundaries are.  Bit-exact, but 3.27-3.63 ms per cfg2 step against 1.68 ms for the replayed launches: a
//  software barrier across the eight non-coherent XCD L2s costs ~25 us, five times the ~5.5 us launch-to-launch floor it was
//  meant to remove.  The in-kernel-barrier idea survives where the barrier is a workgroup one: k_island_substeps.)
// One colour: manifolds [offsets[c], offsets[c+1]) read from device memory so that a captured graph stays
// valid while the colour populations drift; the grid is a multiple of 8 blocks and remapped per XCD.
template <class T, int PASS>
__global__ __launch_bounds__(CONTACT_THREADS) void k_color_pass(DW<T> w, StepParams<T> p, uint32_t color, uint32_t arg_base, uint32_t arg_end) {
    // arg_end != 0: the colour range travels in the kernel arguments (one dependent scalar load less in a latency-bound
    // launch; the host re-captures the graph when the ranges change) -- used for host-uploaded manifold sets, which are
    // static between uploads.  arg_end == 0: read the live range from device memory (handle mode: colours drift every step).
    uint32_t base = arg_base, end = arg_end;
    if (arg_end == 0u) { base = w.color_offsets[color]; end = w.color_offsets[color + 1]; }
    // block b runs on XCD b % 8; each XCD walks one contiguous eighth of the colour.  The eighths are cut from the colour's REAL
    // tile count, not from the launched grid: grids carry up to 25 % slack (colours drift between re-captures), and an eighth of
    // the padded grid would leave the last XCDs with the empty tail while the first ones carry 110 tiles instead of 88.
    const uint32_t tiles = (end - base + CONTACT_THREADS - 1u) / CONTACT_THREADS, per = (tiles + 7u) >> 3, row = blockIdx.x >> 3;
    if (row >= per) return;
    const uint32_t blk = (blockIdx.x & 7u) * per + row;
    uint32_t m = base + blk * CONTACT_THREADS + threadIdx.x;
    if (m >= end) return;
    pass_one<T, PASS>(w, p, m);
}
// The overflow colour: the reference solves it strictly serially in manifold_handles order (solver/plugin.rs:461-467).
// Only the relative order of manifolds that SHARE A BODY can change a result, so the host turns the list into a level
// schedule (JointSchedule: level(m) = 1 + max level of the earlier overflow manifolds that share a body with m; connected
// components are independent): one workgroup per component walks its levels in order, the manifolds of a level run in
// parallel, a workgroup barrier separates levels.  Every body still sees its overflow manifolds in list order: the result is
// bit-identical to the serial loop, without serialising a dense pile's few thousand overflow contacts on one lane.
#define OVERFLOW_THREADS 256
template <class T, int PASS>
__global__ __launch_bounds__(OVERFLOW_THREADS) void k_overflow_pass(DW<T> w, StepParams<T> p, const uint32_t* __restrict__ comp_level_begin,
                                                                    const uint32_t* __restrict__ level_offsets, const uint32_t* __restrict__ order) {
    const uint32_t c = blockIdx.x;
    const uint32_t l0 = comp_level_begin[c], l1 = comp_level_begin[c + 1];
    for (uint32_t l = l0; l < l1; ++l) {
        const uint32_t k0 = level_offsets[l], k1 = level_offsets[l + 1];
        for (uint32_t k = k0 + threadIdx.x; k < k1; k += OVERFLOW_THREADS) pass_one<T, PASS>(w, p, order[k]);
        __syncthreads();  // same CU: the level's body writes are in its L1 / L2 before the next level gathers them
    }
}

// The overflow colour as ONE launch per pass (device closed loop): dataflow instead of levels.  Lane i owns the i-th manifold of the
// colour's list.  For each of its bodies that has a SolverBody the manifold knows its RANK among that body's overflow manifolds in
// list order (k_ovf_post) and the body carries a ticket counting the overflow manifolds solved on it since the step began: the lane
// waits until ticket == epoch * (manifolds on the body) + rank on both bodies, solves with agent-scope accesses of the velocities,
// and bumps the tickets.  Every body therefore sees its overflow manifolds in list order -- the reference's serial result -- while
// unrelated manifolds run concurrently and a dependent one starts as soon as its predecessors are through (no level barrier, no
// launch per level: a pile's overflow colour is hundreds of levels deep while it settles).  Tiles are handed out by an atomic
// counter, so a lane only waits for lanes of waves that already run; spins are bounded (PGC_ERROR is raised instead of hanging).
template <class T, int PASS>
__global__ __launch_bounds__(CONTACT_THREADS) void k_overflow_flow(DW<T> w, StepParams<T> p, OverflowFlow of, uint32_t epoch) {
    __shared__ uint32_t s_tile;
    const uint32_t lane = threadIdx.x;
    if (lane == 0) s_tile = atomicAdd(&of.tiles[epoch], 1u);
    __syncthreads();
    const uint32_t o0 = w.color_offsets[AVN_COLOR_OVERFLOW_INDEX], n23 = w.color_offsets[AVN_COLOR_OVERFLOW_INDEX + 1] - o0;
    const uint32_t i = s_tile * CONTACT_THREADS + lane;
    const bool valid = i < n23;
    const uint32_t m = o0 + (valid ? i : 0u);
    uint32_t r1 = 0xFFFFFFFFu, r2 = 0xFFFFFFFFu, t1 = 0, t2 = 0;
    int2 b = make_int2(0, 0);
    if (valid) {
        b = w.m_bodies[m];
        r1 = of.rank[2 * i]; r2 = of.rank[2 * i + 1];
        if (r1 != 0xFFFFFFFFu) t1 = epoch * (w.inc_off[b.x + 1] - w.inc_off[b.x]) + r1;
        if (r2 != 0xFFFFFFFFu) t2 = epoch * (w.inc_off[b.y + 1] - w.inc_off[b.y]) + r2;
    }
    bool done = !valid;
    for (uint32_t it = 0;; ++it) {
        if (!done) {
            const bool ready = (r1 == 0xFFFFFFFFu || __hip_atomic_load(&of.ticket[b.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == t1) &&
                               (r2 == 0xFFFFFFFFu || __hip_atomic_load(&of.ticket[b.y], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == t2);
            if (ready) {
                asm volatile("" ::: "memory");
                pass_one<T, PASS, true>(w, p, m);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the velocity stores are performed before the tickets move
                if (r1 != 0xFFFFFFFFu) __hip_atomic_fetch_add(&of.ticket[b.x], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (r2 != 0xFFFFFFFFu && !(r1 != 0xFFFFFFFFu && b.x == b.y)) __hip_atomic_fetch_add(&of.ticket[b.y], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                done = true;
            }
        }
        if (__all(done)) break;
        if (it > (1u << 20)) { if (lane == 0) atomicOr(of.error, 2u); break; }
        __builtin_amdgcn_s_sleep(8);   // (thousands of waves poll: a short sleep floods the fabric with sc1 loads and slows the lanes that work)
    }
}
// Round 5, f32: the hand-over without tickets.  The w lane of a body's two velocity records (zero after k_prepare_solver_bodies, passed
// through by every kernel of the substep loop) carries a TAG = (epoch + 1) << 20 | (overflow manifolds of this launch already solved on
// the body).  A record is one aligned 16-byte access, so it validates itself: a lane polls the four velocity records of its two bodies
// (agent scope) until each carries the tag its rank expects -- and then HAS the velocities; it stores them back with the next tag and is
// done.  Against the ticket form a hop loses the writer's "stores performed -> ticket atomic" (two fabric round trips in sequence) and the
// reader's gather after the poll (a third); and everything a manifold reads that no predecessor writes (constraint records, delta
// position / rotation, inertia) is in registers BEFORE the wait instead of being fetched on the dependency chain.  Rank 0 on a body waits
// for nothing (its predecessor is the previous launch); tags of other launches never match (epochs are unique inside a step, the w lanes
// restart at zero with every step).  Same order per body as the tickets give: the reference's serial loop (solver/plugin.rs:461-467).
// WHAT THIS RELIES ON: (1) an aligned 16-byte buffer_store_dwordx4 / buffer_load_dwordx4 with the sc1 bit is SINGLE-COPY ATOMIC at agent scope across
// the XCDs -- a reader never sees the new tag in .w next to old .x / .y / .z.  The CDNA ISA performs a naturally aligned dwordx4 access as one 16-byte request to
// one 128-byte line (it is never split across channels), and that is the whole of the guarantee used; nothing orders two DIFFERENT records, which is why every
// record carries its own tag.  There is no time-out that would catch a torn read: the check is empirical -- tests/test_gpu_overflow_stress.py runs the tag form
// against the ticket form (which does not depend on it: data stores are drained before the ticket moves) and against the oracle, bit for bit, over repeated
// collapses with 10^4..10^5 overflow manifolds; tools/stress_ovf.py is the same at cfg2's size.  (2) every kernel of the substep loop passes the w lanes of the
// velocity records through untouched (k_prepare_solver_bodies zeroes them once per step): the same test fails on the first step if one does not.
#define OVF_TAG_SHIFT 20
template <int PASS>
__global__ __launch_bounds__(CONTACT_THREADS) void k_overflow_flow_tag(DW<float> w, StepParams<float> p, OverflowFlow of, uint32_t epoch) {
    typedef float T;
    __shared__ uint32_t s_tile;
    const uint32_t lane = threadIdx.x;
    if (lane == 0) s_tile = atomicAdd(&of.tiles[epoch], 1u);
    __syncthreads();
    const uint32_t o0 = w.color_offsets[AVN_COLOR_OVERFLOW_INDEX], n23 = w.color_offsets[AVN_COLOR_OVERFLOW_INDEX + 1] - o0;
    const uint32_t i = s_tile * CONTACT_THREADS + lane;
    const bool valid = i < n23;
    const uint32_t m = o0 + (valid ? i : 0u);
    const uint32_t tag0 = (epoch + 1u) << OVF_TAG_SHIFT;
    uint32_t r1 = 0xFFFFFFFFu, r2 = 0xFFFFFFFFu, cm = 0u;
    int2 b = make_int2(0, 0);
    FlowPre pre;
    T restitution = T(0);
    const BodyView<T> bv = global_bodies(w);
    const Vec4<T> zero = make4<T>(0, 0, 0, 0), ident = make4<T>(0, 0, 0, 1);
    pre.h0 = pre.h1 = pre.h2 = zero;
    pre.dp1 = pre.dp2 = zero; pre.dq1 = pre.dq2 = ident; pre.sa1 = pre.sb1 = pre.sa2 = pre.sb2 = zero;
#pragma unroll
    for (uint32_t k = 0; k < AVN_MAX_MANIFOLD_POINTS; ++k) pre.pa[k] = pre.pb[k] = pre.pc[k] = pre.pd[k] = zero;
    if (valid) {
        b = w.m_bodies[m];
        r1 = of.rank[2 * i]; r2 = of.rank[2 * i + 1];
        pre.h1 = w.c_h1[m]; pre.h0 = w.m_n[m];
        if (PASS == PASS_RESTITUTION) restitution = w.m_tv[m].w;
        else {
            pre.h2 = w.m_tv[m];
            const uint32_t S = w.m_stride;
#pragma unroll
            for (uint32_t k = 0; k < AVN_MAX_MANIFOLD_POINTS; ++k) {
                const uint32_t s = k * S + m;
                pre.pa[k] = w.c_pa[s]; pre.pb[k] = w.c_pb[s]; pre.pc[k] = w.c_pc[s]; pre.pd[k] = w.c_pd[s];
            }
        }
        cm = scalar_to_bits(pre.h1.w);
        const size_t o1 = (size_t)b.x * 2, o2 = (size_t)b.y * 2;
        if (PASS != PASS_RESTITUTION) { pre.dp1 = bv.dp[o1]; pre.dq1 = bv.dq[o1]; pre.dp2 = bv.dp[o2]; pre.dq2 = bv.dq[o2]; }
        pre.sa1 = bv.sia[o1]; pre.sb1 = bv.sib[o1]; pre.sa2 = bv.sia[o2]; pre.sb2 = bv.sib[o2];
    }
    const bool has1 = r1 != 0xFFFFFFFFu, has2 = r2 != 0xFFFFFFFFu;
    pre.no1 = !has1; pre.no2 = !has2; pre.tag1 = tag0 | (r1 + 1u); pre.tag2 = tag0 | (r2 + 1u);
    // (the ranks come from the bodies' meta words, the constraint's flags from the SolverBody flags prepare wrote: one criterion, checked)
    if (valid && (has1 == ((cm & AVN_CM_NOBODY1) != 0u) || has2 == ((cm & AVN_CM_NOBODY2) != 0u))) atomicOr(of.error, 8u);
    bool done = !valid;
    for (uint32_t it = 0;; ++it) {
        asm volatile("" ::: "memory");
        if (!done) {
            bool ready = true;
            pre.l1 = pre.a1 = pre.l2 = pre.a2 = zero;
            if (has1) {
                pre.l1 = ld_rec_agent(bv.lin, (size_t)b.x * 2); pre.a1 = ld_rec_agent(bv.ang, (size_t)b.x * 2);
                if (r1 != 0u) ready = __float_as_uint(pre.l1.w) == (tag0 | r1) && __float_as_uint(pre.a1.w) == (tag0 | r1);
            }
            if (has2) {
                pre.l2 = ld_rec_agent(bv.lin, (size_t)b.y * 2); pre.a2 = ld_rec_agent(bv.ang, (size_t)b.y * 2);
                if (r2 != 0u) ready = ready && __float_as_uint(pre.l2.w) == (tag0 | r2) && __float_as_uint(pre.a2.w) == (tag0 | r2);
            }
            if (ready) {
                if (PASS == PASS_RESTITUTION) {
                    BodyRef<T> b1, b2;
                    body_from_recs<T, false>(pre.l1, pre.a1, pre.dp1, pre.dq1, pre.sa1, pre.sb1, !has1, cm & AVN_CM_DOM1, b1);
                    body_from_recs<T, false>(pre.l2, pre.a2, pre.dp2, pre.dq2, pre.sa2, pre.sb2, !has2, cm & AVN_CM_DOM2, b2);
                    const uint32_t np = cm & 7u;
                    if (np != 0u && restitution != T(0)) restitution_chain<T>(w, p, m, np, restitution, xyz<T>(pre.h0), b1, b2);
                    // (always stored, also where nothing changed: the tag is the hand-over)
                    b1.lin_w = b1.ang_w = __uint_as_float(pre.tag1); b2.lin_w = b2.ang_w = __uint_as_float(pre.tag2);
                    store_body<T, 2, true>(bv, b.x, !has1, b1);
                    store_body<T, 2, true>(bv, b.y, !has2, b2);
                } else {
                    solve_core_packed<PASS == PASS_BIAS, 2, true, FlowPre>(w, p, m, bv, b.x, b.y, pre);
                }
                done = true;
            }
        }
        if (__all(done)) break;
        if (it > (1u << 20)) { if (lane == 0) atomicOr(of.error, 2u); break; }
        for (uint32_t j = 0; j < of.poll_sleep; ++j) __builtin_amdgcn_s_sleep(1);
    }
}
// (Tried and rejected, round 2: a whole pass as ONE dataflow launch over all colours -- every manifold waits for per-body tickets in
//  "overflow list order, then colour order", its constraint records already loaded.  Bit-identical (the closed loop tracked the oracle),
//  but 219 us per cfg2 pass against 15 x 8.4 = 126 us for the colour launches: a body's 15 manifolds are 15 dependent hops of
//  poll + sc1 gather + ~1 500 issues + write-through + ticket, ~14 us each with ~10 000 waves polling.  The ticket pass stays where the
//  alternative is hundreds of launches: the overflow colour, k_overflow_flow.
//  Round 3: fewer manifolds per wave (32 / 16 / 8 instead of 64), against the solves of lanes that become ready in different polling rounds
//  running one after the other: cfg2's collapse window 7.00 -> 7.10 / 7.29 / 7.71 ms per step, the settled step 2.98 -> 2.95 (A/B on one box):
//  more waves polling cost more than the serialisation.  Not kept.  Neither was touching a manifold's constraint records BEFORE its ticket
//  wait (so that a hop would find them in L2): 7.22 -> 7.27 ms, nothing gained -- the hop is the sc1 gather and the in-lane chain.
//  Round 4: the settled pile's ~350 overflow manifolds (40 us per pass) in ONE 512-lane workgroup with workgroup-scope fences and L2-served
//  tickets instead of agent-scope accesses across XCDs: 40-49 us per pass, the same (A/B on one box, 3.049 vs 3.054 ms per step).  The pass is
//  (chain depth) x (one manifold's solve latency: records + gathers + ~1 500 dependent issues + stores), not the hand-over distance.  Not kept.)
// tickets and tile counters restart with every step.  A KERNEL, not hipMemsetAsync: inside the captured substep graph a memset node is
// not reliably ordered against a synchronous (null-stream) hipMemcpy issued between replays on ROCm 7.2 (found by the closed-loop tests:
// a whole step of wrong impulses after avn_pipeline_handles_get had copied with hipMemcpy); kernel nodes keep the chain.
__global__ __launch_bounds__(256) void k_overflow_reset(uint32_t* __restrict__ ticket, uint32_t n_ticket, uint32_t* __restrict__ tiles, uint32_t n_tiles) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < n_ticket) ticket[i] = 0u;
    if (i < n_tiles) tiles[i] = 0u;
}
void launch_overflow_reset(uint32_t* ticket, uint32_t n_ticket, uint32_t* tiles, uint32_t n_tiles, hipStream_t s) {
    const uint32_t n = n_ticket > n_tiles ? n_ticket : n_tiles;
    hipLaunchKernelGGL(k_overflow_reset, dim3((n + 255) / 256), dim3(256), 0, s, ticket, n_ticket, tiles, n_tiles);
}
template <class T> struct OverflowFlowLaunch {
    static bool run(const DW<T>&, const StepParams<T>&, int, const OverflowFlow&, uint32_t, uint32_t, hipStream_t) { return false; }
};
template <> struct OverflowFlowLaunch<float> {
    static bool run(const DW<float>& w, const StepParams<float>& p, int pass, const OverflowFlow& of, uint32_t epoch, uint32_t grid_blocks, hipStream_t s) {
        static const bool tickets = avn_env("AVN_OVF_TICKETS") != nullptr;   // (measure build: the round-2 ticket form, for A/B runs)
        if (tickets) return false;
        if (pass == PASS_BIAS) hipLaunchKernelGGL((k_overflow_flow_tag<PASS_BIAS>), dim3(grid_blocks), dim3(CONTACT_THREADS), 0, s, w, p, of, epoch);
        else if (pass == PASS_RELAX) hipLaunchKernelGGL((k_overflow_flow_tag<PASS_RELAX>), dim3(grid_blocks), dim3(CONTACT_THREADS), 0, s, w, p, of, epoch);
        else hipLaunchKernelGGL((k_overflow_flow_tag<PASS_RESTITUTION>), dim3(grid_blocks), dim3(CONTACT_THREADS), 0, s, w, p, of, epoch);
        return true;
    }
};
template <class T> void launch_overflow_flow(const DW<T>& w, const StepParams<T>& p, int pass, const OverflowFlow& of, uint32_t epoch, uint32_t grid_blocks, hipStream_t s) {
    if (!grid_blocks) return;
    if (OverflowFlowLaunch<T>::run(w, p, pass, of, epoch, grid_blocks, s)) return;   // f32: tags in the velocity records; f64 (two accesses per record): tickets
    if (pass == PASS_BIAS) hipLaunchKernelGGL((k_overflow_flow<T, PASS_BIAS>), dim3(grid_blocks), dim3(CONTACT_THREADS), 0, s, w, p, of, epoch);
    else if (pass == PASS_RELAX) hipLaunchKernelGGL((k_overflow_flow<T, PASS_RELAX>), dim3(grid_blocks), dim3(CONTACT_THREADS), 0, s, w, p, of, epoch);
    else hipLaunchKernelGGL((k_overflow_flow<T, PASS_RESTITUTION>), dim3(grid_blocks), dim3(CONTACT_THREADS), 0, s, w, p, of, epoch);
}

template <class T>
__global__ __launch_bounds__(256) void k_store_contact_impulses(DW<T> w) {
    uint32_t m = blockIdx.x * 256 + threadIdx.x;
    if (m >= w.n_manifolds) return;
    uint32_t cm = scalar_to_bits(w.c_h1[m].w);
    uint32_t np = cm & 7u;
    uint32_t S = w.m_stride;
    for (uint32_t k = 0; k < np; ++k) {
        uint32_t s = k * S + m;
        Vec4<T> d = w.c_pd[s];
        bool t = cm & AVN_CM_TANGENT;
        w.mp_w[s] = make4<T>(d.x, t ? d.z : T(0), t ? d.w : T(0), d.y);
    }
}

// ---- launchers ------------------------------------------------------------------------------------------
template <class T> void launch_prepare_contact_constraints(const DW<T>& w, const StepParams<T>& p, hipStream_t s, bool count_clean, const RowsView<T>* rows) {
    if (!count_clean) (void)hipMemsetAsync(w.constraint_count, 0, sizeof(uint32_t), s);
    if (!w.n_manifolds) return;
    if (rows) hipLaunchKernelGGL((k_prepare_contact_constraints<T, true>), dim3((w.n_manifolds + 255) / 256), dim3(256), 0, s, w, p, *rows);
    else hipLaunchKernelGGL((k_prepare_contact_constraints<T, false>), dim3((w.n_manifolds + 255) / 256), dim3(256), 0, s, w, p, RowsView<T>{});
}
template <class T> void launch_store_contact_impulses(const DW<T>& w, hipStream_t s) {
    if (w.n_manifolds) hipLaunchKernelGGL(k_store_contact_impulses<T>, dim3((w.n_manifolds + 255) / 256), dim3(256), 0, s, w);
}
uint32_t color_grid_blocks(uint32_t count) {
    uint32_t nb = (count + CONTACT_THREADS - 1) / CONTACT_THREADS;
    return ((nb + 7u) / 8u) * 8u;
}
// one level of a big overflow colour, device-wide: lane -> manifold order[first + i]
template <class T, int PASS>
__global__ __launch_bounds__(CONTACT_THREADS) void k_overflow_level(DW<T> w, StepParams<T> p, const uint32_t* __restrict__ order, uint32_t first, uint32_t count) {
    uint32_t i = xcd_block(blockIdx.x, gridDim.x) * CONTACT_THREADS + threadIdx.x;
    if (i >= count) return;
    pass_one<T, PASS>(w, p, order[first + i]);
}
// eight lanes per manifold for the colours of `oct_mask` (f32 biased solve / relax only): eight times the workgroups of the lane form's grid
template <class T, int PASS> struct OctLaunch { static bool run(const DW<T>&, const StepParams<T>&, uint32_t, uint32_t, uint32_t, uint32_t, hipStream_t) { return false; } };
template <int PASS> struct OctLaunch<float, PASS> {
    static bool run(const DW<float>& w, const StepParams<float>& p, uint32_t c, uint32_t blocks, uint32_t b, uint32_t e, hipStream_t s) {
        if constexpr (PASS == PASS_BIAS || PASS == PASS_RELAX) {
            hipLaunchKernelGGL((k_color_pass_oct<PASS>), dim3(blocks * (CONTACT_THREADS / OCT_MANIFOLDS_PER_WAVE)), dim3(CONTACT_THREADS), 0, s, w, p, c, b, e);
            return true;
        }
        return false;
    }
};
template <class T, int PASS> static uint32_t launch_pass(const DW<T>& w, const StepParams<T>& p, const uint32_t* grid_blocks, const uint32_t* arg_offsets, const OverflowSchedule& ovf, hipStream_t s, uint32_t oct_mask) {
    uint32_t launches = 0;
    if (grid_blocks[AVN_COLOR_OVERFLOW_INDEX] && ovf.n_glevels) {
        for (uint32_t l = 0; l < ovf.n_glevels; ++l) {
            uint32_t first = ovf.glevel_offsets[l], count = ovf.glevel_offsets[l + 1] - first;
            if (!count) continue;
            hipLaunchKernelGGL((k_overflow_level<T, PASS>), dim3(color_grid_blocks(count)), dim3(CONTACT_THREADS), 0, s, w, p, ovf.gorder, first, count);
            ++launches;
        }
    } else if (grid_blocks[AVN_COLOR_OVERFLOW_INDEX] && ovf.n_components) {
        hipLaunchKernelGGL((k_overflow_pass<T, PASS>), dim3(ovf.n_components), dim3(OVERFLOW_THREADS), 0, s, w, p, ovf.comp_level_begin, ovf.level_offsets, ovf.order);
        ++launches;
    }
    for (uint32_t c = 0; c < AVN_COLOR_OVERFLOW_INDEX; ++c)
        if (grid_blocks[c]) {
            uint32_t b = arg_offsets ? arg_offsets[c] : 0u, e = arg_offsets ? arg_offsets[c + 1] : 0u;
            if (arg_offsets && e == b) continue;  // (a captured range is exact: an empty colour needs no launch)
            if (!((oct_mask >> c) & 1u) || !OctLaunch<T, PASS>::run(w, p, c, grid_blocks[c], b, e, s))
                hipLaunchKernelGGL((k_color_pass<T, PASS>), dim3(grid_blocks[c]), dim3(CONTACT_THREADS), 0, s, w, p, c, b, e);
            ++launches;
        }
    return launches;
}
// inc_slot[colour][body] <- manifold | side << 31 for the manifolds of colours 0..22 (the table was cleared to EMPTY before).
// A dynamic body is in at most one manifold per colour (constraint_graph.rs:36-48): no two lanes write the same slot of a
// body that ever reads it (static bodies collect many writes per colour, and never run the warm start).
template <class T>
__global__ __launch_bounds__(256) void k_build_incidence_slots(DW<T> w) {
    uint32_t m = blockIdx.x * 256 + threadIdx.x;
    if (m >= w.color_offsets[AVN_COLOR_OVERFLOW_INDEX]) return;
    uint32_t lo = 0, hi = AVN_COLOR_OVERFLOW_INDEX;  // colour c: offsets[c] <= m < offsets[c + 1]
    while (hi - lo > 1) { uint32_t mid = (lo + hi) >> 1; if (w.color_offsets[mid] <= m) lo = mid; else hi = mid; }
    int2 b = w.m_bodies[m];
    w.inc_slot[(size_t)lo * w.inc_stride + (uint32_t)b.x] = m;
    w.inc_slot[(size_t)lo * w.inc_stride + (uint32_t)b.y] = m | 0x80000000u;
}
template <class T> void launch_build_incidence_slots(const DW<T>& w, hipStream_t s, bool cleared) {
    if (!cleared) (void)hipMemsetAsync(w.inc_slot, 0xFF, (size_t)AVN_COLOR_OVERFLOW_INDEX * w.inc_stride * sizeof(uint32_t), s);
    if (w.n_manifolds) hipLaunchKernelGGL(k_build_incidence_slots<T>, dim3((w.n_manifolds + 255) / 256), dim3(256), 0, s, w);
}
// ------------------------------------------------------------------------------------------------------
// Island blocks: the WHOLE substep loop of a block of contact islands in one workgroup, bodies staged in LDS.
//
// A colour launch over a few thousand manifolds is pure latency (launch + two dependent memory levels + the in-lane chain:
// ~5.5 us whatever the size), and a scene of many small islands (the reference's "Many Pyramids" bench) pays that floor
// colours x passes x substeps times.  Islands never exchange data inside the solver: a block = a set of whole islands (host:
// World::rebuild_island_blocks) owns its bodies for the entire substep loop, so one workgroup stages their six
// SolverBody / SolverBodyInertia records in LDS once, runs every system of every substep on them -- integrate_velocities,
// warm start, biased solve, integrate_positions + inertia refresh, relax -- with a workgroup barrier where the device-wide
// path has a kernel boundary, and writes the records back once.  Constraint records stay in HBM/L2 (read-mostly, 300 B per
// manifold); the per-body dependent gathers, the part a colour pass waits on, become LDS reads.
// Order: within the block the entries of a colour touch disjoint bodies (the reference's colouring invariant), colours run
// in solve order (overflow first, serially in list order on one lane, then 0..22) -- every body sees exactly the operation
// sequence of the device-wide path, hence of the reference: bit-identical.
template <class T, int PASS, bool CACHE>
__device__ __forceinline__ void island_colours(const DW<T>& w, const StepParams<T>& p, const uint2* ent, uint32_t e0, const BodyView<T>& bv, const uint32_t* col) {
    const uint32_t t = threadIdx.x;
    for (uint32_t c = 0; c < AVN_GRAPH_COLOR_COUNT; ++c) {
        const uint32_t begin = col[c], end = col[c + 1];
        if (begin == end) continue;  // workgroup-uniform
        // slot 0 = the overflow colour: one lane walks it in list order; every other colour: one lane per manifold
        const bool serial = c == 0;
        for (uint32_t e = begin + (serial ? 0u : t); e < end; e += serial ? 1u : ISLAND_THREADS) {
            if (serial && t != 0) continue;
            const uint2 en = ent[CACHE ? e - e0 : e];
            const uint32_t m = CACHE ? e - e0 : en.x;   // CACHE: `w` is the block's LDS copy of the constraint records, indexed by entry
            const int i1 = (int)(en.y & 0xFFFFu), i2 = (int)(en.y >> 16);
            if (PASS == PASS_WARM) warm_core<T, 1>(w, p, m, bv, i1, i2);
            else if (PASS == PASS_BIAS) SolveDispatch<T, true, 1>::run(w, p, m, bv, i1, i2);
            else SolveDispatch<T, false, 1>::run(w, p, m, bv, i1, i2);
        }
        __syncthreads();
    }
}
// CACHE = true: the block's constraint records (19 Vec4 per manifold) and its entry list are staged in LDS next to the bodies
// (host: every block fits ISLAND_LDS_VEC4), so a colour pass touches no global memory at all -- on a block whose colours hold
// ~17 manifolds the two dependent L2 round trips (entry -> records) were ~45 % of a 3.3 us pass.  The accumulated impulses go
// back to HBM once, after the last substep.  CACHE = false: bodies only (any block up to ISLAND_MAX_BODIES bodies).
template <class T, bool CACHE>
__global__ __launch_bounds__(ISLAND_THREADS) void k_island_substeps(DW<T> w, StepParams<T> p, IslandBlocks ib, uint32_t substeps, uint32_t iterations) {
    __shared__ Vec4<T> lds[ISLAND_LDS_VEC4];
    __shared__ uint32_t l_col[AVN_GRAPH_COLOR_COUNT + 1];
    const uint32_t t = threadIdx.x, blk = blockIdx.x;
    const uint32_t b0 = ib.body_off[blk], nb = ib.body_off[blk + 1] - b0;
    const uint32_t nbmax = CACHE ? ib.max_bodies : ISLAND_MAX_BODIES, LM = ib.max_manifolds;
    Vec4<T>* const l_lin = lds, *const l_ang = lds + nbmax, *const l_dp = lds + 2 * nbmax, *const l_dq = lds + 3 * nbmax, *const l_sia = lds + 4 * nbmax,
                   *const l_sib = lds + 5 * nbmax;
    Vec4<T>* const R = lds + 6 * nbmax;   // CACHE: h1[LM] n[LM] tv[LM] pa[4 LM] pb[4 LM] pc[4 LM] pd[4 LM] ent[LM / 2]
    if (t <= AVN_GRAPH_COLOR_COUNT) l_col[t] = ib.col_off[(size_t)blk * AVN_GRAPH_COLOR_COUNT + t];
    for (uint32_t l = t; l < nb; l += ISLAND_THREADS) {
        const uint32_t g = ib.bodies[b0 + l];
        l_lin[l] = w.sb_lin[g]; l_ang[l] = w.sb_ang[g]; l_dp[l] = w.sb_dp[g]; l_dq[l] = w.sb_dq[g]; l_sia[l] = w.si_a[g]; l_sib[l] = w.si_b[g];
    }
    const uint32_t e0 = ib.col_off[(size_t)blk * AVN_GRAPH_COLOR_COUNT], e1 = ib.col_off[(size_t)(blk + 1) * AVN_GRAPH_COLOR_COUNT];
    DW<T> wl = w;
    const uint2* ent = ib.ent;
    if (CACHE) {
        wl.c_h1 = R; wl.m_n = R + LM; wl.m_tv = R + 2 * LM; wl.c_pa = R + 3 * LM; wl.c_pb = R + 7 * LM; wl.c_pc = R + 11 * LM; wl.c_pd = R + 15 * LM;
        wl.m_stride = LM;
        uint2* l_ent = reinterpret_cast<uint2*>(R + 19 * LM);
        ent = l_ent;
        for (uint32_t e = e0 + t; e < e1; e += ISLAND_THREADS) {
            const uint2 en = ib.ent[e];
            const uint32_t m = en.x, lm = e - e0;
            l_ent[lm] = en;
            wl.c_h1[lm] = w.c_h1[m]; wl.m_n[lm] = w.m_n[m]; wl.m_tv[lm] = w.m_tv[m];
#pragma unroll
            for (uint32_t k = 0; k < AVN_MAX_MANIFOLD_POINTS; ++k) {
                const size_t s = (size_t)k * w.m_stride + m; const uint32_t d = k * LM + lm;
                wl.c_pa[d] = w.c_pa[s]; wl.c_pb[d] = w.c_pb[s]; wl.c_pc[d] = w.c_pc[s]; wl.c_pd[d] = w.c_pd[s];
            }
        }
    }
    __syncthreads();
    const BodyView<T> bv{l_lin, l_ang, l_dp, l_dq, l_sia, l_sib};
    for (uint32_t s = 0; s < substeps; ++s) {   // SubstepSchedule order, as World::substep()
        for (uint32_t l = t; l < nb; l += ISLAND_THREADS) {   // integrate_velocities
            const uint32_t g = ib.bodies[b0 + l];
            const Vec4<T> l4 = l_lin[l], a4 = l_ang[l];
            V3<T> v = xyz<T>(l4), om = xyz<T>(a4);
            if (integrate_velocities_one<T>(w, p, g, w.sb_flags[g], v, om, &l_dq[l])) { l_lin[l] = make4<T>(v, l4.w); l_ang[l] = make4<T>(om, a4.w); }
        }
        __syncthreads();
        island_colours<T, PASS_WARM, CACHE>(wl, p, ent, e0, bv, l_col);
        for (uint32_t it = 0; it < iterations; ++it) island_colours<T, PASS_BIAS, CACHE>(wl, p, ent, e0, bv, l_col);
        for (uint32_t l = t; l < nb; l += ISLAND_THREADS) {   // integrate_positions + update_solver_body_angular_inertia
            const uint32_t g = ib.bodies[b0 + l];
            Vec4<T> dp4 = l_dp[l], dq4 = l_dq[l], sa = l_sia[l], sb = l_sib[l];
            integrate_positions_one<T>(w, p, g, xyz<T>(l_lin[l]), xyz<T>(l_ang[l]), dp4, dq4, sa, sb);
            l_dp[l] = dp4; l_dq[l] = dq4; l_sia[l] = sa; l_sib[l] = sb;
        }
        __syncthreads();
        for (uint32_t it = 0; it < iterations; ++it) island_colours<T, PASS_RELAX, CACHE>(wl, p, ent, e0, bv, l_col);
    }
    for (uint32_t l = t; l < nb; l += ISLAND_THREADS) {
        const uint32_t g = ib.bodies[b0 + l];
        w.sb_lin[g] = l_lin[l]; w.sb_ang[g] = l_ang[l]; w.sb_dp[g] = l_dp[l]; w.sb_dq[g] = l_dq[l]; w.si_a[g] = l_sia[l]; w.si_b[g] = l_sib[l];
    }
    if (CACHE) {   // the accumulated impulses (the only constraint records the passes write)
        for (uint32_t e = e0 + t; e < e1; e += ISLAND_THREADS) {
            const uint32_t lm = e - e0, m = ent[lm].x;
#pragma unroll
            for (uint32_t k = 0; k < AVN_MAX_MANIFOLD_POINTS; ++k) w.c_pd[(size_t)k * w.m_stride + m] = wl.c_pd[k * LM + lm];
        }
    }
}
void launch_island_substeps(const DW<float>& w, const StepParams<float>& p, const IslandBlocks& ib, uint32_t substeps, uint32_t iterations, hipStream_t s) {
    if (!ib.n_blocks) return;
    if (ib.cache_records) hipLaunchKernelGGL((k_island_substeps<float, true>), dim3(ib.n_blocks), dim3(ISLAND_THREADS), 0, s, w, p, ib, substeps, iterations);
    else hipLaunchKernelGGL((k_island_substeps<float, false>), dim3(ib.n_blocks), dim3(ISLAND_THREADS), 0, s, w, p, ib, substeps, iterations);
}
template <class T> void launch_body_warm_start(const DW<T>& w, const StepParams<T>& p, bool fuse_integrate_velocities, bool quads, hipStream_t s) {
    if (!w.n_bodies) return;
    uint32_t nb = (w.n_bodies + WS_THREADS - 1) / WS_THREADS;
    nb = ((nb + 7u) / 8u) * 8u;
    // quads: four lanes per body (k_body_warm_start_quad) -- the device closed loop, whose handle lists are in history order (2.78 -> 2.72 ms per settled cfg2
    // step).  With host-uploaded manifolds neighbouring bodies' manifolds are neighbours in the colour-major arrays and the lane-per-body form's gathers
    // coalesce: there the quads LOSE (cfg2 frozen 2 497 -> 2 382 substeps/s, same box), so the caller chooses.
    static const bool lane_per_body = avn_env("AVN_WS_LANE_PER_BODY") && avn_env("AVN_WS_LANE_PER_BODY")[0] == '1';   // (A/B: round 3's form everywhere; the results do not depend on it)
    if (quads && !lane_per_body && w.inc_slot) {
        uint32_t nq = (uint32_t)(((uint64_t)w.n_bodies * 4u + WSQ_THREADS - 1) / WSQ_THREADS);
        nq = ((nq + 7u) / 8u) * 8u;
        if (fuse_integrate_velocities) hipLaunchKernelGGL((k_body_warm_start_quad<T, true>), dim3(nq), dim3(WSQ_THREADS), 0, s, w, p);
        else hipLaunchKernelGGL((k_body_warm_start_quad<T, false>), dim3(nq), dim3(WSQ_THREADS), 0, s, w, p);
        return;
    }
    if (fuse_integrate_velocities) hipLaunchKernelGGL((k_body_warm_start<T, true>), dim3(nb), dim3(WS_THREADS), 0, s, w, p);
    else hipLaunchKernelGGL((k_body_warm_start<T, false>), dim3(nb), dim3(WS_THREADS), 0, s, w, p);
}
template <class T> uint32_t launch_contact_pass(const DW<T>& w, const StepParams<T>& p, int pass, const uint32_t* grid_blocks, const uint32_t* arg_offsets, const OverflowSchedule& ovf, hipStream_t s, uint32_t oct_mask) {
    switch (pass) {
        case PASS_WARM: return 0;  // warm start is body-centric: launch_body_warm_start
        case 4: return launch_pass<T, PASS_WARM>(w, p, grid_blocks, arg_offsets, ovf, s, 0u);   // PASS_WARM_START_COLORS: colour by colour
#ifdef AVN_MEASURE
        case 5: return launch_pass<T, PASS_SKELETON>(w, p, grid_blocks, arg_offsets, ovf, s, 0u);   // PASS_MEMORY_SKELETON (measurement aid, `make measure` only)
#endif
        case PASS_BIAS: return launch_pass<T, PASS_BIAS>(w, p, grid_blocks, arg_offsets, ovf, s, oct_mask);
        case PASS_RELAX: return launch_pass<T, PASS_RELAX>(w, p, grid_blocks, arg_offsets, ovf, s, oct_mask);
        default: return launch_pass<T, PASS_RESTITUTION>(w, p, grid_blocks, arg_offsets, ovf, s, 0u);
    }
}

#define INST(T)                                                                                             \
    template void launch_prepare_contact_constraints<T>(const DW<T>&, const StepParams<T>&, hipStream_t, bool, const RowsView<T>*);   \
    template void launch_store_contact_impulses<T>(const DW<T>&, hipStream_t);                              \
    template void launch_body_warm_start<T>(const DW<T>&, const StepParams<T>&, bool, bool, hipStream_t);   \
    template void launch_build_incidence_slots<T>(const DW<T>&, hipStream_t, bool);                               \
    template uint32_t launch_contact_pass<T>(const DW<T>&, const StepParams<T>&, int, const uint32_t*, const uint32_t*, const OverflowSchedule&, hipStream_t, uint32_t); \
    template void launch_overflow_flow<T>(const DW<T>&, const StepParams<T>&, int, const OverflowFlow&, uint32_t, uint32_t, hipStream_t);
INST(float)
INST(double)
#undef INST

}  // namespace avn
