// k_narrow.hip — narrow-phase kernels (one lane per collider pair).  Device functions: avn_narrow.h.
//
//   k_contact_manifolds_query   batch form of contact_query::contact_manifolds
//                               (reference collision/collider/parry/contact_query.rs:156-261)
#include "avn_kernels.h"
#include "avn_narrow.h"

// measurement cut-offs (AVN_NP_DEBUG, tools/np_phases.sh): compiled in only by `make measure`; the default library has no switch that changes results
#ifdef AVN_MEASURE
#define NP_DEBUG(p) ((p).np_debug)
#else
#define NP_DEBUG(p) 0u
#endif

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

// =====================================================================================================================
// Part 2: NarrowPhase::update_contacts over the device-resident contact table
// (reference collision/narrow_phase/system_param.rs:437-830; ContactManifold::{prune_points, match_contacts}
//  collision/contact_types/mod.rs:425-566; CoefficientCombine dynamics/rigid_body/physics_material.rs:28-36,205-214,372-380)
// =====================================================================================================================
namespace avn {

template <class T> struct NpPt {  // ContactPoint being built
    V3<T> anchor1, anchor2;
    T penetration, normal_speed, warm_n, warm_tx, warm_ty;
    uint32_t fid1, fid2;
};

template <class T> __device__ __forceinline__ T np_combine(T a, uint32_t ra, T b, uint32_t rb) {
    uint32_t rule = ra > rb ? ra : rb;
    if (rule == AVN_COMBINE_GEOMETRIC_MEAN) return sqrt_t(a * b);
    if (rule == AVN_COMBINE_MIN) return smin(a, b);
    if (rule == AVN_COMBINE_MULTIPLY) return a * b;
    if (rule == AVN_COMBINE_MAX) return smax(a, b);
    return (a + b) * T(0.5);
}

// The raw points of ONE pair in LDS: word w of point k of lane l at col[(w * AVN_NP_MAX_RAW + k) * NP_THREADS] with col = base + l -- every
// lane has its own column, neighbouring lanes neighbouring banks.  6 words per point: anchor1 (as contact_query returns it), penetration,
// the two feature ids; anchor2 = anchor1 + (position1 - position2) is recomputed where it is needed.
#define NP_THREADS 64          // the heavy kernel's workgroup (one wave): 24 KB (f32) / 48 KB (f64) of LDS
#define NP_LIGHT_THREADS 256   // the light kernel's workgroup: no LDS
#define NP_POINT_WORDS 6
template <class T> struct NpLdsSink {
    T* col;
    int cnt;
    static __device__ __forceinline__ NpLdsSink make(T* c) { return NpLdsSink{c, 0}; }
    __device__ __forceinline__ int n() const { return cnt; }
    __device__ __forceinline__ T& at(int w, int k) const { return col[(size_t)(w * AVN_NP_MAX_RAW + k) * NP_THREADS]; }
    __device__ __forceinline__ void put(V3<T> a1, T pen, uint32_t f1, uint32_t f2) {
        at(0, cnt) = a1.x; at(1, cnt) = a1.y; at(2, cnt) = a1.z; at(3, cnt) = pen; at(4, cnt) = bits_to_scalar(f1, T(0)); at(5, cnt) = bits_to_scalar(f2, T(0));
        ++cnt;
    }
    __device__ __forceinline__ void get(int k, V3<T>& a1, T& pen, uint32_t& f1, uint32_t& f2) const {
        a1 = {at(0, k), at(1, k), at(2, k)}; pen = at(3, k); f1 = scalar_to_bits(at(4, k)); f2 = scalar_to_bits(at(5, k));
    }
    __device__ __forceinline__ void move(int dst, int src) const {
#pragma unroll
        for (int w = 0; w < NP_POINT_WORDS; ++w) at(w, dst) = at(w, src);
    }
};
// The light kernel's sink: the paths it finishes itself (a ball against a ball or a cuboid) produce at most ONE raw point: registers.
template <class T> struct NpOneSink {
    V3<T> a1_;
    T pen_;
    uint32_t f1_, f2_;
    int cnt;
    static __device__ __forceinline__ NpOneSink make(T*) { return NpOneSink{vzero<T>(), T(0), 0u, 0u, 0}; }
    __device__ __forceinline__ int n() const { return cnt; }
    __device__ __forceinline__ void put(V3<T> a1, T pen, uint32_t f1, uint32_t f2) { a1_ = a1; pen_ = pen; f1_ = f1; f2_ = f2; ++cnt; }
    __device__ __forceinline__ void get(int, V3<T>& a1, T& pen, uint32_t& f1, uint32_t& f2) const { a1 = a1_; pen = pen_; f1 = f1_; f2 = f2_; }
    __device__ __forceinline__ void move(int, int) const {}
};
template <class T, bool HEAVY> struct NpSinkOf { typedef NpOneSink<T> type; };
template <class T> struct NpSinkOf<T, true> { typedef NpLdsSink<T> type; };

// One pair.  HEAVY = false: the whole update unless the pair is a cuboid-cuboid one that survives the SAT -- then nothing is written,
// *deferred = true and *axis holds the separating direction.  HEAVY = true: the deferred pair again, from the top (every input is re-read:
// nothing was written), with the SAT replaced by *axis.
// HS (host shapes, round 6: include/avian_mi355x.h "host shapes"): a pair with an AVN_SHAPE_HOST collider.  HEAVY = false: nothing is written, *deferred = true and
// *hq holds the query contact_manifolds_with_context gets (system_param.rs:700-712); HEAVY = true: the pair again from the top with the manifold the host returned
// (*hm) in the place of contact_query::contact_manifolds' -- margins, speculative filter, pruning, matching, status as for every other pair.
template <class T> struct NpHostQ { uint32_t contact_id, collider1, collider2, reserved; T position1[3], rotation1[4], position2[3], rotation2[4], max_contact_distance; };   // == avn_host_manifold_query_fNN
template <class T> struct NpHostM { uint32_t point_count; T normal[3]; T anchor1[3 * AVN_MAX_QUERY_POINTS]; T penetration[AVN_MAX_QUERY_POINTS]; uint32_t fid1[AVN_MAX_QUERY_POINTS], fid2[AVN_MAX_QUERY_POINTS]; };   // == avn_host_manifold_fNN
static_assert(sizeof(NpHostQ<float>) == sizeof(avn_host_manifold_query_f32) && sizeof(NpHostQ<double>) == sizeof(avn_host_manifold_query_f64), "host query layout");
static_assert(sizeof(NpHostM<float>) == sizeof(avn_host_manifold_f32) && sizeof(NpHostM<double>) == sizeof(avn_host_manifold_f64) && offsetof(NpHostM<double>, normal) == offsetof(avn_host_manifold_f64, normal), "host manifold layout");
static_assert(AVN_MAX_QUERY_POINTS <= AVN_NP_MAX_RAW, "a host manifold fits the raw-point column");
// collision hooks (include/avian_mi355x.h "collision hooks", NpHookList in avn_kernels.h): the ContactPair as CollisionHooks::modify_contacts sees / leaves it
template <class T> struct NpHookC { uint32_t contact_id, collider1, collider2, body1, body2, flags, touching, manifold_count, point_count, reserved;
                                    T normal[3], friction, restitution, tangent_velocity[3], anchor1[12], anchor2[12], penetration[4], normal_speed[4]; uint32_t fid1[4], fid2[4]; };   // == avn_hook_contact_fNN
static_assert(sizeof(NpHookC<float>) == sizeof(avn_hook_contact_f32) && sizeof(NpHookC<double>) == sizeof(avn_hook_contact_f64) && offsetof(NpHookC<double>, normal) == offsetof(avn_hook_contact_f64, normal) &&
              offsetof(NpHookC<float>, fid1) == offsetof(avn_hook_contact_f32, feature_id1), "hook record layout");
template <class T> struct NpHookCtx { NpHookC<T>* rec; uint32_t* count; uint32_t cap, phase; const NpHookC<T>* in; };   // in != nullptr: phase 3, this pair's record
template <class T> __host__ __device__ __forceinline__ NpHookCtx<T> np_hook_ctx(const NpHookList& l) { return NpHookCtx<T>{(NpHookC<T>*)l.records, l.count, l.cap, l.phase, nullptr}; }
template <class T, bool DENSE, bool HEAVY, bool HS = false>
__device__ void np_update_pair(const DW<T>& w, const BP<T>& bp, const CT<T>& ct, const StepParams<T>& p, const uint32_t c,
                               avn_contact_change* __restrict__ changes, uint32_t* __restrict__ n_changes, uint32_t* __restrict__ chg,
                               uint32_t* __restrict__ has, bool* deferred, V3<T>* axis, T* lds_col, NpHostQ<T>* hq = nullptr, const NpHostM<T>* hm = nullptr,
                               const NpHookCtx<T>* hk = nullptr) {
    uint4 meta = ct.meta[c];
    // (collision hooks: phase 2 visits only the rows phase 1 left pending; the bit itself is not a flag of the pair)
    const bool hk_on = HS && hk != nullptr && hk->count != nullptr;
    const bool hooked = HS && HEAVY && hk != nullptr && hk->in != nullptr;   // phase 3: the hook's answer is this pair's manifold
    if (hk_on && hk->phase == 2u && !(meta.z & AVN_CP_ROW_HOOK_PENDING)) return;
    if (HS) meta.z &= ~(uint32_t)AVN_CP_ROW_HOOK_PENDING;
    // (a free id; or a pair of ContactGraph::sleeping_pairs: update_contacts walks the ACTIVE pairs only, system_param.rs:437-475)
    if (DENSE && (!(meta.z & AVN_CP_ROW_USED) || (meta.z & AVN_CP_ROW_SLEEPING))) { chg[c] = 0u; has[c] = 0u; return; }
    const uint32_t slot1 = meta.x, slot2 = meta.y;
    uint32_t flags = meta.z;
    const uint32_t old_nman = meta.w & 0xFFu, old_pc = (meta.w >> 8) & 0xFFu;
    bool status = false;
    int32_t dcount = ct.dcount[c];
    uint32_t n_manifolds = old_nman, point_count = old_pc;
    const uint4 ci1 = bp.col_info[slot1], ci2 = bp.col_info[slot2];  // (entity, body, shape | cflags << 8, -)
    const uint2 ly1 = bp.col_layers[slot1], ly2 = bp.col_layers[slot2];
    const Vec4<T> mn1 = bp.aabb_min[slot1], mx1 = bp.aabb_max[slot1], mn2 = bp.aabb_min[slot2], mx2 = bp.aabb_max[slot2];
    const bool overlap = mn1.x <= mx2.x && mx1.x >= mn2.x && mn1.y <= mx2.y && mx1.y >= mn2.y && mn1.z <= mx2.z && mx1.z >= mn2.z;
    const bool interacts = (ly1.x & ly2.y) != 0 && (ly2.x & ly1.y) != 0;
    // (HS retry -- the query list overflowed and was grown: only the pairs that were handed to the host are visited again; they wrote nothing the first time)
    const bool hs_retry = HS && !HEAVY && hq->reserved != 0u;
    if (hs_retry && (((ci1.z & 0xFFu) != AVN_SHAPE_HOST && (ci2.z & 0xFFu) != AVN_SHAPE_HOST) || !overlap || !interacts)) return;
    if (!overlap || !interacts) {
        flags |= AVN_CP_DISJOINT_AABB;
        status = true;
    } else {
        const int body1 = (int)ci1.y, body2 = (int)ci2.y;
        const uint32_t bm1 = w.bmeta[body1], bm2 = w.bmeta[body2];
        const bool have1 = !(meta_flags(bm1) & AVN_BODY_DISABLED), have2 = !(meta_flags(bm2) & AVN_BODY_DISABLED);
        const Vec4<T> pos1 = w.pos[body1], pos2 = w.pos[body2], rot1 = w.rot[body1], rot2 = w.rot[body2];
        const bool is_static1 = have1 && meta_rb_type(bm1) == AVN_RB_STATIC, is_static2 = have2 && meta_rb_type(bm2) == AVN_RB_STATIC;
        const V3<T> bx1 = xyz<T>(pos1), bx2 = xyz<T>(pos2);   // the BODIES' poses ...
        const Q4<T> bq1 = quat<T>(rot1), bq2 = quat<T>(rot2);
        V3<T> x1 = bx1, x2 = bx2;                              // ... and the COLLIDERS': the same unless the collider is a child entity (HS instantiations, BP::col_lpos)
        Q4<T> q1 = bq1, q2 = bq2;
        if (HS && bp.col_lpos) { collider_pose<T>(bp, slot1, bx1, bq1, x1, q1); collider_pose<T>(bp, slot2, bx2, bq2, x2, q2); }
        // collider.position - body.position (zero for a collider on the body's entity)
        const V3<T> collider_offset1 = have1 ? x1 - bx1 : vzero<T>(), collider_offset2 = have2 ? x2 - bx2 : vzero<T>();
        // what only a pair WITH a manifold needs (centres of mass, angular velocities, materials): the heavy kernel loads it up front, the light
        // one -- whose cuboid pairs end apart or deferred -- only for a ball pair that touches (6 of its ~26 scattered 16-byte gathers per pair)
        V3<T> world_com1 = vzero<T>(), world_com2 = vzero<T>(), ang_vel1 = vzero<T>(), ang_vel2 = vzero<T>();
        T friction = T(0), restitution = T(0);
        auto load_manifold_inputs = [&]() {
            world_com1 = have1 ? qrot(bq1, xyz<T>(w.com[body1])) : vzero<T>(); world_com2 = have2 ? qrot(bq2, xyz<T>(w.com[body2])) : vzero<T>();
            ang_vel1 = have1 ? xyz<T>(w.avel[body1]) : vzero<T>(); ang_vel2 = have2 ? xyz<T>(w.avel[body2]) : vzero<T>();
            const Vec4<T> m1 = ct.col_mat[slot1], m2 = ct.col_mat[slot2];
            const uint32_t r1 = scalar_to_bits(m1.z), r2 = scalar_to_bits(m2.z);
            friction = np_combine<T>(m1.x, r1 & 0xFFu, m2.x, r2 & 0xFFu);
            restitution = np_combine<T>(m1.y, (r1 >> 8) & 0xFFu, m2.y, (r2 >> 8) & 0xFFu);
        };
        if (HEAVY) load_manifold_inputs();
        V3<T> lin_vel1 = have1 ? xyz<T>(w.lvel[body1]) : vzero<T>(), lin_vel2 = have2 ? xyz<T>(w.lvel[body2]) : vzero<T>();
        flags = (flags & ~(uint32_t)(AVN_CP_STATIC1 | AVN_CP_STATIC2)) | (is_static1 ? (uint32_t)AVN_CP_STATIC1 : 0u) | (is_static2 ? (uint32_t)AVN_CP_STATIC2 : 0u);
        const uint32_t cf1 = (ci1.z >> 8) & 0xFFu, cf2 = (ci2.z >> 8) & 0xFFu;
        const bool is_disabled = !have1 || !have2 || (cf1 & AVN_COLLIDER_SENSOR) || (cf2 & AVN_COLLIDER_SENSOR);
        if (!is_disabled && !(flags & AVN_CP_GENERATE_CONSTRAINTS)) { flags |= AVN_CP_STARTED_GENERATING_CONSTRAINTS; status = true; }
        flags = is_disabled ? (flags & ~(uint32_t)AVN_CP_GENERATE_CONSTRAINTS) : (flags | AVN_CP_GENERATE_CONSTRAINTS);
        const Vec4<T> he1 = bp.col_he[slot1], he2 = bp.col_he[slot2];  // (half_extents, collision_margin)
        const T collision_margin_sum = he1.w + he2.w;
        const T sp1 = bp.col_spec[slot1], sp2 = bp.col_spec[slot2];
        const T speculative_margin1 = sp1 >= T(0) ? sp1 : p.default_speculative_margin, speculative_margin2 = sp2 >= T(0) ? sp2 : p.default_speculative_margin;
        const T delta_secs = p.dt_adj;
        const T inv_delta_secs = T(1) / delta_secs;
        if (speculative_margin1 < Limits<T>::max) lin_vel1 = clamp_length_max(lin_vel1, speculative_margin1 * inv_delta_secs);
        if (speculative_margin2 < Limits<T>::max) lin_vel2 = clamp_length_max(lin_vel2, speculative_margin2 * inv_delta_secs);
        const V3<T> relative_linear_velocity = lin_vel2 - lin_vel1;
        const T effective_speculative_margin = delta_secs * length(relative_linear_velocity);
        const T max_contact_distance = smax(effective_speculative_margin, p.contact_tolerance) + collision_margin_sum;
        if (HS && !HEAVY && ((ci1.z & 0xFFu) == AVN_SHAPE_HOST || (ci2.z & 0xFFu) == AVN_SHAPE_HOST)) {   // the manifold is the host's: hand the query over, write nothing
            if (hk_on && hk->phase == 2u) return;   // (its record is collected by k_narrow_phase_host's second pass)
            hq->contact_id = c; hq->collider1 = ci1.x; hq->collider2 = ci2.x; hq->reserved = 0u;
            hq->position1[0] = x1.x; hq->position1[1] = x1.y; hq->position1[2] = x1.z; hq->rotation1[0] = q1.x; hq->rotation1[1] = q1.y; hq->rotation1[2] = q1.z; hq->rotation1[3] = q1.w;
            hq->position2[0] = x2.x; hq->position2[1] = x2.y; hq->position2[2] = x2.z; hq->rotation2[0] = q2.x; hq->rotation2[1] = q2.y; hq->rotation2[2] = q2.z; hq->rotation2[3] = q2.w;
            hq->max_contact_distance = max_contact_distance;
            *deferred = true;
            return;
        }
        const bool was_touching = flags & AVN_CP_TOUCHING;
        // old_manifolds = contacts.manifolds.clone(): only what match_contacts reads (constant indices: registers)
        V3<T> old_a1[AVN_MAX_MANIFOLD_POINTS], old_a2[AVN_MAX_MANIFOLD_POINTS];
        T old_wn[AVN_MAX_MANIFOLD_POINTS], old_wx[AVN_MAX_MANIFOLD_POINTS], old_wy[AVN_MAX_MANIFOLD_POINTS];
        uint2 old_fid[AVN_MAX_MANIFOLD_POINTS];
        const uint32_t old_n = old_nman ? old_pc : 0u;
        // (the heavy kernel issues these loads before the clipping that hides them; the light one only for a pair that turns out to touch --
        //  a ball pair: most of its pairs are cuboids, apart or deferred, and must not carry 44 registers of old points through the SAT)
        auto load_old_points = [&]() {
#pragma unroll
            for (uint32_t k = 0; k < AVN_MAX_MANIFOLD_POINTS; ++k) {
                old_a1[k] = vzero<T>(); old_a2[k] = vzero<T>(); old_wn[k] = T(0); old_wx[k] = T(0); old_wy[k] = T(0); old_fid[k] = make_uint2(0u, 0u);
                if (k < old_n) {
                    Vec4<T> oa = ct.a1(c, k), ob = ct.a2(c, k), ow = ct.w(c, k);
                    old_a1[k] = xyz<T>(oa); old_a2[k] = xyz<T>(ob); old_wn[k] = ow.x; old_wx[k] = ow.y; old_wy[k] = ow.z; old_fid[k] = ct.fid(c, k);
                }
            }
        };
        if (HEAVY) load_old_points();
        if (HEAVY && NP_DEBUG(p) == 3u) {   // timing cut-off: every input loaded, nothing computed or written
            T acc = ((x1.x + x2.y) + (q1.w + q2.x)) + ((world_com1.x + world_com2.y) + (lin_vel1.z + lin_vel2.x)) + ((ang_vel1.x + ang_vel2.y) + (friction + restitution)) + (sp1 + sp2);
#pragma unroll
            for (uint32_t k = 0; k < AVN_MAX_MANIFOLD_POINTS; ++k) acc += (old_a1[k].x + old_a2[k].y) + (old_wn[k] + bits_to_scalar(old_fid[k].x ^ old_fid[k].y, T(0)));
            if (acc == T(123456.75)) chg[c] = 7u;
            return;
        }
        // the raw points of the manifold go to this lane's LDS column (NpLdsSink): the only dynamically indexed storage of the update
        typedef typename NpSinkOf<T, HEAVY>::type Sink;
        Sink sink = Sink::make(lds_col);
        V3<T> normal = vzero<T>();
        bool defer = HEAVY;
        bool has_manifold;
        if (hooked) has_manifold = false;   // (nothing to compute: the pair's points are in the record)
        else if (HS && HEAVY && hm) {   // contact_manifolds_with_context answered by the host
            const uint32_t n = hm->point_count < (uint32_t)AVN_MAX_QUERY_POINTS ? hm->point_count : (uint32_t)AVN_MAX_QUERY_POINTS;
            normal = V3<T>{hm->normal[0], hm->normal[1], hm->normal[2]};
            for (uint32_t k = 0; k < n; ++k) sink.put(V3<T>{hm->anchor1[3 * k], hm->anchor1[3 * k + 1], hm->anchor1[3 * k + 2]}, hm->penetration[k], hm->fid1[k], hm->fid2[k]);
            has_manifold = n != 0u;
        } else
        has_manifold = NP_DEBUG(p) == 1u ? false
            : contact_manifolds_pair_sink<T, Sink, HEAVY ? 2 : 1>(ci1.z & 0xFFu, xyz<T>(he1), x1, q1, ci2.z & 0xFFu, xyz<T>(he2), x2, q2, max_contact_distance, sink, normal, &defer, axis);
        if (!HEAVY && defer) { if (NP_DEBUG(p) == 2u) defer = false; else { *deferred = true; return; } }
        if (!HEAVY && has_manifold) load_manifold_inputs();
        if (HEAVY && NP_DEBUG(p) == 4u) {   // timing cut-off: the manifold's raw points are in LDS, nothing converted or written
            if (has_manifold && normal.x + T(sink.cnt) == T(123456.75)) chg[c] = 7u;
            return;
        }
        const V3<T> d12 = x1 - x2;
        // one raw point -> the ContactPoint being built (system_param.rs:590-640), always from the same expressions
        auto build = [&](int k, NpPt<T>& pt) {
            V3<T> ra1; T rpen;
            sink.get(k, ra1, rpen, pt.fid1, pt.fid2);
            pt.anchor1 = (ra1 + collider_offset1) - world_com1;
            pt.anchor2 = ((ra1 + d12) + collider_offset2) - world_com2;
            pt.penetration = rpen + collision_margin_sum;
            V3<T> relative_velocity = (relative_linear_velocity + cross(ang_vel2, pt.anchor2)) - cross(ang_vel1, pt.anchor1);
            pt.normal_speed = dot(relative_velocity, normal);
            pt.warm_n = T(0); pt.warm_tx = T(0); pt.warm_ty = T(0);
        };
        int nk = 0;
        if (has_manifold)
            for (int k = 0; k < sink.cnt; ++k) {
                NpPt<T> pt;
                build(k, pt);
                const bool keep = -pt.penetration < effective_speculative_margin || (pt.normal_speed * delta_secs - pt.penetration < effective_speculative_margin);
                if (keep) { if (nk != k) sink.move(nk, k); ++nk; }   // compaction in place: nk <= k
            }
        int o0 = 0, o1 = 1, o2 = 2, o3 = 3;   // the (up to four) points to keep, in the reference's output order
        n_manifolds = 0; point_count = 0;
        if (nk > 0) {
            point_count = (uint32_t)nk;
            if (HEAVY && nk > 4) {   // ContactManifold::prune_points (contact_types/mod.rs:425-520): three passes over the kept points
                const T MIN_DISTANCE_SQUARED = T(1e-6);
                auto projected = [&](int i, T& pen_sq) {
                    V3<T> ra1; T rpen; uint32_t f1, f2;
                    sink.get(i, ra1, rpen, f1, f2);
                    const V3<T> a1 = (ra1 + collider_offset1) - world_com1;
                    const T pen = rpen + collision_margin_sum;
                    pen_sq = smax(pen * pen, MIN_DISTANCE_SQUARED);
                    return a1 - normal * dot(a1, normal);
                };
                int p1 = 0; T value = -Limits<T>::max; V3<T> proj1 = vzero<T>();
                for (int i = 0; i < nk; ++i) {
                    T ps; const V3<T> pr = projected(i, ps);
                    const T v = smax(length_squared(pr), MIN_DISTANCE_SQUARED) * ps;
                    if (v > value) { value = v; p1 = i; proj1 = pr; }
                }
                int p2 = -1; T max_distance = -Limits<T>::max; V3<T> proj2 = vzero<T>();
                for (int i = 0; i < nk; ++i) {
                    if (i == p1) continue;
                    T ps; const V3<T> pr = projected(i, ps);
                    const T v = smax(length_squared(pr - proj1), MIN_DISTANCE_SQUARED) * ps;
                    if (v > max_distance) { max_distance = v; p2 = i; proj2 = pr; }
                }
                int p3 = -1, p4 = -1; T min_value = T(0), max_value = T(0);
                const V3<T> perp = cross(proj2 - proj1, normal);
                for (int i = 0; i < nk; ++i) {
                    if (i == p1 || i == p2) continue;
                    T ps; const V3<T> pr = projected(i, ps);
                    const T v = dot(perp, pr - proj1);
                    if (v < min_value) { min_value = v; p3 = i; }
                    else if (v > max_value) { max_value = v; p4 = i; }
                }
                int k = 1;          // output order: p1, [p3], p2, [p4]
                o0 = p1;
                if (p3 >= 0) { o1 = p3; k = 2; }
                if (k == 1) o1 = p2; else o2 = p2;
                ++k;
                if (p4 >= 0) { if (k == 2) o2 = p4; else o3 = p4; ++k; }
                point_count = (uint32_t)k;
            }
            n_manifolds = 1;
        }
        if (HEAVY && NP_DEBUG(p) == 5u) {   // timing cut-off: points kept / pruned, nothing matched or written
            if (T(o0 + o1 + o2 + o3) + T(point_count) == T(123456.75)) chg[c] = 7u;
            return;
        }
        bool touching = n_manifolds != 0;
        V3<T> tangent_velocity = vzero<T>();   // system_param.rs:722-729; only a hook changes it
        // CollisionHooks::modify_contacts (system_param.rs:770-778): `touching && flags.contains(MODIFY_CONTACTS)`
        if (hk_on && !hooked && touching && (flags & AVN_CP_MODIFY_CONTACTS)) {
            if (hk->phase != 2u) {   // phase 1: counted and left pending; the row is otherwise untouched
                atomicAdd(hk->count, 1u);
                ct.meta[c].z = meta.z | AVN_CP_ROW_HOOK_PENDING;
                return;
            }
            // (the bit goes here: a row whose id was recycled this step is in the range of the launch over the old rows AND in the list of the new ones)
            ct.meta[c].z = meta.z;
            const uint32_t slot = atomicAdd(hk->count, 1u);
            if (slot < hk->cap) {
                NpHookC<T>& r = hk->rec[slot];
                r.contact_id = c; r.collider1 = ci1.x; r.collider2 = ci2.x; r.body1 = ci1.y; r.body2 = ci2.y; r.flags = flags & 0xFFFFu;
                r.touching = 1u; r.manifold_count = 1u; r.point_count = point_count; r.reserved = 0u;
                r.normal[0] = normal.x; r.normal[1] = normal.y; r.normal[2] = normal.z; r.friction = friction; r.restitution = restitution;
                r.tangent_velocity[0] = T(0); r.tangent_velocity[1] = T(0); r.tangent_velocity[2] = T(0);
                for (uint32_t k = 0; k < (uint32_t)AVN_MAX_MANIFOLD_POINTS; ++k) {
                    NpPt<T> pt;
                    pt.anchor1 = vzero<T>(); pt.anchor2 = vzero<T>(); pt.penetration = T(0); pt.normal_speed = T(0); pt.fid1 = 0u; pt.fid2 = 0u;
                    if (k < point_count) build(k == 0 ? o0 : (k == 1 ? o1 : (k == 2 ? o2 : o3)), pt);
                    r.anchor1[3 * k] = pt.anchor1.x; r.anchor1[3 * k + 1] = pt.anchor1.y; r.anchor1[3 * k + 2] = pt.anchor1.z;
                    r.anchor2[3 * k] = pt.anchor2.x; r.anchor2[3 * k + 1] = pt.anchor2.y; r.anchor2[3 * k + 2] = pt.anchor2.z;
                    r.penetration[k] = pt.penetration; r.normal_speed[k] = pt.normal_speed; r.fid1[k] = pt.fid1; r.fid2[k] = pt.fid2;
                }
            }
            return;
        }
        if (hooked) {   // the pair as the hook left it; !touching: manifolds.clear()
            const NpHookC<T>& r = *hk->in;
            touching = r.touching != 0u;
            n_manifolds = touching && r.manifold_count ? 1u : 0u;
            point_count = n_manifolds ? (r.point_count < (uint32_t)AVN_MAX_MANIFOLD_POINTS ? r.point_count : (uint32_t)AVN_MAX_MANIFOLD_POINTS) : 0u;
            normal = V3<T>{r.normal[0], r.normal[1], r.normal[2]}; friction = r.friction; restitution = r.restitution;
            tangent_velocity = V3<T>{r.tangent_velocity[0], r.tangent_velocity[1], r.tangent_velocity[2]};
        }
        flags = touching ? (flags | AVN_CP_TOUCHING) : (flags & ~(uint32_t)AVN_CP_TOUCHING);
        if (touching && (!hooked || n_manifolds)) {
            if (!HEAVY) load_old_points();
            const T thr = T(0.1) * p.length_unit;
            const T thr2 = thr * thr;
            ct.n(c) = make4<T>(normal, friction);
            ct.tv(c) = make4<T>(tangent_velocity, restitution);
            for (uint32_t k = 0; k < point_count; ++k) {
                NpPt<T> pt;
                if (hooked) {
                    const NpHookC<T>& r = *hk->in;
                    pt.anchor1 = V3<T>{r.anchor1[3 * k], r.anchor1[3 * k + 1], r.anchor1[3 * k + 2]}; pt.anchor2 = V3<T>{r.anchor2[3 * k], r.anchor2[3 * k + 1], r.anchor2[3 * k + 2]};
                    pt.penetration = r.penetration[k]; pt.normal_speed = r.normal_speed[k]; pt.fid1 = r.fid1[k]; pt.fid2 = r.fid2[k];
                    pt.warm_n = T(0); pt.warm_tx = T(0); pt.warm_ty = T(0);
                } else
                build(k == 0 ? o0 : (k == 1 ? o1 : (k == 2 ? o2 : o3)), pt);
                if (p.match_contacts && old_n) {  // ContactManifold::match_contacts
                    bool matched = false;
#pragma unroll
                    for (uint32_t j = 0; j < AVN_MAX_MANIFOLD_POINTS; ++j) {
                        if (j < old_n && !matched) {
                            const bool unknown = pt.fid1 == 0u || pt.fid2 == 0u;
                            if (((pt.fid1 == old_fid[j].x && pt.fid2 == old_fid[j].y) || (pt.fid2 == old_fid[j].x && pt.fid1 == old_fid[j].y)) ||
                                (unknown && (length_squared(pt.anchor1 - old_a1[j]) < thr2 && length_squared(pt.anchor2 - old_a2[j]) < thr2)) ||
                                (length_squared(pt.anchor1 - old_a2[j]) < thr2 && length_squared(pt.anchor2 - old_a1[j]) < thr2)) {
                                pt.warm_n = old_wn[j]; pt.warm_tx = old_wx[j]; pt.warm_ty = old_wy[j];
                                matched = true;
                            }
                        }
                    }
                }
                ct.a1(c, k) = make4<T>(pt.anchor1, pt.penetration);
                ct.a2(c, k) = make4<T>(pt.anchor2, pt.normal_speed);
                ct.w(c, k) = make4<T>(pt.warm_n, pt.warm_tx, pt.warm_ty, T(0));  // ContactPoint::new: normal_impulse = 0
                ct.fid(c, k) = make_uint2(pt.fid1, pt.fid2);
            }
        }
        dcount = (int32_t)n_manifolds - (int32_t)old_nman;
        if (touching && !was_touching) { flags |= AVN_CP_STARTED_TOUCHING; status = true; }
        else if (!touching && was_touching) { flags |= AVN_CP_STOPPED_TOUCHING; status = true; }
        else if (dcount != 0) status = true;
    }
    if (DENSE) {
        chg[c] = status ? ((flags & 0xFFFFu) | (n_manifolds << 16) | (((uint32_t)(dcount + 128) & 0xFFu) << 24)) : 0u;
        has[c] = status ? 1u : 0u;
        if (flags & AVN_CP_DISJOINT_AABB) atomicAdd(n_changes, 1u);   // rows to remove (here `n_changes` is the removal counter)
    } else if (status) {
        uint32_t slot = atomicAdd(n_changes, 1u);
        avn_contact_change ch;
        ch.contact_id = c; ch.flags = flags & ~(uint32_t)AVN_CP_ROW_USED; ch.manifold_count_change = dcount; ch.manifold_count = n_manifolds;
        changes[slot] = ch;
    }
    // the transient status flags are handled (and cleared) by the host's status processing, system_param.rs:141-389
    const uint32_t kept_flags = flags & ~(uint32_t)(AVN_CP_STARTED_TOUCHING | AVN_CP_STOPPED_TOUCHING | AVN_CP_STARTED_GENERATING_CONSTRAINTS);
    ct.meta[c] = make_uint4(slot1, slot2, kept_flags, n_manifolds | (point_count << 8));
    ct.dcount[c] = dcount;
}

// DENSE = false: the pairs are active[0 .. n_active), changes are appended to `changes` in arbitrary order (the host sorts them).
// DENSE = true (device closed loop, k_graph.hip): every row id < n_active with AVN_CP_ROW_USED is a pair; a row's change is left in
// chg[id] / has[id] (`changes` / `n_changes` reinterpreted), so that a scan over the rows numbers the changes in ascending ContactId.
//
// Two kernels.  k_narrow_phase, one lane per pair: everything that is cheap -- AABB / layer tests, ball paths, and for two cuboids the SAT,
// after which ~60 % of a settled pile's AABB-overlapping pairs are done (apart).  It holds no LDS and half the registers of the whole
// update, so it runs four waves deep and leaves a CU's LDS to whatever shares the chip (the sweep, when the narrow phase overlaps the broad
// phase).  The survivors' (row, separating axis) go to a list in HBM (wave-aggregated append: order arbitrary, the outputs are per row).
// k_narrow_phase_heavy, one lane per SURVIVOR: support faces, clipping, point conversion, pruning, matching, with its raw points in an LDS
// column per lane -- every wave full (round 2 ran this half inside the first kernel's workgroups: 60 % of the lanes of a wave that existed
// per 128 pairs, one such wave per workgroup while 48 KB of LDS per workgroup capped the CU at three of them).  Its grid is sized for the
// worst case (every pair survives) and the workgroups beyond the list leave at once; the last workgroup to finish clears the list's counter
// for the next launch, which saves a memset launch per narrow phase.
// The survivor list is NP_LISTS lists: a workgroup of the first kernel appends to list (blockIdx mod NP_LISTS), whose counter has a cache line
// of its own -- one list would mean one atomic per wave on ONE address: measured 6.4 ns each, 125 us for cfg2's 19 k waves, the largest single
// item of the kernel.  List k owns np_row[k * seg .. (k + 1) * seg) with seg = NP_LIGHT_THREADS * ceil(workgroups / NP_LISTS): it cannot overflow.
#define NP_LISTS 64u
#define NP_CTR_STRIDE 16u   // words between the counters of two lists: (entries, workgroups of the second kernel that are done, -, ...)
__host__ __device__ __forceinline__ uint32_t np_list_segment(uint32_t n_pairs) {
    const uint32_t wgs = (n_pairs + NP_LIGHT_THREADS - 1) / NP_LIGHT_THREADS;
    return NP_LIGHT_THREADS * ((wgs + NP_LISTS - 1) / NP_LISTS);
}
// HS: the world holds AVN_SHAPE_HOST colliders -- their pairs' queries are appended to hostq[0 .. *hostq_n) (arbitrary order: the host sorts by contact id)
template <class T, bool DENSE, bool HS = false>
__global__ __launch_bounds__(NP_LIGHT_THREADS) void k_narrow_phase(DW<T> w, BP<T> bp, CT<T> ct, StepParams<T> p, const uint32_t* __restrict__ active, uint32_t n_active,
                                                                   avn_contact_change* __restrict__ changes, uint32_t* __restrict__ n_changes, uint32_t* __restrict__ chg,
                                                                   uint32_t* __restrict__ has, uint32_t n_list, uint32_t range_base, NpHostQ<T>* __restrict__ hostq = nullptr,
                                                                   uint32_t* __restrict__ hostq_n = nullptr, uint32_t hostq_cap = 0u, uint32_t host_only = 0u, NpHookCtx<T> hk = NpHookCtx<T>()) {
    const uint32_t a = blockIdx.x * NP_LIGHT_THREADS + threadIdx.x;
    bool deferred = false;
    V3<T> axis = vzero<T>();
    uint32_t c = 0;
    if (HS) {
        NpHostQ<T> q;
        q.contact_id = 0xFFFFFFFFu; q.reserved = host_only;
        if (a < n_active) {
            c = !DENSE ? active[a] : (active || n_list || range_base) ? (a < n_list ? active[a] : range_base + (a - n_list)) : a;
            np_update_pair<T, DENSE, false, true>(w, bp, ct, p, c, changes, n_changes, chg, has, &deferred, &axis, nullptr, &q, nullptr, &hk);
        }
        if (deferred && q.contact_id != 0xFFFFFFFFu && hostq_n) {   // a host pair (few of them: one atomic each); past the capacity only the count grows and the host retries larger
            const uint32_t k = atomicAdd(hostq_n, 1u);
            if (k < hostq_cap) hostq[k] = q;
            deferred = false;
        }
    } else
    if (a < n_active) {
        // DENSE with a row list (the rows a step ADDED, narrow phase overlapped with the broad phase): ids list[0 .. n_list), then the
        // contiguous fresh ids range_base ..; DENSE without one: every row id; sparse: the active list
        c = !DENSE ? active[a] : (active || n_list || range_base) ? (a < n_list ? active[a] : range_base + (a - n_list)) : a;
        np_update_pair<T, DENSE, false>(w, bp, ct, p, c, changes, n_changes, chg, has, &deferred, &axis, nullptr);
    }
    const uint64_t m = __ballot(deferred);
    if (m) {
        const uint32_t list = blockIdx.x % NP_LISTS;
        const uint32_t lane = threadIdx.x & 63u, leader = (uint32_t)__ffsll((unsigned long long)m) - 1u;
        uint32_t base = 0;
        if (lane == leader) base = atomicAdd(ct.np_ctr + list * NP_CTR_STRIDE, (uint32_t)__popcll(m));
        base = __shfl(base, (int)leader);
        if (deferred) {
            const size_t k = (size_t)list * np_list_segment(n_active) + base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
            ct.np_row[k] = c;
            ct.np_axis[3 * k] = axis.x; ct.np_axis[3 * k + 1] = axis.y; ct.np_axis[3 * k + 2] = axis.z;
        }
    }
}
// workgroup (chunk, list) = (blockIdx / NP_LISTS, blockIdx mod NP_LISTS): the populated chunks come first in the grid.  The last workgroup of
// a list to finish clears the list's counters for the next launch (every workgroup that takes part has read the count before it reports);
// the workgroups beyond a list's end leave without touching anything.
template <class T, bool DENSE, bool HS = false>
__global__ __launch_bounds__(NP_THREADS) void k_narrow_phase_heavy(DW<T> w, BP<T> bp, CT<T> ct, StepParams<T> p, avn_contact_change* __restrict__ changes,
                                                                   uint32_t* __restrict__ n_changes, uint32_t* __restrict__ chg, uint32_t* __restrict__ has, uint32_t n_pairs,
                                                                   NpHookCtx<T> hk = NpHookCtx<T>()) {
    __shared__ T s_pts[NP_POINT_WORDS * AVN_NP_MAX_RAW * NP_THREADS];   // f32: 24 KB, f64: 48 KB
    const uint32_t list = blockIdx.x % NP_LISTS, chunk = blockIdx.x / NP_LISTS;
    uint32_t* ctr = ct.np_ctr + list * NP_CTR_STRIDE;
    const uint32_t n = __hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (chunk * NP_THREADS >= n) return;
    const uint32_t i = chunk * NP_THREADS + threadIdx.x;
    if (i < n) {
        const size_t k = (size_t)list * np_list_segment(n_pairs) + i;
        V3<T> axis{ct.np_axis[3 * k], ct.np_axis[3 * k + 1], ct.np_axis[3 * k + 2]};
        bool deferred = false;
        if (HS) np_update_pair<T, DENSE, true, true>(w, bp, ct, p, ct.np_row[k], changes, n_changes, chg, has, &deferred, &axis, s_pts + threadIdx.x, nullptr, nullptr, &hk);
        else
        np_update_pair<T, DENSE, true>(w, bp, ct, p, ct.np_row[k], changes, n_changes, chg, has, &deferred, &axis, s_pts + threadIdx.x);
    }
    __syncthreads();
    if (threadIdx.x == 0 && atomicAdd(ctr + 1, 1u) == (n + NP_THREADS - 1) / NP_THREADS - 1u) { atomicExch(ctr, 0u); atomicExch(ctr + 1, 0u); }
}
// the pairs the host answered: one lane each, its raw points in the lane's LDS column like a surviving cuboid pair's
template <class T, bool DENSE>
__global__ __launch_bounds__(NP_THREADS) void k_narrow_phase_host(DW<T> w, BP<T> bp, CT<T> ct, StepParams<T> p, avn_contact_change* __restrict__ changes, uint32_t* __restrict__ n_changes,
                                                                  uint32_t* __restrict__ chg, uint32_t* __restrict__ has, const NpHostQ<T>* __restrict__ q, const NpHostM<T>* __restrict__ m, uint32_t n,
                                                                  NpHookCtx<T> hk) {
    __shared__ T s_pts[NP_POINT_WORDS * AVN_NP_MAX_RAW * NP_THREADS];
    const uint32_t i = blockIdx.x * NP_THREADS + threadIdx.x;
    if (i >= n) return;
    bool deferred = false;
    V3<T> axis = vzero<T>();
    np_update_pair<T, DENSE, true, true>(w, bp, ct, p, q[i].contact_id, changes, n_changes, chg, has, &deferred, &axis, s_pts + threadIdx.x, nullptr, m + i, &hk);
}
// collision hooks, phase 3: the pairs CollisionHooks::modify_contacts answered, one lane each -- the remainder of update_contacts from the returned record
template <class T, bool DENSE>
__global__ __launch_bounds__(NP_THREADS) void k_narrow_phase_hooked(DW<T> w, BP<T> bp, CT<T> ct, StepParams<T> p, avn_contact_change* __restrict__ changes, uint32_t* __restrict__ n_changes,
                                                                    uint32_t* __restrict__ chg, uint32_t* __restrict__ has, const NpHookC<T>* __restrict__ rec, uint32_t n) {
    __shared__ T s_pts[NP_POINT_WORDS * AVN_NP_MAX_RAW * NP_THREADS];
    const uint32_t i = blockIdx.x * NP_THREADS + threadIdx.x;
    if (i >= n) return;
    bool deferred = false;
    V3<T> axis = vzero<T>();
    const NpHookCtx<T> hk{nullptr, nullptr, 0u, 3u, rec + i};
    np_update_pair<T, DENSE, true, true>(w, bp, ct, p, rec[i].contact_id, changes, n_changes, chg, has, &deferred, &axis, s_pts + threadIdx.x, nullptr, nullptr, &hk);
}
size_t np_survivor_list_slack() { return (size_t)NP_LISTS * NP_LIGHT_THREADS; }
size_t np_survivor_counter_bytes() { return (size_t)NP_LISTS * NP_CTR_STRIDE * sizeof(uint32_t); }
template <class T, bool DENSE>
static void launch_np_heavy(const DW<T>& w, const BP<T>& bp, const CT<T>& ct, const StepParams<T>& p, avn_contact_change* changes, uint32_t* n_changes, uint32_t* chg, uint32_t* has,
                            uint32_t n_pairs, hipStream_t st, const NpHookList& hook = NpHookList(), bool hs = false) {
    const uint32_t chunks = np_list_segment(n_pairs) / NP_THREADS;
    if (hook.count || hs) hipLaunchKernelGGL((k_narrow_phase_heavy<T, DENSE, true>), dim3(chunks * NP_LISTS), dim3(NP_THREADS), 0, st, w, bp, ct, p, changes, n_changes, chg, has, n_pairs, np_hook_ctx<T>(hook));
    else
    hipLaunchKernelGGL((k_narrow_phase_heavy<T, DENSE>), dim3(chunks * NP_LISTS), dim3(NP_THREADS), 0, st, w, bp, ct, p, changes, n_changes, chg, has, n_pairs, NpHookCtx<T>());
}

template <class T>
__global__ __launch_bounds__(256) void k_init_contact_rows(CT<T> ct, const uint32_t* __restrict__ ids, const uint32_t* __restrict__ s1, const uint32_t* __restrict__ s2,
                                                           const uint32_t* __restrict__ pf, uint32_t n) {
    uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    uint32_t f = pf[i];
    uint32_t flags = ((f & AVN_PAIR_GENERATE_CONSTRAINTS) ? (uint32_t)AVN_CP_GENERATE_CONSTRAINTS : 0u) | ((f & AVN_PAIR_MODIFY_CONTACTS) ? (uint32_t)AVN_CP_MODIFY_CONTACTS : 0u) |
                     ((f & AVN_PAIR_CONTACT_EVENTS) ? (uint32_t)AVN_CP_CONTACT_EVENTS : 0u);
    ct.meta[ids[i]] = make_uint4(s1[i], s2[i], flags | AVN_CP_ROW_USED, 0u);
    ct.dcount[ids[i]] = 0;
}
template <class T>
__global__ __launch_bounds__(256) void k_clear_contact_rows(CT<T> ct, const uint32_t* __restrict__ ids, uint32_t n) {
    uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    ct.meta[ids[i]] = make_uint4(0u, 0u, 0u, 0u);
    ct.dcount[ids[i]] = 0;
}
// (k_gather_manifolds -- rows -> colour-major arrays through the handle lists -- is gone since round 4: k_prepare_contact_constraints<T, ROWS> reads the rows itself)
template <class T>
__global__ __launch_bounds__(256) void k_scatter_impulses(DW<T> w, CT<T> ct, const uint32_t* __restrict__ handles) {
    uint32_t m = blockIdx.x * 256 + threadIdx.x;
    if (m >= w.n_manifolds) return;
    const uint32_t c = handles[m];
    const uint32_t np = scalar_to_bits(w.c_h1[m].w) & 7u;  // points the constraint has (0 = constraint absent: nothing stored)
    for (uint32_t k = 0; k < np; ++k) ct.w(c, k) = w.mp_w[(size_t)k * w.m_stride + m];
}
template <class T>
__global__ __launch_bounds__(256) void k_unpack_contacts(CT<T> ct, const uint32_t* __restrict__ ids, uint32_t n, ContactsStage<T> o) {
    uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint32_t c = ids[i];
    uint4 meta = ct.meta[c];
    if (!(meta.z & AVN_CP_ROW_USED)) meta = make_uint4(0u, 0u, 0u, 0u);   // a free id: reads back as an empty pair (flags 0, no points), never as stale data
    const uint32_t np = (meta.w & 0xFFu) ? ((meta.w >> 8) & 0xFFu) : 0u;
    if (o.flags) o.flags[i] = meta.z & ~(uint32_t)(AVN_CP_ROW_USED | AVN_CP_ROW_SLEEPING | AVN_CP_ROW_HOOK_PENDING);
    if (o.point_count) o.point_count[i] = (uint8_t)np;
    Vec4<T> n4 = np ? ct.n(c) : make4<T>(0, 0, 0, 0), tv = np ? ct.tv(c) : make4<T>(0, 0, 0, 0);
    st3(o.normal, i, xyz<T>(n4));
    if (o.friction) o.friction[i] = n4.w;
    if (o.restitution) o.restitution[i] = tv.w;
    for (uint32_t k = 0; k < AVN_MAX_MANIFOLD_POINTS; ++k) {
        const size_t d = 4 * (size_t)i + k;
        const bool live = k < np;
        Vec4<T> a1 = live ? ct.a1(c, k) : make4<T>(0, 0, 0, 0), a2 = live ? ct.a2(c, k) : make4<T>(0, 0, 0, 0), ww = live ? ct.w(c, k) : make4<T>(0, 0, 0, 0);
        uint2 f = live ? ct.fid(c, k) : make_uint2(0u, 0u);
        st3(o.anchor1, d, xyz<T>(a1)); st3(o.anchor2, d, xyz<T>(a2));
        if (o.penetration) o.penetration[d] = a1.w;
        if (o.normal_speed) o.normal_speed[d] = a2.w;
        if (o.warm_n) o.warm_n[d] = ww.x;
        if (o.warm_t) { o.warm_t[2 * d] = ww.y; o.warm_t[2 * d + 1] = ww.z; }
        if (o.normal_impulse) o.normal_impulse[d] = ww.w;
        if (o.feature_id1) o.feature_id1[d] = f.x;
        if (o.feature_id2) o.feature_id2[d] = f.y;
    }
}

// avn_contacts_upload: the inverse record copy (rows that migrate between worlds); slots and manifold_count_change stay.
// A row must exist (avn_contact_pairs_add, or the closed loop's own k_pg_add_pairs): an id whose row is not live is skipped and
// reported through `error` (bit 2 of the closed loop's error word) -- writing it would create a pair no list knows about.
template <class T>
__global__ __launch_bounds__(256) void k_pack_contacts(CT<T> ct, const uint32_t* __restrict__ ids, uint32_t n, ContactsStage<T> in, uint32_t* error) {
    uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint32_t c = ids[i];
    uint4 meta = ct.meta[c];
    if (!(meta.z & AVN_CP_ROW_USED)) { if (error) atomicOr(error, 4u); return; }
    const uint32_t np = min((uint32_t)in.point_count[i], (uint32_t)AVN_MAX_MANIFOLD_POINTS);
    meta.z = (in.flags[i] & ~(uint32_t)AVN_CP_ROW_USED) | AVN_CP_ROW_USED;
    meta.w = (np ? 1u : 0u) | (np << 8);
    ct.meta[c] = meta;
    V3<T> nn = ld3(in.normal, i);
    ct.n(c) = make4<T>(nn.x, nn.y, nn.z, in.friction[i]);
    ct.tv(c) = make4<T>(T(0), T(0), T(0), in.restitution[i]);
    for (uint32_t k = 0; k < np; ++k) {
        const size_t s = 4 * (size_t)i + k;
        V3<T> a1 = ld3(in.anchor1, s), a2 = ld3(in.anchor2, s);
        ct.a1(c, k) = make4<T>(a1.x, a1.y, a1.z, in.penetration[s]);
        ct.a2(c, k) = make4<T>(a2.x, a2.y, a2.z, in.normal_speed[s]);
        ct.w(c, k) = make4<T>(in.warm_n[s], in.warm_t[2 * s], in.warm_t[2 * s + 1], in.normal_impulse[s]);
        ct.fid(c, k) = make_uint2(in.feature_id1[s], in.feature_id2[s]);
    }
}

// avn_colliders_upload with a changed slot assignment while rows are live: meta.x / meta.y are collider slots.  apply = 0 counts the live
// rows that name a slot without a successor (nothing is written), apply = 1 renumbers.
template <class T>
__global__ __launch_bounds__(256) void k_remap_row_slots(CT<T> ct, const uint32_t* __restrict__ map, uint32_t n_old, uint32_t apply, uint32_t* n_orphans) {
    uint32_t c = blockIdx.x * 256 + threadIdx.x;
    if (c >= ct.cap) return;
    uint4 meta = ct.meta[c];
    if (!(meta.z & AVN_CP_ROW_USED)) return;
    const uint32_t a = meta.x < n_old ? map[meta.x] : 0xFFFFFFFFu, b = meta.y < n_old ? map[meta.y] : 0xFFFFFFFFu;
    if (!apply) { if (a == 0xFFFFFFFFu || b == 0xFFFFFFFFu) atomicAdd(n_orphans, 1u); return; }
    meta.x = a; meta.y = b;
    ct.meta[c] = meta;
}
template <class T> void launch_remap_row_slots(const CT<T>& ct, const uint32_t* map, uint32_t n_old, uint32_t apply, uint32_t* n_orphans, hipStream_t st) {
    if (ct.cap) hipLaunchKernelGGL(k_remap_row_slots<T>, dim3((ct.cap + 255) / 256), dim3(256), 0, st, ct, map, n_old, apply, n_orphans);
}
template <class T>
__global__ __launch_bounds__(256) void k_rows_set_sleeping(CT<T> ct, const uint32_t* __restrict__ cids, uint32_t n, uint32_t sleeping) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    uint4 meta = ct.meta[cids[i]];
    meta.z = sleeping ? (meta.z | AVN_CP_ROW_SLEEPING) : (meta.z & ~(uint32_t)AVN_CP_ROW_SLEEPING);
    ct.meta[cids[i]] = meta;
}
template <class T> void launch_rows_set_sleeping(const CT<T>& ct, const uint32_t* cids, uint32_t n, uint32_t sleeping, hipStream_t st) {
    if (n) hipLaunchKernelGGL(k_rows_set_sleeping<T>, dim3((n + 255) / 256), dim3(256), 0, st, ct, cids, n, sleeping);
}
template <class T> void launch_init_contact_rows(const CT<T>& ct, const uint32_t* ids, const uint32_t* s1, const uint32_t* s2, const uint32_t* pf, uint32_t n, hipStream_t st) {
    if (n) hipLaunchKernelGGL(k_init_contact_rows<T>, dim3((n + 255) / 256), dim3(256), 0, st, ct, ids, s1, s2, pf, n);
}
template <class T> void launch_clear_contact_rows(const CT<T>& ct, const uint32_t* ids, uint32_t n, hipStream_t st) {
    if (n) hipLaunchKernelGGL(k_clear_contact_rows<T>, dim3((n + 255) / 256), dim3(256), 0, st, ct, ids, n);
}
template <class T> void launch_narrow_phase(const DW<T>& w, const BP<T>& bp, const CT<T>& ct, const StepParams<T>& p, const uint32_t* active, uint32_t n_active,
                                            avn_contact_change* changes, uint32_t* n_changes, hipStream_t st, const NpHostList& hl) {
    if (!hl.host_only && hl.hook.phase != 2u) (void)hipMemsetAsync(n_changes, 0, sizeof(uint32_t), st);
    if (!n_active) return;
    if (hl.any()) hipLaunchKernelGGL((k_narrow_phase<T, false, true>), dim3((n_active + NP_LIGHT_THREADS - 1) / NP_LIGHT_THREADS), dim3(NP_LIGHT_THREADS), 0, st, w, bp, ct, p, active, n_active, changes, n_changes, nullptr, nullptr, 0u, 0u, (NpHostQ<T>*)hl.queries, hl.count, hl.cap, hl.host_only, np_hook_ctx<T>(hl.hook));
    else
    hipLaunchKernelGGL((k_narrow_phase<T, false>), dim3((n_active + NP_LIGHT_THREADS - 1) / NP_LIGHT_THREADS), dim3(NP_LIGHT_THREADS), 0, st, w, bp, ct, p, active, n_active, changes, n_changes, nullptr, nullptr, 0u, 0u);
    if (hl.host_only) return;   // (a retry hands no cuboid pair over: nothing for the second kernel)
    launch_np_heavy<T, false>(w, bp, ct, p, changes, n_changes, nullptr, nullptr, n_active, st, hl.hook, hl.any());
}
template <class T> void launch_narrow_phase_dense(const DW<T>& w, const BP<T>& bp, const CT<T>& ct, const StepParams<T>& p, uint32_t n_rows, uint32_t* chg, uint32_t* has,
                                                  uint32_t* n_remove, hipStream_t st, bool reset_counter, const NpHostList& hl) {
    if (reset_counter) (void)hipMemsetAsync(n_remove, 0, sizeof(uint32_t), st);
    if (!n_rows) return;
    if (hl.any()) hipLaunchKernelGGL((k_narrow_phase<T, true, true>), dim3((n_rows + NP_LIGHT_THREADS - 1) / NP_LIGHT_THREADS), dim3(NP_LIGHT_THREADS), 0, st, w, bp, ct, p, nullptr, n_rows, nullptr, n_remove, chg, has, 0u, 0u, (NpHostQ<T>*)hl.queries, hl.count, hl.cap, hl.host_only, np_hook_ctx<T>(hl.hook));
    else
    hipLaunchKernelGGL((k_narrow_phase<T, true>), dim3((n_rows + NP_LIGHT_THREADS - 1) / NP_LIGHT_THREADS), dim3(NP_LIGHT_THREADS), 0, st, w, bp, ct, p, nullptr, n_rows, nullptr, n_remove, chg, has, 0u, 0u);
    if (hl.host_only) return;
    launch_np_heavy<T, true>(w, bp, ct, p, nullptr, n_remove, chg, has, n_rows, st, hl.hook, hl.any());
}
// the rows list[0 .. n_list) followed by range_base .. range_base + n_range: same per-row work and outputs as the dense form; the
// removal counter is NOT reset (it continues the count of the launch over the older rows)
template <class T> void launch_narrow_phase_rows(const DW<T>& w, const BP<T>& bp, const CT<T>& ct, const StepParams<T>& p, const uint32_t* list, uint32_t n_list, uint32_t range_base,
                                                 uint32_t n_range, uint32_t* chg, uint32_t* has, uint32_t* n_remove, hipStream_t st, const NpHostList& hl) {
    const uint32_t n = n_list + n_range;
    if (!n) return;
    // (range_base = 0 with an empty list would read as the plain dense form: a world's first pairs take the dense launch instead)
    if (hl.any()) {
        if (!n_list && !range_base) hipLaunchKernelGGL((k_narrow_phase<T, true, true>), dim3((n_range + NP_LIGHT_THREADS - 1) / NP_LIGHT_THREADS), dim3(NP_LIGHT_THREADS), 0, st, w, bp, ct, p, nullptr, n_range, nullptr, n_remove, chg, has, 0u, 0u, (NpHostQ<T>*)hl.queries, hl.count, hl.cap, hl.host_only, np_hook_ctx<T>(hl.hook));
        else hipLaunchKernelGGL((k_narrow_phase<T, true, true>), dim3((n + NP_LIGHT_THREADS - 1) / NP_LIGHT_THREADS), dim3(NP_LIGHT_THREADS), 0, st, w, bp, ct, p, list, n, nullptr, n_remove, chg, has, n_list, range_base, (NpHostQ<T>*)hl.queries, hl.count, hl.cap, hl.host_only, np_hook_ctx<T>(hl.hook));
    } else
    if (!n_list && !range_base) hipLaunchKernelGGL((k_narrow_phase<T, true>), dim3((n_range + NP_LIGHT_THREADS - 1) / NP_LIGHT_THREADS), dim3(NP_LIGHT_THREADS), 0, st, w, bp, ct, p, nullptr, n_range, nullptr, n_remove, chg, has, 0u, 0u);
    else hipLaunchKernelGGL((k_narrow_phase<T, true>), dim3((n + NP_LIGHT_THREADS - 1) / NP_LIGHT_THREADS), dim3(NP_LIGHT_THREADS), 0, st, w, bp, ct, p, list, n, nullptr, n_remove, chg, has, n_list, range_base);
    if (hl.host_only) return;
    launch_np_heavy<T, true>(w, bp, ct, p, nullptr, n_remove, chg, has, n, st, hl.hook, hl.any());
}
// the pairs the host answered (queries sorted by contact id on the host, manifolds in the same order): dense = the closed loop's chg / has outputs, else the change list
template <class T> void launch_narrow_phase_host(const DW<T>& w, const BP<T>& bp, const CT<T>& ct, const StepParams<T>& p, bool dense, avn_contact_change* changes, uint32_t* n_changes,
                                                 uint32_t* chg, uint32_t* has, const void* queries, const void* manifolds, uint32_t n, hipStream_t st, const NpHookList& hook) {
    if (!n) return;
    if (dense) hipLaunchKernelGGL((k_narrow_phase_host<T, true>), dim3((n + NP_THREADS - 1) / NP_THREADS), dim3(NP_THREADS), 0, st, w, bp, ct, p, nullptr, n_changes, chg, has, (const NpHostQ<T>*)queries, (const NpHostM<T>*)manifolds, n, np_hook_ctx<T>(hook));
    else hipLaunchKernelGGL((k_narrow_phase_host<T, false>), dim3((n + NP_THREADS - 1) / NP_THREADS), dim3(NP_THREADS), 0, st, w, bp, ct, p, changes, n_changes, nullptr, nullptr, (const NpHostQ<T>*)queries, (const NpHostM<T>*)manifolds, n, np_hook_ctx<T>(hook));
}
template <class T> void launch_narrow_phase_hooked(const DW<T>& w, const BP<T>& bp, const CT<T>& ct, const StepParams<T>& p, bool dense, avn_contact_change* changes, uint32_t* n_changes,
                                                   uint32_t* chg, uint32_t* has, const void* records, uint32_t n, hipStream_t st) {
    if (!n) return;
    if (dense) hipLaunchKernelGGL((k_narrow_phase_hooked<T, true>), dim3((n + NP_THREADS - 1) / NP_THREADS), dim3(NP_THREADS), 0, st, w, bp, ct, p, nullptr, n_changes, chg, has, (const NpHookC<T>*)records, n);
    else hipLaunchKernelGGL((k_narrow_phase_hooked<T, false>), dim3((n + NP_THREADS - 1) / NP_THREADS), dim3(NP_THREADS), 0, st, w, bp, ct, p, changes, n_changes, nullptr, nullptr, (const NpHookC<T>*)records, n);
}
template <class T> void launch_scatter_impulses(const DW<T>& w, const CT<T>& ct, const uint32_t* handles, hipStream_t st) {
    if (w.n_manifolds) hipLaunchKernelGGL(k_scatter_impulses<T>, dim3((w.n_manifolds + 255) / 256), dim3(256), 0, st, w, ct, handles);
}
template <class T> void launch_unpack_contacts(const CT<T>& ct, const uint32_t* ids, uint32_t n, const ContactsStage<T>& o, hipStream_t st) {
    if (n) hipLaunchKernelGGL(k_unpack_contacts<T>, dim3((n + 255) / 256), dim3(256), 0, st, ct, ids, n, o);
}
template <class T> void launch_pack_contacts(const CT<T>& ct, const uint32_t* ids, uint32_t n, const ContactsStage<T>& in, uint32_t* error, hipStream_t st) {
    if (n) hipLaunchKernelGGL(k_pack_contacts<T>, dim3((n + 255) / 256), dim3(256), 0, st, ct, ids, n, in, error);
}
#define INST(T)                                                                                                                                      \
    template void launch_rows_set_sleeping<T>(const CT<T>&, const uint32_t*, uint32_t, uint32_t, hipStream_t);                                          \
    template void launch_remap_row_slots<T>(const CT<T>&, const uint32_t*, uint32_t, uint32_t, uint32_t*, hipStream_t);                               \
    template void launch_pack_contacts<T>(const CT<T>&, const uint32_t*, uint32_t, const ContactsStage<T>&, uint32_t*, hipStream_t);                              \
    template void launch_init_contact_rows<T>(const CT<T>&, const uint32_t*, const uint32_t*, const uint32_t*, const uint32_t*, uint32_t, hipStream_t); \
    template void launch_clear_contact_rows<T>(const CT<T>&, const uint32_t*, uint32_t, hipStream_t);                                                \
    template void launch_narrow_phase<T>(const DW<T>&, const BP<T>&, const CT<T>&, const StepParams<T>&, const uint32_t*, uint32_t, avn_contact_change*, uint32_t*, hipStream_t, const NpHostList&); \
    template void launch_narrow_phase_dense<T>(const DW<T>&, const BP<T>&, const CT<T>&, const StepParams<T>&, uint32_t, uint32_t*, uint32_t*, uint32_t*, hipStream_t, bool, const NpHostList&); \
    template void launch_narrow_phase_rows<T>(const DW<T>&, const BP<T>&, const CT<T>&, const StepParams<T>&, const uint32_t*, uint32_t, uint32_t, uint32_t, uint32_t*, uint32_t*, uint32_t*, hipStream_t, const NpHostList&); \
    template void launch_narrow_phase_host<T>(const DW<T>&, const BP<T>&, const CT<T>&, const StepParams<T>&, bool, avn_contact_change*, uint32_t*, uint32_t*, uint32_t*, const void*, const void*, uint32_t, hipStream_t, const NpHookList&); \
    template void launch_narrow_phase_hooked<T>(const DW<T>&, const BP<T>&, const CT<T>&, const StepParams<T>&, bool, avn_contact_change*, uint32_t*, uint32_t*, uint32_t*, const void*, uint32_t, hipStream_t); \
    template void launch_scatter_impulses<T>(const DW<T>&, const CT<T>&, const uint32_t*, hipStream_t);                                               \
    template void launch_unpack_contacts<T>(const CT<T>&, const uint32_t*, uint32_t, const ContactsStage<T>&, hipStream_t);
INST(float)
INST(double)
#undef INST

}  // namespace avn
