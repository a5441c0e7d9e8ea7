// Host shapes on the C ABI (include/avian_mi355x.h "host shapes"): a scene whose boxes are device shapes and whose CAPSULES exist only in this file.
// The library has no capsule kernel; the collider is uploaded as AVN_SHAPE_HOST and the two callbacks below are this host's AnyCollider implementation
// (aabb_with_context / swept_aabb_with_context and contact_manifolds_with_context, collision/collider/mod.rs:214-258): capsule against the ground slab and capsule
// against capsule.  Everything else -- broad phase, layers, speculative filter, pruning, warm-start matching, ContactGraph / ConstraintGraph bookkeeping, solver -- is
// the library's closed loop, and the boxes never leave the device.  Boxes and capsules are on different collision layers (no capsule-box manifold is implemented here).
//
//   examples/host_shapes_demo [nx ny nz] [capsules] [steps]      prints ms per step with and without the capsules and what crossed the bus
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "avian_mi355x.h"

#define CHECK(call)                                                                                         \
    do {                                                                                                    \
        avn_status st_ = (call);                                                                            \
        if (st_ != AVN_OK) { std::fprintf(stderr, "%s -> %d: %s\n", #call, st_, avn_last_error(world)); return 1; } \
    } while (0)

struct V3 { float x, y, z; };
static V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
static V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
static V3 operator*(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
static float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
static V3 qrot(const float* q, V3 v) { V3 u{q[0], q[1], q[2]}; return v + cross(u, cross(u, v) + v * q[3]) * 2.f; }
static float clamp01(float x) { return x < 0.f ? 0.f : (x > 1.f ? 1.f : x); }

struct Capsules { float half_height, radius, ground_top; uint32_t ground_entity; uint64_t aabb_queries = 0, manifold_queries = 0; };
static void ends(const Capsules& c, const float* p, const float* q, V3& a, V3& b) {
    V3 pos{p[0], p[1], p[2]};
    a = pos + qrot(q, {0.f, -c.half_height, 0.f}); b = pos + qrot(q, {0.f, c.half_height, 0.f});
}
// AnyCollider::aabb_with_context / swept_aabb_with_context
static void capsule_aabbs(void* user, uint32_t, uint32_t n, const void* queries, void* out) {
    Capsules& c = *(Capsules*)user;
    const avn_host_aabb_query_f32* q = (const avn_host_aabb_query_f32*)queries;
    avn_host_aabb_f32* o = (avn_host_aabb_f32*)out;
    c.aabb_queries += n;
    for (uint32_t i = 0; i < n; ++i) {
        V3 p[4]; int k = 2;
        ends(c, q[i].start_position, q[i].start_rotation, p[0], p[1]);
        if (q[i].swept) { ends(c, q[i].end_position, q[i].end_rotation, p[2], p[3]); k = 4; }
        V3 mn = p[0], mx = p[0];
        for (int j = 1; j < k; ++j) { mn = {std::min(mn.x, p[j].x), std::min(mn.y, p[j].y), std::min(mn.z, p[j].z)}; mx = {std::max(mx.x, p[j].x), std::max(mx.y, p[j].y), std::max(mx.z, p[j].z)}; }
        o[i] = {{mn.x - c.radius, mn.y - c.radius, mn.z - c.radius}, {mx.x + c.radius, mx.y + c.radius, mx.z + c.radius}};
    }
}
// AnyCollider::contact_manifolds_with_context: the manifold in contact_query::contact_manifolds' conventions (contact_query.rs:233-252)
static void capsule_manifolds(void* user, uint32_t, uint32_t n, const void* queries, void* out) {
    Capsules& c = *(Capsules*)user;
    const avn_host_manifold_query_f32* q = (const avn_host_manifold_query_f32*)queries;
    avn_host_manifold_f32* o = (avn_host_manifold_f32*)out;
    c.manifold_queries += n;
    for (uint32_t i = 0; i < n; ++i) {
        const float mcd = q[i].max_contact_distance;
        V3 p1{q[i].position1[0], q[i].position1[1], q[i].position1[2]};
        V3 normal{0, 0, 0};
        uint32_t k = 0;
        auto put = [&](V3 on1 /* point on shape 1, relative to position1 */, float dist, uint32_t fid) {
            V3 a = on1 + normal * (dist * 0.5f);
            o[i].anchor1[3 * k] = a.x; o[i].anchor1[3 * k + 1] = a.y; o[i].anchor1[3 * k + 2] = a.z;
            o[i].penetration[k] = -dist; o[i].feature_id1[k] = fid; o[i].feature_id2[k] = fid; ++k;
        };
        if (q[i].collider1 == c.ground_entity || q[i].collider2 == c.ground_entity) {
            const bool cap2 = q[i].collider1 == c.ground_entity;
            V3 e[2];
            if (cap2) ends(c, q[i].position2, q[i].rotation2, e[0], e[1]); else ends(c, q[i].position1, q[i].rotation1, e[0], e[1]);
            normal = cap2 ? V3{0, 1, 0} : V3{0, -1, 0};
            for (uint32_t j = 0; j < 2; ++j) {
                const float dist = e[j].y - c.radius - c.ground_top;
                if (dist < mcd) put((cap2 ? V3{e[j].x, c.ground_top, e[j].z} : e[j] - V3{0, c.radius, 0}) - p1, dist, j + 1);
            }
        } else {   // closest points of the two segments (Ericson 5.1.9)
            V3 a1, b1, a2, b2;
            ends(c, q[i].position1, q[i].rotation1, a1, b1); ends(c, q[i].position2, q[i].rotation2, a2, b2);
            V3 d1 = b1 - a1, d2 = b2 - a2, r = a1 - a2;
            float a = dot(d1, d1), e = dot(d2, d2), f = dot(d2, r), cc = dot(d1, r), b = dot(d1, d2), den = a * e - b * b;
            float s = den > 1e-12f ? clamp01((b * f - cc * e) / den) : 0.f, t = (b * s + f) / e;
            if (t < 0.f) { t = 0.f; s = clamp01(-cc / a); } else if (t > 1.f) { t = 1.f; s = clamp01((b - cc) / a); }
            V3 c1 = a1 + d1 * s, c2 = a2 + d2 * t, d = c2 - c1;
            float L = std::sqrt(dot(d, d));
            if (L > 1e-9f && L - 2.f * c.radius < mcd) { normal = d * (1.f / L); put(c1 + normal * c.radius - p1, L - 2.f * c.radius, 1); }
        }
        o[i].point_count = k;
        o[i].normal[0] = normal.x; o[i].normal[1] = normal.y; o[i].normal[2] = normal.z;
    }
}

static avn_config config() {
    avn_config cfg;
    std::memset(&cfg, 0, sizeof cfg);
    cfg.struct_size = sizeof cfg; cfg.scalar_bits = 32; cfg.device = 0; cfg.substeps = 4; cfg.dt_ns = 1000000000ull / 60;
    cfg.gravity[1] = -9.81; cfg.length_unit = 1.0;
    cfg.contact_damping_ratio = 10.0; cfg.contact_frequency_factor = 1.5; cfg.max_overlap_solve_speed = 4.0; cfg.warm_start_coefficient = 1.0;
    cfg.restitution_threshold = 1.0; cfg.restitution_iterations = 1; cfg.match_contacts = 1; cfg.default_speculative_margin = 3.5e38;
    cfg.contact_tolerance = 0.005; cfg.solver_iterations = 1; cfg.use_graph = 1;
    return cfg;
}

// one world: the box stack and `n_caps` capsules lying in a grid beside it; returns ms per step over the last `timed` steps
static int run(int nx, int ny, int nz, uint32_t n_caps, int steps, int timed, double* ms_out, avn_host_shape_stats* stats_out, float* worst_y_error) {
    const uint32_t n_box = (uint32_t)(nx * ny * nz), n = 1u + n_box + n_caps;
    avn_config cfg = config();
    avn_world* world = nullptr;
    if (avn_world_create(&cfg, &world) != AVN_OK) { std::fprintf(stderr, "avn_world_create: %s\n", avn_last_error(nullptr)); return 2; }
    Capsules caps{0.4f, 0.25f, 0.f, 0u};
    std::vector<float> pos(3 * n, 0.f), rot(4 * n, 0.f), lin(3 * n, 0.f), ang(3 * n, 0.f), inv_m(n, 1.f), inv_i(6 * n, 0.f), he(3 * n, 0.5f);
    std::vector<uint8_t> rb(n, AVN_RB_DYNAMIC), shape(n, AVN_SHAPE_CUBOID);
    std::vector<uint32_t> entity(n), member(n, 2u), filter(n, 1u | 2u);
    std::vector<int32_t> col_body(n);
    for (uint32_t i = 0; i < n; ++i) { rot[4 * i + 3] = 1.f; entity[i] = i; col_body[i] = (int32_t)i; inv_i[6 * i] = inv_i[6 * i + 3] = inv_i[6 * i + 5] = 6.f; }
    rb[0] = AVN_RB_STATIC; inv_m[0] = 0.f; inv_i[0] = inv_i[3] = inv_i[5] = 0.f; pos[1] = -20.f; he[0] = 4000.f; he[1] = 20.f; he[2] = 4000.f;
    member[0] = 1u; filter[0] = 0xFFFFFFFFu;
    uint32_t b = 1;
    for (int j = 0; j < ny; ++j) for (int k = 0; k < nz; ++k) for (int i = 0; i < nx; ++i, ++b) {
        pos[3 * b] = (float)i - 0.5f * (float)(nx - 1); pos[3 * b + 1] = (2.f * (float)j + 1.f) * 0.5f * 0.99f; pos[3 * b + 2] = (float)k - 0.5f * (float)(nz - 1);
    }
    // capsules: a grid beside the stack, dropped from 0.6 above the slab lying along x with a small tilt, in rows one diameter apart so that they also meet each other
    const uint32_t side = (uint32_t)std::ceil(std::sqrt((double)std::max(n_caps, 1u)));
    const float vol = 3.14159265f * caps.radius * caps.radius * 2.f * caps.half_height + 4.f / 3.f * 3.14159265f * caps.radius * caps.radius * caps.radius;
    for (uint32_t c = 0; c < n_caps; ++c, ++b) {
        pos[3 * b] = 0.5f * (float)nx + 3.f + 1.45f * (float)(c % side); pos[3 * b + 1] = 0.6f + 0.05f * (float)(c % 3); pos[3 * b + 2] = 0.5f * (float)(c / side) - 0.25f * (float)side;   // rows one diameter apart: neighbours touch (capsule-capsule manifolds)
        const float ang_z = 1.5707963f + 0.1f * (float)((c * 7u) % 5u) - 0.2f;   // local y (the capsule's axis) turned towards world x
        rot[4 * b + 2] = std::sin(0.5f * ang_z); rot[4 * b + 3] = std::cos(0.5f * ang_z);
        shape[b] = AVN_SHAPE_HOST; he[3 * b] = he[3 * b + 1] = he[3 * b + 2] = 0.f;
        member[b] = 4u; filter[b] = 1u | 4u;
        inv_m[b] = 1.f / vol;
        const float iy = 0.5f * vol * caps.radius * caps.radius, ix = vol * (3.f * caps.radius * caps.radius + 4.f * caps.half_height * caps.half_height) / 12.f + 0.3f * vol * caps.half_height * caps.half_height;
        inv_i[6 * b] = 1.f / ix; inv_i[6 * b + 3] = 1.f / iy; inv_i[6 * b + 5] = 1.f / ix;
    }
    avn_bodies bodies; std::memset(&bodies, 0, sizeof bodies);
    bodies.count = n; bodies.position = pos.data(); bodies.rotation = rot.data(); bodies.linear_velocity = lin.data(); bodies.angular_velocity = ang.data();
    bodies.inv_mass = inv_m.data(); bodies.inv_inertia_local = inv_i.data(); bodies.rb_type = rb.data();
    CHECK(avn_bodies_upload(world, &bodies));
    avn_colliders cols; std::memset(&cols, 0, sizeof cols);
    cols.count = n; cols.entity_index = entity.data(); cols.body = col_body.data(); cols.shape = shape.data(); cols.half_extents = he.data();
    cols.memberships = member.data(); cols.filters = filter.data();
    CHECK(avn_colliders_upload(world, &cols));
    CHECK(avn_existing_pairs_upload(world, nullptr, 0));
    avn_collider_materials mats; std::memset(&mats, 0, sizeof mats); mats.count = n;
    CHECK(avn_collider_materials_upload(world, &mats));
    if (n_caps) CHECK(avn_host_shapes_set(world, capsule_aabbs, capsule_manifolds, &caps));
    CHECK(avn_pipeline_enable(world, 1));
    for (int s = 0; s < steps - timed; ++s) CHECK(avn_step(world));
    CHECK(avn_synchronize(world));
    const auto t0 = std::chrono::steady_clock::now();
    for (int s = 0; s < timed; ++s) CHECK(avn_step(world));
    CHECK(avn_synchronize(world));
    *ms_out = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / timed;
    CHECK(avn_host_shape_stats_get(world, stats_out));
    avn_bodies_out out; std::memset(&out, 0, sizeof out);
    out.position = pos.data();
    CHECK(avn_bodies_download(world, &out));
    *worst_y_error = 0.f;
    for (uint32_t c = 0; c < n_caps; ++c) *worst_y_error = std::max(*worst_y_error, std::fabs(pos[3 * (1 + n_box + c) + 1] - caps.radius));
    std::printf("  %u boxes + %u capsules: %.3f ms per step (last %d of %d); per step %u aabb queries, %u manifold queries (%u with points); callbacks %.3f ms; since the start %.1f MB to the host, %.1f MB back\n",
                n_box, n_caps, *ms_out, timed, steps, stats_out->last_aabb_queries, stats_out->last_manifold_queries, stats_out->last_manifolds_with_points, stats_out->last_callback_ms,
                stats_out->bytes_to_host / 1e6, stats_out->bytes_from_host / 1e6);
    avn_world_destroy(world);
    return 0;
}

int main(int argc, char** argv) {
    const int nx = argc > 3 ? std::atoi(argv[1]) : 12, ny = argc > 3 ? std::atoi(argv[2]) : 10, nz = argc > 3 ? std::atoi(argv[3]) : 12;
    const uint32_t n_caps = argc > 4 ? (uint32_t)std::atoi(argv[4]) : 400u;
    const int steps = argc > 5 ? std::atoi(argv[5]) : 120, timed = std::min(steps, 20);
    double ms_boxes = 0, ms_both = 0; avn_host_shape_stats s0, s1; float e0 = 0, e1 = 0;
    std::printf("host shapes demo: capsules implemented in this file (AnyCollider callbacks), boxes in device kernels\n");
    int rc = run(nx, ny, nz, 0u, steps, timed, &ms_boxes, &s0, &e0);
    if (rc) return rc;
    if ((rc = run(nx, ny, nz, n_caps, steps, timed, &ms_both, &s1, &e1))) return rc;
    std::printf("  the capsules cost %.3f ms per step (%.1f us per capsule); worst |centre height - radius| of a capsule: %.4f\n", ms_both - ms_boxes, n_caps ? 1e3 * (ms_both - ms_boxes) / n_caps : 0.0, e1);
    const bool ok = s1.host_colliders == n_caps && s1.last_aabb_queries == n_caps && s1.last_manifold_queries >= n_caps && e1 < 0.05f && s0.host_colliders == 0;
    std::printf(ok ? "HOST_SHAPES_DEMO_OK\n" : "HOST_SHAPES_DEMO_FAILED\n");
    return ok ? 0 : 3;
}
