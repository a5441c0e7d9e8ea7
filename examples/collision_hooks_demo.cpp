// Collision hooks on the C ABI (include/avian_mi355x.h "collision hooks"): Avian's CollisionHooks::filter_pairs / modify_contacts (collision/hooks.rs:137-231) as the
// two callbacks of a compiled host, inside the library's closed loop.  A pile of boxes nobody hooks, next to
//   * a CONVEYOR BELT: a static slab with ActiveCollisionHooks::MODIFY_CONTACTS; modify_contacts gives its contacts a tangent_velocity (the reference's own example
//     of the hook, hooks.rs:60-110) and the boxes on it ride along;
//   * GHOSTS: boxes with ActiveCollisionHooks::FILTER_PAIRS whose filter_pairs rejects ghost-ghost pairs: dropped onto each other they pass through one another
//     and all come to rest on the ground (one-way / team-based filtering, hooks.rs:24-58).
// Only the hooked pairs cross the bus: 12 B per pair the filter is asked about, 232 B each way per contact pair the hook is shown.  Until round 6 one such collider
// sent the whole world to the HostNarrowPhase mode (every manifold re-sent every step).
//
//   examples/collision_hooks_demo [nx ny nz] [belt boxes] [ghost stacks] [steps]
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

struct Hooks {
    uint32_t belt_entity, ghost_first, ghost_last;   // entity indices
    float belt_speed;
    uint64_t asked = 0, rejected = 0, shown = 0;
    bool ghost(uint32_t e) const { return e >= ghost_first && e <= ghost_last; }
};
// CollisionHooks::filter_pairs: ghosts do not collide with ghosts
static void filter_pairs(void* user, uint32_t n, const avn_hook_pair* pairs, uint8_t* should_collide) {
    Hooks& h = *(Hooks*)user;
    h.asked += n;
    for (uint32_t i = 0; i < n; ++i)
        if (h.ghost(pairs[i].collider1) && h.ghost(pairs[i].collider2)) { should_collide[i] = 0; ++h.rejected; }
}
// CollisionHooks::modify_contacts: the belt's surface moves along x (relative to whichever collider of the pair the belt is)
static void modify_contacts(void* user, uint32_t scalar_bits, uint32_t n, void* contacts) {
    Hooks& h = *(Hooks*)user;
    if (scalar_bits != 32) return;
    avn_hook_contact_f32* c = (avn_hook_contact_f32*)contacts;
    h.shown += n;
    for (uint32_t i = 0; i < n; ++i) {
        if (c[i].collider1 != h.belt_entity && c[i].collider2 != h.belt_entity) continue;
        c[i].tangent_velocity[0] = c[i].collider1 == h.belt_entity ? h.belt_speed : -h.belt_speed;
        c[i].friction = 0.9f;
    }
}

int main(int argc, char** argv) {
    const int nx = argc > 3 ? std::atoi(argv[1]) : 12, ny = argc > 3 ? std::atoi(argv[2]) : 10, nz = argc > 3 ? std::atoi(argv[3]) : 12;
    const int n_belt = argc > 4 ? std::atoi(argv[4]) : 8, n_ghost = argc > 5 ? std::atoi(argv[5]) : 16, steps = argc > 6 ? std::atoi(argv[6]) : 150;
    const uint32_t n_pile = (uint32_t)(nx * ny * nz);
    const uint32_t belt = 1 + n_pile, belt_box0 = belt + 1, ghost0 = belt_box0 + (uint32_t)n_belt, n = ghost0 + 2u * (uint32_t)n_ghost;

    avn_config cfg;
    std::memset(&cfg, 0, sizeof cfg);
    cfg.struct_size = sizeof cfg; cfg.scalar_bits = 32; cfg.device = 0; cfg.substeps = 4; cfg.dt_ns = 1000000000ull / 60;
    cfg.gravity[1] = -9.81; cfg.length_unit = 1.0;
    cfg.contact_damping_ratio = 10.0; cfg.contact_frequency_factor = 1.5; cfg.max_overlap_solve_speed = 4.0; cfg.warm_start_coefficient = 1.0;
    cfg.restitution_threshold = 1.0; cfg.restitution_iterations = 1; cfg.match_contacts = 1; cfg.default_speculative_margin = 3.5e38;
    cfg.contact_tolerance = 0.005; cfg.solver_iterations = 1; cfg.use_graph = 1;
    avn_world* world = nullptr;
    if (avn_world_create(&cfg, &world) != AVN_OK) { std::fprintf(stderr, "avn_world_create: %s\n", avn_last_error(nullptr)); return 2; }

    std::vector<float> pos(3 * n, 0.f), rot(4 * n, 0.f), lin(3 * n, 0.f), ang(3 * n, 0.f), inv_m(n, 1.f), inv_i(6 * n, 0.f), he(3 * n, 0.5f);
    std::vector<uint8_t> rb(n, AVN_RB_DYNAMIC), shape(n, AVN_SHAPE_CUBOID), cflags(n, 0);
    std::vector<uint32_t> entity(n);
    std::vector<int32_t> col_body(n);
    for (uint32_t i = 0; i < n; ++i) { rot[4 * i + 3] = 1.f; entity[i] = 1000u + i; col_body[i] = (int32_t)i; inv_i[6 * i] = inv_i[6 * i + 3] = inv_i[6 * i + 5] = 6.f; }
    auto make_static = [&](uint32_t b, float x, float y, float z, float hx, float hy, float hz) {
        rb[b] = AVN_RB_STATIC; inv_m[b] = 0.f; inv_i[6 * b] = inv_i[6 * b + 3] = inv_i[6 * b + 5] = 0.f;
        pos[3 * b] = x; pos[3 * b + 1] = y; pos[3 * b + 2] = z; he[3 * b] = hx; he[3 * b + 1] = hy; he[3 * b + 2] = hz;
    };
    make_static(0, 0.f, -20.f, 0.f, 400.f, 20.f, 400.f);   // the ground: top at y = 0
    uint32_t b = 1;
    for (int j = 0; j < ny; ++j) for (int k = 0; k < nz; ++k) for (int i = 0; i < nx; ++i, ++b) {
        pos[3 * b] = (float)i - 0.5f * (float)(nx - 1); pos[3 * b + 1] = (2.f * (float)j + 1.f) * 0.5f * 0.99f; pos[3 * b + 2] = (float)k - 0.5f * (float)(nz - 1);
    }
    const float far_x = 0.5f * (float)nx + 30.f;
    const float belt_hx = 2.f * (float)n_belt + 8.f, belt_x = far_x + belt_hx;   // the belt: beside the pile, top at y = 1
    make_static(belt, belt_x, 0.5f, 0.f, belt_hx, 0.5f, 2.f);
    cflags[belt] = AVN_COLLIDER_MODIFY_CONTACTS;
    for (int i = 0; i < n_belt; ++i) { const uint32_t k = belt_box0 + (uint32_t)i; pos[3 * k] = belt_x + 3.f * (float)i - 1.5f * (float)n_belt; pos[3 * k + 1] = 1.52f; }
    for (int i = 0; i < n_ghost; ++i)
        for (int l = 0; l < 2; ++l) {   // two ghosts above each other: the upper one falls THROUGH the lower one
            const uint32_t k = ghost0 + 2u * (uint32_t)i + (uint32_t)l;
            pos[3 * k] = -far_x - 3.f * (float)(i % 8); pos[3 * k + 1] = 0.52f + 1.3f * (float)l; pos[3 * k + 2] = 3.f * (float)(i / 8);
            cflags[k] = AVN_COLLIDER_FILTER_PAIRS;
        }
    avn_bodies bodies; std::memset(&bodies, 0, sizeof bodies);
    bodies.count = n; bodies.position = pos.data(); bodies.rotation = rot.data(); bodies.linear_velocity = lin.data(); bodies.angular_velocity = ang.data();
    bodies.inv_mass = inv_m.data(); bodies.inv_inertia_local = inv_i.data(); bodies.rb_type = rb.data();
    CHECK(avn_bodies_upload(world, &bodies));
    avn_colliders cols; std::memset(&cols, 0, sizeof cols);
    cols.count = n; cols.entity_index = entity.data(); cols.body = col_body.data(); cols.shape = shape.data(); cols.half_extents = he.data(); cols.collider_flags = cflags.data();
    CHECK(avn_colliders_upload(world, &cols));
    CHECK(avn_existing_pairs_upload(world, nullptr, 0));
    avn_collider_materials mats; std::memset(&mats, 0, sizeof mats); mats.count = n;
    CHECK(avn_collider_materials_upload(world, &mats));
    Hooks hooks{entity[belt], entity[ghost0], entity[n - 1], 1.25f};
    CHECK(avn_collision_hooks_set(world, filter_pairs, modify_contacts, &hooks));
    CHECK(avn_pipeline_enable(world, 1));

    double total_ms = 0;
    for (int s = 0; s < steps; ++s) {
        const auto t0 = std::chrono::steady_clock::now();
        CHECK(avn_step(world));
        CHECK(avn_synchronize(world));
        if (s >= steps / 2) total_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
    avn_collision_hook_stats hs; CHECK(avn_collision_hook_stats_get(world, &hs));
    avn_pipeline_stats ps; CHECK(avn_pipeline_stats_get(world, &ps));
    std::vector<float> out_pos(3 * n), out_vel(3 * n);
    avn_bodies_out out; std::memset(&out, 0, sizeof out); out.position = out_pos.data(); out.linear_velocity = out_vel.data();
    CHECK(avn_bodies_download(world, &out));
    // the belt's boxes ride at the belt's speed, all in one direction, still on the belt
    float vmin = 1e30f, vmax = -1e30f; bool on_belt = true;
    for (int i = 0; i < n_belt; ++i) { const uint32_t k = belt_box0 + (uint32_t)i; vmin = std::fmin(vmin, out_vel[3 * k]); vmax = std::fmax(vmax, out_vel[3 * k]); on_belt = on_belt && std::fabs(out_pos[3 * k + 1] - 1.5f) < 0.05f; }
    const bool belt_ok = n_belt == 0 || (on_belt && vmin * vmax > 0.f && std::fabs(std::fabs(vmin) - hooks.belt_speed) < 0.1f && std::fabs(std::fabs(vmax) - hooks.belt_speed) < 0.1f);
    // every ghost rests on the ground: the upper ones fell through the lower ones
    bool ghosts_ok = true;
    for (uint32_t k = ghost0; k < n; ++k) ghosts_ok = ghosts_ok && std::fabs(out_pos[3 * k + 1] - 0.5f) < 0.05f;
    float ymax = -1e30f;
    for (uint32_t i = 1; i <= n_pile; ++i) ymax = std::fmax(ymax, out_pos[3 * i + 1]);
    std::printf("collision_hooks_demo: %u boxes in the pile + a belt with %d boxes + %d ghost stacks, %d steps, %.3f ms/step over the last %d\n", n_pile, n_belt, n_ghost, steps, total_ms / (steps - steps / 2), steps - steps / 2);
    std::printf("  filter_pairs: asked about %llu pairs, rejected %llu (last step %u / %u); modify_contacts: shown %llu contact pairs (last step %u)\n", (unsigned long long)hooks.asked,
                (unsigned long long)hooks.rejected, hs.last_filter_queries, hs.last_filter_rejected, (unsigned long long)hooks.shown, hs.last_modify_queries);
    std::printf("  bus: %.1f kB to the host, %.1f kB back over the whole run (%u manifolds in the solver per step); callbacks %.3f ms last step\n", hs.bytes_to_host / 1e3, hs.bytes_from_host / 1e3, ps.manifolds, hs.last_callback_ms);
    std::printf("  belt boxes vx in [%.3f, %.3f] (belt speed %.2f), ghosts on the ground: %s, top of the pile y = %.3f\n", vmin, vmax, hooks.belt_speed, ghosts_ok ? "yes" : "NO", ymax);
    const bool sane = belt_ok && ghosts_ok && hooks.rejected > 0 && hooks.shown > 0 && ymax > 0.5f * (float)ny * 0.9f && ymax < (float)ny + 1.f;
    std::printf("%s\n", sane ? "HOOKS_DEMO_OK" : "HOOKS_DEMO_SUSPECT");
    avn_world_destroy(world);
    return sane ? 0 : 3;
}
