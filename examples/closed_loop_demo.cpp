// A compiled host on the C ABI and nothing else (no Python, no torch): what a non-Rust engine -- or the body of Avian's replacement
// systems once translated -- does with include/avian_mi355x.h.  Builds a stack of cuboids on a static slab, hands the library the whole
// per-step pipeline (avn_pipeline_enable: broad phase -> narrow phase -> contact bookkeeping -> solver, all on the device), steps it, and
// prints what a host reads back: body state, diagnostics (SolverDiagnostics / CollisionDiagnostics), pipeline counters, the sleeping decision.
//
//   make -C examples            (g++ + the built libavian_mi355x.so)
//   examples/closed_loop_demo [nx ny nz] [steps]
#include <chrono>
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

int main(int argc, char** argv) {
    const int nx = argc > 3 ? std::atoi(argv[1]) : 12, ny = argc > 3 ? std::atoi(argv[2]) : 10, nz = argc > 3 ? std::atoi(argv[3]) : 12;
    const int steps = argc > 4 ? std::atoi(argv[4]) : (argc == 2 ? std::atoi(argv[1]) : 60);
    const uint32_t n = 1u + (uint32_t)(nx * ny * nz);

    avn_config cfg;
    std::memset(&cfg, 0, sizeof cfg);
    cfg.struct_size = sizeof cfg; cfg.scalar_bits = 32; cfg.device = 0; cfg.substeps = 4; cfg.dt_ns = 1000000000ull / 60;
    cfg.gravity[1] = -9.81; cfg.length_unit = 1.0;
    cfg.contact_damping_ratio = 10.0; cfg.contact_frequency_factor = 1.5; cfg.max_overlap_solve_speed = 4.0; cfg.warm_start_coefficient = 1.0;
    cfg.restitution_threshold = 1.0; cfg.restitution_iterations = 1; cfg.match_contacts = 1; cfg.default_speculative_margin = 3.5e38;
    cfg.contact_tolerance = 0.005; cfg.solver_iterations = 1; cfg.use_graph = 1;
    avn_world* world = nullptr;
    if (avn_world_create(&cfg, &world) != AVN_OK) { std::fprintf(stderr, "avn_world_create: %s\n", avn_last_error(nullptr)); return 2; }

    // bodies: index 0 = static slab, then unit cubes (density 1: m = 1, I = 1/6), spacing of the reference's large_pyramid bench in y
    std::vector<float> pos(3 * n, 0.f), rot(4 * n, 0.f), lin(3 * n, 0.f), ang(3 * n, 0.f), inv_m(n, 1.f), inv_i(6 * n, 0.f), he(3 * n, 0.5f);
    std::vector<uint8_t> rb(n, AVN_RB_DYNAMIC), shape(n, AVN_SHAPE_CUBOID);
    std::vector<uint32_t> entity(n);
    std::vector<int32_t> col_body(n);
    for (uint32_t i = 0; i < n; ++i) { rot[4 * i + 3] = 1.f; entity[i] = i; col_body[i] = (int32_t)i; inv_i[6 * i] = inv_i[6 * i + 3] = inv_i[6 * i + 5] = 6.f; }
    rb[0] = AVN_RB_STATIC; inv_m[0] = 0.f; inv_i[0] = inv_i[3] = inv_i[5] = 0.f; pos[1] = -20.f; he[0] = 400.f; he[1] = 20.f; he[2] = 400.f;
    uint32_t b = 1;
    for (int j = 0; j < ny; ++j) for (int k = 0; k < nz; ++k) for (int i = 0; i < nx; ++i, ++b) {
        pos[3 * b] = (float)i - 0.5f * (float)(nx - 1); pos[3 * b + 1] = (2.f * (float)j + 1.f) * 0.5f * 0.99f; pos[3 * b + 2] = (float)k - 0.5f * (float)(nz - 1);
    }
    avn_bodies bodies; std::memset(&bodies, 0, sizeof bodies);
    bodies.count = n; bodies.position = pos.data(); bodies.rotation = rot.data(); bodies.linear_velocity = lin.data(); bodies.angular_velocity = ang.data();
    bodies.inv_mass = inv_m.data(); bodies.inv_inertia_local = inv_i.data(); bodies.rb_type = rb.data();
    CHECK(avn_bodies_upload(world, &bodies));
    avn_colliders cols; std::memset(&cols, 0, sizeof cols);
    cols.count = n; cols.entity_index = entity.data(); cols.body = col_body.data(); cols.shape = shape.data(); cols.half_extents = he.data();
    CHECK(avn_colliders_upload(world, &cols));
    CHECK(avn_existing_pairs_upload(world, nullptr, 0));
    avn_collider_materials mats; std::memset(&mats, 0, sizeof mats); mats.count = n;   // defaults: friction 0.5, restitution 0, Average
    CHECK(avn_collider_materials_upload(world, &mats));
    CHECK(avn_pipeline_enable(world, 1));   // ContactGraph / ConstraintGraph bookkeeping on the device

    avn_sleep_params sp; std::memset(&sp, 0, sizeof sp);
    sp.struct_size = sizeof sp; sp.time_to_sleep = 0.5f; sp.linear_threshold = 0.15f; sp.angular_threshold = 0.15f; sp.delta_secs = 1.f / 60.f; sp.length_unit = 1.0;
    avn_sleep_stats ss; std::memset(&ss, 0, sizeof ss);
    double total_ms = 0;
    for (int s = 0; s < steps; ++s) {
        const auto t0 = std::chrono::steady_clock::now();
        CHECK(avn_step(world));
        CHECK(avn_synchronize(world));
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        if (s >= steps / 2) total_ms += ms;
        CHECK(avn_sleep_update(world, &sp, &ss));
    }
    avn_pipeline_stats ps; CHECK(avn_pipeline_stats_get(world, &ps));
    avn_diagnostics dg; CHECK(avn_diagnostics_get(world, &dg));
    std::vector<float> out_pos(3 * n), out_vel(3 * n);
    avn_bodies_out out; std::memset(&out, 0, sizeof out); out.position = out_pos.data(); out.linear_velocity = out_vel.data();
    CHECK(avn_bodies_download(world, &out));
    float ymax = -1e30f, vmax = 0.f;
    for (uint32_t i = 1; i < n; ++i) { ymax = out_pos[3 * i + 1] > ymax ? out_pos[3 * i + 1] : ymax; for (int c = 0; c < 3; ++c) { float v = out_vel[3 * i + c]; v = v < 0 ? -v : v; vmax = v > vmax ? v : vmax; } }
    std::printf("closed_loop_demo: %u boxes, %d steps, %.3f ms/step over the last %d\n", n - 1, steps, total_ms / (steps - steps / 2), steps - steps / 2);
    std::printf("  contacts: %u active pairs, %u manifolds, %u status changes last step, host bookkeeping %.3f ms\n", ps.active_pairs, ps.manifolds, ps.last_status_changes, ps.last_host_ms);
    std::printf("  diagnostics (ms): broad %.3f narrow %.3f prepare %.3f substeps %.3f store %.3f\n", dg.broad_phase_ms, dg.narrow_phase_ms, dg.prepare_constraints_ms, dg.substeps_ms, dg.store_impulses_ms);
    std::printf("  islands %u, awake bodies %u, resting islands %u; top of the pile y = %.3f, max |v| = %.3f\n", ss.n_islands, ss.n_awake_bodies, ss.n_resting_islands, ymax, vmax);
    const bool sane = ps.manifolds > (uint32_t)(nx * nz) && ymax > 0.5f * (float)ny * 0.9f && ymax < (float)ny + 1.f && vmax < 50.f;
    std::printf("%s\n", sane ? "DEMO_OK" : "DEMO_SUSPECT");
    avn_world_destroy(world);
    return sane ? 0 : 3;
}
