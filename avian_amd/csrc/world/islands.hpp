// world/islands.hpp -- fragment of the body of `template <class T> struct World` (avn_world.hip includes it inside the class):
// islands and the sleeping decision.

    // ---- islands and sleeping (include/avian_mi355x.h: avn_islands_get / avn_sleep_update; k_islands.hip) --------------------------------
    DevBuf b_isl_parent, b_isl_label, b_isl_ctr, b_sleep_timer, b_isl_awake, b_isl_rests, b_isl_wakes;
    DevBuf b_isl_blk_label;     // the island-BLOCK builder's labels (solver-node rule, reused across closed-loop steps): never what avn_sleep_get reports
    uint32_t sleep_n = 0;       // body count the timers belong to (a different count restarts them)
    bool islands_fresh = false; // labels on the device describe the current constraint graph
    avn_status island_buffers() {
        const size_t n = std::max<uint32_t>(dw.n_bodies, 1);
        hipError_t err;
        for (DevBuf* b : {&b_isl_parent, &b_isl_label, &b_isl_awake, &b_isl_blk_label}) { b->ensure(n * 4, err); if (err != hipSuccess) { error = "hipMalloc failed"; return AVN_ERR_OOM; } }
        b_isl_rests.ensure(n, err); if (err != hipSuccess) { error = "hipMalloc failed"; return AVN_ERR_OOM; }
        b_isl_wakes.ensure(n, err); if (err != hipSuccess) { error = "hipMalloc failed"; return AVN_ERR_OOM; }
        b_isl_ctr.ensure(64, err); if (err != hipSuccess) { error = "hipMalloc failed"; return AVN_ERR_OOM; }
        const bool grown = b_sleep_timer.ensure(n * 4, err); if (err != hipSuccess) { error = "hipMalloc failed"; return AVN_ERR_OOM; }
        if (grown || sleep_n != dw.n_bodies) { HIPCHK(hipMemsetAsync(b_sleep_timer.p, 0, n * 4, stream)); sleep_n = dw.n_bodies; }
        return AVN_OK;
    }
    avn_status islands_compute() {
        avn_status st = need_bodies();
        if (st != AVN_OK) return st;
        if ((st = island_buffers()) != AVN_OK) return st;
        HIPCHK(hipMemsetAsync(b_isl_ctr.p, 0, 64, stream));
        launch_islands<T>(dw, b_isl_parent.as<uint32_t>(), b_isl_label.as<uint32_t>(), b_isl_ctr.as<uint32_t>(), stream);
        launches += 2 + (dw.n_manifolds ? 1 : 0) + (dw.n_joints ? 1 : 0);
        HIPCHK(hipGetLastError());
        return AVN_OK;
    }
    avn_status islands_get(uint32_t* island_of_body, uint32_t* n_islands) override {
        avn_status st = islands_compute();
        if (st != AVN_OK) return st;
        uint32_t ctr[2] = {0, 0};
        HIPCHK(hipMemcpyAsync(ctr, b_isl_ctr.p, 8, hipMemcpyDeviceToHost, stream));
        if (island_of_body && dw.n_bodies) HIPCHK(hipMemcpyAsync(island_of_body, b_isl_label.p, (size_t)dw.n_bodies * 4, hipMemcpyDeviceToHost, stream));
        HIPCHK(hipStreamSynchronize(stream));
        if (n_islands) *n_islands = ctr[0];
        return AVN_OK;
    }
    avn_status sleep_update(const avn_sleep_params* sp, avn_sleep_stats* out) override {
        if (!sp || sp->struct_size != sizeof(avn_sleep_params)) { error = "sleep_update: bad params"; return AVN_ERR_BAD_ARG; }
        avn_status st = islands_compute();
        if (st != AVN_OK) return st;
        SleepParams<T> k;
        k.length_unit_squared = (T)sp->length_unit * (T)sp->length_unit;
        k.lin_threshold_squared = (T)(sp->linear_threshold * std::fabs(sp->linear_threshold));   // f32 product, "keep signs", then `as Scalar`
        k.ang_threshold_squared = (T)(sp->angular_threshold * std::fabs(sp->angular_threshold));
        k.delta_secs = sp->delta_secs; k.time_to_sleep = sp->time_to_sleep;
        k.body_lin = nullptr; k.body_ang = nullptr; k.body_disabled = nullptr;
        if (sp->body_linear_threshold || sp->body_angular_threshold || sp->body_sleeping_disabled) {
            const size_t n = dw.n_bodies;
            if ((st = stage_reserve(al(4 * n) * 2 + al(n) + 1024)) != AVN_OK) return st;
            if ((st = stage_in<float>(sp->body_linear_threshold, n, &k.body_lin)) != AVN_OK) return st;
            if ((st = stage_in<float>(sp->body_angular_threshold, n, &k.body_ang)) != AVN_OK) return st;
            if ((st = stage_in<uint8_t>(sp->body_sleeping_disabled, n, &k.body_disabled)) != AVN_OK) return st;
        }
        HIPCHK(hipMemsetAsync(b_isl_awake.p, 0, (size_t)std::max<uint32_t>(dw.n_bodies, 1) * 4, stream));
        launch_sleep_update<T>(dw, k, b_isl_label.as<uint32_t>(), b_sleep_timer.as<float>(), b_isl_awake.as<uint32_t>(), b_isl_rests.as<uint8_t>(), b_isl_wakes.as<uint8_t>(), b_isl_ctr.as<uint32_t>(), stream);
        launches += 2;
        HIPCHK(hipGetLastError());
        if (out) {
            uint32_t ctr[8] = {0};
            HIPCHK(hipMemcpyAsync(ctr, b_isl_ctr.p, 32, hipMemcpyDeviceToHost, stream));
            HIPCHK(hipStreamSynchronize(stream));
            out->n_islands = ctr[0]; out->n_island_bodies = ctr[1]; out->n_sleeping_bodies = ctr[6]; out->n_awake_bodies = ctr[1] - ctr[6];
            out->n_resting_islands = ctr[2]; out->n_resting_bodies = ctr[3]; out->n_waking_islands = ctr[4]; out->n_waking_bodies = ctr[5];
        }
        return AVN_OK;
    }
    avn_status sleep_get(const avn_sleep_out* o) override {
        if (!o) { error = "sleep_get: null"; return AVN_ERR_BAD_ARG; }
        avn_status st = need_bodies();
        if (st != AVN_OK) return st;
        if (sleep_n != dw.n_bodies || !b_isl_rests.p) { error = "sleep_get: call avn_sleep_update first"; return AVN_ERR_STATE; }
        const size_t n = dw.n_bodies;
        if (o->sleep_timer) HIPCHK(hipMemcpyAsync(o->sleep_timer, b_sleep_timer.p, n * 4, hipMemcpyDeviceToHost, stream));
        if (o->island) HIPCHK(hipMemcpyAsync(o->island, b_isl_label.p, n * 4, hipMemcpyDeviceToHost, stream));
        if (o->island_rests) HIPCHK(hipMemcpyAsync(o->island_rests, b_isl_rests.p, n, hipMemcpyDeviceToHost, stream));
        if (o->island_wakes) HIPCHK(hipMemcpyAsync(o->island_wakes, b_isl_wakes.p, n, hipMemcpyDeviceToHost, stream));
        HIPCHK(hipStreamSynchronize(stream));
        return AVN_OK;
    }
    avn_status sleep_reset(const uint32_t* bodies, size_t n) override {
        avn_status st = need_bodies();
        if (st != AVN_OK) return st;
        if ((st = island_buffers()) != AVN_OK) return st;
        if (!bodies || n == 0) { launch_sleep_reset(b_sleep_timer.as<float>(), nullptr, dw.n_bodies, dw.n_bodies, stream); HIPCHK(hipStreamSynchronize(stream)); return AVN_OK; }
        if ((st = stage_reserve(al(4 * n) + 1024)) != AVN_OK) return st;
        const uint32_t* d = nullptr;
        if ((st = stage_in<uint32_t>(bodies, n, &d)) != AVN_OK) return st;
        launch_sleep_reset(b_sleep_timer.as<float>(), d, (uint32_t)n, dw.n_bodies, stream);
        HIPCHK(hipStreamSynchronize(stream));
        return AVN_OK;
    }
