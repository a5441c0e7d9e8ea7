// world/timers.hpp -- fragment of the body of `template <class T> struct World` (avn_world.hip includes it inside the class):
// avn_timers_get.

    avn_status diagnostics(avn_diagnostics* d) override;
    avn_status timers(avn_timers* t) override {
        if (!t) return AVN_ERR_BAD_ARG;
        HIPCHK(hipStreamSynchronize(stream));
        HIPCHK(hipStreamSynchronize(stream_bp));
        if (ev_valid) {
            float a = 0, b = 0, c = 0, d = 0, e = 0;
            HIPCHK(hipEventElapsedTime(&a, ev[0], ev[1]));
            HIPCHK(hipEventElapsedTime(&b, ev[1], ev[2]));
            HIPCHK(hipEventElapsedTime(&c, ev[2], ev[3]));
            HIPCHK(hipEventElapsedTime(&d, ev[3], ev[4]));
            HIPCHK(hipEventElapsedTime(&e, ev[0], ev[4]));
            if (avn_env("AVN_HOST_TRACE")) std::fprintf(stderr, "[avn events] step start -> solver start %.4f ms, prepare %.4f, substeps %.4f, finalize %.4f, whole %.4f\n", a, b, c, d, e);
            // overlapped broad phase: its own duration on its own stream (it is NOT a term of step_ms then)
            if (bp_timed) HIPCHK(hipEventElapsedTime(&a, ev_bp_t0, ev_bp_t1));
            last_timers.broad_phase_ms = a; last_timers.prepare_ms = b; last_timers.substeps_ms = c; last_timers.finalize_ms = d;
            last_timers.step_ms = e;
            last_timers.bias_pass_ms = 0; last_timers.bias_pass_launches = 0;
            last_timers.island_blocks = islands_active() ? islands.n_blocks : 0u; last_timers.side_island_bodies = groups_active ? side_bodies : 0u;
            if (bias_timed) {   // mean over the step's substeps
                double sum = 0;
                for (uint32_t k = 0; k < bias_timed; ++k) { float f = 0; HIPCHK(hipEventElapsedTime(&f, ev_bias[2 * k], ev_bias[2 * k + 1])); sum += f; }
                last_timers.bias_pass_ms = sum / bias_timed; last_timers.bias_pass_launches = bias_launches;
            }
        }
        uint32_t cc = 0;
        HIPCHK(hipMemcpy(&cc, dw.constraint_count, 4, hipMemcpyDeviceToHost));
        last_timers.contact_constraint_count = cc;
        *t = last_timers;
        return AVN_OK;
    }
