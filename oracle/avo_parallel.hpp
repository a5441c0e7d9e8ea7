// ORACLE — TEST INFRASTRUCTURE ONLY (see the header of avo_world.hpp).
//
// avo_parallel.hpp: the CPU baseline's thread pool.  Restates crate::utils::par_for_each (src/utils.rs:57-87) -- serial when
// the pool has one thread or the slice is shorter than min_len, else chunks of max(len / threads, 1) items handed to the
// ComputeTaskPool -- and Bevy's Query::par_iter_mut for the per-body systems (integrator/mod.rs:278,322,356,512,
// solver_body/plugin.rs:188,268,291).  Only loops whose items touch disjoint state are run through it (the manifolds of one
// graph colour, bodies), so every result is bit-identical to the single-thread run (tests/test_oracle_threads.py).
// Thread count: AVO_THREADS in the environment at world creation (default 1).
#pragma once
#include <atomic>
#include <condition_variable>
#include <cstdlib>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace avo {

class Pool {
    std::vector<std::thread> workers_;
    std::mutex mu_;
    std::condition_variable cv_start_, cv_done_;
    const std::function<void(size_t)>* job_ = nullptr;
    size_t n_chunks_ = 0, done_ = 0, active_ = 0;
    std::atomic<size_t> next_{0};
    uint64_t gen_ = 0;
    bool stop_ = false;

    void drain() {   // pull chunk indices until none are left
        size_t mine = 0, i;
        while ((i = next_.fetch_add(1, std::memory_order_relaxed)) < n_chunks_) { (*job_)(i); ++mine; }
        std::lock_guard<std::mutex> lk(mu_);
        done_ += mine;
        --active_;
        if (done_ == n_chunks_ && active_ == 0) cv_done_.notify_one();
    }
    void worker() {
        uint64_t seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_start_.wait(lk, [&] { return stop_ || gen_ != seen; });
                if (stop_) return;
                seen = gen_;
                if (done_ == n_chunks_) continue;   // woke up after the job finished
                ++active_;
            }
            drain();
        }
    }

public:
    const unsigned threads;
    static unsigned from_env() {
        const char* e = std::getenv("AVO_THREADS");
        long v = e ? std::strtol(e, nullptr, 10) : 1;
        return (unsigned)(v < 1 ? 1 : v > 1024 ? 1024 : v);
    }
    explicit Pool(unsigned n) : threads(n < 1 ? 1 : n) {
        for (unsigned t = 1; t < threads; ++t) workers_.emplace_back([this] { worker(); });
    }
    ~Pool() {
        { std::lock_guard<std::mutex> lk(mu_); stop_ = true; }
        cv_start_.notify_all();
        for (std::thread& t : workers_) t.join();
    }
    Pool(const Pool&) = delete;
    Pool& operator=(const Pool&) = delete;

    // f(chunk index) for every chunk in [0, chunks); returns when all have run.  The caller works too.
    void run_chunks(size_t chunks, const std::function<void(size_t)>& f) {
        if (chunks == 0) return;
        if (threads == 1 || chunks == 1) { for (size_t i = 0; i < chunks; ++i) f(i); return; }
        {
            std::lock_guard<std::mutex> lk(mu_);
            job_ = &f; n_chunks_ = chunks; done_ = 0; next_.store(0, std::memory_order_relaxed); ++gen_;
            ++active_;   // the caller
        }
        cv_start_.notify_all();
        drain();
        std::unique_lock<std::mutex> lk(mu_);
        cv_done_.wait(lk, [&] { return done_ == n_chunks_ && active_ == 0; });
    }
    // crate::utils::par_for_each (src/utils.rs:57-87): f(begin, end) over chunks of max(n / threads, 1) items; serial when the
    // pool has one thread or n < min_len
    template <class F> void par_for_each(size_t n, size_t min_len, F&& f) {
        if (threads == 1 || n < min_len) { if (n) f((size_t)0, n); return; }
        const size_t chunk = std::max<size_t>(n / threads, 1), chunks = (n + chunk - 1) / chunk;
        std::function<void(size_t)> job = [&](size_t c) { f(c * chunk, std::min(n, (c + 1) * chunk)); };
        run_chunks(chunks, job);
    }
};

}  // namespace avo
