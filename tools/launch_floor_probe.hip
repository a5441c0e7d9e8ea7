// Diagnostic (tools only, not part of the library): what a dependent kernel-to-kernel boundary costs inside a replayed hipGraph on this
// box, and what each DEPENDENT memory level inside a small kernel adds.  A chain of N kernels, 1024 workgroups of 64 lanes each (a cfg2
// colour launch), captured once and replayed; per-kernel time = replay time / N.
//   empty      : no memory access at all
//   level k    : k dependent loads (each address comes from the previous load's value; the tables are 64 MB apart so nothing is
//                cached from the previous kernel's lines), then one store
// build: hipcc --offload-arch=gfx950 -O2 -o launch_floor_probe launch_floor_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <numeric>
#include <random>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void k_empty() {}
template <int LEVELS>
__global__ __launch_bounds__(64) void k_chase(const uint32_t* __restrict__ t0, const uint32_t* __restrict__ t1, const uint32_t* __restrict__ t2,
                                               const uint32_t* __restrict__ t3, uint32_t* __restrict__ out, uint32_t salt) {
    uint32_t i = blockIdx.x * 64 + threadIdx.x;
    uint32_t v = i ^ salt;
    if (LEVELS >= 1) v = t0[v & 0xFFFFu];
    if (LEVELS >= 2) v = t1[v & 0xFFFFu];
    if (LEVELS >= 3) v = t2[v & 0xFFFFu];
    if (LEVELS >= 4) v = t3[v & 0xFFFFu];
    out[i] = v;
}
int main() {
    const int N = 256, GRID = 1024, REPS = 20;
    hipStream_t st; CK(hipStreamCreate(&st));
    uint32_t* t[4]; uint32_t* out;
    std::vector<uint32_t> perm(1 << 16);
    std::iota(perm.begin(), perm.end(), 0u);
    std::mt19937 rng(7);
    for (int k = 0; k < 4; ++k) {
        std::shuffle(perm.begin(), perm.end(), rng);
        CK(hipMalloc(&t[k], 64u << 20));
        CK(hipMemcpy(t[k], perm.data(), perm.size() * 4, hipMemcpyHostToDevice));
    }
    CK(hipMalloc(&out, GRID * 64 * 4));
    const char* names[] = {"empty", "0 loads + store", "1 level", "2 levels", "3 levels", "4 levels"};
    for (int variant = 0; variant < 6; ++variant) {
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
        for (int i = 0; i < N; ++i) {
            switch (variant) {
                case 0: hipLaunchKernelGGL(k_empty, dim3(GRID), dim3(64), 0, st); break;
                case 1: hipLaunchKernelGGL(k_chase<0>, dim3(GRID), dim3(64), 0, st, t[0], t[1], t[2], t[3], out, (uint32_t)i); break;
                case 2: hipLaunchKernelGGL(k_chase<1>, dim3(GRID), dim3(64), 0, st, t[0], t[1], t[2], t[3], out, (uint32_t)i); break;
                case 3: hipLaunchKernelGGL(k_chase<2>, dim3(GRID), dim3(64), 0, st, t[0], t[1], t[2], t[3], out, (uint32_t)i); break;
                case 4: hipLaunchKernelGGL(k_chase<3>, dim3(GRID), dim3(64), 0, st, t[0], t[1], t[2], t[3], out, (uint32_t)i); break;
                default: hipLaunchKernelGGL(k_chase<4>, dim3(GRID), dim3(64), 0, st, t[0], t[1], t[2], t[3], out, (uint32_t)i); break;
            }
        }
        CK(hipStreamEndCapture(st, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
        for (int w = 0; w < 3; ++w) CK(hipGraphLaunch(ge, st));
        CK(hipStreamSynchronize(st));
        CK(hipEventRecord(a, st));
        for (int r = 0; r < REPS; ++r) CK(hipGraphLaunch(ge, st));
        CK(hipEventRecord(b, st));
        CK(hipEventSynchronize(b));
        float ms = 0; CK(hipEventElapsedTime(&ms, a, b));
        std::printf("%-16s %7.3f us per kernel (graph of %d kernels, %d workgroups x 64 lanes, %d replays)\n", names[variant], ms * 1e3 / (N * REPS), N, GRID, REPS);
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }
    // the same chain as plain stream launches (no graph), empty kernels only
    {
        hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
        for (int i = 0; i < 64; ++i) hipLaunchKernelGGL(k_empty, dim3(GRID), dim3(64), 0, st);
        CK(hipStreamSynchronize(st));
        CK(hipEventRecord(a, st));
        for (int i = 0; i < N * 4; ++i) hipLaunchKernelGGL(k_empty, dim3(GRID), dim3(64), 0, st);
        CK(hipEventRecord(b, st));
        CK(hipEventSynchronize(b));
        float ms = 0; CK(hipEventElapsedTime(&ms, a, b));
        std::printf("%-16s %7.3f us per kernel (stream launches, no graph)\n", "empty", ms * 1e3 / (N * 4));
    }
    return 0;
}
