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
// What one wave's VALU stream costs when it has a SIMD to itself (the colour kernels run at <= 1 wave per SIMD at cfg2): a chain of N
// dependent / independent f32 and packed-f32 multiplies, timed with the 100 MHz s_memrealtime next to the shader clock (s_memtime).
typedef float f2v __attribute__((ext_vector_type(2)));
template <int KIND>
__global__ __launch_bounds__(64) void k_valu(float* out, unsigned long long* stamp, float seed) {
    float a = seed + threadIdx.x, b = 1.0000001f, c = seed, d = seed * 2;
    f2v pa = {a, c}, pb2 = {b, b}, pc = {c, a}, pd = {d, a};
    unsigned long long r0 = __builtin_amdgcn_s_memrealtime(), t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int i = 0; i < 64; ++i) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if (KIND == 0) { a = a * b; }                                    // dependent v_mul_f32
            else if (KIND == 1) { a = a * b; c = c * b; d = d * b; }         // three independent chains of v_mul_f32
            else if (KIND == 2) { pa = pa * pb2; }                           // dependent v_pk_mul_f32
            else { pa = pa * pb2; pc = pc * pb2; pd = pd * pb2; }            // three independent chains of v_pk_mul_f32
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * 64 + threadIdx.x] = a + c + d + pa.x + pa.y + pc.x + pc.y + pd.x + pd.y;
    if (blockIdx.x == 0 && threadIdx.x == 0) { stamp[0] = t1 - t0; stamp[1] = r1 - r0; }
}
template <int KIND> static int valu(const char* name, int per_iter, float* out, unsigned long long* stamp) {
    unsigned long long h[2];
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(k_valu<KIND>, dim3(1024), dim3(64), 0, 0, out, stamp, 1.0f + rep);
        CK(hipDeviceSynchronize());
    }
    CK(hipMemcpy(h, stamp, 16, hipMemcpyDeviceToHost));
    const double n = 64.0 * 16 * per_iter, us = h[1] * 0.01;
    std::printf("%-44s %6.0f instr in %6.2f us = %5.2f ns per instr | s_memtime ticks per instr %5.2f (tick rate %.0f MHz)\n", name, n, us, us * 1e3 / n, (double)h[0] / n, h[0] / us);
    return 0;
}
int main() {
    {
        float* vo; unsigned long long* vs;
        CK(hipMalloc(&vo, 1024 * 64 * 4)); CK(hipMalloc(&vs, 16));
        if (valu<0>("v_mul_f32, one dependent chain", 1, vo, vs)) return 1;
        if (valu<1>("v_mul_f32, three independent chains", 3, vo, vs)) return 1;
        if (valu<2>("v_pk_mul_f32, one dependent chain", 1, vo, vs)) return 1;
        if (valu<3>("v_pk_mul_f32, three independent chains", 3, vo, vs)) return 1;
    }
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
