// Diagnostic (tools only): do lines a kernel pulled into an XCD's L2 survive the kernel boundary, so that the NEXT kernel of a replayed
// graph hits them?  (The premise of prefetching colour c + 1's manifold constants during colour c's impulse chain.)
//   k_touch        : block b loads slice b of a 16 MB table (2 MB per XCD if blocks are dealt round-robin: fits the 4 MB L2)
//   k_read_same    : block b loads slice b again            -> L2 hits if lines survive the boundary AND block b lands on the same XCD
//   k_read_shift   : block b loads slice b + 1              -> another XCD's lines: the control (must miss L2)
//   k_read_cold    : block b loads slice b of ANOTHER table -> never touched: the cold reference
// Run under `rocprofv3 --pmc FETCH_SIZE --kernel-trace` for the L2 -> fabric request bytes per kernel, and plain for the timings.
// build: hipcc --offload-arch=gfx950 -O2 -o l2_retention_probe l2_retention_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
constexpr int GRID = 1024, PER = 64;   // 1024 blocks x 64 lanes x 64 x 4 B... = 16 MB: lane loads PER float4
__device__ __forceinline__ float4 slice_sum(const float4* __restrict__ t, uint32_t slice) {
    float4 a = make_float4(0, 0, 0, 0);
#pragma unroll 8
    for (int j = 0; j < PER / 4; ++j) {   // 16 float4 per lane: 64 lanes x 16 x 16 B = 16 KB per block, 16 MB per grid
        float4 v = t[(size_t)slice * 64 * (PER / 4) + j * 64 + threadIdx.x];
        a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
    return a;
}
__global__ __launch_bounds__(64) void k_touch(const float4* t, float* out) { float4 a = slice_sum(t, blockIdx.x); if (a.x == 123.f) out[0] = a.y; }
__global__ __launch_bounds__(64) void k_read_same(const float4* t, float* out) { float4 a = slice_sum(t, blockIdx.x); if (a.x == 123.f) out[0] = a.y; }
__global__ __launch_bounds__(64) void k_read_shift(const float4* t, float* out) { float4 a = slice_sum(t, (blockIdx.x + 1) % GRID); if (a.x == 123.f) out[0] = a.y; }
__global__ __launch_bounds__(64) void k_read_cold(const float4* t, float* out) { float4 a = slice_sum(t, blockIdx.x); if (a.x == 123.f) out[0] = a.y; }
__global__ __launch_bounds__(256) void k_flush(const float4* t, float* out, size_t n) {   // streams 512 MB: nothing of the tables is left in L2 / Infinity Cache
    float a = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) a += t[i].x;
    if (a == 123.f) out[0] = a;
}
int main() {
    hipStream_t st; CK(hipStreamCreate(&st));
    float4 *A, *B, *F; float* out;
    const size_t bytes = (size_t)GRID * 64 * (PER / 4) * 16, fbytes = 512u << 20;
    CK(hipMalloc(&A, bytes)); CK(hipMalloc(&B, bytes)); CK(hipMalloc(&F, fbytes)); CK(hipMalloc(&out, 64));
    CK(hipMemset(A, 0, bytes)); CK(hipMemset(B, 0, bytes)); CK(hipMemset(F, 0, fbytes));
    hipEvent_t e[8]; for (auto& x : e) CK(hipEventCreate(&x));
    const int REPS = 50;
    double t_same = 0, t_shift = 0, t_cold = 0, t_touch = 0;
    for (int r = 0; r < REPS; ++r) {
        hipLaunchKernelGGL(k_flush, dim3(2048), dim3(256), 0, st, F, out, fbytes / 16);
        CK(hipEventRecord(e[0], st));
        hipLaunchKernelGGL(k_touch, dim3(GRID), dim3(64), 0, st, A, out);
        CK(hipEventRecord(e[1], st));
        hipLaunchKernelGGL(k_read_same, dim3(GRID), dim3(64), 0, st, A, out);
        CK(hipEventRecord(e[2], st));
        hipLaunchKernelGGL(k_read_shift, dim3(GRID), dim3(64), 0, st, A, out);
        CK(hipEventRecord(e[3], st));
        hipLaunchKernelGGL(k_read_cold, dim3(GRID), dim3(64), 0, st, B, out);
        CK(hipEventRecord(e[4], st));
        CK(hipStreamSynchronize(st));
        float ms;
        CK(hipEventElapsedTime(&ms, e[0], e[1])); t_touch += ms;
        CK(hipEventElapsedTime(&ms, e[1], e[2])); t_same += ms;
        CK(hipEventElapsedTime(&ms, e[2], e[3])); t_shift += ms;
        CK(hipEventElapsedTime(&ms, e[3], e[4])); t_cold += ms;
    }
    std::printf("16 MB per kernel, %d blocks x 64 lanes; stream launches with events (event overhead included in every figure)\n", GRID);
    std::printf("k_touch (after a 512 MB flush)      %7.2f us\n", t_touch * 1e3 / REPS);
    std::printf("k_read_same  (same block, same data) %7.2f us\n", t_same * 1e3 / REPS);
    std::printf("k_read_shift (neighbour block's data) %6.2f us\n", t_shift * 1e3 / REPS);
    std::printf("k_read_cold  (untouched table)       %7.2f us\n", t_cold * 1e3 / REPS);
    return 0;
}
