// Diagnostic: which XCDs / CUs does a CU-masked stream reach?  (tools only)
// build: hipcc --offload-arch=gfx950 -O2 -o cumask_probe cumask_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <set>
#include <vector>
__global__ void k_probe(uint32_t* out) {
    uint32_t x, h;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(h));
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = x; out[2 * blockIdx.x + 1] = h; }
}
static void run(const char* name, hipStream_t s, uint32_t* d, int g) {
    hipLaunchKernelGGL(k_probe, dim3(g), dim3(256), 0, s, d);
    hipStreamSynchronize(s);
    std::vector<uint32_t> h(2 * g);
    hipMemcpy(h.data(), d, 8 * g, hipMemcpyDeviceToHost);
    int per_xcc[16] = {0};
    std::set<uint32_t> cus;
    for (int b = 0; b < g; ++b) { per_xcc[h[2 * b] & 0xF]++; cus.insert(((h[2 * b] & 0xF) << 16) | (h[2 * b + 1] & 0xFF00)); }  // CU_ID [11:8], SH [12], SE [15:13]
    printf("%-28s blocks per XCC:", name);
    for (int x = 0; x < 8; ++x) printf(" %4d", per_xcc[x]);
    printf("   distinct (xcc, se, sh, cu): %zu\n", cus.size());
}
int main() {
    uint32_t* d;
    hipMalloc(&d, 1 << 20);
    hipStream_t s0; hipStreamCreateWithFlags(&s0, hipStreamNonBlocking);
    run("unmasked", s0, d, 4096);
    uint32_t m1[8]; for (int i = 0; i < 8; ++i) m1[i] = 0x000000FFu;   // 8 CUs of every 32
    hipStream_t s1; hipError_t e = hipExtStreamCreateWithCUMask(&s1, 8, m1);
    printf("create masked(0xFF per word): %s\n", hipGetErrorName(e));
    if (e == hipSuccess) run("mask 0x000000FF x8", s1, d, 4096);
    uint32_t m2[8]; for (int i = 0; i < 8; ++i) m2[i] = 0xFFFFFF00u;
    hipStream_t s2; e = hipExtStreamCreateWithCUMask(&s2, 8, m2);
    if (e == hipSuccess) run("mask 0xFFFFFF00 x8", s2, d, 4096);
    uint32_t m3[8] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0, 0, 0, 0, 0, 0};
    hipStream_t s3; e = hipExtStreamCreateWithCUMask(&s3, 8, m3);
    if (e == hipSuccess) run("mask first 64 bits", s3, d, 4096);
    uint32_t m4[8]; for (int i = 0; i < 8; ++i) m4[i] = 0x11111111u;   // every 4th bit
    hipStream_t s4; e = hipExtStreamCreateWithCUMask(&s4, 8, m4);
    if (e == hipSuccess) run("mask 0x11111111 x8", s4, d, 4096);
    return 0;
}
