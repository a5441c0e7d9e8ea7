// Diagnostic: which XCD does workgroup b of a launch run on?  (tools only, not part of the library)
// build: hipcc --offload-arch=gfx950 -O2 -o xcc_probe xcc_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k_probe(uint32_t* out) {
    uint32_t x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    if (threadIdx.x == 0) out[blockIdx.x] = x;
}
__global__ void k_spin(uint32_t* out, int n) { uint32_t v = 0; for (int i = 0; i < n; ++i) v += __shfl_xor(v + i, 1); if (v == 12345) out[0] = v; }
int main() {
    const int grids[] = {6256, 872, 391, 1568};
    uint32_t* d;
    hipMalloc(&d, 1 << 20);
    for (int rep = 0; rep < 3; ++rep)
        for (int g : grids) {
            hipLaunchKernelGGL(k_spin, dim3(391), dim3(256), 0, 0, d + 100000, 100);
            hipLaunchKernelGGL(k_probe, dim3(g), dim3(256), 0, 0, d);
            std::vector<uint32_t> h(g);
            hipMemcpy(h.data(), d, g * 4, hipMemcpyDeviceToHost);
            int match[8] = {0}, tot = 0;
            // best rotation r such that xcc == (b + r) % 8
            for (int r = 0; r < 8; ++r) for (int b = 0; b < g; ++b) if ((h[b] & 0xF) == (uint32_t)((b + r) % 8)) ++match[r];
            int best = 0; for (int r = 1; r < 8; ++r) if (match[r] > match[best]) best = r;
            (void)tot;
            printf("grid %5d: xcc of blocks 0..15:", g);
            for (int b = 0; b < 16; ++b) printf(" %u", h[b] & 0xF);
            printf("  | best rotation %d matches %d / %d\n", best, match[best], g);
        }
    return 0;
}
