// Diagnostic (tools only): what ONE hand-over of a 16-byte record between two waves costs on this chip, by cache-coherence scope and by where the two waves run.
// Two single-wave workgroups play ping-pong through two records (value + tag in the w lane, as k_overflow_flow_tag hands a body's velocity over): A stores
// (r, tag 2r+1), B polls until it sees that tag, stores (r, tag 2r+2), A polls for it.  One-way latency = time / (2 * rounds).  Workgroups are dealt to the
// XCDs round-robin (block b on XCD b % 8): blocks (0, 1) are on different XCDs, (0, 8) on the same one.
// aux bits of the raw buffer instructions on gfx942 / gfx950: 1 = sc0, 16 = sc1 (sc1 alone = agent scope, what the solver uses), 2 = nt.
// A combination under which the partner's store never becomes visible (served from a stale cache forever) shows up as TIMEOUT: that scope is not coherent there.
// build: hipcc --offload-arch=gfx950 -O2 -o handover_probe handover_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
constexpr int ROUNDS = 2000;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(void* base) { return __builtin_amdgcn_make_buffer_rsrc(base, 0, 0x7FFFFFFF, 0x00020000); }
template <int LD_AUX, int ST_AUX>
__global__ __launch_bounds__(64) void k_pingpong(uint32_t* rec, uint32_t block_a, uint32_t block_b, uint32_t* out) {
    const bool is_a = blockIdx.x == block_a, is_b = blockIdx.x == block_b;
    if (!is_a && !is_b) return;
    if (threadIdx.x != 0) return;   // (one lane: the latency of the path, not its throughput)
    const __amdgpu_buffer_rsrc_t rs = rsrc(rec);
    const int mine = is_a ? 0 : 64, theirs = is_a ? 64 : 0;   // two records in different 64-byte lines
    uint32_t bad = 0, timeout = 0;
    for (uint32_t r = 1; r <= ROUNDS && !timeout; ++r) {
        if (is_a) __builtin_amdgcn_raw_buffer_store_b128(u32x4{r, r * 3u, r * 5u, 2u * r + 1u}, rs, mine, 0, ST_AUX);
        const uint32_t want = is_a ? 2u * r + 2u : 2u * r + 1u;
        u32x4 v;
        uint32_t spins = 0;
        for (;;) {
            asm volatile("" ::: "memory");   // (a fresh load every round: the builtin is not volatile)
            v = __builtin_amdgcn_raw_buffer_load_b128(rs, theirs, 0, LD_AUX);
            if (v.w == want) break;
            if (++spins > (1u << 18)) { timeout = 1; break; }
        }
        if (!timeout && (v.x != r || v.y != r * 3u || v.z != r * 5u)) ++bad;
        if (is_b && !timeout) __builtin_amdgcn_raw_buffer_store_b128(u32x4{r, r * 3u, r * 5u, 2u * r + 2u}, rs, mine, 0, ST_AUX);
    }
    out[is_a ? 0 : 2] = bad; out[is_a ? 1 : 3] = timeout;
}
template <int LD_AUX, int ST_AUX> static int run(const char* name, uint32_t* rec, uint32_t* out, uint32_t a, uint32_t b) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f; uint32_t h[4] = {0, 0, 0, 0};
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipMemset(rec, 0, 256)); CK(hipMemset(out, 0, 16)); CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((k_pingpong<LD_AUX, ST_AUX>), dim3(16), dim3(64), 0, 0, rec, a, b, out);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
        CK(hipMemcpy(h, out, 16, hipMemcpyDeviceToHost));
        if (h[1] || h[3]) break;
    }
    if (h[1] || h[3]) std::printf("%-58s blocks (%2u, %2u): TIMEOUT (the partner's store never became visible)\n", name, a, b);
    else std::printf("%-58s blocks (%2u, %2u): %6.2f us one way%s\n", name, a, b, best * 1e3f / (2.0f * ROUNDS), (h[0] || h[2]) ? "  PAYLOAD MISMATCH" : "");
    return 0;
}
int main() {
    uint32_t *rec, *out; CK(hipMalloc(&rec, 256)); CK(hipMalloc(&out, 16));
    const uint32_t pairs[3][2] = {{0, 1}, {0, 8}, {0, 4}};
    for (auto& p : pairs) {
        if (run<16, 16>("loads sc1, stores sc1 (agent scope: the solver's hand-over)", rec, out, p[0], p[1])) return 1;
        if (run<17, 17>("loads sc0 sc1, stores sc0 sc1 (system scope)", rec, out, p[0], p[1])) return 1;
        if (run<1, 16>("loads sc0, stores sc1", rec, out, p[0], p[1])) return 1;
        if (run<1, 1>("loads sc0, stores sc0 (group scope)", rec, out, p[0], p[1])) return 1;
        if (run<1, 0>("loads sc0, plain stores", rec, out, p[0], p[1])) return 1;
        if (run<0, 0>("plain loads, plain stores", rec, out, p[0], p[1])) return 1;
    }
    return 0;
}
