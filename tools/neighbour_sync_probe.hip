// Diagnostic (tools only): what a NEIGHBOUR-ONLY synchronisation between the workgroups of one persistent launch costs per round on this
// chip, next to the device-wide barrier that was measured (and rejected) in round 1.  The question behind it: could
// the colour passes of one big island run as ONE persistent launch of spatial tiles that exchange their boundary bodies through HBM and
// wait only for their ~6 neighbours per colour, instead of one kernel boundary per colour (1.6 us + cold caches)?
//   per round, workgroup g:  write a halo buffer (HALO bytes, agent-scope stores) -> release -> flag[g] = round
//                            -> wait until flag[n] >= round for its neighbours n -> acquire -> read the neighbours' halo buffers
//   neighbours of g: g +- 1, g +- 8, g +- 64 (mod G): +-1 are on other XCDs (workgroups are dealt round-robin), +-8 / +-64 on the same XCD
//   variant "barrier": one arrival counter for all G workgroups instead of the neighbour flags
// build: hipcc --offload-arch=gfx950 -O2 -o neighbour_sync_probe neighbour_sync_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
constexpr int G = 256, T = 256, ROUNDS = 400, HALO_WORDS = 1024;   // 4 KB per workgroup per round
__device__ __forceinline__ void st_agent(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ uint32_t ld_agent(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <int MODE, bool FENCES>   // 0 = neighbour flags, 1 = device-wide counter, 2 = no synchronisation at all (the halo traffic alone)
__global__ __launch_bounds__(T) void k_rounds(uint32_t* flags, uint32_t* counter, uint32_t* halo, uint32_t* sink, uint32_t* timeout) {
    const int g = blockIdx.x;
    const int nb[6] = {(g + 1) % G, (g + G - 1) % G, (g + 8) % G, (g + G - 8) % G, (g + 64) % G, (g + G - 64) % G};
    uint32_t acc = 0;
    for (uint32_t r = 1; r <= ROUNDS; ++r) {
        uint32_t* mine = halo + ((size_t)(r & 1) * G + g) * HALO_WORDS;   // double-buffered by round parity
        for (int i = threadIdx.x; i < HALO_WORDS; i += T) st_agent(mine + i, r * 31u + i + acc);
        // FENCES: a real agent-scope release (buffer_wbl2 sc1: the XCD's L2 is written back) / acquire (buffer_inv sc1: it is invalidated).
        // !FENCES: every shared access above and below is an agent-scope (sc1, write-through / L2-bypassing) access already, so only the ORDER
        // "my stores have completed before my flag is stored" is needed: a workgroup-scope fence = s_waitcnt vmcnt(0), no cache maintenance.
        if (FENCES) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); else __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __syncthreads();
        if (MODE == 0) {
            if (threadIdx.x == 0) st_agent(flags + g, r);
            if (threadIdx.x < 6) {
                uint32_t spins = 0;
                while (ld_agent(flags + nb[threadIdx.x]) < r)
                    if (++spins > (1u << 22)) { *timeout = 1; break; }
            }
        } else if (MODE == 1) {
            if (threadIdx.x == 0) {
                __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                uint32_t spins = 0;
                while (ld_agent(counter) < r * G)
                    if (++spins > (1u << 22)) { *timeout = 1; break; }
            }
        }
        __syncthreads();
        if (FENCES) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        for (int k = 0; k < 6; ++k) {
            const uint32_t* theirs = halo + ((size_t)(r & 1) * G + nb[k]) * HALO_WORDS;
            for (int i = threadIdx.x; i < HALO_WORDS / 6; i += T) acc += ld_agent(theirs + i);
        }
    }
    if (acc == 0x12345u) sink[g] = acc;
}
template <int MODE, bool FENCES> static int run(const char* name, uint32_t* flags, uint32_t* counter, uint32_t* halo, uint32_t* sink, uint32_t* timeout) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
        CK(hipMemset(flags, 0, G * 4)); CK(hipMemset(counter, 0, 4)); CK(hipMemset(timeout, 0, 4));
        CK(hipEventRecord(a, 0));
        hipLaunchKernelGGL((k_rounds<MODE, FENCES>), dim3(G), dim3(T), 0, 0, flags, counter, halo, sink, timeout);
        CK(hipEventRecord(b, 0));
        CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        if (ms < best) best = ms;
    }
    uint32_t to = 0; CK(hipMemcpy(&to, timeout, 4, hipMemcpyDeviceToHost));
    std::printf("%-40s %7.2f us per round%s\n", name, best * 1e3 / ROUNDS, to ? "   (a spin TIMED OUT: workgroups not co-resident?)" : "");
    return 0;
}
int main() {
    uint32_t *flags, *counter, *halo, *sink, *timeout;
    CK(hipMalloc(&flags, G * 4)); CK(hipMalloc(&counter, 4)); CK(hipMalloc(&halo, (size_t)2 * G * HALO_WORDS * 4)); CK(hipMalloc(&sink, G * 4)); CK(hipMalloc(&timeout, 4));
    CK(hipMemset(halo, 0, (size_t)2 * G * HALO_WORDS * 4));
    std::printf("%d workgroups x %d lanes, %d rounds, %d B halo written + ~%d B read per workgroup per round\n", G, T, ROUNDS, HALO_WORDS * 4, HALO_WORDS * 4);
    if (run<2, false>("traffic only, no cache maintenance", flags, counter, halo, sink, timeout)) return 1;
    if (run<0, false>("6-neighbour flags, sc1 accesses only", flags, counter, halo, sink, timeout)) return 1;
    if (run<1, false>("device-wide counter, sc1 only", flags, counter, halo, sink, timeout)) return 1;
    if (run<2, true>("traffic only + wbl2 / inv fences", flags, counter, halo, sink, timeout)) return 1;
    if (run<0, true>("6-neighbour flags + fences", flags, counter, halo, sink, timeout)) return 1;
    if (run<1, true>("device-wide counter + fences", flags, counter, halo, sink, timeout)) return 1;
    return 0;
}
