// tools/xcd_probe.hip -- can a small graph's sweeps run as ONE persistent launch confined to one XCD?
//   1. does hipExtStreamCreateWithCUMask confine a launch to the CUs of one XCD, and which mask bits are those?
//   2. what does a barrier among the workgroups of one XCD cost (relaxed agent-scope atomics: same L2, no cache maintenance),
//      against a barrier among 256 workgroups on all XCDs with release / acquire fences, and against a kernel boundary?
// hipcc --offload-arch=gfx950 -O3 tools/xcd_probe.hip -o /tmp/xcd_probe && /tmp/xcd_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <map>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void k_where(int *out)
{
    if (threadIdx.x == 0) {
        unsigned xcc = 0, hwid = 0;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        out[2 * blockIdx.x] = (int)(xcc & 0xf);
        out[2 * blockIdx.x + 1] = (int)hwid;
    }
    // stay a little so that the grid spreads out instead of reusing the first CU
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < 2000) {}
}

// `rounds` barriers among gridDim.x workgroups on one counter (monotonic), bounded spin; fenced = agent-scope release / acquire
template <bool FENCED>
__global__ void k_barrier(unsigned *ctr, int rounds, int *fail, long long *ticks)
{
    const long long t0 = wall_clock64();
    for (int r = 1; r <= rounds; ++r) {
        __syncthreads();
        if (threadIdx.x == 0) {
            if (FENCED) __atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE); else __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned want = (unsigned)r * gridDim.x;
            const long long s0 = wall_clock64();
            for (;;) {
                const unsigned v = FENCED ? __atomic_load_n(ctr, __ATOMIC_ACQUIRE) : __hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (v >= want) break;
                if (wall_clock64() - s0 > 100000000LL) { *fail = 1; break; }      // 1 s at 100 MHz: give up instead of hanging
                __builtin_amdgcn_s_sleep(1);
            }
        }
        __syncthreads();
        if (*fail) break;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) *ticks = wall_clock64() - t0;
}

__global__ void k_empty(int *p) { if (p && threadIdx.x == 12345) *p = 1; }

static int place(hipStream_t s, int blocks, const char *what)
{
    int *d; CHECK(hipMalloc(&d, sizeof(int) * 2 * blocks));
    hipLaunchKernelGGL(k_where, dim3(blocks), dim3(64), 0, s, d);
    CHECK(hipStreamSynchronize(s));
    std::vector<int> h(2 * blocks); CHECK(hipMemcpy(h.data(), d, sizeof(int) * 2 * blocks, hipMemcpyDeviceToHost));
    std::map<int, int> per_xcc; std::map<long long, int> per_cu;
    for (int b = 0; b < blocks; ++b) { per_xcc[h[2 * b]]++; per_cu[((long long)h[2 * b] << 32) | (unsigned)(h[2 * b + 1] & 0xffff0f00)]++; }
    printf("%-34s %3d workgroups on XCCs:", what, blocks);
    for (auto &kv : per_xcc) printf(" %d:%d", kv.first, kv.second);
    printf("   distinct (xcc, se/cu) places: %zu\n", per_cu.size());
    (void)hipFree(d);
    return 0;
}

int main()
{
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    printf("%s, %d CUs\n", prop.name, prop.multiProcessorCount);
    hipStream_t plain; CHECK(hipStreamCreate(&plain));
    place(plain, 256, "no mask");
    // candidate masks for "one XCD": the first 32 bits; every 8th bit
    struct Cand { const char *name; std::vector<uint32_t> m; };
    std::vector<Cand> cands;
    { Cand c{"bits 0..31", std::vector<uint32_t>(8, 0u)}; c.m[0] = 0xffffffffu; cands.push_back(c); }
    { Cand c{"every 8th bit (i % 8 == 0)", std::vector<uint32_t>(8, 0x01010101u)}; cands.push_back(c); }
    { Cand c{"bits 32..63", std::vector<uint32_t>(8, 0u)}; c.m[1] = 0xffffffffu; cands.push_back(c); }
    hipStream_t one_xcd = nullptr;
    for (auto &c : cands) {
        hipStream_t s;
        hipError_t e = hipExtStreamCreateWithCUMask(&s, (uint32_t)c.m.size(), c.m.data());
        if (e != hipSuccess) { printf("mask %-28s hipExtStreamCreateWithCUMask -> %s\n", c.name, hipGetErrorString(e)); continue; }
        char what[96]; snprintf(what, sizeof what, "mask %s", c.name);
        place(s, 64, what);
        if (!one_xcd && c.name[0] == 'e') one_xcd = s;
    }
    unsigned *ctr; int *fail; long long *ticks;
    CHECK(hipMalloc(&ctr, 4)); CHECK(hipMalloc(&fail, 4)); CHECK(hipMalloc(&ticks, 8));
    auto run = [&](hipStream_t s, int blocks, bool fenced, const char *what) {
        const int rounds = 2000;
        (void)hipMemsetAsync(ctr, 0, 4, s); (void)hipMemsetAsync(fail, 0, 4, s);
        if (fenced) hipLaunchKernelGGL(k_barrier<true>, dim3(blocks), dim3(512), 0, s, ctr, rounds, fail, ticks);
        else hipLaunchKernelGGL(k_barrier<false>, dim3(blocks), dim3(512), 0, s, ctr, rounds, fail, ticks);
        (void)hipStreamSynchronize(s);
        long long t = 0; int f = 0; (void)hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost); (void)hipMemcpy(&f, fail, 4, hipMemcpyDeviceToHost);
        printf("%-58s %6.2f us per barrier%s\n", what, t / 100.0 / rounds, f ? "  (TIMED OUT: not co-resident?)" : "");
    };
    run(plain, 32, false, "32 workgroups anywhere, relaxed agent-scope atomics");
    run(plain, 32, true, "32 workgroups anywhere, release / acquire");
    run(plain, 256, false, "256 workgroups, relaxed agent-scope atomics");
    run(plain, 256, true, "256 workgroups, release / acquire");
    if (one_xcd) {
        run(one_xcd, 32, false, "32 workgroups on the masked stream, relaxed");
        run(one_xcd, 32, true, "32 workgroups on the masked stream, release / acquire");
    }
    // kernel boundary: empty launches back to back
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 100; ++i) hipLaunchKernelGGL(k_empty, dim3(256), dim3(512), 0, plain, nullptr);
    hipEventRecord(e0, plain);
    for (int i = 0; i < 2000; ++i) hipLaunchKernelGGL(k_empty, dim3(256), dim3(512), 0, plain, nullptr);
    hipEventRecord(e1, plain); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-58s %6.2f us per launch\n", "empty 256 x 512 kernels back to back", ms * 1e3 / 2000);
    return 0;
}
