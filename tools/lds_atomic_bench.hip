// tools/lds_atomic_bench.hip -- how fast is ds_add_f64 (the camera accumulation of the fused sweep)?
//   hipcc --offload-arch=gfx950 -O3 -o tools/lds_atomic_bench.bin tools/lds_atomic_bench.hip && tools/lds_atomic_bench.bin
// One workgroup per CU, W waves; every wave does R rounds of 27 adds into a [500][27] table in LDS.  Patterns:
//   0 random cameras per lane (the sweep)   1 cameras = lane (64 distinct, stride 27 doubles)   2 all lanes 8 cameras
//   3 conflict-free by construction: lane l adds to double index l (+64 per k)                  4 pattern 0 with ds_add_f32
//   5 pattern 0 as read / add / write       6 pattern 0, only 4 lanes active (a duplicate round)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int PAT>
__global__ __launch_bounds__(512) void k_bench(const int *__restrict__ cams, int rounds, double *out, long long *ticks)
{
    extern __shared__ double acc[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 500 * 27; i += blockDim.x) acc[i] = 0.0;
    __syncthreads();
    double v[27];
#pragma unroll
    for (int k = 0; k < 27; ++k) v[k] = 1.0 + 0.001 * k + lane;
    const long long t0 = wall_clock64();
    for (int r = 0; r < rounds; ++r) {
        const int cam = cams[(size_t)((blockIdx.x * 8 + wave) * rounds + r) * 64 + lane];
        if (PAT == 6 && lane >= 4) continue;
        if (PAT == 3) {
#pragma unroll
            for (int k = 0; k < 27; ++k) unsafeAtomicAdd(acc + lane + 64 * k, v[k]);
        } else if (PAT == 4) {
            float *f = reinterpret_cast<float *>(acc) + cam * 54;
#pragma unroll
            for (int k = 0; k < 27; ++k) unsafeAtomicAdd(f + 2 * k, (float)v[k]);
        } else if (PAT == 5) {
            double *dst = acc + cam * 27, w[27];
#pragma unroll
            for (int k = 0; k < 27; ++k) w[k] = dst[k];
#pragma unroll
            for (int k = 0; k < 27; ++k) dst[k] = w[k] + v[k];
        } else {
            double *dst = acc + cam * 27;
#pragma unroll
            for (int k = 0; k < 27; ++k) unsafeAtomicAdd(dst + k, v[k]);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    const long long t1 = wall_clock64();
    __syncthreads();
    if (threadIdx.x == 0) { ticks[blockIdx.x] = t1 - t0; out[blockIdx.x] = acc[27] + acc[28]; }
}

template <int PAT>
static void run(const char *name, int waves, int n_wg, int rounds, const int *d_cams, double *d_out, long long *d_ticks)
{
    hipFuncSetAttribute(reinterpret_cast<const void *>(&k_bench<PAT>), hipFuncAttributeMaxDynamicSharedMemorySize, 500 * 27 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k_bench<PAT>, dim3(n_wg), dim3(waves * 64), 500 * 27 * 8, 0, d_cams, rounds, d_out, d_ticks);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
    }
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> t(n_wg); hipMemcpy(t.data(), d_ticks, n_wg * sizeof(long long), hipMemcpyDeviceToHost);
    double mean = 0; for (auto x : t) mean += (double)x; mean /= n_wg;
    // wall_clock64 ticks at 100 MHz
    const double us = mean / 100.0;
    printf("%-52s waves %d: %8.1f us for %d rounds x 27 adds per wave -> %6.1f ns per wave-instruction, %6.2f ns per CU-instruction (kernel %.1f us)\n",
           name, waves, us, rounds, 1e3 * us / (rounds * 27.0), 1e3 * us / (rounds * 27.0 * waves), ms * 1e3);
}

int main()
{
    const int n_wg = 256, rounds = 200;
    std::vector<int> cams((size_t)n_wg * 8 * rounds * 64);
    unsigned s = 12345;
    for (size_t i = 0; i < cams.size(); ++i) { s = s * 1664525u + 1013904223u; cams[i] = (int)((s >> 8) % 500u); }
    std::vector<int> c1 = cams, c2 = cams;
    for (size_t i = 0; i < c1.size(); ++i) { c1[i] = (int)(i & 63); c2[i] = (int)(i & 7); }
    int *d0, *d1, *d2; double *d_out; long long *d_ticks;
    hipMalloc(&d0, cams.size() * 4); hipMalloc(&d1, cams.size() * 4); hipMalloc(&d2, cams.size() * 4);
    hipMalloc(&d_out, n_wg * 8); hipMalloc(&d_ticks, n_wg * 8);
    hipMemcpy(d0, cams.data(), cams.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(d1, c1.data(), cams.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(d2, c2.data(), cams.size() * 4, hipMemcpyHostToDevice);
    for (int waves : {1, 8}) {
        run<0>("0 ds_add_f64, 64 random cameras", waves, n_wg, rounds, d0, d_out, d_ticks);
        run<0>("1 ds_add_f64, cameras = lane (stride 27 doubles)", waves, n_wg, rounds, d1, d_out, d_ticks);
        run<0>("2 ds_add_f64, 8 cameras (8 lanes per address)", waves, n_wg, rounds, d2, d_out, d_ticks);
        run<3>("3 ds_add_f64, consecutive doubles (conflict-free)", waves, n_wg, rounds, d0, d_out, d_ticks);
        run<4>("4 ds_add_f32, 64 random cameras", waves, n_wg, rounds, d0, d_out, d_ticks);
        run<5>("5 read / add / write, 64 random cameras", waves, n_wg, rounds, d0, d_out, d_ticks);
        run<6>("6 ds_add_f64, 4 lanes active", waves, n_wg, rounds, d0, d_out, d_ticks);
    }
    return 0;
}
