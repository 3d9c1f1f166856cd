// gbp_fused_plan.hpp -- host-side description of the fused sweep's launch (no kernels): the constants that size its LDS, the kernel
// argument block and the per-handle plan.  Shared by every translation unit that sees the handle (gbp_handle.hpp); the kernels and the
// functions that build and launch a plan are in gbp_fused.hpp (gbp_capi_sweep.hip only).
#pragma once
#include "gbp_kernels.hpp"
#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <vector>

namespace gbp {

constexpr int LDS_BYTES = 160 * 1024;
constexpr int TROW = 28;                            // doubles per row of the workgroup tables in HBM (27 + pad: 16-byte stores / loads)
#ifndef GBP_WAT_WAVES
#define GBP_WAT_WAVES 8
#endif
constexpr int WAT_WAVES = GBP_WAT_WAVES;            // two waves per SIMD.  Round 4 built the three-waves-per-SIMD variant the covariance-form
                                                    // factor core makes possible (-DGBP_WAT_WAVES=12: <= 168 VGPRs, twelve [64][9] message
                                                    // scratches beside a 500-camera table) and measured it SLOWER on MI355X: 118 us per sweep
                                                    // with the relinearisation path in the kernel (55 registers spilled), 99-102 us without it
                                                    // (9 spilled, nothing in the loop) against 66.8 us for eight waves -- every phase that
                                                    // only issues vector-memory instructions takes twice as long, and the slowest workgroup
                                                    // finishes 33 % after the mean (profiles/r04_waves12_*.txt).  The CU's memory pipeline is
                                                    // the limit; more waves queue in front of it.
constexpr int WAVE_LDS_DOUBLES = WTILE * 9;        // per-wave scratch: [24][10] landmark heads (mean | covariance | rows), then [64][9] messages
static_assert(TILE_LMKS * LHEAD <= WAVE_LDS_DOUBLES, "landmark heads of a tile must fit the wave scratch");

// Instrumentation of the persistent loop -- the GBP_FUSED_DBG ablation switches and the per-phase s_memtime profile -- lives in
// experimental/gbp_instrument.hpp and is compiled only into the scratch libraries that tools/profile_round.sh (-DGBP_FUSED_DBG_SWITCHES)
// and tools/phase_profile.py (-DGBP_PHASE_TIMING) build.  The product build sees empty macros: no code, no kernel arguments.
constexpr int NPHASE = 12;          // marks of the phase profile (experimental/gbp_instrument.hpp)
#if defined(GBP_FUSED_DBG_SWITCHES) || defined(GBP_PHASE_TIMING)
#include "experimental/gbp_instrument.hpp"
#else
#define GBP_DBG(a, bit) 0
#define GBP_PH_DECL
#define GBP_PH(i)
#define GBP_PH_NOWAIT(i)
#define GBP_PH_FLUSH(ptr, row)
#define GBP_INSTRUMENT_ARGS
#endif

struct FusedArgs {
    double *block_partials;     // [C][n_blocks][TROW]: 27 sums + one pad double, so that a row starts on 16 bytes (WINDOWED: the rows of a
                                // camera are those of the workgroups whose camera set holds it, in workgroup order: FusedPlan::d_cam_rows)
    int acc_doubles;            // cameras of the group * 27
    int cam_base, cam_count;    // the cameras whose messages THIS launch adds up in its LDS table (all of them when C fits)
    int reverse;                // walk the workgroup's tile range backwards (every other sweep: see fused_launch)
    int nt;                     // which factor streams bypass the memory-side cache (issue_streams)
    GBP_INSTRUMENT_ARGS         // (scratch builds of tools/ only: `int dbg; unsigned long long *phase;`)
    unsigned long long *clk;    // instrumented runs (gbp_ba_set_kernel_timing): where workgroup 0 stores the device's constant-rate clock
                                // (wall_clock64) when it starts, or NULL.  HIP events around a launch also time the dispatch after the
                                // event's barrier packet (~5-8 us) and serialise the stream; this does neither.
    int pin;                    // the first `pin` tiles of every workgroup's range use the memory-side cache as `nt` says; the rest stream
                                // PAST it altogether, loads and message stores (fused_plan: graphs beyond the cache size)
    const int4 *win;            // WINDOWED: per workgroup {lowest camera, cameras in its set, offset into wgcams / rowidx, width of the interval the set lies in}, else NULL
    const int *wgcams;          // WINDOWED: wgcams[offset + k] = the k-th camera of the workgroup's set (ascending): its table row k
    const int *rowidx;          // WINDOWED: rowidx[offset + k] = the row of block_partials that table row is written to
    int full_rows;              // STAGED: write whole camera-message rows (the staged x0 halves cannot be trusted: first staged sweep after
                                // create / restore / a sweep of another kind); else only tiles in which a factor relinearised do
};

struct FusedPlan {
    bool enabled = false;
    int n_groups = 0, group_cams = 0;                // (one camera group: the whole table in LDS)
    int n_blocks = 0;
    int xchg_blocks = 0;                             // grid of the merged reduce-exchange-finish launch (0: not asked yet)
    long long table_rows = 0;                        // rows of block_partials
    int windowed = 0, max_window = 0, max_width = 0; // camera WINDOWS: every workgroup's table covers the cameras its own tiles meet (fused_plan): the largest
                                                     // set, and the widest interval lowest .. highest camera (the 16-bit map camera -> table row covers it)
    int rows_wave = 0;                               // windowed and at most ROWS_WAVE_MAX rows per camera on average: the reduce runs one wave per camera
    const int2 *d_cam_rows = nullptr;                // windowed: per camera {first row, rows} of block_partials, else NULL (camera c: rows c n .. c n + n - 1)
    int single = 0;                                  // launch the SINGLE variant (all same-camera lanes of a tile in one ds_add_f64 per entry)
    int single_probe = -1;                           // -1: SINGLE not wanted, no probe; 1: the device's lane order was verified; 0: it failed, rounds variant instead
    int single_probe_mask = 0;                       // failing patterns of k_single_probe
    size_t shmem = 0;
    FusedArgs args{};
    std::vector<void *> allocs;
    void *(*alloc)(void *ctx, size_t bytes) = nullptr;   // optional: take device memory from the owner's arena (else hipMalloc)
    void *alloc_ctx = nullptr;
};

inline void fused_destroy(FusedPlan &pl)
{
    for (void *q : pl.allocs) (void)hipFree(q);
    pl.allocs.clear();
    pl.enabled = false;
}

template <typename T>
inline int fused_upload(FusedPlan &pl, T **dst, const T *src, size_t n, hipStream_t stream)
{
    void *q = pl.alloc ? pl.alloc(pl.alloc_ctx, std::max<size_t>(n, 1) * sizeof(T)) : nullptr;
    if (!q) {
        if (hipMalloc(&q, std::max<size_t>(n, 1) * sizeof(T)) != hipSuccess) return -1;
        pl.allocs.push_back(q);
    }
    if (src && n) {
        if (hipMemcpyAsync(q, src, n * sizeof(T), hipMemcpyHostToDevice, stream) != hipSuccess) return -1;
        if (hipStreamSynchronize(stream) != hipSuccess) return -1;
    }
    *dst = static_cast<T *>(q);
    return 0;
}

#if defined(GBP_FUSED_DBG_SWITCHES) || defined(GBP_PHASE_TIMING)
GBP_INSTRUMENT_PLAN
#endif

inline size_t fused_shmem(int C)
{
    const int acc_doubles = C * 27;
    return sizeof(double) * ((size_t)((acc_doubles + 1) & ~1) + WAT_WAVES * WAVE_LDS_DOUBLES + 1);
}

// camera windows: a table of `set` cameras, the per-wave scratch, the rows' destinations (one int per table row) and the 16-bit map
// over an interval of `width` cameras
__host__ __device__ constexpr int win_rows_ints(int set) { return (set + 1) & ~1; }
inline size_t fused_shmem_windows(int set, int width)
{
    return fused_shmem(std::max(set, 1)) + (size_t)win_rows_ints(std::max(set, 1)) * sizeof(int) + (((size_t)width * 2 + 7) & ~(size_t)7);
}

// most cameras whose table + the per-wave scratch fit the LDS (the plan falls back to the general sweep above it)
inline int fused_max_cams()
{
    int c = 0;
    while (fused_shmem(c + 1) <= (size_t)LDS_BYTES) ++c;
    return c;
}

}  // namespace gbp
