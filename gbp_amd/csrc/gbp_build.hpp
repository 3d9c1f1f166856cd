// gbp_build.hpp -- device side of gbp_ba_create: the reference's create_ba_graph (gbp/gbp_ba.py:97-150) as kernels.
//
// The reference walks all observations once per camera (an O(C F) Python loop, gbp_ba.py:128-130) to put the factors in
// camera-major order, and generate_priors_var (gbp_ba.py:20-34) walks every variable's adjacent factors.  Here everything
// that is F-sized happens on the GPU:
//   1. k_check_ids            ids inside [0,C) x [0,L)?  camera ids already non-decreasing (every shipped file is)?
//   2. sort_pairs             stable by camera      -> reference factor r = file row ref_file[r]       (skipped when sorted)
//      k_lower_bounds         camera CSR offsets cptr
//   3. sort_pairs             stable by landmark    -> landmark-major list lm2ref (adj_factors order), offsets lptr
//   4. tile packing           next-fit over the landmark degrees (64 slots / 24 landmarks per tile, landmarks above 64 factors
//                             cut into chunk tiles).  Next-fit is a chain -- where a tile starts depends on where the previous one
//                             ended -- so it is done as LIST RANKING, in two levels: k_pack_next gives every landmark the
//                             landmark the next tile would start at IF a tile started here; k_pack_block_walk condenses that
//                             into a list over (block of 128 landmarks, entry offset) nodes; pointer doubling (k_pack_double)
//                             builds 2^k-hop jump tables over THAT list with the number of tiles they cross, k_pack_mark walks
//                             them from the top level down, k_pack_block_fill hands every landmark that really starts a tile
//                             its tile index; k_pack_emit writes the tile table and the slot ranges.  Two ints (tile count,
//                             over-sized landmarks) come back to the host
//   5. k_build_tiles          one wave per tile: every slot finds its factor, writes x0 | z | variance, meta, state (with the
//                             factor's rank among the same-camera factors of its tile) and both directions of the
//                             reference-id <-> slot map
//   6. k_init_vars            landmark records (mean, slot range) and camera records (mean)
// and the priors (gbp_ba.py:20-34) are a per-slot max of Lambda_f (k_factor_lambda_max), a segmented max per landmark over its
// contiguous slots, a per-camera max through the camera's adj_factors list, and k_prior_scalars -- no F-sized array crosses PCIe
// after the observations themselves went up.
#pragma once
#include "gbp_kernels.hpp"

namespace gbp {

size_t sort_pairs_tmp_bytes(size_t n, int bits);
int sort_pairs(void *tmp, size_t tmp_bytes, const int *keys_in, int *keys_out, const int *vals_in, int *vals_out, size_t n, int bits,
               hipStream_t stream);

// flags[0]: an id outside its range; flags[1]: camera ids not sorted (a stable sort is needed)
__global__ __launch_bounds__(BLOCK) void k_check_ids(const int *__restrict__ cam, const int *__restrict__ lmk, int F, int C, int L,
                                                     int *__restrict__ flags)
{
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= F) return;
    const int c = cam[i], l = lmk[i];
    if (c < 0 || c >= C || l < 0 || l >= L) flags[0] = 1;
    if (i > 0 && cam[i - 1] > c) flags[1] = 1;
}

__global__ __launch_bounds__(BLOCK) void k_iota(int *__restrict__ a, int n)
{
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i < n) a[i] = i;
}

__global__ __launch_bounds__(BLOCK) void k_gather_int(const int *__restrict__ src, const int *__restrict__ idx, int *__restrict__ dst, int n)
{
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i < n) dst[i] = src[idx[i]];
}

// ptr[v] = number of keys below v in the sorted list, v = 0 .. n_vals (CSR offsets of a sorted key list)
__global__ __launch_bounds__(BLOCK) void k_lower_bounds(const int *__restrict__ keys, int n, int *__restrict__ ptr, int n_vals)
{
    const int v = blockIdx.x * BLOCK + threadIdx.x;
    if (v > n_vals) return;
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (keys[mid] < v) lo = mid + 1; else hi = mid;
    }
    ptr[v] = lo;
}

// ---- tile packing as list ranking ------------------------------------------------------------------------------------------
// nxt[l] = first landmark of the tile after the one that would start at l; w[l] = tiles that start at l (1, or the chunk count of an
// over-sized landmark).  Entry L is the end of the chain.
__global__ __launch_bounds__(BLOCK) void k_pack_next(const int *__restrict__ lptr, int L, int *__restrict__ nxt, int *__restrict__ w)
{
    const int l = blockIdx.x * BLOCK + threadIdx.x;
    if (l > L) return;
    if (l == L) { nxt[l] = L; w[l] = 0; return; }
    const int deg = lptr[l + 1] - lptr[l];
    if (deg > WTILE) { nxt[l] = l + 1; w[l] = (deg + WTILE - 1) / WTILE; return; }
    int e = l, nf = 0, nl = 0;
    while (e < L && nl < TILE_LMKS) {
        const int d = lptr[e + 1] - lptr[e];
        if (d > WTILE || nf + d > WTILE) break;
        // The belief phase of a tile adds up seven landmarks per pass and writes the sums of pass b into scratch rows 7 b .. 7 b + 6
        // while the messages of LATER passes still wait in their rows (tile_landmark_beliefs): the first factor of a landmark of pass
        // b must sit in row 7 b or behind.  Landmarks without factors take no row, so a run of them could pull a later landmark's
        // rows forward: such a landmark starts a new tile instead.  (Never the case when every landmark has a factor: nf >= nl.)
        if (d > 0 && nf < 7 * (nl / 7)) break;
        nf += d; ++nl; ++e;
    }
    nxt[l] = e; w[l] = 1;
}

// The chain is ranked in two levels so that the scratch stays O(L) (log L jump tables over all landmarks were 1.9 GB of transient
// memory at 10M landmarks).  Landmarks are cut into blocks of PACK_BLOCK; a tile spans at most TILE_LMKS landmarks, so the chain can
// enter a block only at one of its first TILE_LMKS landmarks.  k_pack_block_walk follows the chain through ONE block from each of
// those possible entries (a short sequential walk) and records where it enters the next block and how many tiles start on the way:
// a list over (block, entry) nodes, ~L/5 of them, which pointer doubling (k_pack_double / k_pack_mark, as before) ranks from node
// (0, 0); k_pack_block_fill then walks every block once more from the entry the chain really takes and hands the landmarks that
// start a tile their tile index.
constexpr int PACK_BLOCK = 128;

// node = b * TILE_LMKS + e; the last node (n_nodes - 1) is the end of the chain
__global__ __launch_bounds__(BLOCK) void k_pack_block_walk(const int *__restrict__ nxt, const int *__restrict__ w, int L, int n_blocks,
                                                           int *__restrict__ bnext, int *__restrict__ bw)
{
    const int node = blockIdx.x * BLOCK + threadIdx.x, end_node = n_blocks * TILE_LMKS;
    if (node > end_node) return;
    if (node == end_node) { bnext[node] = node; bw[node] = 0; return; }
    const int b = node / TILE_LMKS, e = node - b * TILE_LMKS, stop = min((b + 1) * PACK_BLOCK, L);
    int l = b * PACK_BLOCK + e, count = 0;
    if (l >= stop && l < L) { bnext[node] = node; bw[node] = 0; return; }      // (an entry offset beyond a short last block: never reached)
    while (l < stop) { count += w[l]; l = nxt[l]; }
    bnext[node] = l >= L ? end_node : (b + 1) * TILE_LMKS + (l - (b + 1) * PACK_BLOCK);
    bw[node] = count;
}

// one thread per block: the entry the chain takes (the only marked node of the block) and the tile index it arrives with
__global__ __launch_bounds__(BLOCK) void k_pack_block_fill(const int *__restrict__ nxt, const int *__restrict__ w, int L, int n_blocks,
                                                           const int *__restrict__ bpos, int *__restrict__ pos)
{
    const int b = blockIdx.x * BLOCK + threadIdx.x;
    if (b >= n_blocks) return;
    int e = -1, t = 0;
    for (int k = 0; k < TILE_LMKS; ++k) {
        const int q = bpos[b * TILE_LMKS + k];
        if (q >= 0) { e = k; t = q; break; }
    }
    if (e < 0) return;                                   // (cannot happen for b * PACK_BLOCK < L: every block is crossed)
    const int stop = min((b + 1) * PACK_BLOCK, L);
    int l = b * PACK_BLOCK + e;
    while (l < stop) { pos[l] = t; t += w[l]; l = nxt[l]; }
    if (l >= L) pos[L] = t;                              // the chain ends in this block: the total number of tiles
}

// one doubling step: 2^(k+1) hops = 2^k hops twice
__global__ __launch_bounds__(BLOCK) void k_pack_double(const int *__restrict__ j0, const int *__restrict__ w0, int *__restrict__ j1,
                                                       int *__restrict__ w1, int n)
{
    const int l = blockIdx.x * BLOCK + threadIdx.x;
    if (l >= n) return;
    const int m = j0[l];
    j1[l] = j0[m];
    w1[l] = w0[l] + w0[m];
}

// pos[l] = index of the tile that starts at l for the landmarks on the chain from 0 (others stay -1); levels from the top down
__global__ __launch_bounds__(BLOCK) void k_pack_mark(const int *__restrict__ jk, const int *__restrict__ wk, int *__restrict__ pos, int n)
{
    const int l = blockIdx.x * BLOCK + threadIdx.x;
    if (l >= n) return;
    const int q = pos[l];
    if (q < 0) return;
    const int m = jk[l];
    if (m != l) pos[m] = q + wk[l];              // (several writers of one entry write the same value)
}

__global__ __launch_bounds__(BLOCK) void k_pack_emit(const int *__restrict__ lptr, int L, const int *__restrict__ nxt, const int *__restrict__ pos,
                                                     int4 *__restrict__ tiles, int *__restrict__ lrow0, int *__restrict__ lrow1,
                                                     int *__restrict__ big_list, int *__restrict__ big_count)
{
    const int l = blockIdx.x * BLOCK + threadIdx.x;
    if (l >= L) return;
    const int t = pos[l];
    if (t < 0) return;                                  // inside somebody else's tile
    const int deg = lptr[l + 1] - lptr[l];
    if (deg > WTILE) {
        for (int o = 0, c = 0; o < deg; o += WTILE, ++c) tiles[t + c] = make_int4(l, 1, min(WTILE, deg - o), 0);      // chunk tiles: a piece of l and nothing else
        lrow0[l] = t * WTILE; lrow1[l] = t * WTILE + deg;      // chunk tiles are full except the last: the slots are contiguous
        big_list[atomicAdd(big_count, 1)] = l;
        return;
    }
    const int e = nxt[l];
    tiles[t] = make_int4(l, e - l, lptr[e] - lptr[l], 0);
    for (int m = l; m < e; ++m) {
        lrow0[m] = t * WTILE + (lptr[m] - lptr[l]);
        lrow1[m] = t * WTILE + (lptr[m + 1] - lptr[l]);
    }
}

// ---- dense packing: tile t = factors [64 t, 64 t + 64) of the landmark-major list (landmarks may span tiles) --------------------------
// Whole-landmark tiles leave slots empty when the landmarks are large -- 40 factors each: one landmark and 24 idle lanes per tile --
// and every idle lane costs what a busy one does in the sweep's memory pipeline.  When every landmark has at least three factors a
// 64-factor window touches at most 22 landmarks (TILE_LMKS = 24 holds), so the windows themselves can be the tiles; the parts of the
// landmarks they cut are summed per tile and finished by k_lmk_finish_parts (gbp_kernels.hpp).  The host picks this packing when the
// whole-landmark one would fill less than 85 % of the slots (gbp_capi.hip: build_graph).
__global__ __launch_bounds__(BLOCK) void k_min_degree(const int *__restrict__ lptr, int L, int *__restrict__ out)
{
    const int l = blockIdx.x * BLOCK + threadIdx.x;
    int d = l < L ? lptr[l + 1] - lptr[l] : 0x7fffffff;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) d = min(d, __shfl_down(d, off, 64));
    if ((threadIdx.x & 63) == 0 && d != 0x7fffffff) atomicMin(out, d);
}

// the landmark that holds position `pos` of the landmark-major list (no landmark is empty here)
GBP_DEV int landmark_of_position(const int *__restrict__ lptr, int L, int pos)
{
    int lo = 0, hi = L;                                  // largest l with lptr[l] <= pos
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (lptr[mid] <= pos) lo = mid; else hi = mid;
    }
    return lo;
}

__global__ __launch_bounds__(BLOCK) void k_dense_tiles(const int *__restrict__ lptr, int L, int F, int T, int4 *__restrict__ tiles)
{
    const int t = blockIdx.x * BLOCK + threadIdx.x;
    if (t >= T) return;
    const int p0 = t * WTILE, p1 = min(F, p0 + WTILE) - 1;
    const int l0 = landmark_of_position(lptr, L, p0), l1 = landmark_of_position(lptr, L, p1);
    tiles[t] = make_int4(l0, l1 - l0 + 1, p1 - p0 + 1, 0);
}

__global__ __launch_bounds__(BLOCK) void k_dense_rows(const int *__restrict__ lptr, int L, int *__restrict__ lrow0, int *__restrict__ lrow1)
{
    const int l = blockIdx.x * BLOCK + threadIdx.x;
    if (l < L) { lrow0[l] = lptr[l]; lrow1[l] = lptr[l + 1]; }
}

struct BuildArgs {
    int4 *tiles;                          // in: {first landmark, landmarks it holds (parts of) , slots used, -}; out: .w = max rank
    const int *lrow0, *lptr;              // per landmark: first slot; offsets into lm2ref
    const int *lm2ref, *ref_cam, *ref_file;   // ref_file may be NULL (reference order == file order)
    const double *cam_means, *lmk_means, *meas;
    int *ref2slot, *cpos;
};

// One wave per tile (create_ba_graph's per-factor part, gbp_ba.py:131-141: linpoint = concat(cam.mu, lmk.mu), measurement,
// adaptive variance = sigma^2, iters_since_relin = 1 gbp.py:249).
__global__ __launch_bounds__(BLOCK) void k_build_tiles(Params p, BuildArgs a)
{
    const int lane = threadIdx.x & 63, t = blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6);
    if (t >= p.T) return;
    const int4 td = a.tiles[t];
    const int slot = t * WTILE + lane;
    const bool active = lane < td.z;
    int cam = -1 - lane, l = td.x, r = 0;       // inactive lanes: distinct negative "cameras" that match nobody
    if (active) {
        int k = 0;
        for (int i = 1; i < td.y; ++i) k += (a.lrow0[td.x + i] <= slot) ? 1 : 0;         // rows ascend with the landmark
        l = td.x + k;
        r = a.lm2ref[a.lptr[l] + (slot - a.lrow0[l])];
        cam = a.ref_cam[r];
    }
    // rank among the same-camera factors of the tile, in slot order (the order of the LDS accumulation of the fused sweep)
    int rank = 0;
#pragma unroll 8
    for (int i = 0; i < WTILE; ++i) {
        const int ci = __shfl(cam, i, 64);
        rank += (i < lane && ci == cam) ? 1 : 0;
    }
    int mr = active ? rank : 0;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mr = max(mr, __shfl_down(mr, off, 64));
    if (lane == 0) a.tiles[t].w = mr;
    slot_words(p, slot)[0] = active ? (((unsigned)cam << META_LMK_BITS) | (unsigned)(l - td.x)) : 0u;
    set_slot_state(p, slot, state_pack(1, p.clk, active ? rank : 0, false, false));      // iters_since_relin = 1, gbp.py:249
    if (p.avar) p.avar[slot] = p.sigma2;                                                 // gbp.py:242
    a.cpos[slot] = active ? r : 0;
    if (!active) return;
    a.ref2slot[r] = slot;
    const int fi = a.ref_file ? a.ref_file[r] : r;
#pragma unroll
    for (int k = 0; k < 6; ++k) p.lin[lin_at(slot, ROW_X0 + k)] = a.cam_means[(size_t)cam * 6 + k];
#pragma unroll
    for (int k = 0; k < 3; ++k) p.lin[lin_at(slot, ROW_X0 + 6 + k)] = a.lmk_means[(size_t)l * 3 + k];
    p.lin[lin_at(slot, ROW_Z)] = a.meas[(size_t)fi * 2];
    p.lin[lin_at(slot, ROW_Z + 1)] = a.meas[(size_t)fi * 2 + 1];
}

// out[b] = {lowest, highest} camera among the factors of the tiles that workgroup b of the fused sweep walks (tiles [b T / n, (b + 1) T / n),
// k_sweep_wat): the interval its camera window lies in (k_wg_cam_sets, fused_plan).  One wave per workgroup, the slot -> factor decode of k_build_tiles.
__global__ __launch_bounds__(BLOCK) void k_wg_cam_range(const int4 *__restrict__ tiles, const int *__restrict__ lrow0, const int *__restrict__ lptr,
                                                        const int *__restrict__ lm2ref, const int *__restrict__ ref_cam, int T, int n_wg,
                                                        int2 *__restrict__ out)
{
    const int lane = threadIdx.x & 63, b = blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6);
    if (b >= n_wg) return;
    const int t0 = (int)((long long)b * T / n_wg), t1 = (int)((long long)(b + 1) * T / n_wg);
    int lo = 0x7fffffff, hi = -1;
    for (int t = t0; t < t1; ++t) {
        const int4 td = tiles[t];
        const int slot = t * WTILE + lane;
        if (lane < td.z) {
            int k = 0;
            for (int i = 1; i < td.y; ++i) k += (lrow0[td.x + i] <= slot) ? 1 : 0;
            const int l = td.x + k, cam = ref_cam[lm2ref[lptr[l] + (slot - lrow0[l])]];
            lo = min(lo, cam); hi = max(hi, cam);
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { lo = min(lo, __shfl_down(lo, off, 64)); hi = max(hi, __shfl_down(hi, off, 64)); }
    if (lane == 0) out[b] = make_int2(lo, hi);
}

// The camera SET of workgroup b: the distinct cameras among the factors of its tiles, ascending, into lists[b * cap ..) (the first `cap`
// of them) and their number into counts[b].  One 256-thread workgroup per workgroup of the sweep: a bitmap over its interval
// rng[b] = [lo, hi] in LDS (dynamic: (widest interval + 31) / 32 + 257 words), then a compaction in index order.
constexpr int SETS_THREADS = 256;
__global__ __launch_bounds__(SETS_THREADS) void k_wg_cam_sets(const int4 *__restrict__ tiles, const int *__restrict__ lrow0, const int *__restrict__ lptr,
                                                              const int *__restrict__ lm2ref, const int *__restrict__ ref_cam, int T, int n_wg,
                                                              const int2 *__restrict__ rng, int cap, int *__restrict__ lists, int *__restrict__ counts)
{
    extern __shared__ unsigned bm[];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const int lo = rng[b].x, width = rng[b].y >= lo ? rng[b].y - lo + 1 : 0, words = (width + 31) >> 5;
    unsigned *scan = bm + words;
    for (int i = tid; i < words; i += SETS_THREADS) bm[i] = 0u;
    __syncthreads();
    const int t0 = (int)((long long)b * T / n_wg), t1 = (int)((long long)(b + 1) * T / n_wg);
    for (int t = t0 + (tid >> 6); t < t1; t += SETS_THREADS / 64) {
        const int4 td = tiles[t];
        const int slot = t * WTILE + lane;
        if (lane < td.z) {
            int k = 0;
            for (int i = 1; i < td.y; ++i) k += (lrow0[td.x + i] <= slot) ? 1 : 0;
            const int l = td.x + k, cam = ref_cam[lm2ref[lptr[l] + (slot - lrow0[l])]] - lo;
            atomicOr(&bm[cam >> 5], 1u << (cam & 31));
        }
    }
    __syncthreads();
    const int per = (words + SETS_THREADS - 1) / SETS_THREADS, w0 = min(tid * per, words), w1 = min(w0 + per, words);
    int n = 0;
    for (int w = w0; w < w1; ++w) n += __popc(bm[w]);
    scan[tid] = (unsigned)n;
    __syncthreads();
    if (tid == 0) {
        unsigned run = 0;
        for (int i = 0; i < SETS_THREADS; ++i) { const unsigned v = scan[i]; scan[i] = run; run += v; }
        scan[SETS_THREADS] = run;
    }
    __syncthreads();
    int pos = (int)scan[tid];
    for (int w = w0; w < w1; ++w)
        for (unsigned bits = bm[w]; bits; bits &= bits - 1u, ++pos)
            if (pos < cap) lists[(size_t)b * cap + pos] = lo + (w << 5) + (__ffs(bits) - 1);
    if (tid == 0) counts[b] = (int)scan[SETS_THREADS];
}

// node.mu = initial estimate (gbp_ba.py:116,123); landmark records also carry their slot range
__global__ __launch_bounds__(BLOCK) void k_init_vars(Params p, const double *__restrict__ cam_means, const double *__restrict__ lmk_means,
                                                     const int *__restrict__ lrow0, const int *__restrict__ lrow1)
{
    const int v = blockIdx.x * BLOCK + threadIdx.x;
    if (v < p.C) {
#pragma unroll
        for (int k = 0; k < 6; ++k) p.cbel[(size_t)v * CAMREC + CAM_MU + k] = cam_means[(size_t)v * 6 + k];
    } else if (v < p.C + p.L) {
        const int l = v - p.C;
        double *lr = p.lrec + (size_t)l * LREC;
#pragma unroll
        for (int k = 0; k < 3; ++k) lr[LR_MU + k] = lmk_means[(size_t)l * 3 + k];
        *reinterpret_cast<int2 *>(lr + LR_ROWS) = make_int2(lrow0[l], lrow1[l]);
    }
}

// Debug check of the layout the build produced (GBP_DEBUG_LAYOUT=1 and the tests): every used slot decodes to the camera and
// landmark of the reference factor it holds, and the two maps invert each other.  err[0] = number of bad slots.
__global__ __launch_bounds__(BLOCK) void k_check_layout(Params p, const int *__restrict__ ref_cam, const int *__restrict__ ref_lmk,
                                                        int *__restrict__ err)
{
    const int slot = blockIdx.x * BLOCK + threadIdx.x;
    int cam, lmk;
    if (slot >= p.T * WTILE || !slot_info(p, slot, cam, lmk)) return;
    const int r = p.cpos[slot];
    if (r < 0 || r >= p.F || p.cadj[r] != slot || ref_cam[r] != cam || ref_lmk[r] != lmk || lmk < 0 || lmk >= p.L) atomicAdd(err, 1);
}

// ---- priors: generate_priors_var gbp_ba.py:20-34 ------------------------------------------------------------------------
// max over a landmark's adjacent factors of max(Lambda_f): its slots are contiguous
__global__ __launch_bounds__(BLOCK) void k_lmk_max(Params p, const double *__restrict__ fm, double *__restrict__ lmk_max)
{
    const int l = blockIdx.x * BLOCK + threadIdx.x;
    if (l >= p.L) return;
    const int2 rows = *reinterpret_cast<const int2 *>(p.lrec + (size_t)l * LREC + LR_ROWS);
    double m = 0.0;                                     // max_factor_lam = 0.  gbp_ba.py:27
    for (int s = rows.x; s < rows.y; ++s) m = fmax(m, fm[s]);
    lmk_max[l] = m;
}

// one workgroup per camera: max over its adj_factors list (cadj = reference id -> slot)
__global__ __launch_bounds__(BLOCK) void k_cam_max(Params p, const double *__restrict__ fm, double *__restrict__ cam_max)
{
    __shared__ double red[BLOCK / 64];
    const int c = blockIdx.x;
    double m = 0.0;
    for (int e = p.cptr[c] + threadIdx.x; e < p.cptr[c + 1]; e += BLOCK) m = fmax(m, fm[p.cadj[e]]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmax(m, __shfl_down(m, off, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int w = 1; w < BLOCK / 64; ++w) m = fmax(m, red[w]);
        cam_max[c] = m;
    }
}

// prior Lambda = (lambda / w2) I, eta = Lambda mu   (gbp_ba.py:32-34; w2 = weaker_factor^2, 1 when the caller gives Lambda)
__global__ __launch_bounds__(BLOCK) void k_prior_scalars(Params p, const double *__restrict__ cam_lambda, const double *__restrict__ lmk_lambda,
                                                         double w2)
{
    const int v = blockIdx.x * BLOCK + threadIdx.x;
    if (v < p.C) {
        const double lam = cam_lambda[v] / w2;
        double *pr = p.cprior + (size_t)v * 27;
#pragma unroll
        for (int k = 0; k < 27; ++k) pr[k] = 0.0;
#pragma unroll
        for (int k = 0; k < 6; ++k) { pr[k] = lam * p.cbel[(size_t)v * CAMREC + CAM_MU + k]; pr[6 + Sym<6>::at(k, k)] = lam; }
    } else if (v < p.C + p.L) {
        const int l = v - p.C;
        const double lam = lmk_lambda[l] / w2;
        double *lr = p.lrec + (size_t)l * LREC;
#pragma unroll
        for (int k = 0; k < 9; ++k) lr[LR_PRIOR + k] = 0.0;
#pragma unroll
        for (int k = 0; k < 3; ++k) { lr[LR_PRIOR + k] = lam * lr[LR_MU + k]; lr[LR_PRIOR + 3 + Sym<3>::at(k, k)] = lam; }
    }
}

// packed landmark priors (eta 3 | Lambda 6) into the landmark records
__global__ __launch_bounds__(BLOCK) void k_scatter_lmk_priors(Params p, const double *__restrict__ pri)
{
    const size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i < (size_t)p.L * 9) p.lrec[(i / 9) * LREC + LR_PRIOR + (i % 9)] = pri[i];
}

// order-independent 64-bit digest of the factor layout (reference id -> slot, camera, landmark): pins a state blob to its graph
__global__ __launch_bounds__(BLOCK) void k_graph_hash(const int *__restrict__ ref2slot, const int *__restrict__ ref_cam,
                                                      const int *__restrict__ ref_lmk, int F, unsigned long long *__restrict__ out)
{
    const int r = blockIdx.x * BLOCK + threadIdx.x;
    unsigned long long v = 0ull;
    if (r < F) {
        v = ((unsigned long long)(unsigned)r << 32) ^ (unsigned)ref2slot[r];
        v = (v ^ (v >> 33)) * 0xff51afd7ed558ccdull;
        v ^= ((unsigned long long)(unsigned)ref_cam[r] << 32) | (unsigned)ref_lmk[r];
        v = (v ^ (v >> 33)) * 0xc4ceb9fe1a85ec53ull;
        v ^= v >> 33;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if ((threadIdx.x & 63) == 0 && v) atomicAdd(out, v);
}

// ---- priors: per-factor maximum of Lambda_f (gbp_ba.py:27-31), weaken_priors (gbp_ba.py:36-42) ----
// np.max(factor.factor.lam) per factor at its current linearisation point (gbp_ba.py:31); 0 for empty slots
__global__ __launch_bounds__(BLOCK) void k_factor_lambda_max(Params p, double *__restrict__ fmax_out)
{
    const int slot = blockIdx.x * BLOCK + threadIdx.x;
    if (slot >= p.T * WTILE) return;
    int cam, lmk;
    if (!slot_info(p, slot, cam, lmk)) { fmax_out[slot] = 0.0; return; }
    double x0[9], Jc[2][6], Jl[2][3], h[2];
#pragma unroll
    for (int k = 0; k < 9; ++k) x0[k] = p.lin[lin_at(slot, ROW_X0 + k)];
    linearise(x0, p.K, Jc, Jl, h);
    const double av = slot_avar(p, slot);
    fmax_out[slot] = factor_lambda_max(Jc, Jl, 1.0 / av);
}


// BAFactorGraph.weaken_priors (gbp_ba.py:36-42): prior eta and Lambda of every variable times `factor`
__global__ __launch_bounds__(BLOCK) void k_weaken_priors(Params p, double factor)
{
    const size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    const size_t nc = (size_t)p.C * 27, nl = (size_t)p.L * 9;
    if (i < nc) p.cprior[i] *= factor;
    else if (i < nc + nl) {
        const size_t j = i - nc;
        p.lrec[(j / 9) * LREC + LR_PRIOR + (j % 9)] *= factor;
    }
}


}  // namespace gbp
