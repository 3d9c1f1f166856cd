// experimental/gbp_instrument.hpp -- instrumentation of the fused sweep's persistent loop.  NOT part of the product build: gbp_fused.hpp
// includes this file only under -DGBP_FUSED_DBG_SWITCHES or -DGBP_PHASE_TIMING, which tools/profile_round.sh, tools/pmc_dbg.sh and
// tools/phase_profile.py pass when they compile their scratch copies of the library (tools/libgbp_dbg.so, tools/libgbp_phase.so).
//
//  * GBP_FUSED_DBG_SWITCHES: the run-time ablation switches of the loop, read from the environment variable GBP_FUSED_DBG at plan time
//    (timing only -- most of them give WRONG results): 1 no ticket wait, 2 no accumulation, 4 no landmark phase, 8 camera records
//    gathered from 8 cameras only (cheap for the address coalescer), 32 only the first round of the accumulation.  As run-time
//    tests they put half a dozen scalar branches into every tile -- 2 us per sweep at the headline size -- which is why the product
//    kernel does not carry them (EXPERIMENTS.md, round 4).
//  * GBP_PHASE_TIMING: every wave adds up the s_memtime ticks it spends between consecutive marks GBP_PH(i) of the loop into its own row
//    of FusedArgs::phase ([workgroup][wave][NPHASE]); gbp_ba_phase_profile reads them back.
//
// (Round 4's compile-time experiments GBP_EXPERIMENT_HALF / GBP_EXPERIMENT_NO_RELIN_PATH -- the two halves of a factor and the loop's
//  skeleton, profiles/r04_factor_halves.json -- were removed from the sources in round 5; they are in the history at 5125c75.)
#pragma once

#define GBP_INSTRUMENT_ARGS int dbg; unsigned long long *phase;

#ifdef GBP_FUSED_DBG_SWITCHES
#define GBP_DBG(a, bit) ((a).dbg & (bit))
#else
#define GBP_DBG(a, bit) 0
#endif

#ifdef GBP_PHASE_TIMING
#define GBP_PH_DECL unsigned long long ph_last = __builtin_amdgcn_s_memtime(), ph_acc[NPHASE] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}
#define GBP_PH(i) do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); const unsigned long long ph_now = __builtin_amdgcn_s_memtime(); \
                       ph_acc[i] += ph_now - ph_last; ph_last = ph_now; } while (0)
#define GBP_PH_NOWAIT(i) do { const unsigned long long ph_now = __builtin_amdgcn_s_memtime(); ph_acc[i] += ph_now - ph_last; ph_last = ph_now; } while (0)
#define GBP_PH_FLUSH(ptr, row) do { if ((ptr) && (threadIdx.x & 63) == 0) for (int i_ = 0; i_ < NPHASE; ++i_) (ptr)[(size_t)(row) * NPHASE + i_] = ph_acc[i_]; } while (0)
#else
#define GBP_PH_DECL
#define GBP_PH(i)
#define GBP_PH_NOWAIT(i)
#define GBP_PH_FLUSH(ptr, row)
#endif

// called by fused_plan (gbp_fused.hpp) after the plan's buffers exist
#define GBP_INSTRUMENT_PLAN                                                                                                   \
    inline int instrument_plan(FusedPlan &pl, hipStream_t stream)                                                             \
    {                                                                                                                         \
        const char *env_dbg = getenv("GBP_FUSED_DBG");                                                                        \
        pl.args.dbg = env_dbg ? atoi(env_dbg) : 0;                                                                            \
        pl.args.phase = nullptr;                                                                                              \
        GBP_INSTRUMENT_PHASE_BUFFER                                                                                           \
        return 0;                                                                                                             \
    }
#ifdef GBP_PHASE_TIMING
#define GBP_INSTRUMENT_PHASE_BUFFER                                                                                           \
        if (fused_upload<unsigned long long>(pl, &pl.args.phase, nullptr, (size_t)pl.n_blocks * WAT_WAVES * NPHASE, stream)) return -1;
#else
#define GBP_INSTRUMENT_PHASE_BUFFER (void)stream;
#endif
