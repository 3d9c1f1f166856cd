/*
 * gbp_ba.h -- C ABI of libgbp_hip.so: the GBP bundle-adjustment sweep on MI355X (gfx950).
 *
 * The reference (joeaortiz/gbp) is pure Python and has no FFI/plugin interface; its "operator
 * boundary" for this path is the Python class API that ba.py drives.  Each
 * entry point below replaces the reference method cited next to it; gbp_amd/compat/ re-exposes
 * them under the reference's class and method names through ctypes.
 *
 * Conventions
 *  - every call returns 0 on success or a negative GBP_E* code; gbp_last_error() gives the
 *    thread-local message.  No exception or abort() crosses the boundary.
 *  - all host pointers are caller-owned, contiguous, C-order float64 / int32; the library copies
 *    in and out and never retains them.  Pointers named *_dev are DEVICE pointers (sharded mode).
 *  - factor-indexed arrays at the boundary are in the REFERENCE's factor order: camera-major,
 *    file order inside a camera (gbp/gbp_ba.py:128-130).  Variables: cameras 0..C-1 then
 *    landmarks 0..L-1 (gbp/gbp_ba.py:114-125).  The internal layout (landmark-major SoA) is
 *    never visible.
 *  - dense matrices are row-major; beliefs/messages/priors are information form (eta, Lambda).
 *  - one host thread per handle; calls that return data synchronise the handle's stream.
 */
#ifndef GBP_BA_H
#define GBP_BA_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GBP_ABI_VERSION 3      /* 3: gbp_ba_info's fused_path is 0 / 1, GBP_FLAG_FORCE_FUSED, gbp_ba_plan_info, gbp_ba_peer_selftest; gbp_ba_grouped_max_cams is gone.  State blobs are version 7. */

enum {
    GBP_OK = 0,
    GBP_EINVAL = -1,   /* bad argument / index out of range */
    GBP_ENOMEM = -2,   /* host or device allocation failed */
    GBP_EHIP = -3,     /* a HIP runtime call failed (message has the HIP error string) */
    GBP_ENODEV = -4,   /* no usable gfx950 device */
    GBP_ESTATE = -5    /* call not valid in the handle's current state */
};

enum { GBP_LOSS_NONE = 0, GBP_LOSS_HUBER = 1, GBP_LOSS_CONSTANT = 2 };   /* Factor.loss, gbp/gbp.py:243 */

typedef struct gbp_ba gbp_ba_t;

/* What create_ba_graph(bal_file, configs) consumes (gbp/gbp_ba.py:97-107,134-135), as arrays.
 * Observations are given in FILE order; the library applies the reference's camera-major order. */
typedef struct gbp_ba_desc {
    int32_t n_cams;               /* C */
    int32_t n_lmks;               /* L */
    int32_t n_factors;            /* F */
    int32_t device;               /* HIP device ordinal */
    double K[4];                  /* fx fy cx cy            utils/read_balfile.py:14-16 */
    const double *cam_means;      /* C*6  t(3), axis-angle(3)   gbp_ba.py:116 */
    const double *lmk_means;      /* L*3                        gbp_ba.py:123 */
    const double *meas;           /* F*2 pixels                 gbp_ba.py:134 */
    const int32_t *cam_idx;       /* F                          read_balfile.py:21 */
    const int32_t *lmk_idx;       /* F                          read_balfile.py:22 */
    double gauss_noise_std;       /* configs['gauss_noise_std'] gbp_ba.py:135 */
    int32_t loss;                 /* GBP_LOSS_*                 configs['loss'] */
    int32_t num_undamped_iters;   /* gbp/gbp.py:33 */
    int32_t min_linear_iters;     /* gbp/gbp.py:34 */
    int32_t flags;                /* GBP_FLAG_* */
    double nstds;                 /* configs['Nstds'] -> mahalanobis_threshold gbp.py:244 */
    double beta;                  /* gbp/gbp.py:32 */
    double eta_damping;           /* gbp/gbp.py:28 */
} gbp_ba_desc_t;

#define GBP_FLAG_NO_FUSED 1       /* force the general 3-kernel sweep (testing / ablation) */
#define GBP_FLAG_FORCE_FUSED 4    /* never leave the fused sweep on sparseness grounds (by default a graph with few factors per camera and
                                     workgroup runs the staged general sweep, which is the faster one there; tests name the sweep they mean) */
#define GBP_FLAG_DEVICE_INPUT 2   /* cam_means / lmk_means / meas / cam_idx / lmk_idx are DEVICE pointers (on desc.device): nothing is
                                     uploaded; the graph is ordered, tiled and linearised where the observations already are */

int gbp_abi_version(void);
const char *gbp_last_error(void);

/* life cycle: create_ba_graph incl. the initial compute_factor of every factor (gbp_ba.py:97-150) */
int gbp_ba_create(gbp_ba_t **out, const gbp_ba_desc_t *desc);
void gbp_ba_destroy(gbp_ba_t *h);
int gbp_ba_set_stream(gbp_ba_t *h, void *hip_stream);     /* NULL = the handle's own stream */
int gbp_ba_sync(gbp_ba_t *h);

/* priors */
int gbp_ba_generate_priors(gbp_ba_t *h, double weaker_factor);          /* BAFactorGraph.generate_priors_var gbp_ba.py:20-34 */
int gbp_ba_factor_lambda_max(gbp_ba_t *h, double *cam_max, double *lmk_max);  /* the max_f max(Lambda_f) half of it (sharded set-up) */
int gbp_ba_set_prior_scalars(gbp_ba_t *h, const double *cam_lambda, const double *lmk_lambda); /* Lambda=l*I, eta=l*mu  gbp_ba.py:32-34 */
int gbp_ba_set_priors(gbp_ba_t *h, const double *cam_eta, const double *cam_lam,
                      const double *lmk_eta, const double *lmk_lam);    /* information form; serves set_priors_var gbp_ba.py:44-52 */
int gbp_ba_weaken_priors(gbp_ba_t *h, double factor);                   /* BAFactorGraph.weaken_priors gbp_ba.py:36-42 */

/* the sweep */
int gbp_ba_update_beliefs(gbp_ba_t *h);                                 /* FactorGraph.update_all_beliefs gbp.py:56-58 */
int gbp_ba_iterate(gbp_ba_t *h, int32_t n_iters, int32_t robustify, int32_t local_relin);
                                                                        /* n x FactorGraph.synchronous_iteration gbp.py:86-92 */

/* the same sweep stage by stage (the reference's FactorGraph exposes all four; synchronous_iteration = robustify ->
 * relinearise -> compute_messages -> update_beliefs, gbp.py:86-92, and that fixed sequence is what gbp_ba_iterate fuses).  A
 * relinearisation decided by gbp_ba_relinearise / gbp_ba_compute_factors takes effect on the messages when they are next
 * computed; the views (gbp_ba_get_factors) show the new linearisation point at once.  Every call order the reference allows is
 * allowed here: when a factor turns out to be DAMPED in the message computation that moves its linearisation point
 * (gbp_ba_compute_factors with the damping on; gbp_ba_relinearise followed by a local_relin = 0 computation) the library switches
 * the dense message remainder on for the handle (9 doubles per factor, allocated then) and runs its general sweep until every
 * remainder has decayed to zero again, then returns to the fused sweep -- results as the reference's either way. */
int gbp_ba_robustify(gbp_ba_t *h);                                      /* FactorGraph.robustify_all_factors gbp.py:82-84 */
int gbp_ba_relinearise(gbp_ba_t *h);                                    /* FactorGraph.relinearise_factors gbp.py:64-80 */
int gbp_ba_compute_messages(gbp_ba_t *h, int32_t local_relin);          /* FactorGraph.compute_all_messages gbp.py:46-54 (no belief changes) */
int gbp_ba_compute_factors(gbp_ba_t *h);                                /* FactorGraph.compute_all_factors gbp.py:60-62 */

/* diagnostics ba.py prints every iteration */
int gbp_ba_are(gbp_ba_t *h, double *out);                               /* BAFactorGraph.are gbp_ba.py:61-69 */
int gbp_ba_energy(gbp_ba_t *h, double *out);                            /* FactorGraph.energy gbp.py:36-44 */
int gbp_ba_residual_sums(gbp_ba_t *h, double out[2]);                   /* {sum ||r||, sum 0.5||r||^2/var}: un-normalised, for shards */

/* state views (reference order, dense); any pointer may be NULL to skip that array */
int gbp_ba_get_beliefs(gbp_ba_t *h, double *cam_eta, double *cam_lam, double *lmk_eta, double *lmk_lam);  /* VariableNode.belief gbp.py:168 */
int gbp_ba_get_means(gbp_ba_t *h, double *cam_mu, double *lmk_mu);                                        /* VariableNode.mu gbp.py:165,193 */
int gbp_ba_get_covariances(gbp_ba_t *h, double *cam_sigma, double *lmk_sigma);                            /* VariableNode.Sigma gbp.py:166,192 */
int gbp_ba_get_priors(gbp_ba_t *h, double *cam_eta, double *cam_lam, double *lmk_eta, double *lmk_lam);   /* VariableNode.prior gbp.py:170 */
int gbp_ba_get_messages(gbp_ba_t *h, int32_t f0, int32_t n, double *cam_eta, double *cam_lam,
                        double *lmk_eta, double *lmk_lam);                                                /* Factor.messages gbp.py:222 */
int gbp_ba_get_factors(gbp_ba_t *h, int32_t f0, int32_t n, double *eta, double *lam, double *linpoint,
                       int32_t *cam, int32_t *lmk, double *meas);                                         /* Factor.factor/.linpoint/.adj_vIDs gbp.py:230-233 */
int gbp_ba_get_relin_state(gbp_ba_t *h, int32_t *iters_since_relin, double *eta_damping,
                           double *adaptive_var, uint8_t *robust_flag);                                   /* gbp.py:242-249 */
int gbp_ba_get_relin_state_range(gbp_ba_t *h, int32_t f0, int32_t n, int32_t *iters_since_relin, double *eta_damping,
                                 double *adaptive_var, uint8_t *robust_flag);                             /* the same for factors [f0, f0+n) only */
int gbp_ba_set_iters_since_relin(gbp_ba_t *h, const int32_t *iters);                                      /* ba.py:91-93 (per factor) */
int gbp_ba_fill_iters_since_relin(gbp_ba_t *h, int32_t value);                                            /* ba.py:91-93 (all factors) */

/* "Num factors relinearising" of ba.py:96-99 without reading F state words back: the number of factors whose
 * iters_since_relin is 0 now (count_relinearising), and the numbers of factors that relinearised in each of the last n
 * sweeps, oldest first (get_relin_counts; n <= 512 and <= sweeps run; the sweep kernels count on the device).
 * iters_since_relin saturates at 524 287 (the reference's Python int is unbounded; only >= min_linear_iters and
 * == num_undamped_iters are ever tested, gbp.py:50,72). */
int gbp_ba_count_relinearising(gbp_ba_t *h, int64_t *count);
int gbp_ba_get_relin_counts(gbp_ba_t *h, int32_t *counts, int32_t n);

/* meas_fn / jac_fn of the reprojection factor at n free-standing 9-vectors x = (t, w, y), evaluated by the device code
 * every sweep kernel inlines (gbp/factors/reprojection.py:12-44, utils/derivatives.py:36-50, utils/lie_algebra.py:32-42):
 * h2[n*2] from the linearisation routine, J18[n*2*9] row-major, hproj2[n*2] from the projection-only routine used by
 * the robust loss and the residual diagnostics.  Any output may be NULL.  No handle: K4 = fx fy cx cy. */
int gbp_ba_eval_fn(const double *K4, int32_t n, const double *x9, double *h2, double *J18, double *hproj2, int32_t device);

/* landmark-sharded multi-GPU sweep (no reference counterpart).  Each rank owns a
 * landmark range and its factors, cameras are replicated.  begin = (if with_messages) robustify /
 * relinearise / messages, then landmark beliefs + this rank's camera partial sums (C*27 packed doubles: eta 6, upper Lambda 21) into
 * partial_dev; the caller all-gathers; end = fixed rank-order sum + prior + camera beliefs. */
int gbp_ba_shard_begin(gbp_ba_t *h, int32_t with_messages, int32_t robustify, int32_t local_relin, double *partial_dev);
int gbp_ba_shard_end(gbp_ba_t *h, const double *gathered_dev, int32_t n_ranks);
#define GBP_CAM_PARTIAL_DOUBLES 27

/* The same sweep with the loop and the camera exchange INSIDE the library (no host round trip per sweep): per iteration
 * local kernels -> camera partial sums -> exchange -> rank-ordered sum + prior + 6x6 solve, all on the handle's stream.
 * The exchange is an all-gather of C*27 doubles per rank:
 *   - gbp_ba_comm_init_rccl: an RCCL communicator owned by the handle (librccl is dlopen()ed on first use; rccl_path NULL =
 *     the copy already mapped in the process, else the system one).  gbp_ba_comm_unique_id on rank 0 makes the 128-byte
 *     id every rank passes in (carry it over any side channel: MPI, torch.distributed, a file);
 *   - gbp_ba_set_exchange: a caller-supplied function (MPI, peer-to-peer copies, a test double).  It must leave
 *     recv_dev[r*count .. (r+1)*count) = rank r's send_dev for every r, ordered after the work already on hip_stream and
 *     before anything enqueued on it afterwards, and return 0.
 * With n_ranks == 1 nothing is exchanged (the path equals gbp_ba_iterate) unless flags has GBP_XCH_ALWAYS. */
typedef int (*gbp_exchange_fn)(void *ctx, const double *send_dev, double *recv_dev, uint64_t count, void *hip_stream);
#define GBP_COMM_ID_BYTES 128
#define GBP_XCH_ALWAYS 1
int gbp_ba_comm_unique_id(void *id128, const char *rccl_path);
int gbp_ba_comm_init_rccl(gbp_ba_t *h, const void *id128, int32_t rank, int32_t n_ranks, int32_t flags, const char *rccl_path);
int gbp_ba_set_exchange(gbp_ba_t *h, gbp_exchange_fn fn, void *ctx, int32_t rank, int32_t n_ranks, int32_t flags);
int gbp_ba_comm_destroy(gbp_ba_t *h);
/*   - gbp_ba_peer_export / gbp_ba_peer_connect: NO collective call.  Every rank owns a mailbox in its own device memory (two
 *     sweep-parity halves of n_ranks x C rows: a camera's 27 partial sums + pad); the kernel that finishes a rank's partial sums
 *     stores each row straight into the mailbox of every rank (peer stores over xGMI on a multi-GPU node).  The data is its own
 *     arrival flag: an empty slot holds a quiet NaN with a payload no arithmetic produces, whoever finishes camera c polls the
 *     n_ranks rows c of its own mailbox until no slot is empty (the poll is the data load) and empties them again for the exchange
 *     after next: one trip per exchange, no acknowledgement wait, no tag, no ordering asked of the link.  After the fused sweep all of that
 *     is ONE launch (reduce -> push -> wait -> rank-ordered sum + prior + 6x6 solve).  export allocates the mailbox for n_ranks and
 *     writes a 64-byte handle (a hipIpcMemHandle_t for other PROCESSES; with GBP_PEER_SAME_PROCESS the raw device address, for
 *     ranks that are threads of one process); carry the handles of all ranks, in rank order, to every rank over any side channel
 *     and pass them to connect.  All ranks must have connected before the first sharded call (barrier on the side channel).
 *     GBP_PEER_RENDEZVOUS keeps the function set with gbp_ba_set_exchange as a hook called with (NULL, NULL, 0, stream) between
 *     the stores and the finish, which then stay two launches (logical ranks on ONE device must not spin on each other).  A
 *     finish wave gives up after GBP_PEER_TIMEOUT_MS (default 20000); from then on everything that hands results to the caller
 *     (get_beliefs / get_means / get_covariances, are / energy, means_snapshot, save_state) returns GBP_ESTATE until gbp_ba_sync has
 *     reported it (and cleared the mark). */
#define GBP_PEER_HANDLE_BYTES 64
#define GBP_PEER_SAME_PROCESS 1
#define GBP_PEER_RENDEZVOUS 2
#define GBP_PEER_MAX_RANKS 16
int gbp_ba_peer_export(gbp_ba_t *h, int32_t n_ranks, void *handle64, int32_t flags);
int gbp_ba_peer_connect(gbp_ba_t *h, int32_t rank, int32_t n_ranks, const void *handles, int32_t flags);
/* after connect (and a side-channel barrier), before the first sharded call, on every rank: one probe row travels to every
 * rank's mailbox exactly as a sweep's rows do and the rows of all ranks are checked on arrival.  GBP_ESTATE names the pair that failed
 * (row late, or wrong contents); the caller then uses the RCCL exchange (gbp_amd/sharded.py does, and records why). */
int gbp_ba_peer_selftest(gbp_ba_t *h, int32_t timeout_ms);
int gbp_ba_iterate_sharded(gbp_ba_t *h, int32_t n_iters, int32_t robustify, int32_t local_relin);   /* n x synchronous_iteration gbp.py:86-92 */
int gbp_ba_update_beliefs_sharded(gbp_ba_t *h);                                                     /* update_all_beliefs gbp.py:56-58 */

/* streaming export of all means for a viewer (the reference's viewer thread reads node.mu of every variable per frame,
 * vis/ba_vis.py:35-55): snapshot = taken in stream order, copied to a pinned host mirror on a copy stream while the
 * following sweeps run; fetch = the newest snapshot that has landed (wait != 0: block for the latest one). */
int gbp_ba_means_snapshot(gbp_ba_t *h);
int gbp_ba_means_fetch(gbp_ba_t *h, double *cam_mu, double *lmk_mu, int32_t wait);

/* BAL-style text files (layout data/README.md:5-14), host only: replaces utils/read_balfile.py:4-37.  First the sizes,
 * then the arrays into caller-owned buffers: K4 = fx fy cx cy, cam_means[C*6], lmk_means[L*3], meas[F*2], ids[F] in FILE
 * order (gbp_ba_create takes them in this order and applies the reference's camera-major factor order itself). */
int gbp_bal_header(const char *path, int32_t *n_cams, int32_t *n_lmks, int32_t *n_obs);
int gbp_bal_read(const char *path, int32_t n_cams, int32_t n_lmks, int32_t n_obs, double *K4, double *cam_means,
                 double *lmk_means, double *meas, int32_t *cam_idx, int32_t *lmk_idx);

/* state checkpoint / restore (the reference holds its state in Python objects and has no
 * counterpart).  The blob holds everything a sweep reads or writes (linearisation points, adaptive variances, messages,
 * relinearisation state, beliefs, means, priors) behind a header that pins the graph; it restores only into a handle
 * created from the same graph.  A restored handle continues bit-identically. */
int gbp_ba_state_size(gbp_ba_t *h, uint64_t *bytes);
int gbp_ba_save_state(gbp_ba_t *h, void *buf, uint64_t bytes);
int gbp_ba_load_state(gbp_ba_t *h, const void *buf, uint64_t bytes);
/* the same checkpoint kept on the device (one slot per handle): restore is a device-to-device copy in stream order */
int gbp_ba_snapshot_state(gbp_ba_t *h);
int gbp_ba_restore_snapshot(gbp_ba_t *h);

/* instrumentation for bench.py: HIP-event time of the dominant (factor) kernel on the handle's stream; enable = n > 1
 * brackets only every n-th launch (two event records per sweep are not free: ~6 us of a 125 us sweep) */
int gbp_ba_set_kernel_timing(gbp_ba_t *h, int32_t enable);
int gbp_ba_get_kernel_timing(gbp_ba_t *h, double *total_ms, int32_t *n_launches, const char **kernel_name);
/* the same instrumented run also stamps, inside the kernels, the device's constant-rate clock (wall_clock64): per sweep since
 * gbp_ba_set_kernel_timing six slots in MICROSECONDS since the first stamp, of which three are used -- [0] fused sweep, [2]
 * camera reduce, [4] camera finish (sharded only): when the kernel's workgroup 0 started; NaN where a kernel did not run.
 * Consecutive START stamps tile the stream's timeline the way rocprofv3's kernel durations do (a kernel's interval includes
 * its own drain and the next dispatch).  HIP events around a launch add the dispatch latency behind the event's barrier
 * packet (5-8 us) and serialise the stream; the stamps are one store per launch.  Up to 4096 sweeps per enable. */
int gbp_ba_get_sweep_clocks(gbp_ba_t *h, double *us6, int32_t cap_sweeps, int32_t *n_sweeps);
/* which exchange the sharded loop uses and what it says about itself: kind GBP_COMM_*, this rank, the rank count (for RCCL:
 * ncclCommCount of the library's communicator) */
enum { GBP_COMM_NONE = 0, GBP_COMM_CALLBACK = 1, GBP_COMM_RCCL = 2, GBP_COMM_PEER = 3 };
int gbp_ba_comm_info(gbp_ba_t *h, int32_t *kind, int32_t *rank, int32_t *n_ranks);
int gbp_ba_get_kernel_times(gbp_ba_t *h, double *ms, int32_t cap, int32_t *n_launches);   /* each bracketed launch, in order; call BEFORE get_kernel_timing (which resets) */
int gbp_ba_info(gbp_ba_t *h, int32_t *fused_path, int32_t *n_tiles, int32_t *n_blocks);   /* fused_path: 0 = general sweep (camera-major staging, any number of cameras), 1 = fused sweep (camera table in LDS) */
/* what the plan of this handle's sweep decided, out[0..n): [0] fused sweep (1) or general sweep (0); [1] the general sweep was picked
 * by the sparseness rule (few factors per camera and workgroup), not asked for; [2] the fused sweep adds all same-camera lanes of a
 * tile in ONE LDS atomic instruction (the SINGLE variant: graphs of few cameras); [3] the probe of the lane order SINGLE relies on,
 * run at create on the handle's device: 1 passed, 0 failed (the rounds variant runs instead), -1 SINGLE was not wanted; [4] tiles per
 * workgroup that keep using the memory-side cache (-1: all of them); [5] workgroups; [6] tiles; [7] tile packing: 0 every landmark
 * inside one 64-slot tile, 1 the same with the landmarks above 64 factors cut into chunk tiles, 2 dense (tile t = factors [64 t, 64 t + 64)
 * of the landmark-major list: chosen when whole landmarks would leave more than 15 % of the slots empty and every landmark has at
 * least three factors).  From 1 on some landmarks span tiles: their beliefs are formed by a small kernel after the sweep; [8] camera
 * windows: the largest per-workgroup camera table of the fused sweep (each workgroup's table covers only the distinct cameras its own
 * tiles meet: sequences and sparse graphs, where those are few however many cameras there are), 0 = every table covers all cameras; [9] rows
 * of all tables together (windows: their sum; else workgroups x cameras), 0 under the general sweep; [10] the reduce behind camera
 * windows adds a camera's rows with one wave (1: at most 16 rows per camera on average) or as a tree (0: also without windows). */
#define GBP_PLAN_INFO_FIELDS 11
int gbp_ba_plan_info(gbp_ba_t *h, int32_t *out, int32_t n);
int gbp_ba_phase_profile(gbp_ba_t *h, uint64_t *out, int32_t cap_rows, int32_t *n_rows, int32_t *n_cols);   /* debug builds with -DGBP_PHASE_TIMING only (tools/phase_profile.py): per-wave time per phase of the last fused sweep */
int gbp_ba_check_layout(gbp_ba_t *h, int32_t *bad_slots);   /* debug: slots whose (camera, landmark) do not match the reference factor they hold (0 = sound) */
int gbp_ba_fused_max_cams(void);    /* most cameras of ONE workgroup's table in the fused sweep (camera table + wave scratch in 160 KB of LDS); graphs above it run the general sweep unless their camera windows fit (gbp_ba_plan_info [8]) */

#ifdef __cplusplus
}
#endif
#endif /* GBP_BA_H */
