/*
 * icmslam.h -- C-ABI of the MI355X-native offline ICM sweep.
 *
 * The reference (Seba-san/icm-slam) is pure Python and has no FFI; the boundary it
 * exposes for this path is the Python method
 *     mapa_refinado, x = ICM_ROS.iterations_process_offline(mapa_viejo, x)
 * (reference scripts/ICM_ROS.py:121-164, called from scripts/example.py:52 and
 * scripts/ICM_ROS.py:301).  The entry points below are what a ctypes binding inside that
 * method binds to; each cites the reference code it replaces.
 *
 * Conventions: plain pointers and sizes only; arrays are C-contiguous float64 in the
 * reference's own layouts ((3,T) poses = all x, then all y, then all theta); every
 * function returns 0 on success or a negative ICM_ERR_* code (never throws);
 * icm_last_error() gives the text.  One handle per GPU / rank; calls on one handle must
 * be serialised by the caller (the reference is non-reentrant as well).
 * Host pointers are borrowed for the duration of the call only.
 */
#ifndef ICMSLAM_H
#define ICMSLAM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ICM_OK 0
#define ICM_ERR_ARG (-1)         /* bad argument / call order                              */
#define ICM_ERR_HIP (-2)         /* HIP runtime error                                      */
#define ICM_ERR_INDEX (-3)       /* reference would raise IndexError (label >= L, or a     */
                                 /* no-beam last pose, scripts/ICM_ROS.py:144)             */
#define ICM_ERR_EMPTY_MAP (-4)   /* Mapa.filtrar left no landmark (reference: ValueError)  */
#define ICM_ERR_CAPACITY (-5)    /* a scan touched more distinct landmarks than supported  */
#define ICM_ERR_UNSUPPORTED (-6) /* input outside what this build implements               */
#define ICM_RETRY_CAREFUL 1000   /* icm_sweep_finish only, after icm_set_optimistic(h, 1): a table overflowed on some
                                    rank, nothing was replaced; repeat the sweep's phase calls with icm_set_optimistic(h, 0) */

#define ICM_SCHEDULE_SEQUENTIAL 0 /* reference Gauss-Seidel order (scripts/ICM_ROS.py:141) */
#define ICM_SCHEDULE_REDBLACK 1   /* odd poses, then even poses (parallel)     */

/* Numeric options of ConfigICM (reference scripts/ICM_SLAM_tools.py:60-102). */
typedef struct icm_config {
    double deltat;          /* sampling period                                            */
    double Q[2];            /* diag of the observation weight                             */
    double R[3];            /* diag of the motion-model weight                            */
    double cte_odom;        /* odometry weight                                            */
    double cota;            /* min. observation count for a landmark to survive filtrar   */
    double dist_thr;        /* association gate / merge distance                          */
    double rango_laser_max; /* max laser range                                            */
    int64_t L;              /* landmark capacity (columns of the running map)             */
} icm_config;

typedef struct icm_handle icm_handle;

/* ---- life cycle ---------------------------------------------------------------------- */
/* Create a solver bound to HIP device `device`.  Fails (ICM_ERR_HIP) if no GPU: there is
 * no CPU fallback. */
int icm_create(const icm_config *cfg, int device, icm_handle **out);
int icm_destroy(icm_handle *h);
const char *icm_last_error(const icm_handle *h); /* h may be NULL: last create error */
/* Launch all work on an existing HIP stream (e.g. torch's current stream) instead of the
 * handle's own. */
int icm_set_stream(icm_handle *h, void *hip_stream);

/* ---- sequence upload + scan pre-filter ------------------------------------------------ */
/* Replaces the attributes the sweep reads from the ICM_ROS object: `mediciones` (B,T),
 * `odometria` (3,T), `u` (2,T) (reference scripts/ICM_ROS.py:20-24,131,141-142).
 *   ranges  [(t_end-t_begin)*B] pose-major: the scans of poses t_begin..t_end-1 of this
 *           rank's shard (the transpose of the reference's beam-major `mediciones`)
 *   odo     [3*T], u [2*T]   full sequence, reference layout
 *   cosb,sinb [B]  cos/sin of the beam bearings (reference: k*pi/180,
 *           scripts/ICM_SLAM_tools.py:44,51), computed by the host
 * Single GPU: t_begin = 0, t_end = T. */
int icm_upload(icm_handle *h, const double *ranges, const double *odo, const double *u,
               const double *cosb, const double *sinb, int64_t T, int64_t B, int64_t t_begin,
               int64_t t_end);
/* Multi-rank jobs, ranks > 0 only (between icm_upload and icm_prefilter): the scan of pose t_begin - 1, the last pose
 * of the shard below.  A shard solves that pose as well (its "ghost pose": the same beams, the same neighbours' values
 * and the same additions over its entries as its owner uses; its TARGETS -- the running means through that pose -- are
 * the same numbers up to the rounding of sums the two ranks add up in different associations, ~1e-16 relative), so that
 * the shard's first even pose finds its odd neighbour without an exchange between the two colours of a sweep (reference
 * loop scripts/ICM_ROS.py:141-158; a pose reads t-1 and t+1 only, :211-214).  A Nelder-Mead result is a simplex vertex,
 * so the ghost's result is its owner's bit for bit unless one of the solve's comparisons flips on those last bits; the
 * owner's value replaces it with the next exchange either way (tests: equal on every pose tested, bounded by 1e-9).
 *   ranges_row [B] */
int icm_upload_ghost_scan(icm_handle *h, const double *ranges_row);
/* filtrar_z for every scan of the shard, once per sequence (reference
 * scripts/ICM_SLAM_tools.py:22-58, called per pose per sweep at scripts/ICM_ROS.py:130,142).
 * nnz_out = number of kept beams. */
int icm_prefilter(icm_handle *h, int64_t *nnz_out);
/* Copy the kept beams out (parity tests): offsets[nloc+1], then per kept beam its scan row
 * index, range d, body x, body y (columns 0,2,3 of filtrar_z's rows; column 1 is
 * bearing(index)).  Any output pointer may be NULL. */
int icm_get_kept(icm_handle *h, int64_t *offsets, int32_t *beam_index, double *d, double *bx,
                 double *by);

/* ---- one sweep through host arrays (the drop-in call) --------------------------------- */
/* iterations_process_offline (reference scripts/ICM_ROS.py:121-164).
 *   x        [3*T] in/out, updated in place like the reference
 *   x0       [3]   self.x0, used to project scan 0 (scripts/ICM_ROS.py:125,137)
 *   map_in   [2*K] mapa_viejo (2,K), not modified
 *   lact_in  Mapa.landmarks_actuales on entry (normally == K)
 *   map_out  [2*L] filtered map, zero padded like Mapa.filtrar's return
 *   counts_out [L] Mapa.cant_obs_i after the sweep
 *   K_out    Mapa.landmarks_actuales after the sweep (columns of mapa_refinado)
 * If scan 0 has no kept beams the reference returns its inputs untouched; so does this
 * (K_out = -1 signals that case). */
int icm_sweep(icm_handle *h, double *x, const double *x0, const double *map_in, int64_t K,
              int64_t lact_in, int schedule, double *map_out, double *counts_out, int64_t *K_out);

/* Optional, for callers that hand the SAME pose array to icm_sweep call after call (the reference's driver loop updates
 * `x` in place and passes it back, scripts/ICM_ROS.py:158,164,298-311): register the array with the GPU runtime once, and
 * every later icm_sweep / icm_set_state / icm_get_state whose `x` lies inside the range reads and writes it in place over
 * PCIe instead of through two staged copies (S2: 0.67 -> 0.4x ms per call).  The range must stay allocated until
 * icm_unpin_host or icm_destroy (a freed and re-used address would alias stale pages): callers that cannot promise that
 * simply do not pin -- results are identical either way. */
int icm_pin_host(icm_handle *h, void *ptr, size_t bytes);
int icm_unpin_host(icm_handle *h, void *ptr);

/* ---- device-resident sweeps (state stays in HBM between sweeps) ------------------------ */
int icm_set_state(icm_handle *h, const double *x, const double *x0, const double *map_in,
                  int64_t K, int64_t lact_in);
/* One sweep on the resident state; the refined map becomes the next sweep's mapa_viejo
 * (reference driver loop scripts/ICM_ROS.py:298-311). */
int icm_sweep_device(icm_handle *h, int schedule);
int icm_get_state(icm_handle *h, double *x, double *map_out, double *counts_out, int64_t *K_out);

/* Snapshot of the sweep state held on the device (poses, mapa_viejo with its counters and search
 * structures) and its restoration by device-to-device copies, stream-ordered, no host round trip:
 * re-running sweeps from the same start (benchmarks, A/B comparisons of the knobs below).  A
 * snapshot belongs to the uploaded sequence; icm_upload discards it. */
int icm_snapshot_state(icm_handle *h);
int icm_restore_state(icm_handle *h);

/* ---- sharded sweep: the same sweep cut at the ONE point where ranks exchange data -------- */
/* Partition: contiguous pose blocks of icm_shard_block(T, world) poses -- ceil(T / world) rounded up to an
 * even number, so every shard starts at an even pose; rank r owns [r blk, min((r+1) blk, T)) and uploads exactly those
 * scans plus, for r > 0, the scan in front (icm_upload_ghost_scan).  odo / u / the pose array / the landmark table are
 * replicated.  One sweep of the reference loop scripts/ICM_ROS.py:141-158, red-black order, is then per rank
 *     icm_sweep_local    phase A over the shard + its landmark sufficient statistics
 *     ONE all-gather     of icm_stats_stride() doubles per rank: [sum x (L) | sum y (L) | n (L) | header (16)], header =
 *                        [0] landmarks created, [1] flags / error code, [2..4] first pose, [5..7] last pose, [8..10] last
 *                        pose but one of the shard -- the previous sweep's values, which the neighbours need as OLD values
 *     icm_sweep_targets  prefix over lower ranks -> running-mean targets, raw map; the neighbours' boundary poses are
 *                        unpacked from their headers; the ghost pose's entries and moments
 *     icm_sweep_solve(REDBLACK, -1)   both colours of the shard in one launch (the ghost pose is its first odd pose)
 *     icm_sweep_finish   Mapa.filtrar, replicated
 * -- no second exchange.  Bind the exchange buffers (device memory owned by the caller, e.g. torch tensors):
 *   stats_all [world * icm_stats_stride()] doubles: rank r's message lives at stats_all + r*stride; the caller
 *             all-gathers it between icm_sweep_local() and icm_sweep_targets(). */
int64_t icm_shard_block(int64_t T, int world);
int64_t icm_stats_stride(const icm_handle *h);
int icm_bind_exchange(icm_handle *h, void *stats_all_dev, int rank, int world);
/* Optional send-side buffer, so that the exchange is one collective call and nothing else on the caller's side
 * (all-gather input and output must not alias): stats_send [icm_stats_stride()] doubles -- icm_sweep_local() writes this
 * rank's message here instead of into its slice of stats_all.  Null = in place. */
int icm_bind_exchange_send(icm_handle *h, void *stats_send_dev);
/* Use caller-owned device memory for the poses: (T,3) doubles, pose-major (a contiguous
 * block of poses is a contiguous block of memory, so shards all-gather in place).  Call
 * before icm_set_state. */
int icm_bind_pose_buffer(icm_handle *h, void *x_dev);
void *icm_pose_buffer(icm_handle *h);                 /* device pointer of x (T,3)          */
/* Collectives issued by the library itself (RCCL over xGMI, resolved with dlopen at run time)
 * instead of by the caller: rank 0 makes an id (icm_comm_unique_id, 128 bytes) and hands it to every
 * rank by whatever means the application has; each rank, after uploading block `rank`, calls icm_comm_init, which
 * creates the communicator and allocates and binds the exchange buffers (statistics, replicated pose array).
 * icm_sweep_sharded then is one whole red-black sweep as listed above with its one ncclAllGather on the handle's stream;
 * a rank that fails on its own in phase A still takes part in the collective and every rank returns that error.
 * icm_gather_poses all-gathers the pose blocks (in place) before icm_get_state. */
int icm_comm_set_library(const char *path);           /* the RCCL copy to load if the process has none loaded yet
                                                          (a process must use ONE: e.g. the copy a PyTorch wheel bundles) */
int icm_comm_available(void);                         /* 1 when an RCCL library can be resolved */
int icm_comm_unique_id(void *id128);
int icm_comm_init(icm_handle *h, const void *id128, int rank, int world);
/* The same driver over a caller-supplied all-gather instead of RCCL (MPI, a test harness that carries the messages
 * through host memory): gather `count` doubles from every rank's `send_dev` into `recv_dev` (rank-major; send_dev may be
 * recv_dev + rank*count), ordered after the work queued on `hip_stream`, complete or stream-ordered on return; 0 = ok. */
typedef int (*icm_allgather_fn)(const void *send_dev, void *recv_dev, size_t count, void *hip_stream, void *user);
int icm_comm_init_transport(icm_handle *h, int rank, int world, icm_allgather_fn fn, void *user);
int icm_comm_destroy(icm_handle *h);
int icm_sweep_sharded(icm_handle *h);
int icm_gather_poses(icm_handle *h);
/* Failing together BEHIND a sweep's exchange (a device error in the targets, the solve launch or Mapa.filtrar of one rank,
 * while its peers' phases ran and they are on their way to the next exchange): icm_sweep_sharded on that rank sends one
 * more message -- its header carrying the code, nothing else -- which the peers receive as their NEXT exchange: the next
 * sweep's, or the closing exchange icm_sharded_end, which every rank of a job calls once after its last sweep
 * (icm_gather_poses runs it first).  Every rank then returns that rank's error, one exchange later; nobody waits for a
 * rank that has left.  A rank that cannot send at all aborts the communicator (RCCL: ncclCommAbort; with a
 * caller-supplied transport the peers are left to that transport's own timeout). */
int icm_sharded_end(icm_handle *h);
/* Phase calls only (icm_sweep_device / icm_sweep / icm_sweep_sharded do this themselves): queue the sweep whole, without
 * the host looking at phase A's counts and overflow flags in the middle -- solves and Mapa.filtrar check the flags on
 * the device (on every rank: the flags travel in the header of the statistics message); icm_sweep_finish then returns
 * ICM_RETRY_CAREFUL if something overflowed.  Red-black sweeps through the default pipeline only; else the request is
 * ignored for that sweep. */
int icm_set_optimistic(icm_handle *h, int on);
int icm_get_optimistic(const icm_handle *h);          /* 1: the sweep icm_sweep_local started WAS queued whole */
int icm_sweep_local(icm_handle *h);                   /* phase A + local statistics          */
int icm_sweep_targets(icm_handle *h);                 /* prefix over ranks -> targets, map   */
int icm_sweep_solve(icm_handle *h, int schedule, int colour); /* colour 1 = odd, 0 = even,  */
                                                      /* -1 = both / sequential             */
int icm_sweep_finish(icm_handle *h);                  /* Mapa.filtrar, next mapa_viejo       */
/* Failing together.  A rank whose icm_sweep_local failed in the careful form (a table too small even at its largest
 * size, labels beyond L: the reference's IndexError, scripts/ICM_SLAM_tools.py:191) must not leave the others waiting in
 * the collective: it calls icm_mark_failed(h, code) -- its message then carries the code in header [1] -- and takes
 * part in the exchange; after it, EVERY rank calls icm_failed_rank and stops if some rank failed (*rank_out >= 0,
 * *code_out its ICM_ERR_* code).  icm_sweep_sharded does both itself. */
int icm_mark_failed(icm_handle *h, int code);   /* code 0: a clean header (the closing exchange of callers that issue the collectives themselves) */
int icm_failed_rank(icm_handle *h, int *rank_out, int *code_out);
/* The same look at every rank's header, plus *retry_out = 1 when some rank reports flags of a sweep it had queued whole
 * (header [1] == 1): that rank and every rank that queued the sweep whole will repeat it (ICM_RETRY_CAREFUL from their
 * icm_sweep_finish), so a rank whose own sweep was NOT queued whole (icm_get_optimistic() == 0 after icm_sweep_local) must
 * call this after the exchange and, on retry, skip icm_sweep_targets / _solve / _finish and repeat the sweep with them.
 * And before ANY rank starts the repeated sweep's exchange it calls this once more: a rank that failed in the first
 * exchange (code in its header) has left and will not join a second one -- every rank then stops with that code.
 * icm_sweep_sharded does all of it. */
int icm_exchange_status(icm_handle *h, int *rank_out, int *code_out, int *retry_out);

/* ---- kernel-level entry points for parity tests ---------------------------------------- */
/* Labels of every kept beam after phase A of the last sweep (reference `c` of
 * Mapa.actualizar, scripts/ICM_SLAM_tools.py:170-181; new landmarks carry their fresh id)
 * and the running-mean targets y[:,c] each pose was solved against (scripts/ICM_ROS.py:152). */
int icm_get_association(icm_handle *h, int32_t *labels, double *target_x, double *target_y);
/* Raw running map before Mapa.filtrar: y (2,L) and cant_obs_i (L), landmarks_actuales. */
int icm_get_raw_map(icm_handle *h, double *y, double *counts, int64_t *lact);
/* One Nelder-Mead solve on the GPU (reference minimizar_xn / minimizar_x,
 * scripts/ICM_ROS.py:209-218,254-260).  two_sided=1: fun_xn with x_pos, u[2x2 col-major as
 * (2,2) numpy u[:,t-1:t+1]], odo (3,3) = odometria[:,t-1:t+2]; two_sided=0: fun_x with
 * odo (3,2).  beams (n): body x,y; targets (n): y[:,c].  out[6] = x,y,theta,f,nit,nfev. */
int icm_solve_one(icm_handle *h, int two_sided, const double *x_ant, const double *x_pos,
                  const double *u, const double *odo, int odo_cols, const double *bx,
                  const double *by, const double *tx, const double *ty, int64_t n, double *out);
/* Energy only, same arguments, at pose `x`: out[0] = fun_xn / fun_x (scripts/ICM_ROS.py:220-278);
 * two_sided = 2 evaluates the observation energy h(x) alone (scripts/ICM_ROS.py:171-200). */
int icm_energy_one(icm_handle *h, int two_sided, const double *x, const double *x_ant,
                   const double *x_pos, const double *u, const double *odo, int odo_cols,
                   const double *bx, const double *by, const double *tx, const double *ty,
                   int64_t n, double *out);

/* ---- initialisation pass (the caller of the sweep's inputs) ----------------- */
/* Flat clusters of the first scan's world points: Mapa.actualizar with Lact == 0 (reference
 * scripts/ICM_SLAM_tools.py:160-165), i.e. fcluster(linkage(pdist(pts)), t) - 1 with SciPy's
 * defaults (single linkage, 'inconsistent' criterion, depth 2).  Host only.  pts (n,2). */
int icm_cluster_first_scan(const double *pts, int64_t n, double t, int32_t *labels_out);
/* The causal pass of inicializar_online / inicializar_online_process (reference
 * scripts/ICM_ROS.py:57-119) over the uploaded sequence: predict, associate against the running
 * map, running-mean update, one-sided solve, for t = 1..T-1.
 *   x0 [3]; y (2,L) / counts (L) / lact: the map seeded from scan 0, updated in place;
 *   x_out [3*T] the initial poses.  Mapa.filtrar (icm_filtrar) is applied by the caller. */
int icm_init_pass(icm_handle *h, const double *x0, double *y, double *counts, int64_t *lact, double *x_out);

/* ---- host-side map prune/merge (no GPU needed) ----------------------------------------- */
/* Mapa.filtrar (reference scripts/ICM_SLAM_tools.py:204-265): y (2,L) row-major, counts (L),
 * lact in/out.  y_out (2,L) zero padded, counts_out (L). */
int icm_filtrar(const icm_config *cfg, const double *y, const double *counts, int64_t lact,
                double *y_out, double *counts_out, int64_t *lact_out);

/* Mapa.filtrar (scripts/ICM_SLAM_tools.py:204-265) on the GPU for a caller-held map: same
 * arguments as icm_filtrar; *path_out as [1] above.  Invalidates the handle's sweep state
 * (icm_set_state again before the next sweep). */
int icm_filtrar_device(icm_handle *h, const double *y, const double *counts, int64_t lact, double *y_out,
                       double *counts_out, int64_t *lact_out, int *path_out);

/* Mapa.actualizar for ONE scan, outside a sweep (reference scripts/ICM_SLAM_tools.py:128-201; the
 * call the reference's online initialisation makes per sample, scripts/ICM_ROS.py:114):
 *   obs (n,2) row-major world points of the scan's kept beams; map_ref (2,K_ref) row-major, NOT
 *   modified; map (2,L) row-major running map and counts (L) = cant_obs_i, both updated in place;
 *   *lact_inout = landmarks_actuales in and out; labels_out (n) = the reference's `c`.
 * landmarks_actuales == 0: the first-scan branch (:160-165) -- single-linkage clusters at
 * dist_thr, cluster centres and sizes.  Otherwise (:167-197): nearest column of
 * map_ref[:, :landmarks_actuales] per observation on the GPU (cdist / argmin, first index on ties),
 * gate at dist_thr, ONE fresh label for all gated-out observations of the scan, then
 * the running-mean recurrence per label.  ICM_ERR_INDEX where the reference raises IndexError
 * (label >= L), ICM_ERR_ARG for an empty reference map. */
int icm_associate(icm_handle *h, const double *obs, int64_t n, const double *map_ref, int64_t K_ref,
                  double *map, double *counts, int64_t *lact_inout, int64_t *labels_out);

/* ---- instrumentation -------------------------------------------------------------------- */
/* When enabled, every kernel launch of a sweep is bracketed by HIP events on the handle's
 * stream; icm_kernel_time() returns accumulated ms and launch count per kernel name. */
int icm_enable_timing(icm_handle *h, int on);
int icm_reset_timing(icm_handle *h);
int icm_kernel_count(const icm_handle *h);
int icm_kernel_time(icm_handle *h, int idx, const char **name, double *ms, int64_t *launches);
/* Counters of the last sweep: [0] kept beams, [1] (pose,landmark) entries, [2] poses with
 * new landmarks, [3] labels in use (landmarks_actuales before filtrar). */
int icm_last_stats(const icm_handle *h, int64_t *out4);
const char *icm_version(void);
const char *icm_build_id(void);   /* first 16 hex digits of the sha256 of the sources the library was built from */
/* FP64 flops (fma = 2) / vector instructions of ONE evaluation of the pose energy (reference
 * fun_xn, scripts/ICM_ROS.py:220-252, in the moment form the solve kernels use), counted at build
 * time from the gfx950 ISA of csrc/eval_probe.hip by tools/count_eval_flops.py. */
int icm_flop_per_eval(void);
int icm_valu_per_eval(void);

#ifdef __cplusplus
}
#endif
#endif /* ICMSLAM_H */
