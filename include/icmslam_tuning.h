/*
 * icmslam_tuning.h -- cross-check forms, test hooks and tuning knobs of the MI355X-native ICM sweep.
 *
 * Nothing here is needed to drive a sweep (include/icmslam.h is the drop-in boundary): these entry points select
 * algebraically identical forms of the same computation for parity tests, expose counters, or switch experiments
 * measured in DESIGN.md.  Every one of them leaves the results bit-identical unless it says otherwise.
 */
#ifndef ICMSLAM_TUNING_H
#define ICMSLAM_TUNING_H

#include "icmslam.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Test hook: run phase A with the brute-force kernel (every beam against every landmark of
 * mapa_viejo, table tiled through LDS) instead of the grid search.  Same results. */
int icm_set_brute_force(icm_handle *h, int on);

/* What phase A associates (Mapa.actualizar's cdist / argmin / gate, scripts/ICM_SLAM_tools.py:168-172):
 *   1 (default) geometric RUNS of each scan's kept beams -- clusters of neighbouring returns cut once per sequence, beside
 *               filtrar_z, each with its bounding circle: one lane per run settles the label of all its beams from the
 *               circle's centre when the nearest landmark wins by more than the circle's diameter (exact: see
 *               k_assoc_runs), and only the runs this does not settle go beam by beam;
 *   0           every kept beam on its own (k_assoc_group): the cross-check form.
 * Labels and counts are identical; the per-entry sums of body points are added up in a different order (~1e-16 relative).
 * icm_get_run_counts: out2 = [0] runs of the uploaded shard, [1] runs that went beam by beam so far, over the handle's
 * life (synchronises the stream). */
int icm_set_assoc_form(icm_handle *h, int form);
int icm_get_run_counts(icm_handle *h, int64_t *out2);
/* The runs themselves (tests): offsets[nloc + 1] by pose, then per run its bounding circle's centre (x, y interleaved, body
 * frame), the sum of its beams' body points (interleaved), the radius (rounded up), its beam count and the offset of its
 * first beam among the pose's kept beams.  Sized by icm_get_run_counts()[0]; any pointer may be NULL. */
int icm_get_runs(icm_handle *h, int64_t *offsets, double *centre_xy, double *sum_xy, float *radius, int32_t *count, int32_t *first);

/* Sweeps queued whole start Mapa.filtrar on a side stream the moment the raw map is out: a one-wave kernel there polls a
 * word the main stream's k_lm_l3 sets (no event packet on the main queue).  Where the two streams cannot run side by side
 * (counter collection with rocprofv3 --pmc, HIP_LAUNCH_BLOCKING / AMD_SERIALIZE_KERNEL, more streams than hardware queues)
 * the wait gives up after ~16 ms, that sweep's Mapa.filtrar runs on the host (same result), and the handle switches to a
 * stop event for the rest of its life.  icm_get_wait_giveups: how many sweeps that happened to (0 in a normal run; a
 * profile taken under serialisation shows the event path from the second sweep on). */
int icm_get_wait_giveups(const icm_handle *h, int64_t *sweeps);
/* Environment knobs read once at icm_create (A/B measurements; results identical): ICM_L3_EVENT=1 starts the side stream
 * by k_lm_l3's stop event from the first sweep (what a give-up switches to); ICM_SOLVE_PPW=32|64 fixes the poses per wave
 * of the one-launch solve (default: 64, 32 for short colours). */

/* Keep the per-beam outputs of a sweep (label and running-mean target of every kept beam)
 * for icm_get_association; off by default (they cost 28 B of HBM traffic per kept beam). */
int icm_set_debug(icm_handle *h, int on);
/* With debug on: per pose (T,3) row-major [final energy, NM iterations, function evaluations]
 * of the last sweep's solve (0 for poses without a solve). */
int icm_get_solve_diag(icm_handle *h, double *out);
/* Form in which the pose solves evaluate the observation energy h(x) of
 * scripts/ICM_ROS.py:171-200 -- the same function in three algebraically identical forms:
 *   0 (default) moment form: quadratic form in (dp, cos d - 1, sin d) about the pose's
 *               previous value, 14 sums per pose; one LANE solves a pose
 *   1           one term per kept beam, literally the reference's sum; one wave per pose
 *   2           one term per (pose, landmark) entry: k |p + R bbar - y|^2_Q + scatter; one
 *               wave per pose
 * Forms 1 and 2 exist to cross-check form 0. */
int icm_set_energy_form(icm_handle *h, int form);

/* Lanes per pose in the red-black solves: 0 / -1 (default) = one lane per pose (throughput form, both colours in one
 * launch), 1 = one DPP quad per pose evaluating the four candidate points of a Nelder-Mead iteration at once (latency
 * form, kept as a cross-check: one launch per colour; measured slower at every size since the folded energy).
 * Bit-identical results. */
int icm_set_solve_lanes(icm_handle *h, int mode);

/* Red-black sweeps in throughput form: 1 (default) = both colours in ONE launch, every
 * even wave starting as soon as the two odd waves holding its poses' neighbours are done
 * (k_solve_m_fused); 0 = one launch per colour.  Bit-identical results. */
int icm_set_colour_fusion(icm_handle *h, int on);
/* How many times an even wave of the one-launch solve polls for its odd neighbours (~0.2 us per
 * poll; default 1 << 17) before it DEFERS: it leaves its poses untouched, and the wave of the launch
 * that finishes last -- every odd pose is final then -- solves the deferred waves before the launch
 * ends.  Forward progress therefore never depends on the order workgroups are dispatched in; 0 defers
 * every wave whose neighbours are not done at its first look.  Bit-identical results for every value.
 * icm_get_fused_deferred: waves deferred so far over the handle's life (synchronises the stream). */
int icm_set_fused_spin_limit(icm_handle *h, int polls);
int icm_get_fused_deferred(icm_handle *h, int64_t *waves);

/* What the Nelder-Mead loop of the one-launch solve evaluates (fun_xn / fun_x, scripts/ICM_ROS.py:220-278):
 *   1  the FOLDED form only -- the whole conditional energy as one quadratic in the planar step with 13 per-pose
 *      coefficients (valid per pose while its heading step stays within 0.25 rad and no angle residual can wrap); a
 *      pose one of whose evaluations leaves that range is solved once more on the spot, by the same wave, with
 *   0  the complete energy (folded where valid, term by term elsewhere) in the loop itself;
 *  -1  (default) 1 with isotropic weights Q0 == Q1, R0 == R1, else 0 (the folded form never holds then).
 * Which road a pose takes is decided from its own data and both evaluate identical arithmetic: bit-identical results.
 * icm_get_fixup_poses: poses that needed the second solve, over the handle's life. */
int icm_set_fold_mode(icm_handle *h, int mode);
int icm_get_fixup_poses(icm_handle *h, int64_t *poses);

/* icm_sweep calls on a registered pose array (icm_pin_host) so far: out3 = [0] calls that started from the device's poses
 * without an upload (the array was the one the call before filled), [1] of those, calls whose check of the array against
 * the device failed -- the caller had changed it -- and that started over with an upload, [2] calls whose solves wrote
 * the poses into the caller's array themselves (no download). */
int icm_get_dropin_counts(const icm_handle *h, int64_t *out3);

/* Diagnostics of a sharded job (bench.py prints them per rank at N > 1): with phase timing on, every sweep queued whole
 * records HIP events on the handle's stream at its phase boundaries and icm_sweep_finish waits for the solves (so the
 * sweeps are a few microseconds slower and no longer overlap: never inside a timed window).  icm_get_phase_times:
 * accumulated ms over *sweeps sweeps: [0] phase A + local statistics, [1] the exchange including the wait for the
 * slowest rank, [2] targets / ghost pose / moments, [3] the solve launch, [4] host time spent waiting in icm_sweep_finish. */
int icm_set_phase_timing(icm_handle *h, int on);
int icm_get_phase_times(const icm_handle *h, double *out5, int64_t *sweeps);

/* Test hook: where = 1 makes the next icm_sweep_local fail with ICM_ERR_HIP before it launches anything (a rank of a
 * sharded job whose device failed: it must still take part in the sweep's collective, icm_sweep_sharded); where = 2 makes
 * the next icm_sweep_targets fail the same way -- BEHIND the sweep's exchange, where the peers no longer wait for this
 * rank in this sweep (icm_sharded_end); 0 = off. */
int icm_set_fault(icm_handle *h, int where);

/* Sizes of the staging area of phase A's (pose, landmark) entries for a shard with nnz kept beams and nloc poses (host
 * arithmetic only, no GPU): out3 = [first place of the sparse area, capacity of the staged-entry arrays, capacity of
 * each per-entry prefix array].  ICM_ERR_CAPACITY when they exceed 32-bit entry offsets. */
int icm_staging_layout(int64_t nnz, int64_t nloc, int64_t *out3);

/* Pipeline that turns the per-pose entries into running-mean targets (the time-ordered
 * per-landmark prefix of Mapa.actualizar, scripts/ICM_SLAM_tools.py:184-196):
 *   1 / -1 (default) = hierarchical running sums (pose chunks -> superchunks -> per-landmark
 *       column prefix; no sort); used for the moment-form solves.  A map so dense that a
 *       64-pose chunk sees more than ~190 distinct landmarks makes the sweep fall back to
 *   0 = the sort-based pipeline (radix sort of the entries by landmark + one wave per
 *       landmark), which has no such limit and also serves energy forms 1/2 and icm_set_debug.
 * The two differ only in the order the per-landmark sums are added up (~1e-15 relative).
 * icm_get_entry_path: pipeline the last sweep actually ran (0 or 1). */
int icm_set_entry_path(icm_handle *h, int mode);
int icm_get_entry_path(const icm_handle *h);

/* Where Mapa.filtrar runs inside a sweep: 1 (default) = on the GPU (the k_fl_* kernel chain on a
 * side stream: prune, grid, nearest-neighbour pairs, and -- when survivors are closer than dist_thr
 * -- label propagation, renumbering and count-weighted means; only coincident landmarks, an empty
 * map or a merge component of more than 7 landmarks go to the host routine), 0 = always the host
 * routine icm_filtrar.  Same results.
 * icm_last_filtrar_info: [0] landmarks_actuales after the last sweep's filter, [1] where it ran
 * (0 GPU, no merges; 1 GPU with merges; 2 host routine), [2] landmarks that had a neighbour closer
 * than dist_thr (-1 on the host path). */
int icm_set_gpu_filtrar(icm_handle *h, int on);
int icm_last_filtrar_info(const icm_handle *h, int64_t *out3);

#ifdef __cplusplus
}
#endif
#endif /* ICMSLAM_TUNING_H */
