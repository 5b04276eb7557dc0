#!/usr/bin/env python3
"""Benchmark of the MI355X-native ICM sweep: ICM pose-updates/s over full sweeps.

    python bench.py                                  # 1 GPU, workload S2 (100k poses / 10k landmarks / 720 beams)
    python bench.py --gpus N --steps K --warmup W    # starts the N ranks itself (fresh child processes)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W      # or is started as one of them

A "step" is one full ICM sweep (reference `iterations_process_offline`, scripts/ICM_ROS.py:121-164:
association of every kept beam, running-mean map, one Nelder-Mead solve per pose, map
prune/merge) over the whole synthetic sequence, which is resident in HBM when timing starts.
Poses are solved in the red-black order (the reference's sequential order is one dependent
chain of T-1 solves and is used for parity, not throughput).

N > 1 (BASELINE.json configs[4]): the fixed S2 sequence is sharded by contiguous pose blocks
over the ranks ("scaling": "strong"), ONE all-gather per sweep (landmark sufficient statistics +
the shards' boundary poses; each shard solves the pose in front of it as a ghost pose instead of
exchanging a halo between the colours).  The same job is then repeated on a sequence N times as
long (every GPU owns a 100k-pose block: fixed work per GPU) and reported as the secondary
record `"weak"` of the same JSON line.  `--scaling weak` swaps the two.

Prints ONE JSON line on rank 0 (contract in the task description) with the extra objects
`roofline` (dominant kernel, HIP events on the launch stream), `cpu_baseline` (the C oracle on a
bounded prefix of the same workload, host cores) and, at N = 1, `dropin` (the same sweep through the
reference's call `ICM_ROS.iterations_process_offline`, host arrays in and out), `config3`
(S1 x 20 consecutive sweeps) and `dataset` (data_IJAC2018: init pass, one sweep in the reference's order, one red-black
sweep, beside the survey's timings of the real reference on the same input).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "icm-slam_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 achievable
FP64_VALU_PEAK_TFLOPS = 78.6  # 256 CUs x 4 SIMDs x 16 FP64 FMA lanes x 2 x 2.4 GHz (no MFMA on this path)
MAX_SWEEPS_WITHOUT_REWIND = 40   # S2 leaves the association gate after ~50 consecutive sweeps
RESET_EVERY = 12


def algorithmic_bytes(kernel, nnz, E, nloc, L, nlaunch, hier=True, runs=0):
    """Compulsory HBM bytes ONE launch of `kernel` moves: every array
    the kernel must read or write once, no re-reads.  nnz = kept beams, E = (pose, landmark)
    entries, nloc = poses of the shard, L = landmark capacity; hier = the hierarchical entry
    pipeline ran (k_pose_moments then reads the staged entries and the prefixes itself); runs =
    geometric runs of the kept beams (what k_assoc_runs associates)."""
    ch = 64 if nloc >= 65536 else (32 if nloc >= 16384 else 16)   # poses per chunk (icm_api.hip)
    nchunks = (nloc + ch - 1) // ch
    group = (nchunks + 63) // 64
    nsuper = (nchunks + group - 1) // max(group, 1)
    per_sweep = {
        # staged entry in (label 4, k 2, sum bx 8, sum by 8); prefix (sum x 8, sum y 8, one word of n and the
        # record slot 4) out; per pose: offsets 16 + pose 24; per chunk: 256 record labels
        "k_chunk_l1": E * 22 + E * 20 + nloc * 40 + nchunks * 256 * 4,
        # dense [superchunks x L] matrix of (sx, sy, n): read once, written once
        "k_lm_l3": 2 * 3 * 8 * nsuper * L + L * 24,
        # read body x,y of every kept beam (16 B) + pose; write one staged entry
        # (label 4, k 2, sum bx 8, sum by 8) per distinct landmark of the scan; counts/flags
        "k_assoc_group": nnz * 16 + E * 22 + nloc * (24 + 8 + 8),
        # phase A by runs: one 24-B record per run (sum of body points 16 -- the circle's centre is that sum / count --, radius
        # 4, count | first beam 4) in -- no beam is read where the bounding-circle test settles the run (S2: every run) --, one
        # staged entry (22) per distinct landmark of the scan out; per pose: pose 24, rotation pair 16, run / beam / plan
        # offsets 12, counts 12
        "k_assoc_runs": runs * 24 + E * 22 + nloc * (24 + 16 + 12 + 12),
        "k_associate_brute": nnz * (16 + 4) + nloc * 32,
        # staged entries in (22); out: key 4, id 4, k 4, world sums 32, rotated mean offset 16;
        # per pose: pose 24, second moments 24, scatter 24, offsets 8
        "k_compact": E * 22 + E * 60 + nloc * 80,
        "radix_sort_pairs": E * 8 * 2 * 2,
        "k_lm_bounds": (L + 1) * 4,
        "k_lm_scan_totals": E * (4 + 32) + L * (8 + 24),
        "k_stats_prefix": L * (24 + 24 + 24),
        # sorted entry ids 4 + gathered world-sum record 32 in; target 16 per entry out; raw map out
        "k_lm_scan": E * (4 + 32 + 16) + L * (8 + 24),
        "k_beam_targets": nnz * (4 + 16) + E * 16,
        # per entry: k 4, rotated mean 16, target 16; per pose: pose 24, scatter 24, 17 moments out
        # hierarchical pipeline: staged k + sums 18, prefix + record slot 20, record prefix 24 (L2-resident gather)
        "k_pose_moments": (E * (18 + 20 + 24) if hier else E * 36) + nloc * (24 + 24 + 136 + 8),
        # both colours together, per pose: 17 moments, own + 2 neighbour poses, odometry 72, u 32, pose out 24, beam offsets 8;
        # trigonometry tables: odometry pairs 32 + own and lower neighbour's pairs 32 in, rotation pair + (cos, sin) pair 32 out
        "k_solve": nloc * (136 + 72 + 72 + 32 + 24 + 8 + 96),
        "k_scan": nloc * 16,
        "k_pose_rot": nloc * (24 + 16),
    }
    return per_sweep.get(kernel, 0) / max(nlaunch, 1)


def survey_bytes_per_pose(B, Kt, K, T):
    """whole-sweep compulsory bytes per pose update."""
    return 8 * B + 88 + 64 * Kt + 40.0 * K / T


# -------------------------------------------------------------------------------------------------
# CPU baselines (the ONLY place bench.py touches oracle/): timed on the host cores, outside the
# timed GPU region, rank 0 at N = 1 only
# -------------------------------------------------------------------------------------------------
def cpu_baseline_numpy(wl, cfg, n_pose, schedule):
    """The NumPy oracle (the reference's own arithmetic, library for library) on one core, on
    the first n_pose poses of the same workload at the full landmark count."""
    from oracle import icm_oracle as o
    ocfg = o.OracleConfig.from_config(cfg)
    scans = np.ascontiguousarray(wl.scans[:n_pose].T)
    u, odo = wl.u[:, :n_pose], wl.odometry[:, :n_pose]
    st = o.MapState(ocfg, wl.K)
    x = np.ascontiguousarray(wl.x_init[:, :n_pose]).copy()
    kept = o.prefilter_all(scans, ocfg)
    t1 = time.perf_counter()
    try:
        o.sweep(ocfg, st, scans, u, odo, wl.x0, wl.map_init.copy(), x, schedule=schedule, kept=kept)
    except ValueError:
        pass  # a short prefix may leave no landmark above `cota`; the sweep work is done by then
    return (n_pose - 1) / (time.perf_counter() - t1)


def cpu_baseline_c(wl, cfg, n_pose, schedule, threads):
    """The compiled C restatement of the same algorithm (oracle/icm_oracle_c.c, gcc -O2):
    literal (brute-force) cdist/argmin association, per-beam energy, SciPy's Nelder-Mead -- what
    the reference's sweep costs without the Python interpreter.  threads = 1: the reference is
    single-threaded."""
    from oracle import c_oracle as co
    scans = np.ascontiguousarray(wl.scans[:n_pose].T)
    u, odo = np.ascontiguousarray(wl.u[:, :n_pose]), np.ascontiguousarray(wl.odometry[:, :n_pose])
    kept = co.prefilter(cfg, scans)
    x = np.ascontiguousarray(wl.x_init[:, :n_pose]).copy()
    co.set_threads(threads)
    co.set_grid(False)
    t1 = time.perf_counter()
    try:
        co.sweep(cfg, kept, u, odo, wl.x0, wl.map_init, x, wl.K, schedule)
    except ValueError:
        pass
    dt = time.perf_counter() - t1
    co.set_grid(True)
    co.set_threads(0)
    return (n_pose - 1) / dt


# -------------------------------------------------------------------------------------------------
# launcher: `python bench.py --gpus N` from a plain shell starts the ranks itself
# -------------------------------------------------------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(args, argv):
    """Parent of an N-rank job.  This process never touches the GPU (no torch.cuda call, no HIP
    library loaded): it starts `python -m torch.distributed.run` as a CHILD process, which starts
    one fresh rank process per GPU, relays rank 0's JSON line and exits with the job's code."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC (RCCL across processes on this driver)
    env.setdefault("OMP_NUM_THREADS", "4")
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in proc.stdout:
        if ln.startswith('{"metric"'):
            line = ln.strip()
        else:
            sys.stderr.write(ln)
    rc = proc.wait()
    if line is not None:
        print(line, flush=True)
    if rc != 0 or line is None:
        raise SystemExit(rc if rc else 1)


# -------------------------------------------------------------------------------------------------
# one rank
# -------------------------------------------------------------------------------------------------
class Job:
    """One workload resident on this rank's GPU and the step that sweeps it."""

    def __init__(self, args, name, T, K, B, rank, world, local_rank, sharded, dist, factory=None):
        from ICM_SLAM_tools import ConfigICM
        from icmslam_hip.synthetic import make_workload
        self.args, self.dist, self.world, self.rank, self.sharded = args, dist, world, rank, sharded
        self.T, self.K, self.B = T, K, B
        from icmslam_hip.sharded import shard_block
        self.blk = shard_block(T, world)          # ceil(T / world) rounded up to an even number of poses
        self.t_begin, self.t_end = min(rank * self.blk, T), min((rank + 1) * self.blk, T)
        t0 = time.perf_counter()
        self.wl = wl = make_workload(T, K, B, t_begin=self.t_begin, t_end=self.t_end)
        self.t_gen = time.perf_counter() - t0
        self.cfg = cfg = ConfigICM(D=dict(wl.config, schedule="redblack"))
        t0 = time.perf_counter()
        if factory is not None:     # test hook (tests/test_bench_launcher.py): an engine double with the phase API
            self.eng = eng = factory(cfg, wl, rank, world, self.t_begin, self.t_end)
        else:
            from icmslam_hip import SweepEngine
            self.eng = eng = SweepEngine(cfg, local_rank)
            eng.upload(wl.scans, wl.odometry, wl.u, t_begin=self.t_begin, t_end=self.t_end, pose_major=True,
                       ghost_scan=wl.ghost_scan if (sharded and rank > 0) else None)
        self.t_upload = time.perf_counter() - t0
        if sharded:
            from icmslam_hip.sharded import LibrarySweep, ShardedSweep
            if args.collectives == "library" and factory is None:
                self.runner = LibrarySweep(eng, rank, world, T)      # RCCL all-gathers issued by the C library
            else:
                self.runner = ShardedSweep(eng, rank, world, T)      # torch.distributed all_gather_into_tensor
            self.runner.set_state(wl.map_init, wl.x_init, wl.x0)
            self.inner = lambda: self.runner.sweep("redblack")
        else:
            eng.set_state(wl.map_init, wl.x_init, wl.x0)
            self.inner = lambda: eng.sweep_device("redblack")
        self.nsweeps = 0
        self.rewind = False

    def enable_rewind(self):
        """ICM on S2 is not a contraction: after ~50 consecutive sweeps poses
        leave the 1 m association gate.  Runs longer than MAX_SWEEPS_WITHOUT_REWIND sweeps rewind
        the state on the device every RESET_EVERY sweeps (inside the timed region, stated in
        config.state_rewind); the default run does not."""
        self.eng.snapshot_state()
        self.rewind = True

    def step(self):
        if self.rewind and self.nsweeps and self.nsweeps % RESET_EVERY == 0:
            self.eng.restore_state()
        self.inner()
        self.nsweeps += 1

    def fence(self):
        if self.args.device == "cuda":
            import torch
            torch.cuda.synchronize()
        if self.dist is not None:
            self.dist.barrier()
            if self.args.device == "cuda":
                import torch
                torch.cuda.synchronize()

    def window(self, steps):
        """Exactly `steps` sweeps between two (barrier + device synchronise) fences; the MAX over ranks of the
        elapsed time."""
        self.fence()
        t0 = time.perf_counter()
        for _ in range(steps):
            self.step()
        self.fence()
        elapsed = time.perf_counter() - t0
        if self.dist is not None:
            import torch
            tt = torch.tensor([elapsed], dtype=torch.float64, device=self.args.device)
            self.dist.all_reduce(tt, op=self.dist.ReduceOp.MAX)
            elapsed = float(tt.item())
        return elapsed

    def timed(self, steps, warmup, windows=1):
        """W untimed sweeps, then `windows` timed windows of exactly K sweeps each.  Every window sweeps the SAME
        states: the state after the warm-up is snapshotted on the device and put back (device-to-device copies,
        outside the timed regions) before each further window, followed by one untimed sweep so that every window
        starts with the steady-state launch sequence.  Returns the list of elapsed times."""
        for _ in range(warmup):
            self.step()
        can_rewind = windows > 1 and hasattr(self.eng, "snapshot_state") and not self.rewind
        if can_rewind:
            self.fence()
            self.eng.snapshot_state()
            self.step()                       # (the same untimed sweep precedes every window)
        out = [self.window(steps)]
        for _ in range(windows - 1 if can_rewind else 0):
            self.eng.restore_state()
            self.step()
            out.append(self.window(steps))
        return out

    def phase_report(self, sweeps=3):
        """After the timed windows of a sharded job: a few more sweeps with events at the phase boundaries
        (icm_set_phase_timing), one line per rank on stderr -- local phase A / the exchange including the wait for the
        slowest rank / targets, ghost pose and moments / the solve launch / host wait at the end -- so that a scaling
        record can be read: which phase, on which rank, is the sweep waiting for."""
        if not self.sharded or not hasattr(self.eng, "set_phase_timing"):
            return
        self.fence()
        self.eng.set_phase_timing(True)
        for _ in range(sweeps):
            self.step()
        self.fence()
        ph, n = self.eng.phase_times()
        self.eng.set_phase_timing(False)
        sys.stderr.write("bench.py phases (ms per sweep, %d diagnostic sweeps) rank %d/%d poses [%d, %d): %s\n"
                         % (n, self.rank, self.world, self.t_begin, self.t_end, json.dumps(ph)))
        sys.stderr.flush()

    def close(self):
        # a sharded job ends with the closing exchange (every rank, same place): where a rank that failed behind the last
        # sweep's exchange delivers its error instead of leaving its peers waiting (icm_sharded_end)
        if getattr(self, "sharded", False) and hasattr(getattr(self, "runner", None), "end"):
            self.runner.end()
        if hasattr(getattr(self, "runner", None), "close"):
            self.runner.close()
        if hasattr(self.eng, "close"):
            self.eng.close()


def choose_collectives(args, rank, world, local_rank, dist):
    """Who issues the all-gather of a sharded sweep.  "library": the C library itself (RCCL on the
    handle's stream, one C call per sweep) -- taken when every rank resolves an RCCL library AND three
    sweeps of the `tiny` workload give bit for bit the state the torch.distributed path gives;
    otherwise "torch" (all_gather_into_tensor from Python).  Both are the product path."""
    import numpy as np
    import torch
    from ICM_SLAM_tools import ConfigICM
    from icmslam_hip import SweepEngine
    from icmslam_hip.sharded import LibrarySweep, ShardedSweep, partition
    from icmslam_hip.synthetic import WORKLOADS, make_workload

    def agree(flag):
        t = torch.tensor([1.0 if flag else 0.0], dtype=torch.float64, device=args.device)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return bool(t.item() > 0.5)

    probe = SweepEngine(ConfigICM(D=make_workload(*WORKLOADS["tiny"]).config), local_rank)
    ok = agree(probe.comm_available())
    probe.close()
    if not ok:
        return "torch", "no RCCL library resolved on some rank"
    wl = make_workload(*WORKLOADS["tiny"])
    cfg = ConfigICM(D=dict(wl.config, schedule="redblack"))
    _, parts = partition(wl.T, world)
    a, b = parts[rank]

    def three_sweeps(kind):
        eng = SweepEngine(cfg, local_rank)
        try:
            eng.upload(wl.scans[a:b], wl.odometry, wl.u, t_begin=a, t_end=b, pose_major=True, ghost_scan=wl.scans[a - 1] if a else None)
            run = kind(eng, rank, world, wl.T)
            try:
                run.set_state(wl.map_init, wl.x_init, wl.x0)
                for _ in range(3):
                    run.sweep("redblack")
                return run.get_state()
            finally:
                if hasattr(run, "close"):
                    run.close()
        finally:
            eng.close()

    # a library communicator that cannot be made (an RCCL build the wheel's torch does not share, a refused
    # ncclCommInitRank) must not end the job: every rank then takes the torch path
    states, why = [], ""
    try:
        states.append(three_sweeps(LibrarySweep))
    except Exception as e:      # noqa: BLE001 -- whatever the library reports, the answer is "torch"
        why = "%s: %s" % (type(e).__name__, e)
    if not agree(len(states) == 1):
        return "torch", "library collectives did not start on some rank" + (" (%s)" % why if why else "")
    states.append(three_sweeps(ShardedSweep))
    same = states[0][3] == states[1][3] and all(np.array_equal(p, q) for p, q in zip(states[0][:3], states[1][:3]))
    if agree(same):
        return "library", "validated against the torch.distributed path on the tiny workload (3 sweeps, bit-identical)"
    return "torch", "library collectives disagreed with torch.distributed on the tiny workload"


def pmc_traffic(workload, kernel):
    """HBM bytes per launch of `kernel` from the committed PMC summary (profiles/traffic.json), and the file it was
    made from; (None, reason) without one -- or when the summary was made with ANOTHER build of the library than the one
    loaded (the file is stamped with icm_build_id(): counter traffic of kernels that have changed since says nothing).
    Only for a one-GPU run of the full workload (that is what was profiled)."""
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
        val, src = int(tj[workload][kernel]), tj.get("_source", {}).get(workload)
        from icmslam_hip import _lib
        have, want = _lib.load().icm_build_id().decode(), (tj.get("_build_id") or {}).get(workload)
        if want != have:
            return None, "stale: %s was profiled on build %s, this is build %s" % (src, want, have)
        return val, src
    except (OSError, KeyError, ValueError, TypeError, AttributeError):
        return None, None


def roofline(job, ms_per_step):
    """Per-kernel launch times from HIP events on the launch stream (icm_enable_timing), over
    three more sweeps; achieved = algorithmic bytes (or flops) per launch / average launch time."""
    eng, args = job.eng, job.args
    eng.enable_timing(True)
    nroof = 3
    for _ in range(nroof):
        job.inner()
    job.fence()
    st = eng.last_stats()      # (after the timing sweeps: they run the entry-offset scan, so the entry count is theirs)
    kt = {k: v for k, v in eng.kernel_times().items() if v[1] > 0 and k != "k_prefilter"}
    hier = eng.entry_path() == "hier"
    eng.enable_timing(False)
    # k_filtrar_* run on a side stream, concurrent with the solves: not on the critical path
    side = tuple(k for k in kt if k.startswith("k_filtrar") or k == "k_neigh_table")
    dom = max((k for k in kt if k not in side), key=lambda k: kt[k][0])

    def ab_of(k, n_k):
        return algorithmic_bytes(k, st["kept_beams"], st["entries"], eng.nloc, eng.L, n_k / nroof, hier, runs=eng.run_counts()[0])

    per_kernel = {}
    for k, (ms_k, n_k) in kt.items():
        ab = ab_of(k, n_k)
        per_kernel[k] = {"ms_per_launch": round(ms_k / n_k, 4), "launches_per_sweep": n_k / nroof,
                         "GBps": round(ab / (ms_k / n_k * 1e-3) / 1e9, 1) if ab else None}

    def hbm_roof(kname):
        ms_k, n_k = kt[kname]
        ab = ab_of(kname, n_k)
        a_gbs = ab / (ms_k / n_k * 1e-3) / 1e9
        # `traffic` is a PMC quantity (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, tools/profile.sh): it cannot
        # be collected from inside this process.  The number reported is the one the committed PMC summary of this
        # kernel holds (profiles/traffic.json, written by tools/pmc_summary.py --traffic from the same workload's
        # passes; `traffic_source` names the summary file it came from), null when there is none for this workload.
        tr, src = pmc_traffic(args.workload, kname)
        return {"bound": "hbm", "kernel": kname, "achieved": round(a_gbs, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(a_gbs / HBM_PEAK_GBS, 5), "traffic": tr, "traffic_source": src,
                "avg_launch_ms": round(ms_k / n_k, 4), "launches_per_sweep": n_k / nroof,
                "algorithmic_bytes_per_launch": int(ab)}

    ms, n = kt[dom]
    avg_ms, nl = ms / n, n / nroof
    stream_kernel = max((k for k in kt if k not in side + ("k_solve",)), key=lambda k: kt[k][0])
    # The solves are not a bandwidth kernel: one lane per pose runs ~80 dependent energy evaluations from
    # registers.  Their yardstick is the FP64 vector rate; flops = (energy evaluations counted by the kernel
    # itself) x (FP64 flops of one evaluation, counted from the kernel's ISA at build time:
    # tools/count_eval_flops.py -> icm_flop_per_eval()).  Reported whichever kernel dominates.
    solve_rec = None
    if "k_solve" in kt:
        ms_s, n_s = kt["k_solve"]
        eng.set_debug(True)
        if hier:
            eng.set_entry_path("hier")   # (debug alone would select the sort-based pipeline)
        job.inner()
        job.fence()
        nfev = float(eng.solve_diag()[:, 2].sum())
        eng.set_debug(False)
        eng.set_entry_path("auto")
        fpe = eng.flop_per_eval()
        nl_s = n_s / nroof
        tfl = nfev * fpe / nl_s / (ms_s / n_s * 1e-3) / 1e12
        solve_rec = {"bound": "valu_fp64", "kernel": "k_solve", "achieved": round(tfl, 3), "peak": FP64_VALU_PEAK_TFLOPS,
                     "unit": "TFLOP/s", "frac": round(tfl / FP64_VALU_PEAK_TFLOPS, 5), "traffic": pmc_traffic(args.workload, "k_solve")[0],
                     "traffic_source": pmc_traffic(args.workload, "k_solve")[1],
                     "avg_launch_ms": round(ms_s / n_s, 4), "launches_per_sweep": nl_s, "energy_evaluations_per_sweep": int(nfev),
                     "flop_per_evaluation": fpe, "flop_per_evaluation_source": "FP64 VALU instructions of one energy evaluation in the built kernel's ISA (fma = 2)",
                     "algorithmic_bytes_per_launch": int(ab_of("k_solve", n_s)),
                     "note": "latency-bound: the launch lasts as long as its slowest dependent pair of poses (odd pose, then its even neighbour), "
                             "each a serial Nelder-Mead chain; the folded energy needs 55 flops per evaluation where the term-by-term form needed 129"}
    roof = dict(solve_rec) if dom == "k_solve" else hbm_roof(dom)
    if solve_rec is not None and dom != "k_solve":
        roof["solve_kernel"] = solve_rec
    roof["hbm_stream_kernel"] = hbm_roof(stream_kernel)
    roof["kernels_ms_per_sweep"] = {k: round(v[0] / nroof, 4) for k, v in sorted(kt.items(), key=lambda kv: -kv[1][0])}
    roof["kernels"] = per_kernel
    roof["side_stream_kernels"] = list(side)
    # The per-kernel pass brackets every launch with events and waits for each (icm_enable_timing), so it runs the
    # sweep's CAREFUL schedule: the entry-offset scan (k_scan, two launches) is part of it.  The timed steps above are
    # queued whole and skip those two launches from the second sweep on (k_scan row).
    if "k_scan" in kt:
        roof["timing_pass_only_kernels"] = ["k_scan"]
    Kt = st["entries"] / max(eng.nloc, 1)
    sweep_bytes = survey_bytes_per_pose(job.B, Kt, job.K, job.T) * (job.T - 1)
    roof["sweep_algorithmic_GBps"] = round(sweep_bytes / (ms_per_step * 1e-3) / 1e9, 2)
    roof["sweep_frac_of_hbm_peak"] = round(roof["sweep_algorithmic_GBps"] / (HBM_PEAK_GBS * job.world), 5)
    return roof


def _assoc_record(eng):
    """What phase A associated: the form, the runs of the sequence and how many of them went beam by beam."""
    if not hasattr(eng, "run_counts"):
        return None
    runs, bbb = eng.run_counts()
    return {"form": "geometric runs of each scan's kept beams, cut once per sequence; a run whose bounding circle settles the argmin and "
                    "the gate for all its beams takes the label from its centre, the others go beam by beam (same labels as the "
                    "reference's per-beam rule, scripts/ICM_SLAM_tools.py:168-172)",
            "runs": runs, "runs_taken_beam_by_beam_so_far": bbb}


def dropin_record(job, steps, warmup):
    """The reference's own call: mapa_refinado, x = ICM.iterations_process_offline(mapa_viejo, x)
    (scripts/ICM_ROS.py:298-311 driver loop), host NumPy arrays in and out every sweep -- PCIe
    included, so this is never `value`."""
    import torch
    from copy import deepcopy as copy
    from ICM_ROS import ICM_ROS
    wl = job.wl
    icm = ICM_ROS(job.cfg)
    icm.attach_engine(job.eng, wl.scans.T, wl.odometry, wl.u)   # the sequence is already in HBM
    icm.x0 = wl.x0.reshape(3, 1)
    icm.set_initial_state(wl.x_init, wl.map_init)
    mapa_viejo, x = copy(icm.mapa_viejo), copy(icm.positions)
    for _ in range(warmup):
        mapa_refinado, x = icm.iterations_process_offline(mapa_viejo, x)
        mapa_viejo = copy(mapa_refinado)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        mapa_refinado, x = icm.iterations_process_offline(mapa_viejo, x)
        mapa_viejo = copy(mapa_refinado)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    return {"call": "ICM_ROS.iterations_process_offline (host arrays in/out, PCIe inclusive)", "steps": steps,
            "ms_per_step": round(1e3 * el / steps, 4), "value": round((job.T - 1) * steps / el, 1), "unit": "pose-updates/s"}


def dataset_cpu_baseline_c(icm, init):
    """The compiled C oracle (oracle/icm_oracle_c.c, gcc -O2, ONE thread) on data_IJAC2018 on THIS box's host: the
    initialisation pass, one sweep in the reference's order and one red-black sweep -- what the same work costs on a
    CPU core without the Python interpreter, beside the GPU's numbers for the same three jobs."""
    from oracle import c_oracle as co
    from oracle import icm_oracle as o
    cfg = icm.config
    zz, odo, u = icm.mediciones, icm.odometria, icm.u
    T = zz.shape[1]
    kept = co.prefilter(cfg, zz)
    co.set_threads(1)
    co.set_grid(False)   # (11 landmarks: the literal scan over all of them, like the reference)

    def med3(f):
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            f()
            ts.append(time.perf_counter() - t0)
        return sorted(ts)[1]

    rec = {"cores": 1, "kind": "port", "host_cores": os.cpu_count(),
           "what": "compiled C oracle, one thread, literal brute-force association and per-beam energy, SciPy's Nelder-Mead restated; median of 3"}
    lact = int(init["landmarks_actuales"])
    for sched in ("sequential", "redblack"):
        def one():
            x = np.ascontiguousarray(init["x_init"]).copy()
            co.sweep(cfg, kept, u, odo, odo[:, 0], init["map_init"], x, lact, sched)
        t = med3(one)
        rec[sched] = {"ms": round(1e3 * t, 3), "pose_updates_per_s": round((T - 1) / t, 1)}
    # the init pass from the first scan's clusters (their SciPy linkage is host work on both sides: not timed here)
    ocfg = o.OracleConfig.from_config(cfg)
    st = o.MapState(ocfg)
    k0 = o.filtrar_z(zz[:, 0], ocfg)
    y0, _ = o.cluster_first_scan(st, np.zeros((2, ocfg.L)), o.project_beams(odo[:, 0].copy(), k0[:, 2:4]))
    t = med3(lambda: co.init_pass(cfg, kept, u, odo, y0, st.cant_obs_i, st.landmarks_actuales))
    rec["init_pass"] = {"ms": round(1e3 * t, 3)}
    co.set_grid(True)
    co.set_threads(0)
    return rec


def dataset_record(device):
    """data_IJAC2018 (BASELINE.json configs[0] / [1]: 1833 poses x 181 beams, config_default.yaml) -- the ONE input the real
    reference was ever timed on (in the build container; its Python never travels to the GPU box):
    the initialisation pass (scripts/ICM_ROS.py:102-119 -> icm_init_pass), one sweep in the reference's sequential order and
    one red-black sweep (scripts/ICM_ROS.py:121-164), each from the state the init pass leaves (fixtures under tests/golden/),
    resident state, median of five; and the reference's own call (host arrays in and out) in the reference's order."""
    import torch
    from copy import deepcopy as copy
    from ICM_ROS import ICM_ROS
    from ICM_SLAM_tools import ConfigICM
    gold = os.path.join(ROOT, "tests", "golden")
    f_data, f_init = os.path.join(gold, "data_IJAC2018.npz"), os.path.join(gold, "init_pass.npz")
    if not (os.path.exists(f_data) and os.path.exists(f_init)):
        return None
    cfg = ConfigICM("config_default.yaml")
    icm = ICM_ROS(cfg)
    icm.device = device
    icm.load_data(f_data)
    init = np.load(f_init)
    x0 = icm.odometria[:, 0]
    eng = icm._get_engine()
    T = int(icm.mediciones.shape[1])

    def med(f, n=5):
        ts = []
        for _ in range(n):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            f()
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        return sorted(ts)[len(ts) // 2]

    t_init = med(lambda: eng.init_pass(x0), 3)
    solves_init = int((np.diff(eng.kept_beams()[0])[1:] > 0).sum())
    out = {"workload": "data_IJAC2018: %d poses x %d beams, config_default.yaml (BASELINE.json configs[0]/[1])" % (T, icm.mediciones.shape[0]),
           "init_pass": {"ms": round(1e3 * t_init, 3), "solves": solves_init, "solves_per_s": round(solves_init / t_init, 1),
                         "what": "icm_init_pass incl. the first scan's clustering on the host and all host <-> device copies"}}
    lact = int(init["landmarks_actuales"])
    for sched in ("sequential", "redblack"):
        def one():
            eng.sweep_device(sched)
        ts = []
        for _ in range(5):
            eng.set_state(init["map_init"], init["x_init"], x0, lact)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            one()
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        t = sorted(ts)[2]
        out[sched] = {"ms": round(1e3 * t, 4), "pose_updates_per_s": round((T - 1) / t, 1),
                      "what": "one sweep from (map_init, x_init), state resident in HBM, %s order" % ("the reference's" if sched == "sequential" else "red-black")}
    icm.set_initial_state(init["x_init"], init["map_init"], init["cant_obs_i"])
    st = {"m": copy(icm.mapa_viejo), "x": copy(icm.positions)}

    def call():
        icm.set_initial_state(init["x_init"], init["map_init"], init["cant_obs_i"])
        st["x"][...] = init["x_init"]
        icm.iterations_process_offline(st["m"], st["x"])
    t_call = med(call)
    out["reference_call"] = {"ms": round(1e3 * t_call, 4), "pose_updates_per_s": round((T - 1) / t_call, 1),
                             "what": "ICM_ROS.iterations_process_offline(mapa_viejo, x), host arrays in and out, schedule = config_default.yaml's (sequential)"}
    ref = {"sweep_s": [15.1, 17.5], "pose_updates_per_s": [105, 121], "init_pass_s": [9.3, 11.9], "cores": 1,
           "provenance": "BASELINE.md section 2: the reference's own Python (imported under a roslibpy stub, ROS-free harness) timed in the "
                         "build container on ONE core of an 8-core Intel Xeon @ 2.10 GHz, NumPy 2.2.6 / SciPy 1.15.3; not a number "
                         "published by the reference and not measured on this box"}
    out["reference_measured"] = ref
    out["cpu_baseline_c"] = dataset_cpu_baseline_c(icm, init)
    cb = out["cpu_baseline_c"]
    out["gpu_over_cpu_c_one_core"] = {"sweep_sequential": round(cb["sequential"]["ms"] / out["sequential"]["ms"], 2),
                                      "sweep_redblack": round(cb["redblack"]["ms"] / out["redblack"]["ms"], 1),
                                      "init_pass": round(cb["init_pass"]["ms"] / out["init_pass"]["ms"], 2)}
    out["speedup_vs_reference_measured"] = {
        "sweep_sequential": [round(ref["sweep_s"][0] / (out["sequential"]["ms"] * 1e-3), 0), round(ref["sweep_s"][1] / (out["sequential"]["ms"] * 1e-3), 0)],
        "sweep_redblack": [round(ref["sweep_s"][0] / (out["redblack"]["ms"] * 1e-3), 0), round(ref["sweep_s"][1] / (out["redblack"]["ms"] * 1e-3), 0)],
        "init_pass": [round(ref["init_pass_s"][0] / t_init, 0), round(ref["init_pass_s"][1] / t_init, 0)]}
    eng.close()
    return out


def run_rank(args):
    from icmslam_hip.synthetic import WORKLOADS
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    factory = None
    if args.engine_factory:       # test hook: "module:function" building an engine double (CPU, gloo)
        import importlib
        mod, fn = args.engine_factory.split(":")
        factory = getattr(importlib.import_module(mod), fn)
    dist = None
    sharded = world > 1 or args.force_sharded
    json_fd = None
    if sharded:
        # RCCL prints a version banner on file descriptor 1 when a communicator is made: keep this
        # rank's stdout for the ONE JSON line, everything else goes to stderr
        sys.stdout.flush()
        json_fd = os.dup(1)
        os.dup2(2, 1)
        import torch
        import torch.distributed as dist
        if "MASTER_ADDR" not in os.environ:  # plain `python bench.py --force-sharded`
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1")
        if args.device == "cuda":
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo")
    ranks_seen = 1
    if dist is not None:
        import torch
        one = torch.ones(1, dtype=torch.float64, device=args.device)
        dist.all_reduce(one)
        ranks_seen = int(one.item())
        assert ranks_seen == dist.get_world_size() == world

    collectives_note = None
    if sharded and factory is None and args.collectives == "auto":
        args.collectives, collectives_note = choose_collectives(args, rank, world, local_rank, dist)
    T1, K, B = WORKLOADS[args.workload]
    modes = [args.scaling] if world == 1 else [args.scaling, "weak" if args.scaling == "strong" else "strong"]
    records = {}
    main_job = None
    for mode in modes:
        T = T1 * world if mode == "weak" else T1
        job = Job(args, args.workload, T, K, B, rank, world, local_rank, sharded, dist, factory)
        total = args.steps + args.warmup
        if total > MAX_SWEEPS_WITHOUT_REWIND and not args.no_rewind and factory is None:
            job.enable_rewind()
        # three timed windows of K sweeps over the same states; the record is the MEDIAN window, the others are listed
        wins = sorted(job.timed(args.steps, args.warmup, windows=args.windows))
        elapsed = wins[len(wins) // 2]
        ms = 1e3 * elapsed / max(args.steps, 1)
        rec = {"value": round((T - 1) * args.steps / elapsed, 1), "ms_per_step": round(ms, 4),
               "ms_per_step_windows": {"n": len(wins), "min": round(1e3 * wins[0] / max(args.steps, 1), 4),
                                       "median": round(ms, 4), "max": round(1e3 * wins[-1] / max(args.steps, 1), 4)},
               "poses": T,
               "poses_per_gpu": job.blk, "scaling": mode,
               "workload": ("%s: synthetic %d poses / %d landmarks / %d beams, red-black ICM sweep" % (args.workload, T, K, B))
               + ("" if world == 1 else
                  (" -- ONE sequence of %d x %d poses, a %d-pose block per GPU" % (world, T1, job.blk) if mode == "weak"
                   else " -- the %d-pose sequence split into %d-pose blocks (BASELINE.json configs[4])" % (T, job.blk)))}
        records[mode] = rec
        try:
            job.phase_report()
        except Exception as e:      # (diagnostics only)
            sys.stderr.write("bench.py: phase report failed on rank %d: %s\n" % (rank, e))
        if mode == args.scaling:
            main_job, main_ms = job, ms
        else:
            job.close()
    job = main_job
    rec = records[args.scaling]
    st = job.eng.last_stats() if hasattr(job.eng, "last_stats") else {"kept_beams": None}
    out = {
        "metric": "ICM pose-updates/sec (full sweep)", "value": rec["value"], "unit": "pose-updates/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": rec["ms_per_step"],
        "ms_per_step_windows": rec["ms_per_step_windows"],
        "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": rec["workload"], "schedule": "redblack", "poses": rec["poses"], "poses_per_gpu": rec["poses_per_gpu"],
                   "landmarks": K, "beams": B, "kept_beams": st["kept_beams"] if world == 1 else None,
                   "parallelism": "pose-shard x%d" % world,
                   "entry_pipeline": job.eng.entry_path() if hasattr(job.eng, "entry_path") else None,
                   "association": _assoc_record(job.eng) if world == 1 else "geometric runs (bounding-circle test, beam by beam where it does not settle)",
                   "fixup_poses": job.eng.fixup_poses() if hasattr(job.eng, "fixup_poses") else None,
                   "state_rewind": ("initial state restored on the device every %d sweeps" % RESET_EVERY) if job.rewind
                   else "none inside a timed window: %d consecutive sweeps from the initial state per window" % (args.steps + args.warmup + 1),
                   "collectives_per_sweep": 0 if not sharded else "1 all-gather of [3L+16] f64 per rank (landmark statistics + the shard's boundary poses); no halo exchange (ghost pose)",
                   "collectives_issued_by": None if not sharded else
                   ("C library (ncclAllGather on the handle's stream)" if args.collectives == "library" and factory is None
                    else "torch.distributed (all_gather_into_tensor)") + ((": " + collectives_note) if collectives_note else "")},
        "ranks_seen": ranks_seen,
        "setup_s": {"generate": round(job.t_gen, 2), "upload_and_prefilter": round(job.t_upload, 2)},
    }
    for mode, r in records.items():
        if mode != args.scaling:
            out[mode] = r
    if factory is None and not args.no_roofline:
        out["roofline"] = roofline(job, main_ms)
    if rank == 0 and world == 1 and factory is None and not args.no_extras:
        out["dropin"] = dropin_record(job, max(args.steps // 2, 5), 2)
    wl, cfg = job.wl, job.cfg
    if rank == 0 and world == 1 and factory is None and args.cpu_poses != 0:
        n_c = args.cpu_poses if args.cpu_poses > 0 else {"S2": 2500, "S1": 10000}.get(args.workload, job.T)
        n_c = min(n_c, job.t_end)
        v_c = cpu_baseline_c(wl, cfg, n_c, "redblack", 1)
        out["cpu_baseline"] = {"value": round(v_c, 1), "unit": "pose-updates/s", "cores": 1, "kind": "port",
                               "sample": "compiled C oracle (oracle/icm_oracle_c.c, gcc -O2, one thread; literal brute-force association "
                                         "and per-beam energy like the reference), one red-black sweep over the first %d poses of "
                                         "the same sequence at the full %d-landmark map; host has %d cores" % (n_c, K, os.cpu_count())}
        n_py = min({"S2": 300, "S1": 1000}.get(args.workload, job.T), job.t_end)
        v_py = cpu_baseline_numpy(wl, cfg, n_py, "redblack")
        out["cpu_baseline_numpy"] = {"value": round(v_py, 2), "unit": "pose-updates/s", "cores": 1, "kind": "port",
                                     "sample": "NumPy oracle (oracle/icm_oracle.py: the reference's arithmetic library for library, "
                                               "interpreter included), first %d poses" % n_py}
        out["gpu_over_cpu"] = {"vs_c_port": round(out["value"] / v_c, 1), "vs_numpy_port": round(out["value"] / v_py, 1)}
    job.close()
    if rank == 0 and world == 1 and factory is None and not args.no_extras and args.workload == "S2":
        # BASELINE.json configs[2]: S1 (10k poses / 1k landmarks / 360 beams), 20 consecutive ICM iterations
        j3 = Job(args, "S1", *WORKLOADS["S1"], rank, world, local_rank, False, None)
        el = j3.timed(20, 2)[0]
        out["config3"] = {"workload": "S1: synthetic 10000 poses / 1000 landmarks / 360 beams, 20 consecutive red-black sweeps (BASELINE.json configs[2])",
                          "steps": 20, "warmup": 2, "ms_per_step": round(1e3 * el / 20, 4), "value": round((j3.T - 1) * 20 / el, 1),
                          "unit": "pose-updates/s"}
        j3.close()
    if rank == 0 and world == 1 and factory is None and not args.no_extras:
        try:
            ds = dataset_record(local_rank)
        except Exception as e:   # (a record beside the headline, never a reason to lose the line)
            ds = {"error": "%s: %s" % (type(e).__name__, e)}
        if ds is not None:
            out["dataset"] = ds
    if rank == 0:
        if json_fd is not None:
            sys.stdout.flush()
            os.write(json_fd, (json.dumps(out) + "\n").encode())
        else:
            print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="S2", help="S2 (BASELINE metric config), S1, tiny")
    ap.add_argument("--cpu-poses", type=int, default=-1, help="prefix length of the CPU baseline (0 = skip)")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the drop-in and config-3 records")
    ap.add_argument("--windows", type=int, default=3, help="timed windows of --steps sweeps each over the same states (median reported)")
    ap.add_argument("--no-rewind", action="store_true", help="never rewind the state, however many sweeps")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="strong",
                    help="N > 1: strong = the workload's sequence split N ways (BASELINE configs[4], the headline); "
                         "weak = the sequence has N x the workload's poses (fixed work per GPU); the other one is "
                         "reported as a secondary record")
    ap.add_argument("--collectives", choices=("auto", "library", "torch"), default="auto",
                    help="sharded runs: all-gathers issued by the C library (RCCL) or by torch.distributed; auto = library when it validates")
    ap.add_argument("--force-sharded", action="store_true",
                    help="drive even a 1-rank run through the sharded path (torch.distributed + RCCL all-gathers)")
    ap.add_argument("--device", choices=("cuda", "cpu"), default="cuda", help="cpu: launcher tests only (with --engine-factory)")
    ap.add_argument("--engine-factory", default="", help="test hook: module:function returning an engine double")
    args = ap.parse_args()
    if args.device == "cpu" and not args.engine_factory:
        raise SystemExit("bench.py: there is no CPU path; --device cpu is for the launcher test with an engine double")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        launch_ranks(args, sys.argv[1:])
        return
    run_rank(args)


if __name__ == "__main__":
    main()
