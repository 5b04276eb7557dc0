#!/usr/bin/env python3
"""Benchmark of the MI355X-native ICM sweep: ICM pose-updates/s over full sweeps.

    python bench.py                      # 1 GPU, workload S2 (100k poses / 10k landmarks / 720 beams)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one full ICM sweep (reference `iterations_process_offline`, scripts/ICM_ROS.py:121-164:
association of every kept beam, running-mean map, one Nelder-Mead solve per pose, map
prune/merge) over the whole synthetic sequence, which is resident in HBM when timing starts.
Poses are solved in the red-black order (the reference's sequential order is one dependent
chain of T-1 solves and is used for parity, not throughput).  With N > 1 ONE sequence is sharded
by contiguous pose blocks over the ranks, with one all-gather of the landmark sufficient
statistics and two 48-byte halo exchanges per sweep.  Default `--scaling weak`: the sequence
grows with the job (N x 100k poses through the same 10k-landmark field; every GPU owns a
100k-pose block, i.e. the per-GPU work is the N = 1 workload).  `--scaling strong` shards the
fixed 100k-pose sequence instead (BASELINE.json configs[4]); one sweep of it is under 1 ms on one
GPU, so that mode is bounded by the serial Nelder-Mead chain and collective latency (DESIGN.md
section 6).

Prints ONE JSON line on rank 0 (contract in the task description) with the two extra objects
`roofline` (dominant kernel, measured with HIP events on the launch stream) and
`cpu_baseline` (the NumPy oracle timed on one host core on a bounded prefix of the same
workload).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "icm-slam_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 achievable
FP64_VALU_PEAK_TFLOPS = 78.6  # 256 CUs x 4 SIMDs x 16 FP64 FMA lanes x 2 x 2.4 GHz (no MFMA on this path)
FLOP_PER_ENERGY_EVAL = 180    # moment-form fun_xn: trig 24, observation quadratic form 67, priors 87


def algorithmic_bytes(kernel, nnz, E, nloc, L, nlaunch, hier=True):
    """Compulsory HBM bytes ONE launch of `kernel` moves: every array
    the kernel must read or write once, no re-reads.  nnz = kept beams, E = (pose, landmark)
    entries, nloc = poses of the shard, L = landmark capacity; hier = the hierarchical entry
    pipeline ran (k_pose_moments then reads the staged entries and the prefixes itself)."""
    ch = 64 if nloc >= 65536 else (32 if nloc >= 16384 else 16)   # poses per chunk (icm_api.hip)
    nchunks = (nloc + ch - 1) // ch
    group = (nchunks + 63) // 64
    nsuper = (nchunks + group - 1) // max(group, 1)
    per_sweep = {
        # staged entry in (label 4, k 4, sum bx 8, sum by 8); prefix (sum x 8, sum y 8, n 4) + record
        # slot (1) out; per pose: offsets 16 + pose 24; per chunk: 256 record labels
        "k_chunk_l1": E * 24 + E * 21 + nloc * 40 + nchunks * 256 * 4,
        # dense [superchunks x L] matrix of (sx, sy, n): read once, written once
        "k_lm_l3": 2 * 3 * 8 * nsuper * L + L * 24,
        # read body x,y of every kept beam (16 B) + pose; write one staged entry
        # (label 4, k 4, sum bx 8, sum by 8) per distinct landmark of the scan; counts/flags
        "k_assoc_group": nnz * 16 + E * 24 + nloc * (24 + 8 + 8),
        "k_associate_brute": nnz * (16 + 4) + nloc * 32,
        # staged entries in (24); out: key 4, id 4, k 4, world sums 32, rotated mean offset 16;
        # per pose: pose 24, second moments 24, scatter 24, offsets 8
        "k_compact": E * 24 + E * 60 + nloc * 80,
        "radix_sort_pairs": E * 8 * 2 * 2,
        "k_lm_bounds": (L + 1) * 4,
        "k_lm_scan_totals": E * (4 + 32) + L * (8 + 24),
        "k_stats_prefix": L * (24 + 24 + 24),
        # sorted entry ids 4 + gathered world-sum record 32 in; target 16 per entry out; raw map out
        "k_lm_scan": E * (4 + 32 + 16) + L * (8 + 24),
        "k_beam_targets": nnz * (4 + 16) + E * 16,
        # per entry: k 4, rotated mean 16, target 16; per pose: pose 24, scatter 24, 17 moments out
        # hierarchical pipeline: staged k + sums 20, prefix 20, record slot 1, record prefix 24 (L2-resident gather)
        "k_pose_moments": (E * (20 + 20 + 1 + 24) if hier else E * 36) + nloc * (24 + 24 + 136 + 8),
        # both colours together, per pose: 17 moments, own + 2 neighbour poses, odometry 72, u 32, pose out 24
        "k_solve": nloc * (136 + 72 + 72 + 32 + 24 + 8),
        "k_scan": nloc * 16,
    }
    return per_sweep.get(kernel, 0) / max(nlaunch, 1)


def survey_bytes_per_pose(B, Kt, K, T):
    """whole-sweep compulsory bytes per pose update."""
    return 8 * B + 88 + 64 * Kt + 40.0 * K / T


def cpu_baseline(wl, cfg, n_pose, schedule):
    """The NumPy oracle (the reference's own arithmetic, library for library) on one core, on
    the first n_pose poses of the same workload at the full landmark count."""
    from oracle import icm_oracle as o
    ocfg = o.OracleConfig.from_config(cfg)
    scans = np.ascontiguousarray(wl.scans[:n_pose].T)
    u, odo = wl.u[:, :n_pose], wl.odometry[:, :n_pose]
    st = o.MapState(ocfg, wl.K)
    x = np.ascontiguousarray(wl.x_init[:, :n_pose]).copy()
    t0 = time.perf_counter()
    kept = o.prefilter_all(scans, ocfg)
    t1 = time.perf_counter()
    try:
        o.sweep(ocfg, st, scans, u, odo, wl.x0, wl.map_init.copy(), x, schedule=schedule, kept=kept)
    except ValueError:
        pass  # a short prefix may leave no landmark above `cota`; the sweep work is done by then
    t2 = time.perf_counter()
    return (n_pose - 1) / (t2 - t1), t1 - t0


def cpu_baseline_c(wl, cfg, n_pose, schedule):
    """The compiled C restatement of the same algorithm (oracle/icm_oracle_c.c, gcc -O2, one
    core): brute-force association, per-beam energy, SciPy's Nelder-Mead -- what the reference's
    sweep costs without the Python interpreter."""
    from oracle import c_oracle as co
    scans = np.ascontiguousarray(wl.scans[:n_pose].T)
    u, odo = np.ascontiguousarray(wl.u[:, :n_pose]), np.ascontiguousarray(wl.odometry[:, :n_pose])
    kept = co.prefilter(cfg, scans)
    x = np.ascontiguousarray(wl.x_init[:, :n_pose]).copy()
    t1 = time.perf_counter()
    try:
        co.sweep(cfg, kept, u, odo, wl.x0, wl.map_init, x, wl.K, schedule)
    except ValueError:
        pass
    t2 = time.perf_counter()
    return (n_pose - 1) / (t2 - t1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="S2", help="S2 (BASELINE metric config), S1, tiny")
    ap.add_argument("--cpu-poses", type=int, default=-1, help="prefix length of the CPU baseline (0 = skip)")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="N > 1: weak = the sequence has N x the workload's poses (fixed work per GPU); "
                         "strong = the workload's sequence split N ways")
    ap.add_argument("--force-sharded", action="store_true",
                    help="drive even a 1-rank run through the sharded path (torch.distributed + RCCL all-gathers)")
    args = ap.parse_args()

    import torch
    from ICM_SLAM_tools import ConfigICM
    from icmslam_hip import SweepEngine
    from icmslam_hip.synthetic import WORKLOADS, make_workload

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs torch.distributed.run with --nproc-per-node %d" % (args.gpus, args.gpus))
    dist = None
    sharded = world > 1 or args.force_sharded
    if sharded:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        if "MASTER_ADDR" not in os.environ:  # plain `python bench.py --force-sharded`
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", RANK="0", WORLD_SIZE="1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    T1, K, B = WORKLOADS[args.workload]
    T = T1 * world if args.scaling == "weak" else T1   # poses of the whole (sharded) sequence
    blk = (T + world - 1) // world
    t_begin, t_end = min(rank * blk, T), min((rank + 1) * blk, T)
    t0 = time.perf_counter()
    wl = make_workload(T, K, B, t_begin=t_begin, t_end=t_end)
    t_gen = time.perf_counter() - t0
    cfg = ConfigICM(D=wl.config)
    schedule = "redblack"

    eng = SweepEngine(cfg, local_rank)
    t0 = time.perf_counter()
    eng.upload(wl.scans, wl.odometry, wl.u, t_begin=t_begin, t_end=t_end, pose_major=True)
    t_upload = time.perf_counter() - t0
    if sharded:
        from icmslam_hip.sharded import ShardedSweep
        runner = ShardedSweep(eng, rank, world, T)
        runner.set_state(wl.map_init, wl.x_init, wl.x0)
        step = lambda: runner.sweep(schedule)  # noqa: E731
    else:
        eng.set_state(wl.map_init, wl.x_init, wl.x0)
        step = lambda: eng.sweep_device(schedule)  # noqa: E731

    # ICM on this synthetic sequence is not a contraction: the noisy odometry prior walks the poses
    # away from the 1 m association gate, after ~50 sweeps landmarks are re-created en masse and at
    # sweep 72 the map capacity L overflows (IndexError, as in the reference; identical in both
    # entry pipelines).  So that ANY --steps measures the same work per step, the state is rewound
    # to the initial one every RESET_EVERY sweeps by a device-side copy (icm_restore_state, ~20 MB
    # device-to-device, stream-ordered, ~10 us) -- inside the timed region, on every rank.
    RESET_EVERY = 12
    eng.snapshot_state()
    inner_step, nsweeps = step, [0]

    def step():  # noqa: F811
        if nsweeps[0] and nsweeps[0] % RESET_EVERY == 0:
            eng.restore_state()
        inner_step()
        nsweeps[0] += 1

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    ms_per_step = 1e3 * elapsed / max(args.steps, 1)
    value = (T - 1) * args.steps / elapsed

    # ---- roofline of the dominant kernel: HIP events around every launch, on the launch stream
    roof = None
    st = eng.last_stats()
    if not args.no_roofline:
        eng.enable_timing(True)
        nroof = 3
        for _ in range(nroof):
            inner_step()   # (no rewind here: at most 4 sweeps past the last one, well inside the stable range)
        fence()
        kt = {k: v for k, v in eng.kernel_times().items() if v[1] > 0 and k != "k_prefilter"}
        hier = eng.entry_path() == "hier"
        eng.enable_timing(False)
        # k_filtrar_grid is one workgroup on a side stream, concurrent with the solves: not on
        # the critical path and not a bandwidth kernel, so it is never the roofline subject
        side = ("k_filtrar_grid", "k_neigh_table")   # side stream, under the solves
        dom = max((k for k in kt if k not in side), key=lambda k: kt[k][0])
        per_kernel = {}
        for k, (ms_k, n_k) in kt.items():
            ab = algorithmic_bytes(k, st["kept_beams"], st["entries"], eng.nloc, eng.L, n_k / nroof, hier)
            per_kernel[k] = {"ms_per_launch": round(ms_k / n_k, 4), "launches_per_sweep": n_k / nroof,
                             "GBps": round(ab / (ms_k / n_k * 1e-3) / 1e9, 1) if ab else None}
        ms, n = kt[dom]
        avg_ms = ms / n
        nl = n / nroof
        abytes = algorithmic_bytes(dom, st["kept_beams"], st["entries"], eng.nloc, eng.L, nl, hier)
        ach = abytes / (avg_ms * 1e-3) / 1e9
        def hbm_roof(kname):
            ms_k, n_k = kt[kname]
            ab = algorithmic_bytes(kname, st["kept_beams"], st["entries"], eng.nloc, eng.L, n_k / nroof, hier)
            a_gbs = ab / (ms_k / n_k * 1e-3) / 1e9
            return {"bound": "hbm", "kernel": kname, "achieved": round(a_gbs, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(a_gbs / HBM_PEAK_GBS, 5), "traffic": traffic_of(kname),
                    "avg_launch_ms": round(ms_k / n_k, 4), "launches_per_sweep": n_k / nroof,
                    "algorithmic_bytes_per_launch": int(ab)}

        def traffic_of(kname):
            tf = os.path.join(ROOT, "profiles", "traffic.json")
            try:
                return json.load(open(tf)).get(args.workload, {}).get(kname)
            except Exception:
                return None

        stream_kernel = max((k for k in kt if k not in side + ("k_solve",)), key=lambda k: kt[k][0])
        if dom == "k_solve":
            # the solves are not a bandwidth kernel: one lane per pose runs ~85 dependent energy
            # evaluations from registers.  Their roofline is the FP64 vector rate; the flops are the
            # measured number of energy evaluations x the flops of one evaluation.
            eng.set_debug(True)
            if hier:
                eng.set_entry_path("hier")   # (debug alone would select the sort-based pipeline)
            inner_step()   # (no rewind here: at most 4 sweeps past the last one, well inside the stable range)
            fence()
            nfev = float(eng.solve_diag()[:, 2].sum())
            eng.set_debug(False)
            eng.set_entry_path("auto")
            flops = nfev * FLOP_PER_ENERGY_EVAL / nl
            tfl = flops / (avg_ms * 1e-3) / 1e12
            roof = {"bound": "valu_fp64", "kernel": dom, "achieved": round(tfl, 3), "peak": FP64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(tfl / FP64_VALU_PEAK_TFLOPS, 5), "traffic": traffic_of(dom), "avg_launch_ms": round(avg_ms, 4),
                    "launches_per_sweep": nl, "energy_evaluations_per_sweep": int(nfev),
                    "flop_per_evaluation": FLOP_PER_ENERGY_EVAL, "algorithmic_bytes_per_launch": int(abytes)}
        else:
            roof = hbm_roof(dom)
        roof["hbm_stream_kernel"] = hbm_roof(stream_kernel)
        roof["kernels_ms_per_sweep"] = {k: round(v[0] / nroof, 4) for k, v in sorted(kt.items(), key=lambda kv: -kv[1][0])}
        roof["kernels"] = per_kernel
        roof["note"] = ("k_solve: FP64-VALU/latency bound (one lane per pose, a launch lasts as long as its slowest pose's chain "
                        "of Nelder-Mead evaluations); k_assoc_group: the HBM-streaming kernel; k_filtrar_grid: one workgroup on "
                        "a side stream under the solves")
        Kt = st["entries"] / max(eng.nloc, 1)
        sweep_bytes = survey_bytes_per_pose(B, Kt, K, T) * (T - 1)
        roof["sweep_algorithmic_GBps"] = round(sweep_bytes / (ms_per_step * 1e-3) / 1e9, 2)
        roof["sweep_frac_of_hbm_peak"] = round(roof["sweep_algorithmic_GBps"] / (HBM_PEAK_GBS * world), 5)

    out = {
        "metric": "ICM pose-updates/sec (full sweep)", "value": round(value, 1), "unit": "pose-updates/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": ("%s: synthetic %d poses / %d landmarks / %d beams, red-black ICM sweep" % (args.workload, T, K, B))
                               + ("" if world == 1 else
                                  (" -- ONE sequence of %d x %d poses, a %d-pose block per GPU" % (world, T1, blk) if args.scaling == "weak"
                                   else " -- the %d-pose sequence split into %d-pose blocks" % (T, blk))),
                   "schedule": schedule, "poses": T, "poses_per_gpu": blk, "landmarks": K, "beams": B,
                   "kept_beams": st["kept_beams"] if world == 1 else None,
                   "parallelism": "pose-shard x%d" % world, "entry_pipeline": eng.entry_path(),
                   "state_rewind": "initial state restored on the device every %d sweeps" % RESET_EVERY,
                   "collectives_per_sweep": 0 if not sharded else "1 all-gather of [3L+8] f64 statistics + 2 halo all-gathers of 48 B per rank"},
        "setup_s": {"generate": round(t_gen, 2), "upload_and_prefilter": round(t_upload, 2)},
    }
    if roof is not None:
        out["roofline"] = roof
    if rank == 0 and world == 1 and args.cpu_poses != 0:
        n_c = args.cpu_poses if args.cpu_poses > 0 else {"S2": 2500, "S1": 10000}.get(args.workload, T)
        n_c = min(n_c, t_end)
        v_c = cpu_baseline_c(wl, cfg, n_c, schedule)
        out["cpu_baseline"] = {"value": round(v_c, 1), "unit": "pose-updates/s", "cores": 1, "kind": "port",
                               "sample": "compiled C oracle (oracle/icm_oracle_c.c, gcc -O2, one core; brute-force association "
                                         "and per-beam energy like the reference), one red-black sweep over the first %d poses of "
                                         "the same sequence at the full %d-landmark map; host has %d cores" % (n_c, K, os.cpu_count())}
        n_py = min({"S2": 300, "S1": 1000}.get(args.workload, T), t_end)
        v_py, _ = cpu_baseline(wl, cfg, n_py, schedule)
        out["cpu_baseline_numpy"] = {"value": round(v_py, 2), "unit": "pose-updates/s", "cores": 1, "kind": "port",
                                     "sample": "NumPy oracle (oracle/icm_oracle.py: the reference's arithmetic library for library, "
                                               "interpreter included), first %d poses" % n_py}
        out["gpu_over_cpu"] = {"vs_c_port": round(value / v_c, 1), "vs_numpy_port": round(value / v_py, 1)}
    if rank == 0:
        print(json.dumps(out))
    eng.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
