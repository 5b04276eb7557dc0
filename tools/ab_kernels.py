"""A/B timing of library builds: sweep time (20 consecutive sweeps) and the solve / association / level-2..3 kernel times.
    VLIB=path/to/other_build.so WL=S2|S1|tiny python tools/ab_kernels.py [solve-lane modes: 0 lane, 1 quad]"""
import os, sys
sys.path.insert(0, 'icm-slam_amd'); sys.path.insert(0, '.')
from icmslam_hip import _lib
if os.environ.get("VLIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["VLIB"])
import numpy as np, time, zlib
from ICM_SLAM_tools import ConfigICM
from icmslam_hip import SweepEngine
from icmslam_hip.synthetic import WORKLOADS, make_workload
import torch
wl = make_workload(*WORKLOADS[os.environ.get("WL", "S2")])
eng = SweepEngine(ConfigICM(D=wl.config))
eng.upload(wl.scans, wl.odometry, wl.u, pose_major=True)
ref = None
for mode in [int(a) for a in sys.argv[1:]] or [0, 1]:
    eng.set_solve_lanes(mode)
    eng.set_state(wl.map_init, wl.x_init, wl.x0)
    for _ in range(3): eng.sweep_device("redblack")
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): eng.sweep_device("redblack")
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 20 * 1e3
    eng.enable_timing(True)
    for _ in range(3): eng.sweep_device("redblack")
    kt = eng.kernel_times(); eng.enable_timing(False)
    x = eng.get_state()[0]
    if ref is None: ref = x
    print(os.environ.get("VLIB", "default"), "lanes-mode", mode, "ms/sweep %.4f" % ms, "k_solve", kt.get("k_solve"), "k_assoc", round(sum(kt[k][0] / kt[k][1] for k in ("k_assoc_group", "k_assoc_runs") if kt.get(k, (0, 0))[1]), 4), "l1", round(kt["k_chunk_l1"][0]/kt["k_chunk_l1"][1],4), "moments", round(kt["k_pose_moments"][0]/kt["k_pose_moments"][1],4), "l2/l3/push", [round(kt[k][0]/kt[k][1],4) for k in ("k_chunk_l2","k_lm_l3","k_rec_push")], "same", bool(np.array_equal(x, ref)), "crc %08x" % zlib.crc32(np.ascontiguousarray(x).tobytes()), flush=True)
eng.close()
