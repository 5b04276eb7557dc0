"""Phase A alone on a frozen state (poses never change: sweep_local only), per-kernel timing pass."""
import os, sys
sys.path.insert(0, 'icm-slam_amd'); sys.path.insert(0, '.')
from icmslam_hip import _lib
if os.environ.get("VLIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["VLIB"])
import numpy as np, torch
from ICM_SLAM_tools import ConfigICM
from icmslam_hip import SweepEngine
from icmslam_hip.synthetic import WORKLOADS, make_workload
wl = make_workload(*WORKLOADS[os.environ.get("WL", "S2")])
eng = SweepEngine(ConfigICM(D=wl.config))
eng.upload(wl.scans, wl.odometry, wl.u, pose_major=True)
eng.set_state(wl.map_init, wl.x_init, wl.x0)
for _ in range(3):
    eng.sweep_local()
eng.enable_timing(True)
for _ in range(5):
    eng.sweep_local()
kt = eng.kernel_times()
print(os.environ.get("VLIB", "default"), {k: round(v[0] / v[1], 4) for k, v in kt.items() if v[1] and k in ("k_assoc_group", "k_chunk_l1", "k_scan")}, eng.last_stats())
eng.close()
