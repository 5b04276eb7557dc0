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
# NPOSE=n: only the first n poses of the sequence (against the full map) -- the counter passes of tools/pmc_assoc.sh
npose = int(os.environ.get("NPOSE", "0"))
T_, K_, B_ = WORKLOADS[os.environ.get("WL", "S2")]
# WL_CACHE=file.npz: the generated workload is kept there (the counter passes start this program once per counter group)
cache = os.environ.get("WL_CACHE")
if cache and os.path.exists(cache):
    class _W: pass
    wl = _W()
    z = np.load(cache, allow_pickle=True)
    for k in ("scans", "odometry", "u", "map_init", "x_init", "x0"):
        setattr(wl, k, z[k])
    wl.config = z["config"].item()
else:
    wl = make_workload(T_, K_, B_, t_end=npose or None)
    if cache:
        np.savez(cache, scans=wl.scans, odometry=wl.odometry, u=wl.u, map_init=wl.map_init, x_init=wl.x_init, x0=wl.x0, config=np.array(wl.config, dtype=object))
eng = SweepEngine(ConfigICM(D=wl.config))
if npose:
    eng.upload(wl.scans, wl.odometry, wl.u, t_begin=0, t_end=npose, pose_major=True)
else:
    eng.upload(wl.scans, wl.odometry, wl.u, pose_major=True)
eng.set_state(wl.map_init, wl.x_init, wl.x0)
if os.environ.get("ASSOC_FORM"):
    eng.set_assoc_form(int(os.environ["ASSOC_FORM"]))   # 1 runs (default), 0 beam by beam
for _ in range(int(os.environ.get("WARM", "3"))):
    eng.sweep_local()
eng.enable_timing(True)
for _ in range(int(os.environ.get("REPS", "5"))):
    eng.sweep_local()
kt = eng.kernel_times()
print(os.environ.get("VLIB", "default"), {k: round(v[0] / v[1], 4) for k, v in kt.items() if v[1] and k in ("k_assoc_group", "k_assoc_runs", "k_chunk_l1", "k_scan")}, eng.last_stats())
eng.close()
