"""The reference's own call on S2 in the driver loop's pattern (the same x array handed back): ms per call, icm_get_dropin_counts
(profiles/r04_dropin_call.txt)."""
import sys, time
sys.path.insert(0, 'icm-slam_amd'); sys.path.insert(0, '.')
import numpy as np, torch
from copy import deepcopy as copy
from ICM_SLAM_tools import ConfigICM
from icmslam_hip import SweepEngine
from icmslam_hip.synthetic import WORKLOADS, make_workload
from ICM_ROS import ICM_ROS
wl = make_workload(*WORKLOADS["S2"])
cfg = ConfigICM(D=dict(wl.config, schedule="redblack"))
eng = SweepEngine(cfg)
eng.upload(wl.scans, wl.odometry, wl.u, pose_major=True)
def t(f, n=10):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
icm = ICM_ROS(cfg)
icm.attach_engine(eng, wl.scans.T, wl.odometry, wl.u)
icm.x0 = wl.x0.reshape(3, 1)
icm.set_initial_state(wl.x_init, wl.map_init)
print("_sequence_key %.4f ms" % t(lambda: icm._sequence_key(), 20))
def reset():
    icm.set_initial_state(wl.x_init, wl.map_init)
    return {"mv": copy(icm.mapa_viejo), "x": copy(icm.positions)}
st = reset()
def call():
    mr, st["x"] = icm.iterations_process_offline(st["mv"], st["x"]); st["mv"] = copy(mr)
def call_fresh_():
    mr, x = icm.iterations_process_offline(st["mv"], st["x"].copy()); st["x"] = x; st["mv"] = copy(mr)
for _ in range(3): call()
print("drop-in call (same x object: pinned from the 2nd call) %.4f ms" % t(call, 16))
print(eng.dropin_counts())
eng.close()
