"""The reference's own pose order on data_IJAC2018 (one dependent chain of T - 1 solves, scripts/ICM_ROS.py:141-158) walked by ONE lane
(icm_set_solve_lanes(0)) and by ONE DPP quad with wave-uniform control flow (icm_set_solve_lanes(1): the straight-line quad form in the shipped build): ms per sweep, and whether the poses agree
bit for bit.    VLIB=other_build.so python tools/seq_chain_timing.py"""
import os, sys, time
sys.path.insert(0, 'icm-slam_amd'); sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from icmslam_hip import _lib
if os.environ.get("VLIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["VLIB"])
import numpy as np, torch
from icmslam_hip import SweepEngine
from util import Cfg, dataset, gold
zz, odo, u = dataset()
init = gold("init_pass.npz")
eng = SweepEngine(Cfg())
eng.upload(zz, odo, u)
res = {}
for mode in (0, 1, 0, 1):
    eng.set_solve_lanes(mode)
    ts = []
    for _ in range(5):
        eng.set_state(init["map_init"], init["x_init"], odo[:, 0], 11)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        eng.sweep_device("sequential")
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    x = eng.get_state()[0]
    res.setdefault(mode, x)
    print("solve lanes %d (%s): sequential sweep %.2f ms (median of 5)  same poses as lanes 0: %s" % (mode, "one DPP quad" if mode else "one lane", 1e3 * sorted(ts)[2], bool(np.array_equal(x, res[0]))), flush=True)
ts = []
for _ in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    eng.init_pass(odo[:, 0])
    torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
print("init pass %.2f ms (median of 3)" % (1e3 * sorted(ts)[1]))
eng.close()
