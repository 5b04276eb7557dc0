"""A/B of phase A's launch form on S2 (one MI355X): k_assoc_group on persistent waves (workgroups per CU) against one
short-lived wave per pose; per-kernel HIP-event times and the whole sweep, same box, interleaved.

    python tools/ab_assoc.py [wg_per_cu ...]      # default -1 0 8 (0: four poses per workgroup, -1: one-wave workgroups, n: persistent)
"""
import sys, time
sys.path.insert(0, 'icm-slam_amd'); sys.path.insert(0, '.')
import numpy as np, torch
from ICM_SLAM_tools import ConfigICM
from icmslam_hip import SweepEngine
from icmslam_hip.synthetic import WORKLOADS, make_workload

wl = make_workload(*WORKLOADS[sys.argv[1] if len(sys.argv) > 1 and sys.argv[1] in WORKLOADS else "S2"])
cfg = ConfigICM(D=wl.config)
modes = [int(a) for a in sys.argv[1:] if a.lstrip('-').isdigit()] or [-1, 0, 8]
eng = SweepEngine(cfg)
eng.upload(wl.scans, wl.odometry, wl.u, pose_major=True)
eng.set_state(wl.map_init, wl.x_init, wl.x0)
for _ in range(3):
    eng.sweep_device("redblack")
eng.snapshot_state()
ref = None
for rep in range(2):
    for m in modes:
        eng.set_assoc_persistence(m)
        eng.restore_state()
        eng.sweep_device("redblack")
        torch.cuda.synchronize(); t0 = time.perf_counter(); n = 20
        for _ in range(n):
            eng.sweep_device("redblack")
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
        st = eng.get_state()
        if ref is None:
            ref = st
        same = all(np.array_equal(a, b) for a, b in zip(ref[:3], st[:3])) and ref[3] == st[3]
        eng.restore_state()
        eng.enable_timing(True)
        for _ in range(3):
            eng.sweep_device("redblack")
        kt = eng.kernel_times(); eng.enable_timing(False)
        print("persistence %d: sweep %.4f ms  k_assoc_group %.4f ms  state == first run: %s" % (m, dt * 1e3, kt["k_assoc_group"][0] / kt["k_assoc_group"][1], same), flush=True)
eng.close()
