import sys, time
sys.path.insert(0, 'icm-slam_amd'); sys.path.insert(0, '.')
import numpy as np, torch
from ICM_SLAM_tools import ConfigICM
from icmslam_hip import SweepEngine
from icmslam_hip.synthetic import WORKLOADS, make_workload
for name, n in (("S1", 4000), ("S2", 3000)):
    T,K,B = WORKLOADS[name]
    wl = make_workload(T,K,B); cfg = ConfigICM(D=wl.config)
    eng = SweepEngine(cfg, 0); eng.upload(wl.scans, wl.odometry, wl.u, pose_major=True)
    eng.set_state(wl.map_init, wl.x_init, wl.x0); eng.snapshot_state()
    ref = None
    t0 = time.perf_counter()
    for i in range(n):
        if i and i % 12 == 0:
            if i == 12:
                ref = eng.get_state()
            elif i % 60 == 0:
                cur = eng.get_state()
                assert all(np.array_equal(a, b) for a, b in zip(ref, cur)), "state after 12 sweeps differs from the first round"
            eng.restore_state()
        eng.sweep_device("redblack")
        if (i+1) % 1000 == 0:
            torch.cuda.synchronize(); print(name, i+1, 'sweeps, %.3f ms each' % ((time.perf_counter()-t0)/(i+1)*1e3), eng.entry_path(), flush=True)
    eng.get_state()
    print(name, 'done', flush=True)
    eng.close()
