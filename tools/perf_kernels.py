import sys, time
sys.path.insert(0, 'icm-slam_amd'); sys.path.insert(0, '.')
import numpy as np, torch
from ICM_SLAM_tools import ConfigICM
from icmslam_hip import SweepEngine
from icmslam_hip.synthetic import WORKLOADS, make_workload
name = sys.argv[1] if len(sys.argv) > 1 else "S2"
T,K,B = WORKLOADS[name]
wl = make_workload(T,K,B); cfg = ConfigICM(D=wl.config)
eng = SweepEngine(cfg, 0); eng.upload(wl.scans, wl.odometry, wl.u, pose_major=True)
out = {}
for mode in ("sort","hier"):
    eng.set_entry_path(mode)
    eng.set_state(wl.map_init, wl.x_init, wl.x0)
    for _ in range(3): eng.sweep_device("redblack")
    torch.cuda.synchronize(); t0=time.perf_counter(); n=20
    for _ in range(n): eng.sweep_device("redblack")
    torch.cuda.synchronize(); dt=(time.perf_counter()-t0)/n
    eng.enable_timing(True)
    for _ in range(3): eng.sweep_device("redblack")
    kt=eng.kernel_times(); eng.enable_timing(False)
    print(mode, eng.entry_path(), '%.3f ms/sweep'%(dt*1e3), {k: round(v[0]/3,3) for k,v in kt.items() if v[1]}, flush=True)
    out[mode] = eng.get_state()
a, b = out["sort"], out["hier"]
print("x diff after 26 sweeps", np.abs(a[0]-b[0]).max(), "map diff", np.abs(a[1]-b[1]).max(), "K", a[3], b[3])
eng.close()
