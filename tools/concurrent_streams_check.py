import sys, time, threading
sys.path.insert(0, 'icm-slam_amd'); sys.path.insert(0, '.')
import numpy as np, torch
from ICM_SLAM_tools import ConfigICM
from icmslam_hip import SweepEngine
from icmslam_hip.synthetic import make_workload
wls = [make_workload(60_000, 10_000, 720, seed=20181 + i) for i in range(3)]
engs = []
for wl in wls:
    cfg = ConfigICM(D=wl.config)
    e = SweepEngine(cfg, 0); e.upload(wl.scans, wl.odometry, wl.u, pose_major=True)
    e.set_state(wl.map_init, wl.x_init, wl.x0); e.snapshot_state()
    engs.append(e)
def run(e, n, out, k):
    st = []
    for i in range(n):
        if i % 10 == 0 and i:
            st.append(e.get_state()[0].copy()); e.restore_state()
        e.sweep_device("redblack")
    out[k] = st
ref = [None]*3
for k, e in enumerate(engs): run(e, 31, ref, k)      # alone
for e in engs: e.restore_state()
torch.cuda.synchronize()
got = [None]*3
th = [threading.Thread(target=run, args=(e, 201, got, k)) for k, e in enumerate(engs)]
t0 = time.perf_counter()
for t in th: t.start()
for t in th: t.join()
torch.cuda.synchronize()
print('3 engines x 201 sweeps concurrently: %.2f s' % (time.perf_counter()-t0))
ok = True
for k in range(3):
    for j, s in enumerate(got[k]):
        if not np.array_equal(s, ref[k][0]):
            ok = False; print('engine', k, 'round', j, 'differs from the solo run: max', np.abs(s-ref[k][0]).max())
print('all rounds of all engines bit-identical to the solo runs:', ok)
