"""Nelder-Mead statistics of one S2 sweep (what bounds k_solve_m_fused): evaluations and iterations
per pose, and per WAVE of the one-lane-per-pose solve (a wave runs as long as its slowest lane)."""
import sys
sys.path.insert(0, 'icm-slam_amd'); sys.path.insert(0, '.')
import numpy as np
from ICM_SLAM_tools import ConfigICM
from icmslam_hip import SweepEngine
from icmslam_hip.synthetic import WORKLOADS, make_workload
name = sys.argv[1] if len(sys.argv) > 1 else "S2"
wl = make_workload(*WORKLOADS[name])
eng = SweepEngine(ConfigICM(D=wl.config))
eng.upload(wl.scans, wl.odometry, wl.u, pose_major=True)
eng.set_state(wl.map_init, wl.x_init, wl.x0)
for sweep in range(1, 6):
    eng.set_debug(sweep in (1, 5)); eng.set_entry_path("hier")
    eng.sweep_device("redblack")
    if sweep in (1, 5):
        d = eng.solve_diag()
        nit, nfev = d[1:, 1], d[1:, 2]
        print("sweep %d: nfev mean %.1f median %.0f p99 %.0f max %.0f | nit mean %.1f max %.0f | nfev/nit %.3f" % (
            sweep, nfev.mean(), np.median(nfev), np.percentile(nfev, 99), nfev.max(), nit.mean(), nit.max(), nfev.sum() / nit.sum()))
        for colour, first in (("odd", 1), ("even", 2)):
            it = d[first::2, 1]
            n = (len(it) // 64) * 64
            w = it[:n].reshape(-1, 64)
            print("   %s waves: %d, per-wave max nit mean %.1f  min %.0f  max %.0f; lane utilisation %.2f" % (
                colour, w.shape[0], w.max(axis=1).mean(), w.max(axis=1).min(), w.max(axis=1).max(), w.mean() / w.max(axis=1).mean()))
eng.close()
