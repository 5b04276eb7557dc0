"""Where a k_assoc_runs wave's life goes at S2: shader-clock stamps of every 16th pose's wave (start, header scalars in, run
records in, grid records in + decisions, table updated, end).  Needs a measurement build (the stamps add full waits):
    cd icm-slam_amd/csrc && hipcc -O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -mllvm -amdgpu-kernarg-preload-count=14 \
        -DICM_ASSOC_TS -shared -o ../../scratch/lib_ats.so icm_api.hip icm_host.cpp
    VLIB=scratch/lib_ats.so python tools/assoc_runs_timeline.py"""
import os, sys, ctypes as C
sys.path.insert(0, 'icm-slam_amd'); sys.path.insert(0, '.')
from icmslam_hip import _lib
_lib.LIB_PATH = os.path.abspath(os.environ.get("VLIB", "scratch/lib_ats.so"))
import numpy as np
from ICM_SLAM_tools import ConfigICM
from icmslam_hip import SweepEngine
from icmslam_hip.synthetic import WORKLOADS, make_workload
wl = make_workload(*WORKLOADS[os.environ.get("WL", "S2")])
eng = SweepEngine(ConfigICM(D=wl.config))
eng.upload(wl.scans, wl.odometry, wl.u, pose_major=True)
eng.set_state(wl.map_init, wl.x_init, wl.x0)
for _ in range(5): eng.sweep_device("redblack")
n = min(8192, wl.scans.shape[0] // 16)
buf = np.zeros(8 * 8192, dtype=np.uint64)
fn = eng.lib.icm_debug_assoc_ts
fn.restype = C.c_int; fn.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
assert fn(eng.h, buf.ctypes.data_as(C.c_void_p), buf.size) == 0
ts = buf.reshape(-1, 8)[:n].astype(np.float64)
t = ts[ts[:, 7] >= 1]
GHZ = float(os.environ.get("GHZ", "2.15"))   # shader clock under this load (s_memtime ticks)
def us(a): return a / (GHZ * 1e3)
life = us(t[:, 6] - t[:, 0])
print("waves sampled %d, run batches per wave mean %.2f" % (len(t), t[:, 7].mean()))
print("kernel span by the stamps: %.1f us" % us(ts[:, 6].max() - ts[ts[:, 0] > 0, 0].min()))
print("wave life            mean %.2f  p50 %.2f  p90 %.2f us" % (life.mean(), np.median(life), np.percentile(life, 90)))
for name, a, b in (("start -> header scalars", 0, 1), ("scalars -> run records", 1, 2), ("run records -> grid records, decisions", 2, 3),
                   ("decisions -> table updated", 3, 4), ("table updated -> end of the run loop", 4, 5), ("loop end -> end (compaction + stores)", 5, 6)):
    d = us(t[:, b] - t[:, a])
    print("%-46s mean %.2f  p50 %.2f  p90 %.2f us  (%.0f %% of the life)" % (name, d.mean(), np.median(d), np.percentile(d, 90), 100 * d.mean() / life.mean()))
eng.close()
