"""Where a k_assoc_group wave's life goes at S2: shader-clock stamps of every 16th pose's wave (start, header scalars in, first
beams in, end of batch 1 / 2 / last, end).  Needs a measurement build (the stamps add two full waits at the head of the wave):
    cd icm-slam_amd/csrc && hipcc -O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -mllvm -amdgpu-kernarg-preload-count=14 \
        -DICM_ASSOC_TS -shared -o ../../scratch/lib_ats.so icm_api.hip icm_host.cpp
    VLIB=scratch/lib_ats.so python tools/assoc_timeline.py"""
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
n = min(8192, (wl.scans.shape[0] if wl.scans.shape[0] > wl.scans.shape[1] else wl.scans.shape[1]) // 16)
buf = np.zeros(8 * 8192, dtype=np.uint64)
fn = eng.lib.icm_debug_assoc_ts
fn.restype = C.c_int; fn.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
assert fn(eng.h, buf.ctypes.data_as(C.c_void_p), buf.size) == 0
ts = buf.reshape(-1, 8)[:n].astype(np.float64)
nb = ts[:, 7]
ok = nb >= 2
t = ts[ok]
GHZ = float(os.environ.get("GHZ", "2.4"))   # shader clock under this load (s_memtime ticks)
def us(a): return a / (GHZ * 1e3)
life = us(t[:, 6] - t[:, 0])
print("waves sampled %d (>= 2 batches: %d), batches per wave mean %.2f" % (n, ok.sum(), nb.mean()))
print("kernel span by the stamps: %.1f us" % us(ts[:, 6].max() - ts[ts[:, 0] > 0, 0].min()))
print("wave life            mean %.2f  p50 %.2f  p90 %.2f us" % (life.mean(), np.median(life), np.percentile(life, 90)))
for name, a, b in (("start -> header scalars", 0, 1), ("scalars -> first beams", 1, 2), ("first beams -> end of batch 1", 2, 3), ("batch 2", 3, 4), ("batch 2 end -> last batch end", 4, 5), ("last batch end -> end (compaction + stores)", 5, 6)):
    d = us(t[:, b] - t[:, a])
    print("%-46s mean %.2f  p50 %.2f  p90 %.2f us  (%.0f %% of the life)" % (name, d.mean(), np.median(d), np.percentile(d, 90), 100 * d.mean() / life.mean()))
rest = us(t[:, 5] - t[:, 4]) / np.maximum(t[:, 7] - 2, 1)
print("per batch after the second: mean %.2f us" % rest[t[:, 7] > 2].mean())
np.save("gpurun_out/assoc_ts.npy", ts)
eng.close()
