"""Per-wave timeline of k_solve_m_fused at S2: when every wave starts, when an even wave's two odd waves have published,
when it ends.  Needs a measurement build of the library (per-wave time stamps and evaluation counters compiled in):
    cd icm-slam_amd/csrc && hipcc -O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -DICM_WAVE_TS -shared \
        -o ../../tools/lib_ts.so icm_api.hip icm_host.cpp
    python tools/wave_timeline.py            (VLIB=path selects another build)"""
import os, sys, ctypes as C
sys.path.insert(0, 'icm-slam_amd'); sys.path.insert(0, '.')
from icmslam_hip import _lib
_lib.LIB_PATH = os.path.abspath(os.environ.get("VLIB", "tools/lib_ts.so"))
import numpy as np
from ICM_SLAM_tools import ConfigICM
from icmslam_hip import SweepEngine
from icmslam_hip.synthetic import WORKLOADS, make_workload
wl = make_workload(*WORKLOADS["S2"])
eng = SweepEngine(ConfigICM(D=wl.config))
eng.upload(wl.scans, wl.odometry, wl.u, pose_major=True)
eng.set_state(wl.map_init, wl.x_init, wl.x0)
for _ in range(5): eng.sweep_device("redblack")
es = eng.lib.icm_debug_eval_stats
es.restype = C.c_int; es.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
st4 = np.zeros(4, dtype=np.uint64)
es(eng.h, st4.ctypes.data_as(C.c_void_p), 1)
eng.set_debug(True)
eng.sweep_device("redblack")
es(eng.h, st4.ctypes.data_as(C.c_void_p), 1)
print("wave evaluations %d, taking the generic call %d (%.2f%%); lane evaluations %d, generic %d (%.4f%%)" % (st4[0], st4[1], 100.0 * st4[1] / max(st4[0], 1), st4[2], st4[3], 100.0 * st4[3] / max(st4[2], 1)))
d = eng.solve_diag()
nw = (50000 + 63) // 64
buf = np.zeros(4 * 2 * nw, dtype=np.uint64)
fn = eng.lib.icm_debug_wave_ts
fn.restype = C.c_int; fn.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
assert fn(eng.h, buf.ctypes.data_as(C.c_void_p), buf.size) == 0
ts = buf.reshape(-1, 4).astype(np.float64)
t0 = ts[:, 0].min()
ts = (ts - t0) / 100.0   # 100 MHz -> microseconds
odd, even = ts[:nw], ts[nw:]
nit = d[:, 1]
def wmax(first):
    it = nit[first::2]; n = (len(it) // 64) * 64
    return it[:n].reshape(-1, 64).max(axis=1)
mo, me = wmax(1), wmax(2)
print("odd : start mean %.1f max %.1f | end mean %.1f max %.1f | dur mean %.1f max %.1f us" % (odd[:, 0].mean(), odd[:, 0].max(), odd[:, 2].mean(), odd[:, 2].max(), (odd[:, 2] - odd[:, 1]).mean(), (odd[:, 2] - odd[:, 1]).max()))
print("even: start mean %.1f max %.1f | go mean %.1f max %.1f | end mean %.1f max %.1f | dur mean %.1f max %.1f us" % (even[:, 0].mean(), even[:, 0].max(), even[:, 1].mean(), even[:, 1].max(), even[:, 2].mean(), even[:, 2].max(), (even[:, 2] - even[:, 1]).mean(), (even[:, 2] - even[:, 1]).max()))
k = min(len(mo), nw)
dur_o = (odd[:k, 2] - odd[:k, 1]); dur_e = (even[:k, 2] - even[:k, 1])
print("us per NM iteration of the slowest lane: odd %.3f  even %.3f" % (np.median(dur_o[mo[:k] > 0] / mo[:k][mo[:k] > 0]), np.median(dur_e[me[:k] > 0] / me[:k][me[:k] > 0])))
print("per-wave max nit: odd mean %.1f max %.0f | even mean %.1f max %.0f" % (mo.mean(), mo.max(), me.mean(), me.max()))
wait = even[:, 1] - even[:, 0]
print("even waves: wait for flags mean %.1f max %.1f us; start-after-kernel-begin percentiles" % (wait.mean(), wait.max()), np.percentile(even[:, 0], [10, 50, 90, 99]).round(1))
print("odd end percentiles", np.percentile(odd[:, 2], [10, 50, 90, 99, 100]).round(1))
print("even go percentiles", np.percentile(even[:, 1], [10, 50, 90, 99, 100]).round(1))
print("even end percentiles", np.percentile(even[:, 2], [10, 50, 90, 99, 100]).round(1))
# what gates each even wave's go: its flags (odd end of wv, wv+1) or its own dispatch
gate = np.maximum(odd[:k, 2], np.r_[odd[1:k, 2], 0])
print("even go - max(odd end of its two): mean %.2f  p99 %.2f us ; dispatch later than flags in %d waves" % ((even[:k, 1] - gate).mean(), np.percentile(even[:k, 1] - gate, 99), int((even[:k, 0] > gate).sum())))
np.save("gpurun_out/wave_ts.npy", ts)
eng.close()
