"""CPU sanitizer pass over the host C / C++ of this repository (AddressSanitizer + UndefinedBehaviorSanitizer; GPU
sanitizers are not available on the pool): the library's host routines (icm-slam_amd/csrc/icm_host.cpp: Mapa.filtrar of
scripts/ICM_SLAM_tools.py:204-265, the search grid, the first scan's clustering :160-165) and the C oracle
(oracle/icm_oracle_c.c).  `make asan` builds both with gcc -fsanitize=address,undefined; a child process with the
sanitizer runtime preloaded drives them over the reference's golden inputs and the edge cases (empty / single / coincident
landmarks, non-finite coordinates, no-beam scans) and must end clean AND reproduce the goldens."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import ctypes as C, os, sys
import numpy as np
ROOT = sys.argv[1]
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "icm-slam_amd"))
from util import Cfg, dataset, gold
dp = C.POINTER(C.c_double)
def P(a): return a.ctypes.data_as(dp)

# ---- the library's host C++ -------------------------------------------------------------------------------------
from icmslam_hip._lib import IcmConfig
lib = C.CDLL(os.path.join(ROOT, "tests", "asan", "libicm_host_asan.so"))
def cfg_of(L, cota=300.0, thr=1.0):
    c = IcmConfig(); c.deltat = 0.1; c.Q[0] = c.Q[1] = 1.0; c.R[0] = c.R[1] = c.R[2] = 1.0
    c.cte_odom = 1.0; c.cota = cota; c.dist_thr = thr; c.rango_laser_max = 10.0; c.L = L
    return c
def filtrar(c, y, cnt, lact):
    L = int(c.L)
    yo, co, lo = np.zeros((2, L)), np.zeros(L), C.c_int64(0)
    rc = lib.asan_filtrar(C.byref(c), P(np.ascontiguousarray(y)), P(np.ascontiguousarray(cnt)), C.c_int64(lact), P(yo), P(co), C.byref(lo))
    return rc, yo, co, int(lo.value)
pp = gold("sweep1_perpose.npz")
L = 1000
y = np.zeros((2, L)); cnt = np.zeros(L); la = int(pp["filtrar_lact_in"])
y[:, :la] = pp["filtrar_y_in"]; cnt[:la] = pp["filtrar_cnt_in"]
rc, yo, co, lo = filtrar(cfg_of(L), y, cnt, la)
assert rc == 0 and lo == int(pp["filtrar_lact_out"]) and np.array_equal(yo[:, :lo], pp["filtrar_y_out"]) and np.array_equal(co[:lo], pp["filtrar_cnt_out"])
rng = np.random.default_rng(3)
for trial in range(60):     # merges, chains, coincident survivors, single / no survivors, a full map
    n = int(rng.integers(0, 40))
    Lt = 64 if trial % 7 else n if n else 1
    pts = rng.uniform(-6, 6, (2, n))
    if n > 3 and trial % 3 == 0: pts[:, 1] = pts[:, 0] + 0.3          # a pair inside the gate
    if n > 5 and trial % 5 == 0: pts[:, 3] = pts[:, 2]                # coincident
    if n > 8 and trial % 4 == 0: pts[:, 5:8] = pts[:, [4]] + np.array([[0.4, 0.8, 1.2], [0, 0, 0]])   # a chain
    yy = np.zeros((2, Lt)); cc = np.zeros(Lt)
    m = min(n, Lt)
    yy[:, :m] = pts[:, :m]; cc[:m] = rng.integers(0, 12, m)
    rc, yo, co, lo = filtrar(cfg_of(Lt, cota=5.0), yy, cc, m)
    assert rc in (0, -4), rc
    if rc == 0: assert 1 <= lo <= m and np.isfinite(yo[:, :lo]).all()
rc, *_ = filtrar(cfg_of(8), np.zeros((2, 8)), np.zeros(8), 9)       # landmarks_actuales beyond L: refused, not read
assert rc == -3
out3 = (C.c_int64 * 3)()
for K, pts in ((0, np.zeros((2, 1))), (1, np.zeros((2, 1))), (5, np.array([[0, 1e9, -1e9, 3, 4.0], [0, 1, 2, 3, 4.0]])),
               (4, np.array([[0, np.nan, 2, 3.0], [0, 1, np.inf, 3.0]])), (500, rng.uniform(-100, 100, (2, 500)))):
    pts = np.ascontiguousarray(pts, dtype=np.float64)
    tot = lib.asan_build_grid(P(pts[0].copy()), P(pts[1].copy()), C.c_int64(K), C.c_double(1.0), out3)
    assert tot == K and out3[2] == K and out3[0] >= 1 and out3[1] >= 1
from scipy.cluster.hierarchy import fcluster, linkage
from scipy.spatial.distance import pdist
for trial in range(40):
    pts = np.concatenate([rng.normal(c, 0.15, (rng.integers(1, 6), 2)) for c in rng.uniform(-6, 6, (rng.integers(1, 6), 2))])
    lab = np.zeros(pts.shape[0], dtype=np.int32)
    rc = lib.asan_cluster_first_scan(P(np.ascontiguousarray(pts)), C.c_int64(pts.shape[0]), C.c_double(1.0), lab.ctypes.data_as(C.POINTER(C.c_int32)))
    assert rc == 0
    if pts.shape[0] >= 2:
        assert np.array_equal(lab, fcluster(linkage(pdist(pts)), 1.0) - 1)
print("host C++ clean")

# ---- the C oracle -----------------------------------------------------------------------------------------------
from oracle import c_oracle as co
co._PATH = os.path.join(ROOT, "oracle", "libicm_oracle_c_asan.so")
from oracle import icm_oracle as o
zz, odo, u = dataset()
cfg = Cfg()
T = 260
scans = zz[:, :T].copy()
scans[:, 7] = cfg.rango_laser_max          # a scan without beams
kept = co.prefilter(cfg, scans)
T = int(np.flatnonzero(np.diff(kept[0]) > 0)[-1]) + 1     # (the reference raises on a no-beam LAST pose: end on a scan with beams)
scans = np.ascontiguousarray(scans[:, :T])
kept = co.prefilter(cfg, scans)
g = gold("filtrar_z.npz")
assert np.array_equal(kept[0][:7], g["offsets"][:7])
init = gold("init_pass.npz")
for sched in ("sequential", "redblack"):
    x = np.ascontiguousarray(init["x_init"][:, :T]).copy()
    a = {}
    try:
        co.sweep(Cfg(cota=5.0), kept, np.ascontiguousarray(u[:, :T]), np.ascontiguousarray(odo[:, :T]), odo[:, 0], init["map_init"], x, 11, sched, assoc=a)
    except ValueError:
        pass
    assert np.isfinite(x).all() and a["labels"].size == kept[0][-1]
ocfg = o.OracleConfig()
st = o.MapState(ocfg)
k0 = o.filtrar_z(scans[:, 0], ocfg)
y0, _ = o.cluster_first_scan(st, np.zeros((2, ocfg.L)), o.project_beams(odo[:, 0].copy(), k0[:, 2:4]))
xi, yi, ci, li = co.init_pass(cfg, kept, np.ascontiguousarray(u[:, :T]), np.ascontiguousarray(odo[:, :T]), y0, st.cant_obs_i, st.landmarks_actuales)
assert np.abs(xi[:, :7] - init["x_init"][:, :7]).max() <= 1e-9         # (identical to the reference up to the scan that was blanked)
s = gold("solve_t100.npz")
out = co.solve_one(cfg, 1, init["x_init"][:, 99], init["x_init"][:, 101], u[:, 99:101], odo[:, 99:102], s["beams"][:, 0:2], s["targets"])
assert np.abs(out[:3] - s["xopt"]).max() <= 1e-9 and int(out[4]) == int(s["nit"]) and int(out[5]) == int(s["nfev"])
print("C oracle clean")
'''


def _runtime(name):
    p = subprocess.run(["gcc", "-print-file-name=" + name], capture_output=True, text=True).stdout.strip()
    return p if os.path.isabs(p) and os.path.exists(p) else None


def test_host_code_and_c_oracle_under_address_and_ub_sanitizers(tmp_path):
    asan = _runtime("libasan.so")
    if asan is None:
        pytest.skip("no libasan in this toolchain")
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "icm-slam_amd", "csrc"), "asan"])
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "asan"])
    script = tmp_path / "child.py"
    script.write_text(CHILD)
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0:exitcode=23:halt_on_error=1",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1:exitcode=24", PYTHONDONTWRITEBYTECODE="1", OMP_NUM_THREADS="1")
    p = subprocess.run([sys.executable, str(script), ROOT], env=env, capture_output=True, text=True, timeout=900)
    tail = (p.stdout + "\n" + p.stderr)[-4000:]
    assert p.returncode == 0, tail
    assert "host C++ clean" in p.stdout and "C oracle clean" in p.stdout, tail
    assert "AddressSanitizer" not in p.stderr and "runtime error" not in p.stderr, tail
