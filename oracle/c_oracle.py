"""ctypes front end of oracle/icm_oracle_c.c -- TEST INFRASTRUCTURE (see that file's header).
Same role as oracle/icm_oracle.py, ~300x faster; used for full-size S1 checks and as the
compiled single-core CPU baseline of bench.py."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_HERE, "libicm_oracle_c.so")
_lib = None
_dp = C.POINTER(C.c_double)
_lp = C.POINTER(C.c_int64)
_ip = C.POINTER(C.c_int32)


class OcConfig(C.Structure):
    _fields_ = [("deltat", C.c_double), ("Q", C.c_double * 2), ("R", C.c_double * 3), ("cte_odom", C.c_double),
                ("cota", C.c_double), ("dist_thr", C.c_double), ("rango_laser_max", C.c_double), ("L", C.c_int64)]


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(_PATH):
            subprocess.check_call(["make", "-C", _HERE])
        _lib = C.CDLL(_PATH)
        _lib.oc_filtrar_z.restype = C.c_int64
        _lib.oc_get_threads.restype = C.c_int
    return _lib


def _cfg(c):
    o = OcConfig()
    o.deltat = float(c.deltat)
    o.Q[0], o.Q[1] = float(c.Q[0, 0]), float(c.Q[1, 1])
    o.R[0], o.R[1], o.R[2] = float(c.R[0, 0]), float(c.R[1, 1]), float(c.R[2, 2])
    o.cte_odom, o.cota, o.dist_thr = float(c.cte_odom), float(c.cota), float(c.dist_thr)
    o.rango_laser_max, o.L = float(c.rango_laser_max), int(c.L)
    return o


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _p(a):
    return a.ctypes.data_as(_dp)


def bearings(B, cfg):
    k = np.arange(B)
    inc = getattr(cfg, "angle_increment", None)
    return k * np.pi / 180.0 if inc is None else k * float(inc)


def prefilter(cfg, scans_BT):
    """filtrar_z for every column; returns CSR (off, k, d, ang, bx, by)."""
    lib = load()
    c = _cfg(cfg)
    scans = _f(np.asarray(scans_BT).T)
    T, B = scans.shape
    ang = bearings(B, cfg)
    cosb, sinb = _f(np.cos(ang)), _f(np.sin(ang))
    off = np.zeros(T + 1, dtype=np.int64)
    ks, ds, bxs, bys = [], [], [], []
    kk = np.zeros(B, dtype=np.int32)
    dd, bx, by = np.zeros(B), np.zeros(B), np.zeros(B)
    for t in range(T):
        n = lib.oc_filtrar_z(C.byref(c), _p(scans[t]), _p(cosb), _p(sinb), C.c_int64(B), kk.ctypes.data_as(_ip), _p(dd), _p(bx), _p(by))
        off[t + 1] = off[t] + n
        ks.append(kk[:n].copy()); ds.append(dd[:n].copy()); bxs.append(bx[:n].copy()); bys.append(by[:n].copy())
    k = np.concatenate(ks) if ks else np.zeros(0, dtype=np.int32)
    return off, k, np.concatenate(ds), ang[k], np.concatenate(bxs), np.concatenate(bys)


def set_threads(n):
    """OpenMP threads of the sweep (0 = all cores).  Results do not depend on it."""
    load().oc_set_threads(C.c_int(int(n)))


def get_threads():
    return int(load().oc_get_threads())


def set_grid(on):
    """True (default): associate through a uniform grid (exactly the brute-force answer);
    False: the literal scan over every landmark."""
    load().oc_set_grid(C.c_int(int(bool(on))))


def sweep(cfg, kept, u, odo, x0, mapa_viejo, x, lact=None, schedule="sequential", assoc=None):
    """One sweep on prefiltered beams `kept` (from prefilter()).  x (3,T) is updated in place.
    Returns (map (2,K'), counts (L), K', raw (y, counts, lact)) or None if scan 0 is empty.
    assoc: optional dict filled with 'labels' (nnz) and 'targets' (2,nnz) = y[:, c] per kept beam."""
    lib = load()
    c = _cfg(cfg)
    off, k, d, ang, bx, by = kept
    T = odo.shape[1]
    L = int(cfg.L)
    mv = _f(mapa_viejo)
    K = mv.shape[1]
    lact = K if lact is None else int(lact)
    assert x.flags.c_contiguous and x.dtype == np.float64 and x.shape == (3, T)
    mo, co = np.zeros((2, L)), np.zeros(L)
    yr, cr = np.zeros((2, L)), np.zeros(L)
    Ko, la = C.c_int64(0), C.c_int64(0)
    d, ang, bx, by = _f(d), _f(ang), _f(bx), _f(by)
    odo, u, x0 = _f(odo), _f(u), _f(np.asarray(x0, dtype=np.float64).reshape(3))
    nnz = int(off[-1])
    lab = np.zeros(max(nnz, 1), dtype=np.int64) if assoc is not None else None
    tg = np.zeros((2, max(nnz, 1))) if assoc is not None else None
    off = np.ascontiguousarray(off, dtype=np.int64)
    rc = lib.oc_sweep2(C.byref(c), C.c_int64(T), off.ctypes.data_as(_lp), _p(d), _p(ang), _p(bx), _p(by), _p(odo), _p(u),
                       _p(x0), _p(mv), C.c_int64(K), C.c_int64(lact), C.c_int({"sequential": 0, "redblack": 1}[schedule]),
                       _p(x), _p(mo), _p(co), C.byref(Ko), _p(yr), _p(cr), C.byref(la),
                       lab.ctypes.data_as(_lp) if lab is not None else None, _p(tg) if tg is not None else None)
    if assoc is not None:
        assoc["labels"], assoc["targets"] = lab[:nnz], tg[:, :nnz]
    if rc == 1:
        return None
    if rc == -3:
        raise IndexError("oracle: label capacity exceeded or no-beam last pose")
    if rc == -4:
        raise ValueError("oracle: no landmark reached cota")
    return mo[:, :Ko.value].copy(), co, int(Ko.value), (yr, cr, int(la.value))


def init_pass(cfg, kept, u, odo, y0, cnt0, lact0):
    """The causal initialisation pass (scripts/ICM_ROS.py:102-119) from the map the first scan's clustering seeds
    (y0 (2,L), cnt0 (L), lact0: oracle/icm_oracle.py cluster_first_scan).  Returns (x (3,T), raw y (2,L), counts (L), lact)."""
    lib = load()
    c = _cfg(cfg)
    off, k, d, ang, bx, by = kept
    T = odo.shape[1]
    y, cnt = _f(y0).copy(), _f(cnt0).copy()
    la = C.c_int64(int(lact0))
    x = np.zeros((3, T))
    off = np.ascontiguousarray(off, dtype=np.int64)
    d, ang, bx, by, odo, u = _f(d), _f(ang), _f(bx), _f(by), _f(odo), _f(u)
    rc = lib.oc_init_pass(C.byref(c), C.c_int64(T), off.ctypes.data_as(_lp), _p(d), _p(ang), _p(bx), _p(by), _p(odo), _p(u),
                          _p(y), _p(cnt), C.byref(la), _p(x))
    if rc == -3:
        raise IndexError("oracle: label capacity exceeded in the initialisation pass")
    return x, y, cnt, int(la.value)


def solve_one(cfg, two_sided, x_ant, x_pos, u, odo, beams_d_ang, targets):
    lib = load()
    c = _cfg(cfg)
    out = np.zeros(6)
    b = _f(beams_d_ang)
    t = _f(targets)
    d, ang, tx, ty = _f(b[:, 0]), _f(b[:, 1]), _f(t[:, 0]), _f(t[:, 1])
    xp = _f(np.asarray(x_pos).reshape(3)) if x_pos is not None else _f(np.zeros(3))
    lib.oc_solve_one(C.byref(c), C.c_int(int(two_sided)), _p(_f(np.asarray(x_ant).reshape(3))), _p(xp), _p(_f(u)), _p(_f(odo)),
                     _p(d), _p(ang), _p(tx), _p(ty), C.c_int64(len(d)), _p(out))
    return out
