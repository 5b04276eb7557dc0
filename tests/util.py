"""Shared helpers of the test-suite: golden fixtures, configs."""
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def gold(name):
    return np.load(os.path.join(GOLD, name))


class Cfg:
    """The numeric options of config_ros.yaml / config_default.yaml (identical values)."""

    def __init__(self, **kw):
        self.N = 2
        self.deltat = 0.1
        self.L = 1000
        self.Q = np.eye(2)
        self.R = np.eye(3)
        self.cte_odom = 1.0
        self.cota = 300.0
        self.dist_thr = 1.0
        self.dist_thr_obs = 1.0
        self.rango_laser_max = 10.0
        self.radio = 0.137
        self.angle_increment = None
        for k, v in kw.items():
            setattr(self, k, v)


def dataset():
    """(zz (181,1833) prepared ranges, odometry (3,T), velocities (2,T)) of data_IJAC2018."""
    d = gold("data_IJAC2018.npz")
    z = d["observations"]
    cfg = Cfg()
    zz = np.minimum(z + cfg.radio, z * 0 + cfg.rango_laser_max)  # scripts/sensors_definitions.py:22
    return zz, d["odometry"], d["velocities"]


def label_digest(off, labels):
    """One 64-bit word per pose from the labels of its kept beams (order-sensitive, wrap-around
    arithmetic): sum_j (label_j + 2) * (j * 2654435761 + 0x9E3779B1) over the pose's beams j = 0..n-1.
    Full-size fixtures keep this instead of the 23 M labels themselves."""
    off = np.asarray(off, dtype=np.int64)
    lab = np.asarray(labels).astype(np.int64).astype(np.uint64)
    n = off[1:] - off[:-1]
    if lab.size == 0:
        return np.zeros(n.size, dtype=np.uint64)
    j = (np.arange(lab.size, dtype=np.int64) - np.repeat(off[:-1], n)).astype(np.uint64)
    with np.errstate(over="ignore"):
        w = (lab + np.uint64(2)) * (j * np.uint64(2654435761) + np.uint64(0x9E3779B1))
        h = np.add.reduceat(w, np.minimum(off[:-1], lab.size - 1))
    h[n == 0] = 0
    return h


def hip_runtime():
    """ctypes handle of the HIP runtime this process ALREADY has loaded (the copy libicmslam_hip.so is bound to: the
    PyTorch wheel's or the system's) -- found by its path in /proc/self/maps, never by a bare name that could pull a
    second runtime into the process."""
    import ctypes
    for ln in open("/proc/self/maps"):
        path = ln.split()[-1]
        if "libamdhip64.so" in path and os.path.exists(path):
            return ctypes.CDLL(path)
    raise RuntimeError("no HIP runtime loaded in this process yet")
