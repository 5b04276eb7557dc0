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
