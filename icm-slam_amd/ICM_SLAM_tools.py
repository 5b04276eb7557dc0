"""Drop-in counterpart of the reference's `scripts/ICM_SLAM_tools.py` (== `scripts/ICM_SLAM.py`)
for the MI355X-native build: the same names (`ConfigICM`, `Mapa`, `ROS`, `Sensor`, `filtrar_z`,
`tras_rot_z`, `Rota`, `entrepi`, `calc_cambio`), the same argument meaning
and error behaviour, so `from ICM_SLAM_tools import *` in a driver keeps working.

What is NOT here: the reference's matplotlib helpers (`graficar*`: visualisation, out of scope,
SURVEY section 2) and the arithmetic of the sweep.  `filtrar_z`, the association/running-mean of
`Mapa.actualizar` and the map prune/merge of `Mapa.filtrar` run inside the HIP library
(`icmslam_hip`); the methods below are thin calls into it.  There is no NumPy fallback.
"""
import os
from copy import deepcopy as copy  # noqa: F401  (re-exported like the reference)

import numpy as np
import yaml

from icmslam_hip import engine as _engine

_DEFAULTS = {
    # keys config_default.yaml lacks although the reference's ConfigICM reads them
    # unconditionally (scripts/ICM_SLAM_tools.py:92-99 -> KeyError)
    "topic_laser": "/pioneer2dx/laser/scan_Lidar_horizontal",
    "topic_laser_msg": "sensor_msgs/LaserScan",
    "topic_odometry": "/pioneer2dx/ground_truth/odom",
    "topic_odometry_msg": "nav_msgs/Odometry",
    "file": "data_IJAC2018.mat",
    "time": 275.0,
    # extensions of this build (absent = reference behaviour)
    "angle_increment": None,   # beam pitch [rad]; None = the reference's hard-coded 1 degree
    "schedule": "sequential",  # 'sequential' = reference pose order, 'redblack' = parallel
}


class ConfigICM:
    """All configuration parameters (reference scripts/ICM_SLAM_tools.py:60-102).
    `ConfigICM(configFile)` reads the YAML's top-level `D` mapping; `ConfigICM(D={...})`
    takes the mapping directly."""

    def __init__(self, configFile="config_default.yaml", D={}):
        if not D:
            if not os.path.isabs(configFile) and not os.path.exists(configFile):
                here = os.path.join(os.path.dirname(os.path.abspath(__file__)), configFile)
                if os.path.exists(here):
                    configFile = here
            with open(configFile, "r") as arch:
                D = yaml.load(arch, Loader=yaml.FullLoader)["D"]
        D = dict(_DEFAULTS, **D)
        self.N = D["N"]
        self.deltat = D["deltat"]
        self.L = D["L"]
        self.Q = np.eye(2)
        self.Q[0, 0] = D["Q"][0]
        self.Q[1, 1] = D["Q"][1]
        self.R = np.eye(3)
        self.R[0, 0] = D["R"][0]
        self.R[1, 1] = D["R"][1]
        self.R[2, 2] = D["R"][2]
        self.cte_odom = D["cte_odom"]
        self.cota = D["cota"]
        self.dist_thr = D["dist_thr"]
        self.dist_thr_obs = D["dist_thr_obs"]
        self.rango_laser_max = D["rango_laser_max"]
        self.radio = D["radio"]
        self.topic_laser = D["topic_laser"]
        self.topic_laser_msg = D["topic_laser_msg"]
        self.topic_odometry = D["topic_odometry"]
        self.topic_odometry_msg = D["topic_odometry_msg"]
        self.file = D["file"]
        self.time = D["time"]
        self.angle_increment = D["angle_increment"]
        self.schedule = D["schedule"]

    def set_Tf(self, Tf):
        self.Tf = Tf


class Mapa:
    """State of the landmark map between sweeps (reference scripts/ICM_SLAM_tools.py:104-265):
    `landmarks_actuales`, `cant_obs_i`.  Inside a sweep the per-scan update is fused into the HIP
    kernels; `actualizar` is the same step for one scan (`icm_associate`), `filtrar` the library's
    map prune/merge."""

    def __init__(self, config):
        self.config = config
        self.L = config.L
        self.cota = config.cota
        self.dist_thr = config.dist_thr
        self.landmarks_actuales = 0
        self.clear_obs()

    def clear_obs(self):
        # landmarks_actuales survives, like the reference (scripts/ICM_SLAM_tools.py:119-126)
        self.cant_obs_i = np.zeros(self.L)

    def filtrar(self, mapa):
        """Drop landmarks seen fewer than `cota` times, merge nearest neighbours closer than
        `dist_thr` (count-weighted), renumber.  Returns the (2,L) zero-padded map."""
        yo, co, lact = _engine.filtrar_map(self.config, mapa, self.cant_obs_i, self.landmarks_actuales)
        self.landmarks_actuales = lact
        self.cant_obs_i = co
        return yo

    def actualizar(self, mapa, mapa_referencia, obs):
        """One scan into the running map (reference scripts/ICM_SLAM_tools.py:128-201): `obs` (n,2)
        world points are matched to the columns of `mapa_referencia[:, :landmarks_actuales]` (the
        first call, with no landmark yet, clusters the scan instead), gated at `dist_thr`; `mapa`
        (2,L) and `cant_obs_i` take the running-mean update.  Returns (mapa, c) like the reference:
        `mapa` is updated in place and returned, `c` are the labels.  Inside a sweep this step is
        fused into the phase A/B kernels; this is the per-scan call of the online initialisation
        (scripts/ICM_ROS.py:114), through `icm_associate`."""
        inplace = isinstance(mapa, np.ndarray) and mapa.dtype == np.float64 and mapa.flags.c_contiguous
        mw = mapa if inplace else np.ascontiguousarray(mapa, dtype=np.float64)
        if not (isinstance(self.cant_obs_i, np.ndarray) and self.cant_obs_i.dtype == np.float64 and self.cant_obs_i.flags.c_contiguous):
            self.cant_obs_i = np.ascontiguousarray(self.cant_obs_i, dtype=np.float64)
        c, lact = _engine.actualizar_scan(self.config, mw, mapa_referencia, obs, self.landmarks_actuales, self.cant_obs_i)
        if not inplace:
            mapa[...] = mw
        self.landmarks_actuales = lact
        return mapa, c


def filtrar_z(z, config):
    """Scan pre-filter (reference scripts/ICM_SLAM_tools.py:22-58): 3-tap median, max-range cut,
    isolated-beam rejection; rows [d, ang, d cos ang, d sin ang].  Runs the HIP kernel the
    sweep uses (`k_prefilter`)."""
    return _engine.prefilter_scans(config, np.asarray(z, dtype=np.float64).reshape(-1))[0]


def tras_rot_z(x, z):
    """Body -> world transform of the kept beams, in place on columns 2:4 like the reference
    (scripts/ICM_SLAM_tools.py:465-480).  Inside a sweep this is fused into `k_assoc_group`."""
    x = np.asarray(x, dtype=np.float64).reshape(3)
    ct = np.cos(x[2] - np.pi / 2.0)
    st = np.sin(x[2] - np.pi / 2.0)
    bx, by = z[:, 2].copy(), z[:, 3].copy()
    z[:, 2] = (bx * ct - by * st) + x[0]
    z[:, 3] = (bx * st + by * ct) + x[1]
    return z


def entrepi(angulo):
    """Equivalent angle in (-pi, pi] (reference scripts/ICM_SLAM_tools.py:455-463)."""
    angulo = np.mod(angulo, 2 * np.pi)
    if angulo > np.pi:
        angulo = angulo - 2 * np.pi
    return angulo


def Rota(theta):
    """2-D rotation [[c, s], [-s, c]] (reference scripts/ICM_SLAM_tools.py:482-488)."""
    return np.array([[np.cos(theta), np.sin(theta)], [-np.sin(theta), np.cos(theta)]])


def calc_cambio(y, mapa_viejo):
    """min / max / mean nearest-neighbour displacement of the map between two sweeps
    (reference scripts/ICM_SLAM_tools.py:490-495)."""
    d = np.sqrt(((mapa_viejo.T[:, None, :] - y.T[None, :, :]) ** 2).sum(axis=2))
    md = d.min(axis=0)
    return md.min(), md.max(), md.mean()


class ROS:
    """Placeholder of the rosbridge client (reference scripts/ICM_SLAM_tools.py:267-341).
    The ROS bridge is sensor I/O, outside the accelerated path; connecting needs `roslibpy`."""

    def __init__(self):
        pass

    def connect_ros(self):
        raise NotImplementedError("the rosbridge client is not part of the MI355X sweep build; "
                                  "use ICM_ROS.load_data()/inicializar_offline() for recorded data")

    def disconnect_ros(self):
        pass


class Sensor:
    """Message buffer of one ROS topic (reference scripts/ICM_SLAM_tools.py:343-449); kept so
    that `Lidar`/`Odometria` definitions import, not used by the offline path."""

    def __init__(self, config="", name="name", topic="", topic_msg="", principalCallback=""):
        self.msgs = []
        self.value = np.array([])
        self.k0 = 0
        self.config = config
        self.name = name
        self.topic = topic
        self.topic_msg = topic_msg
        self.principalCallback = principalCallback
        self.c = 0
