"""Drop-in counterpart of the reference's `scripts/ICM_ROS.py`: class `ICM_ROS` with the same
constructor, attributes and methods, whose offline sweep
    mapa_refinado, x = ICM.iterations_process_offline(mapa_viejo, x)
(reference scripts/ICM_ROS.py:121-164) runs in hand-written HIP kernels on an MI355X through
the C-ABI of include/icmslam.h.  `x` is updated in place and returned, `mapa_viejo` is not
modified, the returned map is a fresh (2,K') array -- exactly the reference's contract.

There is no CPU path: without the built library or without a GPU the calls raise.
"""
import os
from copy import deepcopy as copy

import numpy as np

from ICM_SLAM_tools import *  # noqa: F401,F403  (the reference star-imports its tools too)
from ICM_SLAM_tools import ConfigICM, Mapa, ROS
from icmslam_hip.engine import SweepEngine


class ICM_ROS(ROS):
    def __init__(self, config, x0=""):
        super().__init__()
        if isinstance(x0, str) and x0 == "":
            self.x0 = np.zeros((3, 1))
        else:
            self.x0 = x0
        self.config = config
        self.new_data = 0
        self.odometria = np.array([])
        self.mediciones = np.array([])
        self.u = np.array([])
        self.iterations_flag = False
        self.seq0 = 0
        self.seq = 0
        self.debug = False
        self.device = int(os.environ.get("LOCAL_RANK", "0")) if os.environ.get("ICMSLAM_DEVICE") is None \
            else int(os.environ["ICMSLAM_DEVICE"])
        self._engine = None
        self._seq_key = None

    # ------------------------------------------------------------------------------------
    # data + initial state (ROS-free counterparts of inicializar_online)
    # ------------------------------------------------------------------------------------
    def load_data(self, file=None):
        """Offline loader keyed on `config.file` (`ICM_method.load_data` of the reference's
        legacy API, scripts/ICM_SLAM_old.py:249): a MATLAB v5 `.mat` or an `.npz` holding
        `observations` (B,T), `odometry` (3,T), `velocities` (2,T).  Ranges are inflated by
        the trunk radius and clipped like the Lidar callback does
        (reference scripts/sensors_definitions.py:22)."""
        file = file or self.config.file
        cands = [file, os.path.join(os.path.dirname(os.path.abspath(__file__)), file)]
        path = next((c for c in cands if os.path.exists(c)), None)
        if path is None:
            stem = os.path.splitext(file)[0]
            alt = [stem + ".npz", os.path.join(os.path.dirname(os.path.abspath(__file__)), os.path.basename(stem) + ".npz")]
            path = next((c for c in alt if os.path.exists(c)), None)
        if path is None:
            raise FileNotFoundError(file)
        if path.endswith(".npz"):
            d = np.load(path)
        else:
            import scipy.io as sio
            d = sio.loadmat(path)
        z = np.asarray(d["observations"], dtype=np.float64)
        self.odometria = np.asarray(d["odometry"], dtype=np.float64)
        self.u = np.asarray(d["velocities"], dtype=np.float64)
        self.mediciones = np.minimum(z + self.config.radio, z * 0.0 + self.config.rango_laser_max)
        self.x0 = np.array([self.odometria[:, 0]]).T
        self._seq_key = None
        return self.mediciones, self.odometria, self.u

    def load_messages(self, lidar, odometria):
        """Sequence from the buffers of the two topic subscribers (`Lidar`, `Odometria` of
        sensors_definitions.py, filled live over rosbridge or by matlab2ros.replay): samples are
        paired by sequence number, scans become the columns of `mediciones`, poses and twists the
        columns of `odometria` / `u` -- the arrays `inicializar_online` accumulates
        (reference scripts/ICM_ROS.py:66-93) and `load_data()` reads from a file."""
        scans = {m['seq']: m['data'] for m in lidar.msgs}
        odos = {m['seq']: m['data'] for m in odometria.msgs}
        seqs = sorted(set(scans) & set(odos))
        if not seqs:
            raise ValueError("load_messages: no sample with both a scan and an odometry message")
        self.mediciones = np.ascontiguousarray(np.hstack([scans[k].reshape(-1, 1) for k in seqs]), dtype=np.float64)
        self.odometria = np.ascontiguousarray(np.hstack([odos[k]['odo'].reshape(3, 1) for k in seqs]), dtype=np.float64)
        self.u = np.ascontiguousarray(np.hstack([odos[k]['u'].reshape(2, 1) for k in seqs]), dtype=np.float64)
        self.x0 = np.array([self.odometria[:, 0]]).T
        self._seq_key = None
        return self.mediciones, self.odometria, self.u

    def set_initial_state(self, positions, mapa, cant_obs_i=None):
        """Install the result of an initialisation pass: poses (3,T) and landmark map (2,K)
        (what `inicializar_online` leaves in `self.positions` / `self.mapa_viejo`,
        reference scripts/ICM_ROS.py:95-100)."""
        self.positions = np.array(positions, dtype=np.float64)
        self.mapa_viejo = np.array(mapa, dtype=np.float64)
        self.mapa_obj = Mapa(self.config)
        self.mapa_obj.landmarks_actuales = self.mapa_viejo.shape[1]
        if cant_obs_i is not None:
            self.mapa_obj.cant_obs_i = np.array(cant_obs_i, dtype=np.float64)
        self.iterations_flag = True

    def inicializar_offline(self):
        """ROS-free `inicializar_online` (reference scripts/ICM_ROS.py:47-100) on the loaded
        sequence: the same causal pass -- first scan clustered into the first landmarks, then
        predict / associate against the running map / one-sided solve per sample -- run by the
        HIP library, followed by `Mapa.filtrar`.  Leaves `self.mapa_viejo`, `self.positions`
        and `self.mapa_obj` exactly where the reference leaves them."""
        if self.mediciones.size == 0:
            self.load_data()
        self.x0 = np.array([self.odometria[:, 0]]).T
        eng = self._get_engine()
        x, y, cnt, lact, _ = eng.init_pass(self.x0)
        self.mapa_obj = Mapa(self.config)
        self.mapa_obj.landmarks_actuales = lact
        self.mapa_obj.cant_obs_i = cnt
        yy = self.mapa_obj.filtrar(y)
        yy = yy[:, :self.mapa_obj.landmarks_actuales]
        self.mapa_viejo = copy(yy)
        self.positions = copy(x)
        self.iterations_flag = True

    def inicializar_online(self):
        """The live rosbridge pass of the reference (scripts/ICM_ROS.py:47-100) is sensor
        I/O and outside this build; recorded data go through load_data() +
        set_initial_state()."""
        self.connect_ros()

    # ------------------------------------------------------------------------------------
    # the hot path
    # ------------------------------------------------------------------------------------
    def _sequence_key(self):
        """Identity of the uploaded sequence: the array objects, their shapes and buffer addresses, and a checksum of a
        strided sample of each (a few microseconds per call).  The reference reads `mediciones`, `odometria` and `u`
        afresh on every call (scripts/ICM_ROS.py:127-158); here they are uploaded and pre-filtered once, so REPLACING any
        of them (also by a same-shaped array at a recycled address) re-uploads, and so does an in-place edit that touches
        a sampled column (16 scans, every 256th odometry / control sample, first and last always) -- an in-place edit
        elsewhere must be announced with invalidate_sequence().  (Documented in INTEGRATION.md; a full checksum of
        `mediciones` would read 576 MB per sweep at the headline size.)"""
        m, o, u = self.mediciones, self.odometria, self.u

        def chk(a, cols):
            if a.ndim != 2 or not a.size:
                return 0.0
            step = max(1, a.shape[1] // cols)
            return float(a[:, ::step].sum()) + float(a[:, -1].sum())

        return (id(m), id(o), id(u), m.shape, o.shape, u.shape,
                m.__array_interface__["data"][0], o.__array_interface__["data"][0], u.__array_interface__["data"][0],
                chk(m, 16), chk(o, 256), chk(u, 256))

    def _get_engine(self):
        key = self._sequence_key()
        if self._engine is None:
            self._engine = SweepEngine(self.config, self.device)
        if key != self._seq_key:
            self._engine.upload(self.mediciones, self.odometria, self.u)
            self._seq_key = key
        return self._engine

    def attach_engine(self, engine, mediciones, odometria, u):
        """Adopt a SweepEngine that already holds this sequence in HBM (uploaded and pre-filtered
        by the caller) instead of uploading it again."""
        self.mediciones, self.odometria, self.u = mediciones, odometria, u
        if (engine.B, engine.T) != tuple(np.shape(mediciones)) or engine.nloc != engine.T:
            raise ValueError("attach_engine: the engine holds a different sequence")
        self._engine = engine
        self._seq_key = self._sequence_key()

    def invalidate_sequence(self):
        """Force a re-upload (and a new scan pre-filter) on the next sweep, e.g. after
        editing `mediciones` in place."""
        self._seq_key = None

    def _check_overrides(self):
        for name in ("h", "g", "fun_x", "fun_xn", "minimizar_x", "minimizar_xn"):
            if getattr(type(self), name) is not getattr(ICM_ROS, name):
                raise NotImplementedError(
                    "%s.%s overrides the built-in model: Python callbacks cannot run inside the HIP "
                    "kernels and this build has no CPU path (reference scripts/ICM_ROS.py:166-169)"
                    % (type(self).__name__, name))

    def iterations_process_offline(self, mapa_viejo, x):
        """One offline ICM sweep over the whole sequence (reference scripts/ICM_ROS.py:121-164)."""
        self._check_overrides()
        eng = self._get_engine()
        if not hasattr(self, "mapa_obj"):
            self.mapa_obj = Mapa(self.config)
            self.mapa_obj.landmarks_actuales = mapa_viejo.shape[1]
        self.mapa_obj.clear_obs()
        inplace = isinstance(x, np.ndarray) and x.dtype == np.float64 and x.flags.c_contiguous
        xw = x if inplace else np.ascontiguousarray(x, dtype=np.float64)
        res = eng.sweep(mapa_viejo, xw, np.asarray(self.x0, dtype=np.float64).reshape(3),
                        self.mapa_obj.landmarks_actuales, getattr(self.config, "schedule", "sequential"))
        if res is None:  # scan 0 without observations: the reference returns its inputs
            return mapa_viejo, x
        if not inplace:
            x[...] = xw
        mo, co, K = res
        self.mapa_obj.cant_obs_i = co
        self.mapa_obj.landmarks_actuales = K
        return mo[:, :K].copy(), x

    # ------------------------------------------------------------------------------------
    # the model functions "the user can configure" (reference scripts/ICM_ROS.py:166-278),
    # evaluated on the GPU with the same stashed-state calling convention
    # ------------------------------------------------------------------------------------
    def _beams_xy(self, zt):
        zt = np.asarray(zt, dtype=np.float64)
        return zt[:, 0] * np.cos(zt[:, 1]), zt[:, 0] * np.sin(zt[:, 1])

    def g(self, xt, ut):
        xt = np.asarray(xt, dtype=np.float64).reshape((3, 1))
        ut = np.asarray(ut, dtype=np.float64).reshape((2, 1))
        S = np.array([[np.cos(xt[2])[0], 0.0], [np.sin(xt[2])[0], 0.0], [0.0, 1.0]])
        return (xt + self.config.deltat * np.matmul(S, ut).reshape((3, 1))).reshape((3, 1))

    def h(self, xt, zt):
        bx, by = self._beams_xy(zt)
        y = np.asarray(self.mapa_visto, dtype=np.float64)
        eng = self._get_engine()
        return eng.energy_one(2, xt, np.zeros(3), None, np.zeros(2), np.zeros((3, 2)), bx, by, y[:, 0], y[:, 1])

    def _pose_args(self, two_sided):
        t = self.t
        bx, by = self._beams_xy(self.medicion_actual)
        y = np.asarray(self.mapa_visto, dtype=np.float64)
        x_ant = np.asarray(self.xt, dtype=np.float64).reshape(3)
        if two_sided:
            return (x_ant, np.asarray(self.x_pos, dtype=np.float64).reshape(3), self.u[:, t - 1:t + 1],
                    self.odometria[:, t - 1:t + 2], bx, by, y[:, 0], y[:, 1])
        return (x_ant, None, self.u[:, t - 1:t], self.odometria[:, t - 1:t + 1], bx, by, y[:, 0], y[:, 1])

    def fun_xn(self, x):
        return self._get_engine().energy_one(1, x, *self._pose_args(True))

    def fun_x(self, x):
        return self._get_engine().energy_one(0, x, *self._pose_args(False))

    def minimizar_xn(self, medicion_actual, mapa_visto, x, t):
        self.x_ant = x[:, t - 1].reshape((3, 1))
        self.x_pos = x[:, t + 1].reshape((3, 1))
        self.xt = x[:, t - 1].reshape((3, 1))
        self.t = t
        self.medicion_actual = medicion_actual
        self.mapa_visto = mapa_visto
        return self._get_engine().solve_one(1, *self._pose_args(True))[:3].copy()

    def minimizar_x(self, medicion_actual, mapa_visto):
        self.medicion_actual = medicion_actual
        self.mapa_visto = mapa_visto
        return self._get_engine().solve_one(0, *self._pose_args(False))[:3].copy()


class ICM_method(ICM_ROS):
    """Names of the reference's legacy API (`ICM_method`, scripts/ICM_SLAM_old.py:59; the
    file itself no longer runs) mapped onto the same HIP sweep."""

    def __init__(self, config, x0=""):
        ICM_ROS.__init__(self, config, x0)

    def inicializar(self, positions, mapa, cant_obs_i=None):
        self.set_initial_state(positions, mapa, cant_obs_i)

    def itererar(self, mapa_viejo, x):
        return self.iterations_process_offline(mapa_viejo, x)


if __name__ == "__main__":
    import sys
    config = ConfigICM(sys.argv[1] if len(sys.argv) > 1 else "config_ros.yaml")
    ICM = ICM_ROS(config)
    ICM.load_data()
    if len(sys.argv) > 2:
        init = np.load(sys.argv[2])
        ICM.set_initial_state(init["x_init"], init["map_init"])
    else:
        ICM.inicializar_offline()
    mapa_viejo = copy(ICM.mapa_viejo)
    x = copy(ICM.positions)
    for iteracionICM in range(config.N):
        print("iteración ICM : ", iteracionICM + 1)
        mapa_refinado, x = ICM.iterations_process_offline(mapa_viejo, x)
        print("Correccion: ", np.linalg.norm(x - ICM.positions, axis=1).sum())
        print("cambios (min, max, medio): ", calc_cambio(mapa_refinado, mapa_viejo))  # noqa: F405
        mapa_viejo = copy(mapa_refinado)
