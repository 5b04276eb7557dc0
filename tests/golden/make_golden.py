#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the REAL reference.

Runs only in the build container, where the reference checkout is mounted at
/root/reference (it never travels to the GPU box).  It imports the reference's
own modules (scripts/ICM_SLAM_tools.py, scripts/ICM_ROS.py) with an empty stub
for the missing `roslibpy` dependency, drives the ROS-free offline path of
SURVEY.md Appendix D on scripts/data_IJAC2018.mat, and dumps inputs/outputs
as small .npz fixtures.  Nothing of the reference's source text is stored:
the fixtures are arrays (inputs and expected outputs) only.

    python tests/golden/make_golden.py            # sweeps 1,2 (about 1 min)
    python tests/golden/make_golden.py --long     # also sweep 30 (about 8 min)

Versions that produced the committed fixtures are recorded inside each file
(`numpy_version`, `scipy_version`); the reference pins numpy 1.19.5 /
scipy 1.5.4 (scripts/requisitos.txt:13,21) which are not installable here.
"""
import argparse
import os
import sys
import types
import warnings

os.environ.setdefault("MPLBACKEND", "Agg")
os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True

import numpy as np
import scipy
import scipy.io as sio

REF = "/root/reference/scripts"
HERE = os.path.dirname(os.path.abspath(__file__))


def load_reference():
    sys.modules["roslibpy"] = types.ModuleType("roslibpy")
    sys.path.insert(0, REF)
    cwd = os.getcwd()
    os.chdir(REF)
    try:
        import ICM_SLAM_tools as tools  # noqa
        import ICM_ROS as icmros  # noqa
    finally:
        os.chdir(cwd)
    return tools, icmros


def ragged(list_of_2d, ncol):
    off = np.zeros(len(list_of_2d) + 1, dtype=np.int64)
    for i, a in enumerate(list_of_2d):
        off[i + 1] = off[i] + (a.shape[0] if a.ndim == 2 else 0)
    flat = np.zeros((off[-1], ncol))
    for i, a in enumerate(list_of_2d):
        if a.ndim == 2 and a.shape[0]:
            flat[off[i]:off[i + 1]] = a
    return off, flat


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--long", action="store_true", help="also run to sweep 30")
    args = ap.parse_args()
    warnings.simplefilter("ignore")
    tools, icmros = load_reference()
    ver = dict(numpy_version=np.__version__, scipy_version=scipy.__version__)

    cfg = tools.ConfigICM(os.path.join(REF, "config_ros.yaml"))
    mat = sio.loadmat(os.path.join(REF, "data_IJAC2018.mat"))
    z = np.asarray(mat["observations"], dtype=np.float64)
    odo = np.asarray(mat["odometry"], dtype=np.float64)
    u = np.asarray(mat["velocities"], dtype=np.float64)
    # (1) dataset as a data fixture
    np.savez_compressed(os.path.join(HERE, "data_IJAC2018.npz"),
                        observations=z, odometry=odo, velocities=u)
    # range preparation as scripts/sensors_definitions.py:22
    zz = np.minimum(z + cfg.radio, z * 0 + cfg.rango_laser_max)
    T = zz.shape[1]

    # (2) filtrar_z on every scan
    fz = [tools.filtrar_z(zz[:, t].copy(), cfg) for t in range(T)]
    off, flat = ragged(fz, 4)
    np.savez_compressed(os.path.join(HERE, "filtrar_z.npz"), offsets=off, rows=flat, **ver)

    # (3) init pass, ROS-free restatement of the driver loop scripts/ICM_ROS.py:57-100
    copy = icmros.copy
    icm = icmros.ICM_ROS(cfg)
    icm.mediciones, icm.u, icm.odometria = zz, u, odo
    icm.x0 = np.array([odo[:, 0]]).T
    xt = copy(icm.x0)
    x = copy(icm.x0)
    y = np.zeros((2, cfg.L))
    icm.mapa_obj = tools.Mapa(cfg)
    z0 = tools.filtrar_z(zz[:, 0].copy(), cfg)
    zt = tools.tras_rot_z(xt, z0)
    y, c = icm.mapa_obj.actualizar(y, y, zt[:, 2:4])
    init_c0 = c.copy()
    for t in range(1, T):
        icm.t = t
        y, xt = icm.inicializar_online_process(y, xt)
        xt = np.reshape(xt, (3, 1))
        x = np.concatenate((x, xt), axis=1)
    init_y_raw = y.copy()
    init_cnt_raw = icm.mapa_obj.cant_obs_i.copy()
    init_lact_raw = int(icm.mapa_obj.landmarks_actuales)
    yy = icm.mapa_obj.filtrar(y)
    yy = yy[:, :icm.mapa_obj.landmarks_actuales]
    map_init = copy(yy)
    x_init = copy(x)
    np.savez_compressed(os.path.join(HERE, "init_pass.npz"), x_init=x_init, map_init=map_init,
                        cant_obs_i=icm.mapa_obj.cant_obs_i.copy(),
                        landmarks_actuales=int(icm.mapa_obj.landmarks_actuales),
                        labels_scan0=init_c0, y_raw=init_y_raw, cant_obs_raw=init_cnt_raw,
                        landmarks_raw=init_lact_raw, **ver)
    print("init: landmarks", map_init.shape[1], "x_init[:,100]", x_init[:, 100])

    # (4) unit pin: single solve t=100 against map_init/x_init
    t = 100
    z100 = tools.filtrar_z(zz[:, t].copy(), cfg)
    zt100 = tools.tras_rot_z(x_init[:, t], z100.copy())
    d = icmros.cdist(map_init.T, zt100[:, 2:4])
    c100 = np.argmin(d, axis=0)
    icm.x_ant = x_init[:, t - 1].reshape((3, 1))
    icm.x_pos = x_init[:, t + 1].reshape((3, 1))
    icm.xt = x_init[:, t - 1].reshape((3, 1))
    icm.t = t
    icm.medicion_actual = z100[:, 0:2]
    icm.mapa_visto = map_init[:, c100].T
    start = ((icm.x_ant + icm.x_pos) / 2.0).reshape(3)
    h_start = float(icm.h(start.reshape((3, 1)), z100[:, 0:2]))
    f_start = float(np.asarray(icm.fun_xn(start)).reshape(-1)[0])
    from scipy.optimize import fmin
    xopt, fopt, nit, nfev, _ = fmin(icm.fun_xn, start, xtol=0.001, disp=0, full_output=1)
    np.savez_compressed(os.path.join(HERE, "solve_t100.npz"), labels=c100, beams=z100,
                        targets=icm.mapa_visto, start=start, h_start=h_start, f_start=f_start,
                        xopt=xopt, fopt=float(np.asarray(fopt).reshape(-1)[0]), nit=nit,
                        nfev=nfev, **ver)
    print("solve t=100:", xopt, fopt, nit, nfev)

    # (5)+(6) sweeps with per-pose instrumentation of sweep 1
    rec = dict(labels=[], targets=[], starts=[], xout=[], t=[])
    orig_act = tools.Mapa.actualizar
    orig_fmin = icmros.fmin
    state = {"on": False, "last_c": None}

    def act_wrap(self, mapa, mapa_ref, obs):
        m, cc = orig_act(self, mapa, mapa_ref, obs)
        if state["on"]:
            state["last_c"] = cc.copy()
            state["last_y"] = m[:, cc].T.copy()
        return m, cc

    def fmin_wrap(f, x0, **kw):
        xo, fo, nit_, nfev_, _ = orig_fmin(f, x0, full_output=1, **kw)
        if state["on"]:
            rec["t"].append(icm.t)
            rec["labels"].append(state["last_c"])
            rec["targets"].append(state["last_y"])
            rec["starts"].append(np.asarray(x0, dtype=float).reshape(3).copy())
            rec["xout"].append(np.r_[xo, float(np.asarray(fo).reshape(-1)[0]), nit_, nfev_])
        return xo

    tools.Mapa.actualizar = act_wrap
    icmros.fmin = fmin_wrap
    orig_filtrar = tools.Mapa.filtrar
    filt = {}

    def filtrar_wrap(self, mapa):
        if state["on"]:
            filt["y_in"] = mapa.copy()
            filt["cnt_in"] = self.cant_obs_i.copy()
            filt["lact_in"] = int(self.landmarks_actuales)
        out = orig_filtrar(self, mapa)
        if state["on"]:
            filt["y_out"] = out.copy()
            filt["cnt_out"] = self.cant_obs_i.copy()
            filt["lact_out"] = int(self.landmarks_actuales)
        return out

    tools.Mapa.filtrar = filtrar_wrap

    mapa_viejo = copy(map_init)
    x = copy(x_init)
    nsw = 30 if args.long else 2
    keep = {1, 2, 30}
    for it in range(1, nsw + 1):
        state["on"] = (it == 1)
        mapa, x = icm.iterations_process_offline(mapa_viejo, x)
        state["on"] = False
        if it == 1:
            loff, lflat = ragged([np.asarray(a, dtype=float).reshape(-1, 1) for a in rec["labels"]], 1)
            _, tflat = ragged(rec["targets"], 2)
            np.savez_compressed(os.path.join(HERE, "sweep1_perpose.npz"),
                                t=np.array(rec["t"]), offsets=loff,
                                labels=lflat[:, 0].astype(np.int64), targets=tflat,
                                starts=np.array(rec["starts"]), solves=np.array(rec["xout"]),
                                filtrar_y_in=filt["y_in"][:, :filt["lact_in"]],
                                filtrar_cnt_in=filt["cnt_in"][:filt["lact_in"]],
                                filtrar_lact_in=filt["lact_in"],
                                filtrar_y_out=filt["y_out"][:, :filt["lact_out"]],
                                filtrar_cnt_out=filt["cnt_out"][:filt["lact_out"]],
                                filtrar_lact_out=filt["lact_out"], **ver)
        if it in keep:
            np.savez_compressed(os.path.join(HERE, "sweep%02d.npz" % it), x=x.copy(), mapa=mapa.copy(),
                                cant_obs_i=icm.mapa_obj.cant_obs_i.copy(),
                                landmarks_actuales=int(icm.mapa_obj.landmarks_actuales), **ver)
            print("sweep", it, "x[:,-1]", x[:, -1], "sum|dx|", np.abs(x - x_init).sum())
        mapa_viejo = copy(mapa)


if __name__ == "__main__":
    main()
