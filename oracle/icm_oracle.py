"""CPU oracle for the offline ICM sweep -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A NumPy restatement of the reference's hot path (Seba-san/icm-slam,
`scripts/ICM_ROS.py:121-278` and the helpers in `scripts/ICM_SLAM_tools.py`),
written from the maths in SURVEY.md Appendix A.  Only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this
module, and only as the checker.  The product path (`icm-slam_amd/`) never
imports it and has no CPU fallback.

Parity pin: every function here is checked bit-for-bit against golden vectors
produced by importing the real reference in the build container
(`tests/golden/make_golden.py`, numpy 2.2.6 / scipy 1.15.3): filtrar_z rows,
per-pose labels/targets/solver outputs of sweep 1, poses and maps after sweeps
1, 2 and 30 (`tests/test_oracle_golden.py`).  The reference itself has no
tests, so those goldens are the only pin.

The optimiser is a restatement of SciPy's Nelder-Mead as called by
`scipy.optimize.fmin(f, x0, xtol=1e-3, disp=0)` (scipy 1.15.3,
`scipy/optimize/_optimize.py::_minimize_neldermead`; the reference pins
scipy==1.5.4 in `scripts/requisitos.txt:21`, same algorithm).
"""
from __future__ import annotations

import math

import numpy as np

TWO_PI = 2 * np.pi
HALF_PI = np.pi / 2.0


class OracleConfig:
    """Numeric options of `ConfigICM` (reference scripts/ICM_SLAM_tools.py:60-102)."""

    def __init__(self, deltat=0.1, L=1000, Q=(1.0, 1.0), R=(1.0, 1.0, 1.0), cte_odom=1.0,
                 cota=300.0, dist_thr=1.0, rango_laser_max=10.0, radio=0.137,
                 angle_increment=None):
        self.deltat = float(deltat)
        self.L = int(L)
        self.Q = np.diag(np.asarray(Q, dtype=float))
        self.R = np.diag(np.asarray(R, dtype=float))
        self.cte_odom = float(cte_odom)
        self.cota = float(cota)
        self.dist_thr = float(dist_thr)
        self.rango_laser_max = float(rango_laser_max)
        self.radio = float(radio)
        self.angle_increment = angle_increment  # None = the reference's hard-coded 1 degree

    @classmethod
    def from_config(cls, c):
        return cls(deltat=c.deltat, L=c.L, Q=np.diag(c.Q), R=np.diag(c.R), cte_odom=c.cte_odom,
                   cota=c.cota, dist_thr=c.dist_thr, rango_laser_max=c.rango_laser_max,
                   radio=c.radio, angle_increment=getattr(c, "angle_increment", None))


def bearings(idx, cfg):
    """Beam bearing of scan row `idx`.  Reference: index*pi/180 (1 degree hard-coded,
    scripts/ICM_SLAM_tools.py:44,51); `angle_increment` is this build's extension."""
    idx = np.asarray(idx)
    if cfg.angle_increment is None:
        return idx * np.pi / 180.0
    return idx * float(cfg.angle_increment)


# ----------------------------------------------------------------------------------------
# geometry helpers
# ----------------------------------------------------------------------------------------
def entrepi(a):
    """Wrap to (-pi, pi].  Reference scripts/ICM_SLAM_tools.py:455-463."""
    a = np.mod(a, TWO_PI)
    if a > np.pi:
        a = a - TWO_PI
    return a


def rota(theta):
    """[[c, s], [-s, c]].  Reference scripts/ICM_SLAM_tools.py:482-488."""
    c = np.cos(theta)
    s = np.sin(theta)
    return np.array([[c, s], [-s, c]])


def project_beams(pose, body_xy):
    """Body -> world projection of the kept beams (reference `tras_rot_z`,
    scripts/ICM_SLAM_tools.py:465-480).  Pure: returns the (n,2) world points."""
    pose = np.asarray(pose, dtype=float).reshape(3)
    ct = np.cos(pose[2] - HALF_PI)
    st = np.sin(pose[2] - HALF_PI)
    rot = np.array([[ct, st], [-st, ct]])
    return np.matmul(body_xy, rot) + pose[0:2].reshape(1, 2)


def _pairwise_dist(a, b):
    """Euclidean distances between rows of a (m,2) and b (n,2): sqrt(dx*dx + dy*dy), the
    arithmetic of scipy's cdist/pdist 'euclidean' (reference uses them at
    scripts/ICM_SLAM_tools.py:46,169,241)."""
    dx = a[:, None, 0] - b[None, :, 0]
    dy = a[:, None, 1] - b[None, :, 1]
    return np.sqrt(dx * dx + dy * dy)


# ----------------------------------------------------------------------------------------
# a2: scan pre-filter
# ----------------------------------------------------------------------------------------
def median3(z):
    """3-tap median with zero padding (scipy.signal.medfilt default kernel, reference
    scripts/ICM_SLAM_tools.py:37).  Selection only, no arithmetic."""
    z = np.asarray(z, dtype=float)
    p = np.concatenate(([0.0], z, [0.0]))
    a, b, c = p[:-2], p[1:-1], p[2:]
    return np.maximum(np.minimum(a, b), np.minimum(np.maximum(a, b), c))


def filtrar_z(z, cfg):
    """Scan pre-filter, SURVEY Appendix A.2 (reference scripts/ICM_SLAM_tools.py:22-58).
    Returns (n,4) rows [d, ang, bx, by]; an empty (0,) array when <=1 beam is in range."""
    m = median3(z)
    idx = np.where(m < cfg.rango_laser_max)[0]
    if len(idx) <= 1:
        return np.array([])
    ang = bearings(idx, cfg)
    pts = np.stack((np.cos(ang) * m[idx], np.sin(ang) * m[idx]), axis=1)
    d = _pairwise_dist(pts, pts)
    d[d == 0] = 100
    nn = np.amin(d, axis=0)
    keep = idx[nn <= cfg.dist_thr]
    out = np.empty((len(keep), 4))
    out[:, 0] = m[keep]
    out[:, 1] = bearings(keep, cfg)
    out[:, 2] = out[:, 0] * np.cos(out[:, 1])
    out[:, 3] = out[:, 0] * np.sin(out[:, 1])
    return out


# ----------------------------------------------------------------------------------------
# a4 / a13 / a14: the running-mean map
# ----------------------------------------------------------------------------------------
class MapState:
    """State of the reference `Mapa` object (scripts/ICM_SLAM_tools.py:104-126)."""

    def __init__(self, cfg, landmarks_actuales=0):
        self.L = cfg.L
        self.cota = cfg.cota
        self.dist_thr = cfg.dist_thr
        self.landmarks_actuales = int(landmarks_actuales)
        self.clear_obs()

    def clear_obs(self):
        # note: landmarks_actuales survives
        self.cant_obs_i = np.zeros(self.L)


def associate(ref_map, lact, obs, dist_thr):
    """Nearest-landmark association with a distance gate (reference
    scripts/ICM_SLAM_tools.py:168-172).  Returns labels (int64, -1 = gated out)."""
    d = _pairwise_dist(ref_map[:, :lact].T, obs)
    c = np.argmin(d, axis=0)
    c[np.amin(d, axis=0) > dist_thr] = -1
    return c


def actualizar(state, mapa, ref_map, obs):
    """Else-branch of `Mapa.actualizar` (reference scripts/ICM_SLAM_tools.py:167-201),
    SURVEY Appendix A.3: associate, give every gated-out beam of the scan the single new
    label `Lact` (Appendix B.1), then fold the scan into the running means."""
    lact = state.landmarks_actuales
    if lact == 0:
        raise NotImplementedError("first-scan clustering branch (a5) is outside the sweep")
    c = associate(ref_map, lact, obs, state.dist_thr)
    if np.any(c == -1):
        c[c == -1] = lact
    lact = int(max(lact, c.max() + 1))
    cnt = state.cant_obs_i
    for i in np.unique(c):
        sel = c == i
        k = int(sel.sum())
        mapa[:, i] = np.sum(obs[sel], axis=0) / (cnt[i] + k) + mapa[:, i] * cnt[i] / (cnt[i] + k)
        cnt[i] = cnt[i] + k
    state.landmarks_actuales = lact
    return mapa, c


def filtrar(state, mapa):
    """Prune + merge of the running map (reference `Mapa.filtrar`,
    scripts/ICM_SLAM_tools.py:204-265).  Returns the (2,L) zero padded filtered map and
    updates `state.landmarks_actuales` / `state.cant_obs_i`."""
    lact = state.landmarks_actuales
    cnt = state.cant_obs_i[0:lact]
    few = np.where(cnt < state.cota)[0]
    if few.size > 0:
        lact = lact - few.size
        many = np.where(cnt >= state.cota)[0]
        mapa = mapa[:, many]
        cnt = cnt[many]
    pts = mapa[:, 0:lact].T
    a = _pairwise_dist(pts, pts)
    np.fill_diagonal(a, 0.0)
    if a.size == 0:
        raise ValueError("zero-size array to reduction operation maximum which has no identity")
    a[a == 0] = np.amax(a)
    b = np.argmin(a, axis=0)
    a = np.amin(a, axis=0)
    close = np.where(a < state.dist_thr)[0]
    c = np.arange(lact)
    for i in close:
        c[c == c[b[i]]] = c[i]
    for i in range(lact - 1, -1, -1):
        if not np.any(c == i):
            c[c >= i] = c[c >= i] - 1
    lact = int(c.max()) + 1
    out = np.zeros((2, state.L))
    cnt_out = np.zeros(state.L)
    for i in range(lact):
        sel = c == i
        cnt_out[i] = np.sum(cnt[sel])
        out[:, i] = np.sum(mapa[:, sel] * np.tile(cnt[sel], (2, 1)), axis=1) / cnt_out[i]
    state.landmarks_actuales = lact
    state.cant_obs_i = cnt_out
    return out


# ----------------------------------------------------------------------------------------
# a6-a10: energies
# ----------------------------------------------------------------------------------------
def g(cfg, x, u):
    """Unicycle step x + dt*[v cos(th), v sin(th), w] (reference scripts/ICM_ROS.py:202-207)."""
    x = np.asarray(x, dtype=float).reshape((3, 1))
    u = np.asarray(u, dtype=float).reshape((2, 1))
    S = np.array([[np.cos(x[2])[0], 0.0], [np.sin(x[2])[0], 0.0], [0.0, 1.0]])
    return (x + cfg.deltat * np.matmul(S, u).reshape((3, 1))).reshape((3, 1))


def h(cfg, x, beams, targets):
    """Observation energy sum_i (w_i - y_i)^T Q (w_i - y_i) (reference
    scripts/ICM_ROS.py:171-200); beams (n,2)=[d, ang], targets (n,2)."""
    x = np.asarray(x, dtype=float).reshape(3)
    alfa = beams[:, 1] + x[2] - HALF_PI
    w = np.stack((x[0] + beams[:, 0] * np.cos(alfa), x[1] + beams[:, 0] * np.sin(alfa)), axis=1)
    r = w - targets
    return np.sum(np.matmul(r, cfg.Q) * r)


def _odom_term(cfg, pose_a, pose_b, odo_a, odo_b):
    """cte_odom * ||q||^2 of SURVEY Appendix A.4 for the motion a -> b."""
    q = np.zeros((3, 1))
    q[0:2] = (np.matmul(rota(odo_a[2]), (odo_b[0:2] - odo_a[0:2]).reshape((2, 1)))
              - np.matmul(rota(pose_a[2][0]), pose_b[0:2] - pose_a[0:2]))
    q[2] = odo_b[2] - odo_a[2] - pose_b[2] + pose_a[2]
    q[2] = entrepi(q[2])
    return cfg.cte_odom * np.matmul(q.T, q)


def energy_prev(cfg, x, x_ant, u_ant, odo_ant, odo_t):
    """prev(x; a) of Appendix A.4 (reference scripts/ICM_ROS.py:238-251, 263-277)."""
    x = np.asarray(x, dtype=float).reshape((3, 1))
    r = x - g(cfg, x_ant, u_ant)
    r[2] = entrepi(r[2])
    return np.matmul(np.matmul(r.T, cfg.R), r), _odom_term(cfg, x_ant, x, odo_ant, odo_t)


def energy_next(cfg, x, x_pos, u_t, odo_t, odo_pos):
    """next(x; b) of Appendix A.4 (reference scripts/ICM_ROS.py:227-236)."""
    x = np.asarray(x, dtype=float).reshape((3, 1))
    r = g(cfg, x, u_t) - x_pos
    r[2] = entrepi(r[2])
    return np.matmul(np.matmul(r.T, cfg.R), r) + _odom_term(cfg, x, x_pos, odo_t, odo_pos)


def fun_xn(cfg, x, x_ant, x_pos, u, odo, t, beams, targets):
    """Two-sided conditional energy of pose t (reference scripts/ICM_ROS.py:220-252)."""
    f = energy_next(cfg, x, x_pos, u[:, t], odo[:, t], odo[:, t + 1])
    pr, po = energy_prev(cfg, x, x_ant, u[:, t - 1], odo[:, t - 1], odo[:, t])
    return float((f + pr + h(cfg, x, beams, targets) + po)[0, 0])


def fun_x(cfg, x, x_ant, u, odo, t, beams, targets):
    """One-sided energy (reference scripts/ICM_ROS.py:262-278)."""
    pr, po = energy_prev(cfg, x, x_ant, u[:, t - 1], odo[:, t - 1], odo[:, t])
    return float((pr + h(cfg, x, beams, targets) + po)[0, 0])


# ----------------------------------------------------------------------------------------
# a11: Nelder-Mead exactly as scipy.optimize.fmin(f, x0, xtol=1e-3, disp=0) runs it
# ----------------------------------------------------------------------------------------
class _MaxFun(Exception):
    pass


def nelder_mead(f, x0, xatol=1e-3, fatol=1e-4, full_output=False):
    """SURVEY Appendix A.5.  rho=1 chi=2 psi=sigma=0.5, maxiter=maxfun=200*N."""
    x0 = np.asarray(x0, dtype=float).flatten()
    N = len(x0)
    maxiter = maxfun = 200 * N
    ncalls = [0]

    def func(v):
        if ncalls[0] >= maxfun:
            raise _MaxFun()
        ncalls[0] += 1
        return f(np.copy(v))

    sim = np.empty((N + 1, N))
    sim[0] = x0
    for k in range(N):
        y = np.array(x0, copy=True)
        y[k] = (1 + 0.05) * y[k] if y[k] != 0 else 0.00025
        sim[k + 1] = y
    fsim = np.full((N + 1,), np.inf)
    try:
        for k in range(N + 1):
            fsim[k] = func(sim[k])
    except _MaxFun:
        pass
    order = np.argsort(fsim)
    sim, fsim = sim[order], fsim[order]
    it = 1
    while ncalls[0] < maxfun and it < maxiter:
        try:
            if (np.max(np.abs(sim[1:] - sim[0])) <= xatol
                    and np.max(np.abs(fsim[0] - fsim[1:])) <= fatol):
                break
            xbar = np.add.reduce(sim[:-1], 0) / N
            xr = 2 * xbar - 1 * sim[-1]
            fr = func(xr)
            shrink = False
            if fr < fsim[0]:
                xe = 3 * xbar - 2 * sim[-1]
                fe = func(xe)
                if fe < fr:
                    sim[-1], fsim[-1] = xe, fe
                else:
                    sim[-1], fsim[-1] = xr, fr
            elif fr < fsim[-2]:
                sim[-1], fsim[-1] = xr, fr
            elif fr < fsim[-1]:
                xc = 1.5 * xbar - 0.5 * sim[-1]
                fc = func(xc)
                if fc <= fr:
                    sim[-1], fsim[-1] = xc, fc
                else:
                    shrink = True
            else:
                xcc = 0.5 * xbar + 0.5 * sim[-1]
                fcc = func(xcc)
                if fcc < fsim[-1]:
                    sim[-1], fsim[-1] = xcc, fcc
                else:
                    shrink = True
            if shrink:
                for j in range(1, N + 1):
                    sim[j] = sim[0] + 0.5 * (sim[j] - sim[0])
                    fsim[j] = func(sim[j])
            it += 1
        except _MaxFun:
            pass
        order = np.argsort(fsim)
        sim, fsim = sim[order], fsim[order]
    if full_output:
        return sim[0].copy(), float(fsim[0]), it, ncalls[0]
    return sim[0].copy()


# ----------------------------------------------------------------------------------------
# a1: one sweep
# ----------------------------------------------------------------------------------------
def prefilter_all(scans, cfg):
    """filtrar_z for every column of `scans` (B,T); cacheable (pure function of the scan)."""
    return [filtrar_z(scans[:, t], cfg) for t in range(scans.shape[1])]


def solve_pose(cfg, x, t, u, odo, beams, targets, x0_pose=None):
    """minimizar_xn / minimizar_x for pose t (reference scripts/ICM_ROS.py:209-218,254-260)."""
    Tf = x.shape[1]
    x_ant = x[:, t - 1].reshape((3, 1)).copy()
    if t + 1 < Tf:
        x_pos = x[:, t + 1].reshape((3, 1)).copy()
        start = (x_ant + x_pos) / 2.0
        return nelder_mead(lambda v: fun_xn(cfg, v, x_ant, x_pos, u, odo, t, beams, targets), start)
    start = g(cfg, x_ant, u[:, t - 1])
    return nelder_mead(lambda v: fun_x(cfg, v, x_ant, u, odo, t, beams, targets), start)


def sweep(cfg, state, scans, u, odo, x0, mapa_viejo, x, schedule="sequential", kept=None,
          trace=None):
    """One offline ICM sweep (reference `ICM_ROS.iterations_process_offline`,
    scripts/ICM_ROS.py:121-164) in the three-phase form of SURVEY Appendix A.6:
    A+B (associate + running means, in pose order, against the fixed `mapa_viejo` and the
    previous-sweep poses) then C (pose solves in `schedule` order) then D (`filtrar`).
    `schedule='sequential'` is the reference's Gauss-Seidel order and reproduces it bit for
    bit; `'redblack'` solves odd poses first, then even poses.  `x` is updated in place and
    returned with the refined (2,K') map."""
    Tf = x.shape[1]
    if kept is None:
        kept = prefilter_all(scans, cfg)
    y = np.zeros((2, cfg.L))
    state.clear_obs()
    if kept[0].shape[0] == 0:
        return mapa_viejo, x
    # ---- phases A + B, pose order --------------------------------------------------
    w0 = project_beams(np.asarray(x0).reshape(3), kept[0][:, 2:4])
    y, _ = actualizar(state, y, mapa_viejo, w0)
    labels = [None] * Tf
    targets = [None] * Tf
    for t in range(1, Tf):
        if kept[t].shape[0] == 0:
            continue
        w = project_beams(x[:, t], kept[t][:, 2:4])
        y, c = actualizar(state, y, mapa_viejo, w)
        labels[t] = c
        targets[t] = y[:, c].T.copy()
    # ---- phase C -------------------------------------------------------------------
    if schedule == "sequential":
        order = list(range(1, Tf))
    elif schedule == "redblack":
        order = list(range(1, Tf, 2)) + list(range(2, Tf, 2))
    else:
        raise ValueError(schedule)
    x0v = np.asarray(x0, dtype=float).reshape(3)
    for t in order:
        if kept[t].shape[0] == 0:
            # reference: (last solved pose + x[:,t+1])/2; the last solved pose is x[:,t-1],
            # except before the first solve where it is x0 (scripts/ICM_ROS.py:125,143-147)
            prev = x0v if t == 1 else x[:, t - 1]
            x[:, t] = (prev + x[:, t + 1]) / 2.0  # IndexError at t = Tf-1 like the reference
            continue
        xt = solve_pose(cfg, x, t, u, odo, kept[t][:, 0:2], targets[t])
        if trace is not None:
            trace(t, labels[t], targets[t], xt)
        x[:, t] = xt
    # ---- phase D -------------------------------------------------------------------
    yy = filtrar(state, y)
    return yy[:, :state.landmarks_actuales].copy(), x


def sweep_interleaved(cfg, state, scans, u, odo, x0, mapa_viejo, x, kept=None):
    """The same sweep with map update and pose solve interleaved per pose, i.e. literally the
    loop order of scripts/ICM_ROS.py:141-158.  Exists to demonstrate (tests) that the
    three-phase `sweep(..., 'sequential')` is the same computation."""
    Tf = x.shape[1]
    if kept is None:
        kept = prefilter_all(scans, cfg)
    y = np.zeros((2, cfg.L))
    state.clear_obs()
    if kept[0].shape[0] == 0:
        return mapa_viejo, x
    xt = np.asarray(x0, dtype=float).reshape(3).copy()
    y, _ = actualizar(state, y, mapa_viejo, project_beams(xt, kept[0][:, 2:4]))
    for t in range(1, Tf):
        if kept[t].shape[0] == 0:
            xt = (xt + x[:, t + 1]) / 2.0
            x[:, t] = xt
            continue
        y, c = actualizar(state, y, mapa_viejo, project_beams(x[:, t], kept[t][:, 2:4]))
        xt = solve_pose(cfg, x, t, u, odo, kept[t][:, 0:2], y[:, c].T.copy())
        x[:, t] = xt
    yy = filtrar(state, y)
    return yy[:, :state.landmarks_actuales].copy(), x


def prepare_ranges(z, cfg):
    """Trunk-radius inflation and clipping zz = min(z + radio, rango_laser_max)
    (reference scripts/sensors_definitions.py:22)."""
    return np.minimum(z + cfg.radio, z * 0 + cfg.rango_laser_max)


def calc_cambio(y, mapa_viejo):
    """min/max/mean nearest-neighbour change of the map (reference
    scripts/ICM_SLAM_tools.py:490-495)."""
    md = np.amin(_pairwise_dist(mapa_viejo.T, y.T), axis=0)
    return np.amin(md), np.amax(md), np.mean(md)


# ----------------------------------------------------------------------------------------
# a5 + the online initialisation pass (reference
# scripts/ICM_ROS.py:47-119 driven ROS-free as in SURVEY Appendix D)
# ----------------------------------------------------------------------------------------
def single_linkage(pts):
    """scipy.cluster.hierarchy.linkage(pdist(pts)) (method 'single'): Prim's MST walk from
    point 0, links sorted by height with a stable sort, then union-find labelling where the
    smaller cluster id goes first.  Returns Z (n-1,4) like SciPy."""
    n = pts.shape[0]
    d = _pairwise_dist(pts, pts)
    Z = np.zeros((n - 1, 4))
    merged = np.zeros(n, dtype=bool)
    D = np.full(n, np.inf)
    x = 0
    for k in range(n - 1):
        merged[x] = True
        cur = np.inf
        y = -1
        for i in range(n):
            if merged[i]:
                continue
            if D[i] > d[x, i]:
                D[i] = d[x, i]
            if D[i] < cur:
                y, cur = i, D[i]
        Z[k, 0], Z[k, 1], Z[k, 2] = x, y, cur
        x = y
    Z = Z[np.argsort(Z[:, 2], kind="mergesort")]
    parent = list(range(2 * n - 1))
    size = [1] * n + [0] * (n - 1)

    def find(a):
        while parent[a] != a:
            a = parent[a]
        return a

    for i in range(n - 1):
        a, b = find(int(Z[i, 0])), find(int(Z[i, 1]))
        Z[i, 0], Z[i, 1] = (a, b) if a < b else (b, a)
        parent[a] = parent[b] = n + i
        size[n + i] = size[a] + size[b]
        Z[i, 3] = size[n + i]
    return Z


def inconsistency(Z, depth=2):
    """scipy.cluster.hierarchy.inconsistent(Z, d)[:, 3]: (h - mean)/std over the link heights
    of the sub-tree down to `depth` levels (sample std from the sums-of-squares formula)."""
    n = Z.shape[0] + 1
    out = np.zeros(n - 1)
    for i in range(n - 1):
        # SciPy's walk is post-order: left child link, right child link, the link itself
        hs = []

        def walk(node, lev):
            if lev < depth - 1:
                for ch in (int(Z[node, 0]), int(Z[node, 1])):
                    if ch >= n:
                        walk(ch - n, lev + 1)
            hs.append(Z[node, 2])
        walk(i, 0)
        cnt = len(hs)
        s = sum(hs)
        ss = sum(h * h for h in hs)
        var = (ss - s * s / cnt) / (cnt - 1) if cnt >= 2 else (ss - s * s / cnt) / cnt
        std = math.sqrt(var) if var > 0 else 0.0
        if std > 0:
            out[i] = (Z[i, 2] - s / cnt) / std
    return out


def fcluster_inconsistent(Z, t):
    """scipy.cluster.hierarchy.fcluster(Z, t) (criterion 'inconsistent', depth 2): a node whose
    whole sub-tree has inconsistency <= t becomes one flat cluster; clusters are numbered 1..
    in left-first depth-first order."""
    n = Z.shape[0] + 1
    inc = inconsistency(Z, 2)
    mx = np.zeros(n - 1)
    for i in range(n - 1):  # children have smaller indices than their parent
        m = inc[i]
        for ch in (int(Z[i, 0]), int(Z[i, 1])):
            if ch >= n:
                m = max(m, mx[ch - n])
        mx[i] = m
    T = np.zeros(n, dtype=np.int64)
    ncl = 0

    def visit(node, leader):
        nonlocal ncl
        if not leader and mx[node] <= t:
            leader = True
            ncl += 1
        for ch in (int(Z[node, 0]), int(Z[node, 1])):
            if ch >= n:
                visit(ch - n, leader)
        for ch in (int(Z[node, 0]), int(Z[node, 1])):
            if ch < n:
                if not leader:
                    ncl += 1
                T[ch] = ncl
    visit(n - 2, False)
    return T


def cluster_first_scan(state, mapa, obs):
    """`Mapa.actualizar` with Lact == 0 (reference scripts/ICM_SLAM_tools.py:160-165): flat
    clusters of the first scan, their means and sizes."""
    if obs.shape[0] == 1:
        c = np.zeros(1, dtype=np.int64)  # SciPy refuses a single observation; one cluster
    else:
        c = fcluster_inconsistent(single_linkage(obs), state.dist_thr) - 1
    lact = int(c.max()) + 1
    for i in range(lact):
        mapa[:, i] = np.mean(obs[c == i, :], axis=0).T
        state.cant_obs_i[i] = np.sum(c == i)
    state.landmarks_actuales = lact
    return mapa, c


def init_pass(cfg, scans, u, odo, kept=None):
    """The causal first pass that builds the initial poses and map (reference
    `inicializar_online` + `inicializar_online_process`, scripts/ICM_ROS.py:57-119, on recorded
    data: scan 0 seeds the map, then predict with g, associate against the RUNNING map,
    one-sided solve).  Returns (x_init (3,T), map_init (2,K), state, labels of scan 0)."""
    T = odo.shape[1]
    if kept is None:
        kept = prefilter_all(scans, cfg)
    x0 = odo[:, 0].copy()
    xt = x0.reshape((3, 1)).copy()
    x = np.zeros((3, T))
    x[:, 0] = x0
    y = np.zeros((2, cfg.L))
    state = MapState(cfg)
    w0 = project_beams(x0, kept[0][:, 2:4])
    y, c0 = cluster_first_scan(state, y, w0)
    for t in range(1, T):
        xtc = g(cfg, xt, u[:, t - 1])
        if kept[t].shape[0] == 0:
            xt = xtc + 0.0
        else:
            w = project_beams(xtc.reshape(3), kept[t][:, 2:4])
            y, c = actualizar(state, y, y, w)
            x_ant = xt.copy()
            tg = y[:, c].T.copy()
            xs = nelder_mead(lambda v: fun_x(cfg, v, x_ant, u, odo, t, kept[t][:, 0:2], tg), g(cfg, x_ant, u[:, t - 1]))
            xt = xs.reshape((3, 1))
        x[:, t] = xt.reshape(3)
    y_raw, cnt_raw, lact_raw = y.copy(), state.cant_obs_i.copy(), state.landmarks_actuales
    yy = filtrar(state, y)
    return x, yy[:, :state.landmarks_actuales].copy(), state, c0, (y_raw, cnt_raw, lact_raw)


# ---------------------------------------------------------------------------------------
# Association by geometric runs (round 5): a CPU statement of the claim the HIP kernels
# k_run_build / k_assoc_runs rest on, so that it can be checked against the reference's
# literal per-beam rule (`associate` above = scripts/ICM_SLAM_tools.py:168-172) without a GPU.
# ---------------------------------------------------------------------------------------
RUN_CAP = 64


def cut_runs(body_xy, dist_thr):
    """Cut a scan's kept beams (n,2) body-frame points, in beam order, into runs: a new run starts where the next point
    is farther than 0.35 dist_thr from the last one, farther than 0.5 dist_thr from the run's first point, or after
    RUN_CAP beams.  Returns a list of (first, count)."""
    gap2, ext2 = (0.35 * dist_thr) ** 2, (0.5 * dist_thr) ** 2
    runs, js = [], 0
    n = body_xy.shape[0]
    for j in range(1, n + 1):
        cut = j == n
        if not cut:
            g = body_xy[j] - body_xy[j - 1]
            e = body_xy[j] - body_xy[js]
            cut = g[0] * g[0] + g[1] * g[1] > gap2 or e[0] * e[0] + e[1] * e[1] > ext2 or j - js >= RUN_CAP
        if cut:
            runs.append((js, j - js))
            js = j
    return runs if n else []


def run_decision(centre_w, radius, ref_map, lact, dist_thr):
    """The bounding-circle test for one run: `centre_w` the run's centre projected with the pose (2,), `radius` a bound
    on the distance of every beam of the run to the centre.  Candidates are the landmarks of the 3x3 cells (edge
    dist_thr (1 + 1e-9), origin at the map's minimum) around the centre's cell, like the kernel's grid record.  Returns
    the landmark index every beam of the run takes, or None when the test does not settle the run (crowded or distant
    landmarks, gated-out beams, more than four candidates)."""
    K = min(int(lact), ref_map.shape[1])
    if K == 0:
        return None
    mx, my = ref_map[0, :K], ref_map[1, :K]
    cell = dist_thr * (1.0 + 1e-9)
    gx0, gy0 = mx.min(), my.min()
    nx, ny = int(np.floor((mx.max() - gx0) / cell)) + 1, int(np.floor((my.max() - gy0) / cell)) + 1

    def cidx(v, g0, n):
        return np.clip(np.floor((v - g0) / cell), 0, n - 1).astype(np.int64)

    cx, cy = int(cidx(centre_w[0], gx0, nx)), int(cidx(centre_w[1], gy0, ny))
    lx, ly = cidx(mx, gx0, nx), cidx(my, gy0, ny)
    cand = np.flatnonzero((np.abs(lx - cx) <= 1) & (np.abs(ly - cy) <= 1))
    if cand.size == 0 or cand.size > 4:
        return None
    d = np.sqrt((mx[cand] - centre_w[0]) ** 2 + (my[cand] - centre_w[1]) ** 2)
    order = np.argsort(d, kind="stable")
    d1 = d[order[0]]
    d2 = d[order[1]] if cand.size > 1 else np.inf
    eps = 1e-4 * dist_thr
    r = float(radius)
    if d1 + r <= dist_thr - eps and (d1 + 2 * r) / cell <= 1.0 - 1e-4 and d2 - d1 >= 2 * r + eps:
        return int(cand[order[0]])
    return None
