"""Synthetic scan sequences for the benchmark configurations of BASELINE.json
(SURVEY.md section 8d): trunks on a jittered grid, a lawn-mower unicycle trajectory through
it, ray-cast 2-D LiDAR scans, noisy velocities and their integrated odometry.

This is workload generation only (no part of the ICM algorithm lives here).  Everything is
seeded, and scans can be generated for a pose sub-range so that every rank of a sharded run
builds only its own shard.

Deviations from the survey's sketch, stated here because they shape the workload:
  * the initial poses are truth + smooth noise (sigma 0.05 m / 0.01 rad), standing in for
    the output of the (out-of-scope) online initialisation pass.  Raw integrated odometry
    drifts by tens of metres over 1e5 poses, far outside the 1 m association gate;
  * `angle_increment` = 2*pi/B (full circle), because the reference's hard-coded 1 degree
    per scan row (scripts/ICM_SLAM_tools.py:44) would make 720 rows two revolutions.
"""
import numpy as np

SPACING = 2.5       # landmark grid pitch [m] (> 2*dist_thr, so Mapa.filtrar never merges)
JITTER = 0.5        # uniform jitter of each trunk around its grid node [m]
RADIO = 0.137       # trunk radius [m] (config `radio`)
RMAX = 10.0         # laser max range [m]
DT = 0.1
V_NOM = 0.4         # [m/s]


class Workload:
    pass


def landmarks(K, rng):
    n = int(np.ceil(np.sqrt(K)))
    gx, gy = np.meshgrid(np.arange(n), np.arange(n), indexing="xy")
    pts = np.stack((gx.ravel(), gy.ravel()), axis=0)[:, :K].astype(float) * SPACING
    pts += rng.uniform(-JITTER, JITTER, size=pts.shape)
    return pts, n


def trajectory(T, n_side, rng):
    """True poses (3,T) and controls (2,T): lanes along grid corridors joined by
    semicircles outside the field, with a gentle sinusoidal meander.  The lane pitch shrinks
    with T so that one pass covers the field; once the pitch is down to one grid row, a still
    longer sequence reverses at the field's edge and sweeps back over it."""
    side = n_side * SPACING
    path_len = T * V_NOM * DT
    lanes = max(1, int(path_len // (side + 8.0)))
    rows_per_lane = max(1, int(n_side // lanes))
    v = np.full(T, V_NOM)
    w = np.zeros(T)
    x = np.zeros((3, T))
    y0 = (0 + 0.5) * SPACING
    x[:, 0] = (-3.0, y0, 0.0)
    r = rows_per_lane * SPACING / 2.0
    turn_steps = int(round(np.pi * r / (V_NOM * DT)))
    lane_steps = int(round((side + 6.0) / (V_NOM * DT)))
    t = 0
    row, heading, sense = 0, 1, 1   # lane's grid row; +1 east / -1 west; +1 field-upwards / -1 back down
    while t < T - 1:
        e = min(t + lane_steps, T - 1)
        # meander: whole cosine periods of yaw rate per lane, so the heading swings by
        # +-0.05 rad and is back to the lane direction at the lane's end
        cycles = max(1, int(round(lane_steps / 157.0)))
        w[t:e] = 0.02 * np.cos(2 * np.pi * cycles * np.arange(e - t) / lane_steps)
        t = e
        if t >= T - 1:
            break
        e = min(t + turn_steps, T - 1)
        # a sequence longer than one coverage of the field comes back down over it
        if not 0 <= row + sense * rows_per_lane <= n_side - 1:
            sense = -sense
        row += sense * rows_per_lane
        w[t:e] = sense * heading * (np.pi / (turn_steps * DT))
        heading = -heading
        t = e
    for k in range(T - 1):
        th = x[2, k]
        x[0, k + 1] = x[0, k] + DT * v[k] * np.cos(th)
        x[1, k + 1] = x[1, k] + DT * v[k] * np.sin(th)
        x[2, k + 1] = th + DT * w[k]
    return x, np.stack((v, w), axis=0)


def _grid_index(pts, cell):
    x0, y0 = pts[0].min(), pts[1].min()
    cx = np.floor((pts[0] - x0) / cell).astype(np.int64)
    cy = np.floor((pts[1] - y0) / cell).astype(np.int64)
    nx, ny = int(cx.max()) + 1, int(cy.max()) + 1
    cid = cy * nx + cx
    order = np.argsort(cid, kind="stable")
    start = np.searchsorted(cid[order], np.arange(nx * ny + 1))
    return dict(x0=x0, y0=y0, nx=nx, ny=ny, cell=cell, order=order, start=start)


NOISE_BLOCK = 256   # poses per block of the range-noise stream


def range_noise(seed, t_begin, t_end, B, sigma):
    """Gaussian range noise of the poses [t_begin, t_end), (P,B): a counter-based stream keyed by
    (seed, block of NOISE_BLOCK poses), so that a pose's scan carries the same noise whichever
    rank of whatever partition generates it (an N-rank job sweeps the N = 1 job's inputs)."""
    out = np.empty((t_end - t_begin, B))
    for b in range(t_begin // NOISE_BLOCK, (max(t_end, t_begin + 1) - 1) // NOISE_BLOCK + 1):
        blk = np.random.default_rng([seed, 0x5CA9, b]).normal(0.0, sigma, size=(NOISE_BLOCK, B))
        a0, a1 = max(b * NOISE_BLOCK, t_begin), min((b + 1) * NOISE_BLOCK, t_end)
        if a1 > a0:
            out[a0 - t_begin:a1 - t_begin] = blk[a0 - b * NOISE_BLOCK:a1 - b * NOISE_BLOCK]
    return out


def raycast(poses, lm, B, inc, noise, chunk=2048):
    """Ranges (P,B) to the trunk surfaces (no hit = RMAX), beam k at bearing k*inc in the
    sensor frame, sensor x-axis to the robot's right (world bearing = ang + theta - pi/2,
    reference scripts/ICM_ROS.py:191).  `noise` (P,B) is added to the beams that hit."""
    P = poses.shape[1]
    out = np.full((P, B), RMAX)
    gi = _grid_index(lm, 4.0)
    reach = int(np.ceil((RMAX + RADIO) / gi["cell"]))
    offs = np.arange(-reach, reach + 1)
    for p0 in range(0, P, chunk):
        p1 = min(p0 + chunk, P)
        px, py, th = poses[0, p0:p1], poses[1, p0:p1], poses[2, p0:p1]
        pcx = np.floor((px - gi["x0"]) / gi["cell"]).astype(np.int64)
        pcy = np.floor((py - gi["y0"]) / gi["cell"]).astype(np.int64)
        pose_l, lm_l = [], []
        for dy in offs:
            cy = pcy + dy
            oky = (cy >= 0) & (cy < gi["ny"])
            c0 = np.clip(pcx - reach, 0, gi["nx"] - 1)
            c1 = np.clip(pcx + reach, 0, gi["nx"] - 1)
            okx = (pcx + reach >= 0) & (pcx - reach < gi["nx"])
            ok = oky & okx
            cyc = np.clip(cy, 0, gi["ny"] - 1)
            s = gi["start"][cyc * gi["nx"] + c0]
            e = gi["start"][cyc * gi["nx"] + c1 + 1]
            cnt = np.where(ok, e - s, 0)
            tot = int(cnt.sum())
            if tot == 0:
                continue
            pi = np.repeat(np.arange(p1 - p0), cnt)
            base = np.repeat(s, cnt)
            within = np.arange(tot) - np.repeat(np.cumsum(cnt) - cnt, cnt)
            pose_l.append(pi)
            lm_l.append(gi["order"][base + within])
        if not pose_l:
            continue
        pi = np.concatenate(pose_l)
        li = np.concatenate(lm_l)
        dx = lm[0, li] - px[pi]
        dy_ = lm[1, li] - py[pi]
        d = np.hypot(dx, dy_)
        sel = (d < RMAX + RADIO) & (d > RADIO * 1.01)
        pi, d, dx, dy_ = pi[sel], d[sel], dx[sel], dy_[sel]
        # bearing of the trunk centre in the sensor frame, and its half-width
        phi = np.mod(np.arctan2(dy_, dx) - th[pi] + np.pi / 2.0, 2 * np.pi)
        half = np.arcsin(np.minimum(RADIO / d, 1.0))
        k_lo = np.ceil((phi - half) / inc).astype(np.int64)
        k_hi = np.floor((phi + half) / inc).astype(np.int64)
        nb = np.maximum(k_hi - k_lo + 1, 0)
        tot = int(nb.sum())
        if tot == 0:
            continue
        pair = np.repeat(np.arange(len(pi)), nb)
        kk = np.repeat(k_lo, nb) + (np.arange(tot) - np.repeat(np.cumsum(nb) - nb, nb))
        delta = kk * inc - phi[pair]
        dd = d[pair]
        disc = RADIO ** 2 - (dd * np.sin(delta)) ** 2
        okb = disc >= 0
        rng_hit = dd * np.cos(delta) - np.sqrt(np.where(okb, disc, 0.0))
        nrev = int(round(2 * np.pi / inc))
        kmod = np.mod(kk, nrev)
        okb &= (kmod < B) & (rng_hit > 0) & (rng_hit < RMAX)
        flat = pi[pair][okb] * B + kmod[okb]
        np.minimum.at(out[p0:p1].reshape(-1), flat, rng_hit[okb])
    hit = out < RMAX
    out[hit] += noise[hit]
    np.clip(out, 0.05, RMAX, out=out)
    return out


def make_workload(T, K, B, seed=20181, t_begin=0, t_end=None, L_margin=4096):
    """Returns a Workload with: config values (dict), map_true/map_init (2,K), x_true/x_init
    (3,T), u (2,T), odometry (3,T), scans (t_end-t_begin, B) pose-major *prepared* ranges
    zz = min(z + radio, rango_laser_max) (reference scripts/sensors_definitions.py:22)."""
    t_end = T if t_end is None else t_end
    rng = np.random.default_rng(seed)
    lm, n_side = landmarks(K, rng)
    x_true, u_true = trajectory(T, n_side, rng)
    u = u_true + np.stack((rng.normal(0, 0.01, T), rng.normal(0, 0.005, T)))
    odo = np.zeros((3, T))
    odo[:, 0] = x_true[:, 0]
    c, s = np.cos, np.sin
    th = odo[2, 0]
    for k in range(T - 1):
        odo[0, k + 1] = odo[0, k] + DT * u[0, k] * c(th)
        odo[1, k + 1] = odo[1, k] + DT * u[0, k] * s(th)
        th = th + DT * u[1, k]
        odo[2, k + 1] = th
    # stand-in for the init pass: truth + smooth (low-pass) noise
    def smooth(sig):
        w = rng.normal(0, 1, T + 200)
        ker = np.ones(201) / np.sqrt(201.0)
        return sig * np.convolve(w, ker, mode="valid")[:T]
    x_init = x_true + np.stack((smooth(0.05), smooth(0.05), smooth(0.01)))
    x_init[:, 0] = x_true[:, 0]
    map_init = lm + rng.normal(0, 0.05, lm.shape)
    inc = 2 * np.pi / B
    angle_increment = None if B == 360 else inc  # 360 rows = the reference's own 1 degree
    z = raycast(x_true[:, t_begin:t_end], lm, B, inc, range_noise(seed, t_begin, t_end, B, 0.01))
    zz = np.minimum(z + RADIO, RMAX)
    wl = Workload()
    # the scan in front of the shard: ranks > 0 of a sharded job solve that pose too (their ghost pose)
    wl.ghost_scan = None
    if t_begin >= 1:
        zg = raycast(x_true[:, t_begin - 1:t_begin], lm, B, inc, range_noise(seed, t_begin - 1, t_begin, B, 0.01))
        wl.ghost_scan = np.minimum(zg + RADIO, RMAX)[0]
    wl.T, wl.K, wl.B, wl.t_begin, wl.t_end = T, K, B, t_begin, t_end
    wl.map_true, wl.map_init, wl.x_true, wl.x_init = lm, map_init, x_true, x_init
    wl.u, wl.odometry, wl.scans = u, odo, zz
    wl.x0 = x_init[:, 0].copy()
    wl.config = dict(N=20, deltat=DT, L=int(K + L_margin), Q=[1.0, 1.0], R=[1.0, 1.0, 1.0], cte_odom=1.0,
                     cota=5.0, dist_thr=1.0, dist_thr_obs=1.0, rango_laser_max=RMAX, radio=RADIO,
                     angle_increment=angle_increment)
    return wl


WORKLOADS = {
    # name: (T poses, K landmarks, B beams)  -- BASELINE.json configs[2], configs[3]
    "S1": (10_000, 1_000, 360),
    "S2": (100_000, 10_000, 720),
    "tiny": (600, 64, 180),
    "mini": (700, 64, 180),      # like tiny, and twice its length (the 2-rank weak-scaling job) also ends on a scan with beams
}
