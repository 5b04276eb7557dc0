"""Test double with the phase API of `icmslam_hip.SweepEngine`, computed by the CPU oracle:
lets `icmslam_hip.sharded.ShardedSweep` (partition arithmetic, buffer layout, collective
order) run under torch.distributed/gloo on machines without a GPU.  Test infrastructure."""
import numpy as np

from oracle import icm_oracle as o


class OracleShardEngine:
    exchange_device = "cpu"

    def __init__(self, ocfg, scans_BT, u, odo, t_begin, t_end):
        self.cfg, self.u, self.odo = ocfg, u, odo
        self.T = odo.shape[1]
        self.a, self.b = t_begin, t_end
        self.kept = {t: o.filtrar_z(scans_BT[:, t], ocfg) for t in range(t_begin, t_end)}
        # the ghost pose: the last pose of the rank below, solved here as well (icmslam_hip/sharded.py)
        self.ghost = t_begin - 1 if t_begin >= 2 else None
        if self.ghost is not None:
            assert t_begin % 2 == 0, "shards start at even poses"
            self.kept[self.ghost] = o.filtrar_z(scans_BT[:, self.ghost], ocfg)
        self.L = ocfg.L

    def stats_stride(self):
        return 3 * self.L + 16

    def bind_tensors(self, stats, poses, rank, world):
        self.stats = stats.numpy().reshape(world, -1)
        self.x = poses.numpy().reshape(-1, 3)  # (Tpad,3) shared with the tensor
        self.rank, self.world = rank, world

    def set_state(self, mapa_viejo, x, x0, lact=None):
        self.x[:self.T] = np.asarray(x).T
        self.x0 = np.asarray(x0, dtype=float).reshape(3)
        self.map = np.array(mapa_viejo, dtype=float)
        self.lact = self.map.shape[1] if lact is None else lact
        self.counts = np.zeros(self.L)

    def sweep_local(self):
        L, lact0 = self.L, self.lact
        S = np.zeros((3, L))
        self.entries = {}
        n_new = 0
        for t in range(self.a, self.b):
            k = self.kept[t]
            if k.ndim != 2 or k.shape[0] == 0:
                continue
            pose = self.x0 if t == 0 else self.x[t]
            w = o.project_beams(pose, k[:, 2:4])
            c = o.associate(self.map, min(lact0, self.map.shape[1]), w, self.cfg.dist_thr)
            if np.any(c == -1):
                c[c == -1] = lact0 + n_new
                n_new += 1
            ent = {}
            for i in np.unique(c):
                sel = c == i
                ent[int(i)] = (w[sel].sum(axis=0), int(sel.sum()))
                S[0:2, i] += ent[int(i)][0]
                S[2, i] += ent[int(i)][1]
            self.entries[t] = (c, ent)
        row = self.stats[self.rank]
        row[:3 * L] = S.reshape(-1)
        row[3 * L] = n_new
        row[3 * L + 1] = 0.0
        # boundary poses as the previous sweep left them: first, last, last but one
        row[3 * L + 2:3 * L + 5] = self.x[self.a]
        row[3 * L + 5:3 * L + 8] = self.x[self.b - 1]
        row[3 * L + 8:3 * L + 11] = self.x[max(self.b - 2, 0)]
        self.n_new = n_new

    def sweep_targets(self):
        L, lact0 = self.L, self.lact
        allS = self.stats[:, :3 * L].reshape(self.world, 3, L)
        # the neighbours' boundary poses (previous-sweep values) out of their headers
        if self.ghost is not None:
            below = self.stats[self.rank - 1]
            self.x[self.a - 1] = below[3 * L + 5:3 * L + 8]
            self.x[self.a - 2] = below[3 * L + 8:3 * L + 11]
        if self.b < self.T and self.rank + 1 < self.world:
            self.x[self.b] = self.stats[self.rank + 1][3 * L + 2:3 * L + 5]
        run = allS[:self.rank, :, :].sum(axis=0)
        run[:, lact0:] = 0.0
        self.targets = {}
        g = self.ghost
        if g is not None and self.kept[g].ndim == 2 and self.kept[g].shape[0] > 0:
            # ghost pose: associated with its owner's previous-sweep value; running means through it inclusive = the
            # totals of all lower ranks; the landmark it created itself is the last new one of the rank below
            w = o.project_beams(self.x[g], self.kept[g][:, 2:4])
            c = o.associate(self.map, min(lact0, self.map.shape[1]), w, self.cfg.dist_thr)
            tg = np.zeros((c.size, 2))
            old = c >= 0
            tg[old] = (run[0:2, c[old]] / run[2, c[old]]).T
            if np.any(~old):
                col = lact0 + int(self.stats[self.rank - 1, 3 * L]) - 1
                sb = allS[self.rank - 1]
                tg[~old] = sb[0:2, col] / sb[2, col]
            self.targets[g] = tg
        for t in range(self.a, self.b):
            if t not in self.entries:
                continue
            c, ent = self.entries[t]
            for i, (sw, k) in ent.items():
                run[0:2, i] += sw
                run[2, i] += k
            self.targets[t] = (run[0:2, c] / run[2, c]).T.copy()
        tot = allS[:, :, :lact0].sum(axis=0)
        y = np.zeros((2, L))
        cnt = np.zeros(L)
        cnt[:lact0] = tot[2]
        nz = tot[2] > 0
        y[:, :lact0][:, nz] = tot[0:2][:, nz] / tot[2][nz]
        col = lact0
        for r in range(self.world):
            nn = int(self.stats[r, 3 * L])
            y[:, col:col + nn] = allS[r, 0:2, lact0:lact0 + nn] / allS[r, 2, lact0:lact0 + nn]
            cnt[col:col + nn] = allS[r, 2, lact0:lact0 + nn]
            col += nn
        self.y_raw, self.cnt_raw, self.lact_raw = y, cnt, col

    def sweep_solve(self, schedule, colour):
        xv = self.x[:self.T].T  # (3,T) view: writes go to this rank's pose array
        first = max(self.a, 1) if self.ghost is None else self.ghost
        order = [t for t in range(first, self.b) if t & 1] + [t for t in range(first, self.b) if not t & 1]
        for t in order:
            if colour >= 0 and (t & 1) != colour:
                continue
            k = self.kept[t]
            if k.ndim != 2 or k.shape[0] == 0:
                prev = self.x0 if t == 1 else xv[:, t - 1]
                xv[:, t] = (prev + xv[:, t + 1]) / 2.0
                continue
            xv[:, t] = o.solve_pose(self.cfg, xv, t, self.u, self.odo, k[:, 0:2], self.targets[t])

    def sweep_finish(self):
        st = o.MapState(self.cfg, self.lact_raw)
        st.cant_obs_i = self.cnt_raw.copy()
        # (live columns only: with nothing to prune the reference indexes its unsliced (2,L) map
        # with a length-Lact mask and raises, scripts/ICM_SLAM_tools.py:259 -- DESIGN.md section 2)
        yy = o.filtrar(st, self.y_raw[:, :self.lact_raw].copy())
        self.lact = st.landmarks_actuales
        self.map = yy[:, :self.lact].copy()
        self.counts = st.cant_obs_i

    def get_state(self):
        mo = np.zeros((2, self.L))
        mo[:, :self.lact] = self.map
        return self.x[:self.T].T.copy(), mo, self.counts.copy(), self.lact
