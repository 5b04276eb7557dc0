"""The sweep's one collective for a rank of a multi-rank job that runs ALONE on a box (tools/shard_cost_strong.py,
tools/shard_cost_weak.py): its own message looped back, its neighbours' slots filled once with stand-ins."""
import numpy as np
import torch

HDR = 16


class SoloComm:
    """The sweep's one collective for a rank that runs alone: unlike NoComm this lets the sweep be queued whole."""

    def __init__(self, wl):
        self.filled, self.wl = False, wl

    def gather_stats(self, sw):
        st = sw.stride
        sw.stats[sw.rank * st:(sw.rank + 1) * st].copy_(sw.stats_send)   # (one copy per sweep, like a collective)
        if self.filled:
            return
        self.filled = True
        a, b = sw.own
        x = torch.tensor(np.ascontiguousarray(self.wl.x_init.T), dtype=torch.float64, device=sw.stats.device)
        if sw.rank > 0:                 # the rank below: every landmark observed 100 times at its map position (so that the
            lo = sw.stats[(sw.rank - 1) * st:sw.rank * st]   # ghost pose's running means exist and nothing is pruned), its last two poses
            lo.zero_()
            L, K = (st - HDR) // 3, self.wl.K
            m = torch.tensor(self.wl.map_init, dtype=torch.float64, device=sw.stats.device)
            lo[0:K].copy_(100.0 * m[0]); lo[L:L + K].copy_(100.0 * m[1]); lo[2 * L:2 * L + K].fill_(100.0)
            lo[st - HDR + 5:st - HDR + 8].copy_(x[a - 1])
            lo[st - HDR + 8:st - HDR + 11].copy_(x[a - 2])
        if sw.rank + 1 < sw.world:      # the rank above: its first pose
            hi = sw.stats[(sw.rank + 1) * st:(sw.rank + 2) * st]
            hi.zero_()
            hi[st - HDR + 2:st - HDR + 5].copy_(x[b])

    def all_gather(self, buf, rank, count):
        pass
