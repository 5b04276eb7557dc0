import sys, time
sys.path.insert(0, 'icm-slam_amd'); sys.path.insert(0, '.')
import numpy as np, torch
from ICM_SLAM_tools import ConfigICM
from icmslam_hip import SweepEngine
from icmslam_hip.sharded import NoComm, ShardedSweep, partition
from icmslam_hip.synthetic import WORKLOADS, make_workload
T,K,B = WORKLOADS["S2"]


class SoloComm:
    """One rank of a job run alone, through the exchange path a real job takes (its own statistics and halo looped
    back, the other ranks' slots left as they are): unlike NoComm this lets the sweep be queued whole."""

    def __init__(self):
        self.filled = False

    def gather_stats(self, sw):
        sw.stats[sw.rank * sw.stride:(sw.rank + 1) * sw.stride].copy_(sw.stats_send)   # (one copy per sweep, like a collective)
        if self.filled:
            return
        self.filled = True
        L3 = sw.stride - 8
        own = sw.stats_send[L3:L3 + 8]
        for r in range(sw.world):          # the absent neighbours' boundary poses, once: this rank's own edge poses stand in
            if r != sw.rank:
                hd = sw.stats[r * sw.stride + L3:r * sw.stride + L3 + 8]
                hd.zero_()
                hd[2:5].copy_(own[5:8] if r > sw.rank else own[2:5])
                hd[5:8].copy_(own[5:8] if r > sw.rank else own[2:5])
                src = own[5:8] if r > sw.rank else own[2:5]
                sw.halo_recv[r * 6:r * 6 + 3].copy_(src)
                sw.halo_recv[r * 6 + 3:r * 6 + 6].copy_(src)

    def halo(self, sw):
        sw.halo_recv[sw.rank * 6:(sw.rank + 1) * 6].copy_(sw.halo_send)
        sw.eng.halo_unpack()

    def all_gather(self, buf, rank, count):
        pass


careful = len(sys.argv) > 1 and sys.argv[1] == "careful"   # host look at phase A's flags in the middle of every sweep
for world in (1,2,4,8):
    blk, parts = partition(T, world)
    a,b = parts[0]
    wl = make_workload(T,K,B,t_begin=a,t_end=b); cfg = ConfigICM(D=wl.config)
    eng = SweepEngine(cfg, 0); eng.upload(wl.scans, wl.odometry, wl.u, t_begin=a, t_end=b, pose_major=True)
    run = ShardedSweep(eng, 0, world, T, comm=NoComm() if careful else SoloComm())
    run.set_state(wl.map_init, wl.x_init, wl.x0)
    for _ in range(3): run.sweep("redblack")
    torch.cuda.synchronize(); t0=time.perf_counter(); n=20
    for _ in range(n): run.sweep("redblack")
    torch.cuda.synchronize(); dt=(time.perf_counter()-t0)/n
    eng.enable_timing(True)
    for _ in range(3): run.sweep("redblack")
    kt=eng.kernel_times(); eng.enable_timing(False)
    print('careful' if careful else 'queued whole', 'world',world,'rank-0 share: %.3f ms/sweep (no collectives)'%(dt*1e3), {k: round(v[0]/3,3) for k,v in kt.items() if v[1]})
    eng.close()
