"""What ONE rank of a strong-scaled S2 job costs per sweep on one MI355X, collectives excluded:
the rank runs alone through the exchange path a real job takes -- its own message looped back, its neighbours' slots
filled once with stand-ins (the rank below: every landmark seen 100 times at its map position, so that the ghost pose's
running means exist and no landmark is pruned; boundary poses from the initial state) -- for world sizes 1 / 2 / 4 / 8, rank 0 (no ghost pose) and rank 1 (ghost pose
and a neighbour on both sides).

    python tools/shard_cost_strong.py [careful]        # careful: host look at phase A's flags in the middle of every sweep
"""
import sys, time
sys.path.insert(0, 'icm-slam_amd'); sys.path.insert(0, '.')
import numpy as np, torch
from ICM_SLAM_tools import ConfigICM
from icmslam_hip import SweepEngine
from icmslam_hip.sharded import ShardedSweep, partition
from icmslam_hip.synthetic import WORKLOADS, make_workload
T, K, B = WORKLOADS["S2"]
sys.path.insert(0, "tools")
from solo_comm import SoloComm   # noqa: E402


careful = len(sys.argv) > 1 and sys.argv[1] == "careful"
for world in (1, 2, 4, 8):
    for rank in ((0,) if world == 1 else (0, 1)):
        blk, parts = partition(T, world)
        a, b = parts[rank]
        wl = make_workload(T, K, B, t_begin=a, t_end=b); cfg = ConfigICM(D=wl.config)
        eng = SweepEngine(cfg, 0)
        eng.upload(wl.scans, wl.odometry, wl.u, t_begin=a, t_end=b, pose_major=True, ghost_scan=wl.ghost_scan if rank else None)
        run = ShardedSweep(eng, rank, world, T, comm=SoloComm(wl))
        if careful:
            eng.set_optimistic = None    # (ShardedSweep then takes the careful form every sweep)
        run.set_state(wl.map_init, wl.x_init, wl.x0)
        for _ in range(3): run.sweep("redblack")
        torch.cuda.synchronize(); t0 = time.perf_counter(); n = 20
        for _ in range(n): run.sweep("redblack")
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
        eng.enable_timing(True)
        for _ in range(3): run.sweep("redblack")
        kt = eng.kernel_times(); eng.enable_timing(False)
        print('careful' if careful else 'queued whole', 'world', world, 'rank', rank, 'share: %.3f ms/sweep (collective excluded)' % (dt * 1e3),
              {k: round(v[0] / 3, 3) for k, v in kt.items() if v[1]}, flush=True)
        eng.close()
