"""What ONE rank of a WEAK-scaled job costs per sweep on one MI355X, collectives excluded: one
sequence of world x 100 000 poses through the S2 landmark field, a 100 000-pose block per rank; the rank runs alone
through the exchange path a real job takes (sweeps queued whole; tools/solo_comm.py), world 1 / 2 / 8, ranks 0 and 5.

    python tools/shard_cost_weak.py
"""
import sys, time
sys.path.insert(0, 'icm-slam_amd'); sys.path.insert(0, '.'); sys.path.insert(0, 'tools')
import numpy as np, torch
from ICM_SLAM_tools import ConfigICM
from icmslam_hip import SweepEngine
from icmslam_hip.sharded import ShardedSweep, partition
from solo_comm import SoloComm
from icmslam_hip.synthetic import WORKLOADS, make_workload
T1,K,B = WORKLOADS["S2"]
for world, rank in ((1,0),(2,1),(8,0),(8,5)):
    T = T1*world
    blk, parts = partition(T, world)
    a,b = parts[rank]
    wl = make_workload(T,K,B,t_begin=a,t_end=b); cfg = ConfigICM(D=wl.config)
    eng = SweepEngine(cfg, 0); eng.upload(wl.scans, wl.odometry, wl.u, t_begin=a, t_end=b, pose_major=True, ghost_scan=wl.ghost_scan if rank else None)
    run = ShardedSweep(eng, rank, world, T, comm=SoloComm(wl))
    run.set_state(wl.map_init, wl.x_init, wl.x0)
    for _ in range(3): run.sweep("redblack")
    torch.cuda.synchronize(); t0=time.perf_counter(); n=20
    for _ in range(n): run.sweep("redblack")
    torch.cuda.synchronize(); dt=(time.perf_counter()-t0)/n
    eng.enable_timing(True)
    for _ in range(3): run.sweep("redblack")
    kt=eng.kernel_times(); eng.enable_timing(False)
    print('weak world',world,'rank',rank,'share: %.3f ms/sweep (no collectives)'%(dt*1e3), {k: round(v[0]/3,3) for k,v in kt.items() if v[1]}, eng.last_stats(), flush=True)
    eng.close()
