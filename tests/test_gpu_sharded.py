"""The sharded sweep on ONE GPU: several in-process ranks bound to the same exchange buffers
(the collectives become shared memory), against the unsharded sweep; plus the RCCL
all-gather path itself with a 1-rank process group."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _workload():
    from ICM_SLAM_tools import ConfigICM
    from icmslam_hip.synthetic import make_workload
    wl = make_workload(1900, 100, 180)  # lane, turn (some poses see nothing), lane
    return wl, ConfigICM(D=wl.config)


def _single(wl, cfg, sweeps):
    from icmslam_hip import SweepEngine
    eng = SweepEngine(cfg)
    eng.upload(wl.scans, wl.odometry, wl.u, pose_major=True)
    eng.set_state(wl.map_init, wl.x_init, wl.x0)
    for _ in range(sweeps):
        eng.sweep_device("redblack")
    out = eng.get_state()
    eng.close()
    return out


@pytest.mark.parametrize("world", [2, 3, 4, 7])      # 7: shards of 272 poses and a last one of 268
def test_virtual_ranks_match_unsharded(world):
    import torch
    from icmslam_hip import SweepEngine
    from icmslam_hip.sharded import NoComm, ShardedSweep, partition, run_virtual_ranks
    wl, cfg = _workload()
    sweeps = 3
    x1, m1, c1, K1 = _single(wl, cfg, sweeps)
    blk, parts = partition(wl.T, world)
    engines, runners, buffers = [], [], None
    for r, (a, b) in enumerate(parts):
        e = SweepEngine(cfg)
        e.upload(wl.scans[a:b], wl.odometry, wl.u, t_begin=a, t_end=b, pose_major=True)
        run = ShardedSweep(e, r, world, wl.T, comm=NoComm(), buffers=buffers)
        buffers = (run.stats, run.poses)
        run.set_state(wl.map_init, wl.x_init, wl.x0)
        engines.append(e)
        runners.append(run)
    run_virtual_ranks(engines, sweeps)
    torch.cuda.synchronize()
    assert all(e.entry_path() == "hier" for e in engines), "the sharded sweep runs the hierarchical pipeline"
    for e in engines:
        x, m, c, K = e.get_state()
        assert K == K1
        assert np.abs(m[:, :K] - m1[:, :K1]).max() <= 1e-9
        assert np.array_equal(c, c1)
        d = np.abs(x - x1).max(axis=0)
        print("world %d: max|dx| %.3e, poses above 1e-9: %d" % (world, d.max(), int((d > 1e-9).sum())))
        assert d.max() <= 1e-9
    for e in engines:
        e.close()


def test_rccl_all_gather_path_one_rank():
    """ShardedSweep with the real torch.distributed 'nccl' (= RCCL) backend, world size 1:
    same result as the plain device-resident sweep."""
    import torch
    import torch.distributed as dist
    from icmslam_hip import SweepEngine
    from icmslam_hip.sharded import ShardedSweep
    wl, cfg = _workload()
    x1, m1, c1, K1 = _single(wl, cfg, 2)
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29517", rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    try:
        e = SweepEngine(cfg)
        e.upload(wl.scans, wl.odometry, wl.u, pose_major=True)
        run = ShardedSweep(e, 0, 1, wl.T)
        run.set_state(wl.map_init, wl.x_init, wl.x0)
        for _ in range(2):
            run.sweep("redblack")
        torch.cuda.synchronize()
        x, m, c, K = run.get_state()
        e.close()
    finally:
        dist.destroy_process_group()
    assert K == K1 and np.array_equal(x, x1) and np.array_equal(m, m1)


def test_library_collectives_one_rank():
    """The collectives issued by the C library itself (icm_comm_init / icm_sweep_sharded /
    icm_gather_poses: RCCL resolved with dlopen, all-gathers on the handle's stream), world size 1 --
    no torch.distributed anywhere: same state as the plain device-resident sweep, bit for bit."""
    from icmslam_hip import SweepEngine
    from icmslam_hip.sharded import LibrarySweep
    wl, cfg = _workload()
    x1, m1, c1, K1 = _single(wl, cfg, 3)
    e = SweepEngine(cfg)
    assert e.comm_available()
    e.upload(wl.scans, wl.odometry, wl.u, pose_major=True)
    run = LibrarySweep(e, 0, 1, wl.T, bcast=lambda payload: payload)
    run.set_state(wl.map_init, wl.x_init, wl.x0)
    for _ in range(3):
        run.sweep("redblack")
    x, m, c, K = run.get_state()
    run.close()
    e.close()
    assert K == K1 and np.array_equal(x, x1) and np.array_equal(m, m1) and np.array_equal(c, c1)


def test_library_collectives_refuse_a_wrong_partition():
    from icmslam_hip import SweepEngine
    from icmslam_hip.engine import IcmError
    wl, cfg = _workload()
    e = SweepEngine(cfg)
    e.upload(wl.scans[:100], wl.odometry, wl.u, t_begin=0, t_end=100, pose_major=True)
    uid = e.comm_unique_id()
    with pytest.raises((IcmError, ValueError)):
        e.comm_init(uid, 0, 1)          # 100 poses are not block 0 of a 1-rank partition of T poses
    with pytest.raises((IcmError, ValueError)):
        e.sweep_sharded()               # no communicator
    e.close()


def _native_worker(rank, world, port, out_path):
    """One rank of a multi-process sharded sweep; every rank uses cuda:0 (one-GPU box), the
    collectives go through gloo -- the path under test is the library's send-side buffers
    (icm_bind_exchange_send), halo pack in icm_sweep_solve and icm_halo_unpack."""
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "icm-slam_amd"), os.path.join(root, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch
    import torch.distributed as dist
    from icmslam_hip import SweepEngine
    from icmslam_hip.sharded import ShardedSweep, TorchComm, partition
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)

    class HostHopComm(TorchComm):
        """TorchComm whose all-gathers hop through host memory (gloo has no device collectives)."""

        def _ag(self, out, inp):
            o, i = out.cpu(), inp.cpu()
            self.dist.all_gather_into_tensor(o, i, group=self.group)
            out.copy_(o)

        def gather_stats(self, sw):
            assert sw.native
            self._ag(sw.stats, sw.stats_send)

        def halo(self, sw):
            assert sw.native
            self._ag(sw.halo_recv, sw.halo_send)
            sw.eng.halo_unpack()

        def all_gather(self, buf, r, count):
            mine = buf[r * count:(r + 1) * count].clone()
            self._ag(buf, mine)

    wl, cfg = _workload()
    _, parts = partition(wl.T, world)
    a, b = parts[rank]
    e = SweepEngine(cfg)
    e.upload(wl.scans[a:b], wl.odometry, wl.u, t_begin=a, t_end=b, pose_major=True)
    run = ShardedSweep(e, rank, world, wl.T, comm=HostHopComm())
    assert run.native
    run.set_state(wl.map_init, wl.x_init, wl.x0)
    for _ in range(3):
        run.sweep("redblack")
    torch.cuda.synchronize()
    x, m, c, K = run.get_state()
    np.savez(out_path % rank, x=x, m=m[:, :K], c=c, K=K, path=e.entry_path())
    e.close()
    dist.barrier()
    dist.destroy_process_group()


def test_three_processes_library_side_halo(tmp_path):
    """Three OS processes (ranks) on the one GPU, real exchanges between them: the library's
    send buffers + halo pack/unpack give the unsharded result; the interior rank has a
    neighbour on both sides."""
    import torch.multiprocessing as mp
    world = 3
    wl, cfg = _workload()
    x1, m1, c1, K1 = _single(wl, cfg, 3)
    out = str(tmp_path / "rank%d.npz")
    mp.spawn(_native_worker, args=(world, 29561, out), nprocs=world, join=True)
    for r in range(world):
        g = np.load(out % r)
        assert str(g["path"]) == "hier"
        assert int(g["K"]) == K1 and np.array_equal(g["c"], c1)
        assert np.abs(g["m"] - m1[:, :K1]).max() <= 1e-9
        d = np.abs(g["x"] - x1).max(axis=0)
        print("rank %d: max|dx| %.3e, poses above 1e-9: %d" % (r, d.max(), int((d > 1e-9).sum())))
        assert d.max() <= 1e-9
