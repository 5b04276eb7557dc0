"""World-size-2 run of the sharded sweep driver (icmslam_hip/sharded.py) over
torch.distributed/gloo on the CPU, with the oracle-backed engine double, against the
unsharded oracle sweep in the same red-black order."""
import os
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

from util import Cfg, ROOT, dataset, gold

T_SUB = 249
SWEEPS = 2


def _worker(rank, world, port, out_path):
    sys.path.insert(0, os.path.join(ROOT, "icm-slam_amd"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from icmslam_hip.sharded import ShardedSweep, partition
    from oracle import icm_oracle as o
    from oracle_shard_engine import OracleShardEngine
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    zz, odo, u = dataset()
    zz, odo, u = zz[:, :T_SUB], odo[:, :T_SUB], u[:, :T_SUB]
    init = gold("init_pass.npz")
    ocfg = o.OracleConfig.from_config(Cfg(cota=20.0))
    _, parts = partition(T_SUB, world)
    a, b = parts[rank]
    eng = OracleShardEngine(ocfg, zz, u, odo, a, b)
    run = ShardedSweep(eng, rank, world, T_SUB)
    run.set_state(init["map_init"], init["x_init"][:, :T_SUB], odo[:, 0], int(init["landmarks_actuales"]))
    for _ in range(SWEEPS):
        run.sweep("redblack")
    x, m, c, K = run.get_state()
    np.savez(out_path % rank, x=x, m=m[:, :K], c=c, K=K)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_ranks_gloo_match_unsharded_oracle(tmp_path, world):
    """world 2: one shard boundary; world 3: an interior rank with a neighbour on both sides (the
    pose blocks are NOT exchanged per sweep: one all-gather of statistics + boundary poses, ghost pose
    solved redundantly)."""
    from oracle import icm_oracle as o
    out = str(tmp_path / "rank%d.npz")
    mp.spawn(_worker, args=(world, 29533 + world, out), nprocs=world, join=True)
    zz, odo, u = dataset()
    zz, odo, u = zz[:, :T_SUB], odo[:, :T_SUB], u[:, :T_SUB]
    init = gold("init_pass.npz")
    ocfg = o.OracleConfig.from_config(Cfg(cota=20.0))
    st = o.MapState(ocfg, int(init["landmarks_actuales"]))
    x = np.ascontiguousarray(init["x_init"][:, :T_SUB]).copy()
    mv = init["map_init"].copy()
    for _ in range(SWEEPS):
        mv, x = o.sweep(ocfg, st, zz, u, odo, odo[:, 0], mv, x, schedule="redblack")
    res = [np.load(out % r) for r in range(world)]
    r0 = res[0]
    for r in res:
        assert int(r["K"]) == mv.shape[1]
        assert np.abs(r["m"] - mv).max() <= 1e-12     # sum/n vs the reference's recurrence
        assert np.abs(r["x"] - x).max() <= 1e-9
        assert np.array_equal(r["c"], st.cant_obs_i)
    for r1 in res[1:]:
        assert np.array_equal(r0["x"], r1["x"]) and np.array_equal(r0["m"], r1["m"])   # replicas agree


def test_partition_refuses_world_sizes_that_leave_a_rank_without_poses():
    """Blocks are rounded up to an even number of poses, so some (T, world) pairs that divide cleanly leave trailing ranks
    empty (T = 21, world = 7: blocks of 4 cover the sequence with six ranks): refused up front, naming the next smaller usable
    world size; every accepted partition covers the sequence with non-empty blocks that start at even poses."""
    import pytest
    from icmslam_hip.sharded import partition, world_fits
    with pytest.raises(ValueError, match="use 6 ranks"):
        partition(21, 7)
    with pytest.raises(ValueError, match="use 3 ranks"):
        partition(10, 4)
    for T in (5, 10, 21, 600, 1833, 100000):
        for world in range(1, 12):
            if not world_fits(T, world):
                with pytest.raises(ValueError):
                    partition(T, world)
                continue
            blk, parts = partition(T, world)
            assert parts[0][0] == 0 and parts[-1][1] == T and all(b > a and a % 2 == 0 for a, b in parts)
            assert all(parts[i][1] == parts[i + 1][0] for i in range(world - 1))
