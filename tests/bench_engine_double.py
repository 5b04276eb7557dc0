"""Engine double for bench.py's launcher test (`--engine-factory bench_engine_double:make`):
the oracle-backed phase-API engine of tests/oracle_shard_engine.py, so that
`python bench.py --gpus 2 --device cpu` exercises the self-launch, the rank environment, the
sharded driver and the JSON relay under torch.distributed/gloo on a machine without a GPU.
TEST INFRASTRUCTURE: bench.py itself never imports oracle/ outside its cpu_baseline leg."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (HERE, ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def make(cfg, wl, rank, world, t_begin, t_end):
    from oracle import icm_oracle as o
    from oracle_shard_engine import OracleShardEngine
    ocfg = o.OracleConfig.from_config(cfg)
    scans_BT = np.zeros((wl.B, wl.T))
    scans_BT[:, t_begin:t_end] = wl.scans.T      # this rank generated only its own shard ...
    if t_begin >= 1 and wl.ghost_scan is not None:
        scans_BT[:, t_begin - 1] = wl.ghost_scan  # ... and the scan of its ghost pose
    return OracleShardEngine(ocfg, scans_BT, wl.u, wl.odometry, t_begin, t_end)
