#!/usr/bin/env python3
"""Fixture generator: the landmark-count trajectory of the S2 workload under the C oracle.

    python tests/golden/make_s2_trajectory.py [sweeps=80]     # ~10 min on 8 cores, CPU only

Runs `sweeps` consecutive red-black ICM sweeps of the synthetic S2 sequence (100 000 poses /
10 000 landmarks / 720 beams, icmslam_hip/synthetic.py, seed 20181) on oracle/icm_oracle_c.c
(grid-accelerated association = the brute-force answer; OpenMP over poses) and records, per
sweep, landmarks_actuales after Mapa.filtrar and the raw label count before it, plus the sweep
at which the oracle raises the reference's IndexError (labels beyond L,
scripts/ICM_SLAM_tools.py:191), if it does.  Writes tests/golden/s2_k_trajectory.json.
tests/test_gpu_scale_parity.py replays the same run on the GPU against it."""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, "icm-slam_amd")]

from ICM_SLAM_tools import ConfigICM  # noqa: E402
from icmslam_hip.synthetic import WORKLOADS, make_workload  # noqa: E402
from oracle import c_oracle as co  # noqa: E402


def main(sweeps):
    wl = make_workload(*WORKLOADS["S2"])
    cfg = ConfigICM(D=wl.config)
    kept = co.prefilter(cfg, wl.scans.T)
    x, mv, la = wl.x_init.copy(), wl.map_init, wl.K
    traj, err = [], None
    for it in range(sweeps):
        try:
            mv, cnt, la, raw = co.sweep(cfg, kept, wl.u, wl.odometry, wl.x0, mv, x, la, "redblack")
        except IndexError:
            err = it + 1
            break
        traj.append([it + 1, int(la), int(raw[2])])
        print(traj[-1], flush=True)
    json.dump({"workload": "S2", "schedule": "redblack", "L": int(cfg.L), "columns": ["sweep", "landmarks_actuales", "labels_before_filtrar"],
               "trajectory": traj, "index_error_sweep": err}, open(os.path.join(HERE, "s2_k_trajectory.json"), "w"))


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 80)
