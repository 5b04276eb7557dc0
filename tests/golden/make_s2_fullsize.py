#!/usr/bin/env python3
"""Fixture generator: pose-level state of the FULL S2 workload after red-black sweeps 1 and 2
under the C oracle.

    python tests/golden/make_s2_fullsize.py        # ~2 min on 8 cores, CPU only

Runs two consecutive red-black ICM sweeps of the synthetic S2 sequence (100 000 poses / 10 000
landmarks / 720 beams, icmslam_hip/synthetic.py, seed 20181 -- the workload bench.py quotes) on
oracle/icm_oracle_c.c (grid-accelerated association = the brute-force answer, per-beam energy,
running-mean recurrence, SciPy's Nelder-Mead; OpenMP over poses) and stores, per sweep: every
pose, the refined map and counters, the raw map and counters before Mapa.filtrar, and one
64-bit digest per pose of the labels of its kept beams (tests/util.py::label_digest).
Writes tests/golden/s2_fullsize.npz (~5 MB).  tests/test_gpu_scale_parity.py compares the HIP
state -- unsharded and on 8 virtual ranks -- against it: labels / counters exact, map and EVERY
pose <= 1e-9.  Semantics pinned: scripts/ICM_SLAM_tools.py:167-197, scripts/ICM_ROS.py:141-158."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, "icm-slam_amd"), os.path.join(ROOT, "tests")]

from ICM_SLAM_tools import ConfigICM  # noqa: E402
from icmslam_hip.synthetic import WORKLOADS, make_workload  # noqa: E402
from oracle import c_oracle as co  # noqa: E402
from util import label_digest  # noqa: E402


def main():
    wl = make_workload(*WORKLOADS["S2"])
    cfg = ConfigICM(D=wl.config)
    kept = co.prefilter(cfg, wl.scans.T)
    x, mv, la = wl.x_init.copy(), wl.map_init, wl.K
    out = {"nnz": np.int64(kept[0][-1]), "kept_digest": label_digest(kept[0], kept[1])}
    for it in (1, 2):
        a = {}
        mv, cnt, la, raw = co.sweep(cfg, kept, wl.u, wl.odometry, wl.x0, mv, x, la, "redblack", assoc=a)
        nz = int(np.flatnonzero(cnt).max()) + 1 if cnt.any() else 0
        out.update({"x%d" % it: x.copy(), "map%d" % it: mv.copy(), "counts%d" % it: cnt[:nz].copy(), "K%d" % it: np.int64(la),
                    "raw_lact%d" % it: np.int64(raw[2]), "raw_map%d" % it: raw[0][:, :raw[2]].copy(),
                    "raw_counts%d" % it: raw[1][:raw[2]].copy(), "labels%d" % it: label_digest(kept[0], a["labels"])})
        print("sweep %d: K %d, raw labels %d, sum|x - x_init| %.6f" % (it, la, raw[2], np.abs(x - wl.x_init).sum()), flush=True)
    np.savez_compressed(os.path.join(HERE, "s2_fullsize.npz"), **out)


if __name__ == "__main__":
    main()
