"""The claim behind phase A by runs, checked on the CPU against the reference's literal per-beam rule (no GPU, no HIP):
whenever the bounding-circle test settles a run, EVERY beam of the run gets that label from `cdist` / `argmin` / gate
(scripts/ICM_SLAM_tools.py:168-172, restated in oracle.associate and pinned to the reference's own labels by
tests/test_oracle_golden.py).  Checked on data_IJAC2018 with the reference's initial state and with perturbed poses and
maps (denser, shifted: the regimes in which the test must refuse), and the share of settled runs is reported -- it is what
the HIP kernel's hot path depends on."""
import numpy as np
import pytest

from oracle import icm_oracle as o
from util import dataset, gold


def _check(kept, poses, ref_map, lact, thr, what):
    settled = runs = beams = wrong = 0
    for t, kz in enumerate(kept):
        if kz.ndim != 2 or kz.shape[0] == 0:
            continue
        body = kz[:, 2:4]
        w = o.project_beams(poses[:, t], body)
        lab = o.associate(ref_map, lact, w, thr)      # the reference's rule, beam by beam
        for first, k in o.cut_runs(body, thr):
            b = body[first:first + k]
            c = b.sum(axis=0) / k
            r = np.sqrt(((b - c) ** 2).sum(axis=1)).max() * 1.000001 + 1e-12
            cw = o.project_beams(poses[:, t], c[None, :])[0]
            got = o.run_decision(cw, r, ref_map, lact, thr)
            runs += 1
            beams += k
            if got is not None:
                settled += 1
                wrong += int((lab[first:first + k] != got).sum())
    print("%s: %d runs (%.2f beams each), %.1f %% settled by the bounding circle, %d beams labelled differently from the per-beam rule"
          % (what, runs, beams / max(runs, 1), 100.0 * settled / max(runs, 1), wrong))
    assert wrong == 0
    return settled, runs


def test_a_settled_run_carries_the_reference_label_of_every_one_of_its_beams():
    cfg = o.OracleConfig()
    zz, odo, u = dataset()
    kept = o.prefilter_all(zz, cfg)
    init = gold("init_pass.npz")
    x, m = init["x_init"], init["map_init"]
    lact = m.shape[1]
    s, n = _check(kept, x, m, lact, cfg.dist_thr, "data_IJAC2018, reference initial state")
    assert s > 0.9 * n
    rng = np.random.default_rng(17)
    # poses off by up to 0.3 m / 0.05 rad: beams straddle the gate, nearest landmarks change
    xp = x + np.vstack((rng.uniform(-0.3, 0.3, (2, x.shape[1])), rng.uniform(-0.05, 0.05, (1, x.shape[1]))))
    _check(kept[:600], xp, m, lact, cfg.dist_thr, "perturbed poses")
    # a denser map: every landmark doubled 0.25 m beside itself, and a crowd around one -- the test must refuse, never err
    v = rng.normal(0, 1, m.shape)
    dense = np.concatenate((m, m + 0.25 * v / np.linalg.norm(v, axis=0),
                            m[:, [3]] + 0.3 * np.stack((np.cos(np.arange(6.0)), np.sin(np.arange(6.0))))), axis=1)
    s2, n2 = _check(kept[:600], x, dense, dense.shape[1], cfg.dist_thr, "doubled and crowded map")
    assert s2 < 0.5 * n2
    # a smaller gate: runs shrink with it (the cutting thresholds scale with dist_thr)
    cfg2 = o.OracleConfig(dist_thr=0.4)
    kept2 = o.prefilter_all(zz[:, :400], cfg2)
    _check(kept2, x, m, lact, cfg2.dist_thr, "dist_thr = 0.4")


def test_runs_partition_a_scan_in_beam_order():
    rng = np.random.default_rng(3)
    for _ in range(50):
        n = int(rng.integers(0, 200))
        pts = np.cumsum(rng.normal(0, 0.12, (n, 2)), axis=0)
        runs = o.cut_runs(pts, 1.0)
        assert sum(k for _, k in runs) == n and all(1 <= k <= o.RUN_CAP for _, k in runs)
        assert [f for f, _ in runs] == list(np.cumsum([0] + [k for _, k in runs[:-1]])) if runs else n == 0
        for f, k in runs:
            b = pts[f:f + k]
            assert (np.hypot(*(b[1:] - b[:-1]).T) <= 0.35 + 1e-12).all() and (np.hypot(*(b - b[0]).T) <= 0.5 + 1e-12).all()
