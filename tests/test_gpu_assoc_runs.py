"""Phase A by geometric runs (k_assoc_runs) against the beam-by-beam rule it replaces on the hot path.

The reference decides every beam on its own: cdist / argmin / gate of Mapa.actualizar (scripts/ICM_SLAM_tools.py:168-172).
The run form settles a whole cluster of neighbouring returns from its bounding circle and falls back to the per-beam rule
where that is not conclusive, so the bar is exactness: the label of EVERY kept beam, every entry count and every counter
equal to the beam-by-beam kernel's (and to the brute-force kernel's and the oracle's); sums of body points are added up
per run instead of per beam, i.e. in another order (~1e-16 relative), so real-valued outputs are held to 1e-9 like
everywhere else in this suite and reported when they are not bit-identical.
"""
import numpy as np
import pytest

from util import Cfg, dataset, gold

pytestmark = pytest.mark.gpu

TOL = 1e-9


def _synthetic(name):
    from ICM_SLAM_tools import ConfigICM
    from icmslam_hip.synthetic import WORKLOADS, make_workload
    wl = make_workload(*WORKLOADS[name])
    return wl, ConfigICM(D=wl.config)


def _engine_dataset():
    from icmslam_hip import SweepEngine
    zz, odo, u = dataset()
    g = gold("init_pass.npz")
    eng = SweepEngine(Cfg())
    eng.upload(zz, odo, u)
    return eng, g["map_init"].copy(), g["x_init"].copy(), g["x_init"][:, 0].copy()


def _engine_synth(name):
    from icmslam_hip import SweepEngine
    wl, cfg = _synthetic(name)
    eng = SweepEngine(cfg)
    eng.upload(wl.scans, wl.odometry, wl.u, pose_major=True)
    return eng, wl.map_init.copy(), wl.x_init.copy(), wl.x0.copy()


def _check_runs(eng, thr):
    """Structure of the runs: a partition of every pose's kept beams into consecutive stretches, cut by the stated rule,
    with a bounding circle that really bounds and the sum that really is the sum."""
    off, bk, d, bx, by = eng.kept_beams()
    roff, c, sb, r, k, f = eng.runs()
    n_runs, _ = eng.run_counts()
    assert roff[0] == 0 and roff[-1] == n_runs == len(k)
    nb = np.diff(off)
    pose_of_run = np.repeat(np.arange(len(nb)), np.diff(roff))
    assert np.array_equal(np.bincount(pose_of_run, weights=k, minlength=len(nb)).astype(np.int64), nb), "the runs of a pose hold all its beams"
    # consecutive: the first run starts at the pose's first beam, each next one where the last ended
    first_of_pose = roff[:-1][np.diff(roff) > 0]
    assert not f[first_of_pose].any()
    same = pose_of_run[1:] == pose_of_run[:-1]
    assert np.array_equal((f[:-1] + k[:-1])[same], f[1:][same])
    assert k.min() >= 1 and k.max() <= 64
    start = off[pose_of_run] + f                    # absolute index of each run's first beam
    run_of_beam = np.repeat(np.arange(n_runs), k)   # (runs are in beam order)
    assert np.array_equal(np.repeat(start, k) + (np.arange(len(run_of_beam)) - np.repeat(np.cumsum(k) - k, k)), np.arange(len(bx)))
    # the sum and the circle
    sx = np.bincount(run_of_beam, weights=bx, minlength=n_runs)
    sy = np.bincount(run_of_beam, weights=by, minlength=n_runs)
    assert np.abs(sb[:, 0] - sx).max() <= 1e-12 * max(1.0, np.abs(sx).max()) and np.abs(sb[:, 1] - sy).max() <= 1e-12 * max(1.0, np.abs(sy).max())
    dist = np.hypot(bx - c[run_of_beam, 0], by - c[run_of_beam, 1])
    assert (dist <= r[run_of_beam].astype(np.float64)).all(), "every beam inside its run's circle"
    assert np.abs(c[:, 0] - sx / k).max() <= 1e-12 and np.abs(c[:, 1] - sy / k).max() <= 1e-12
    # the cutting rule: inside a run, neighbours within the gap and everybody within the extent of the first beam
    gap, ext = 0.35 * thr, 0.5 * thr
    inner = np.ones(len(bx), dtype=bool)
    inner[start] = False
    j = np.flatnonzero(inner)
    assert (np.hypot(bx[j] - bx[j - 1], by[j] - by[j - 1]) <= gap * (1 + 1e-12)).all()
    assert (np.hypot(bx - bx[start][run_of_beam], by - by[start][run_of_beam]) <= ext * (1 + 1e-12)).all()
    # ... and a run begins only where the rule asks for it
    heads = start[f > 0]
    prev_start = start[np.flatnonzero(f > 0) - 1]
    g_ = np.hypot(bx[heads] - bx[heads - 1], by[heads] - by[heads - 1])
    e_ = np.hypot(bx[heads] - bx[prev_start], by[heads] - by[prev_start])
    full = k[np.flatnonzero(f > 0) - 1] == 64
    assert ((g_ > gap * (1 - 1e-12)) | (e_ > ext * (1 - 1e-12)) | full).all()
    return n_runs, float(k.mean()), float(r.max())


@pytest.mark.parametrize("which", ["dataset", "tiny", "S1"])
def test_runs_partition_the_kept_beams(which):
    eng = _engine_dataset()[0] if which == "dataset" else _engine_synth(which)[0]
    n, kmean, rmax = _check_runs(eng, 1.0)
    print("%s: %d runs, %.2f beams per run, largest radius %.3f m" % (which, n, kmean, rmax))
    eng.close()


def test_device_runs_equal_the_oracle_cut_and_its_settled_share():
    """k_run_build against the CPU statement of the cutting rule (oracle.cut_runs) on data_IJAC2018: the same runs, beam
    for beam; and the device's count of runs that went beam by beam in sweep 1 equals the number of runs the oracle's
    bounding-circle test (oracle.run_decision) does not settle."""
    from oracle import icm_oracle as o
    eng, map0, x_init, x0 = _engine_dataset()
    off, bk, d, bx, by = eng.kept_beams()
    roff, c, sb, r, k, f = eng.runs()
    mine = [[(int(f[q]), int(k[q])) for q in range(roff[t], roff[t + 1])] for t in range(len(off) - 1)]
    unsettled = 0
    for t in range(len(off) - 1):
        body = np.stack((bx[off[t]:off[t + 1]], by[off[t]:off[t + 1]]), axis=1)
        want = o.cut_runs(body, 1.0)
        assert mine[t] == want, "runs of scan %d" % t
        pose = x0 if t == 0 else x_init[:, t]
        for first, cnt in want:
            b = body[first:first + cnt]
            cc = b.sum(axis=0) / cnt
            rr = np.sqrt(((b - cc) ** 2).sum(axis=1)).max() * 1.000001 + 1e-12
            if o.run_decision(o.project_beams(pose, cc[None, :])[0], rr, map0, map0.shape[1], 1.0) is None:
                unsettled += 1
    eng.set_assoc_form("runs")
    eng.set_state(map0, x_init, x0)
    before = eng.run_counts()[1]
    eng.sweep_device("sequential")
    bbb = eng.run_counts()[1] - before
    print("data_IJAC2018: %d runs equal the oracle's cut; beam by beam: device %d, oracle's rule %d" % (len(k), bbb, unsettled))
    assert bbb == unsettled
    eng.close()


def _sweep_both_forms(eng, map0, x0_, xstart, sweeps=2, schedule="redblack"):
    out = {}
    for form in ("beams", "runs"):
        eng.set_assoc_form(form)
        eng.set_debug(True)
        eng.set_entry_path("hier")
        eng.set_state(map0, x0_, xstart)
        per = []
        for _ in range(sweeps):
            eng.sweep_device(schedule)
            lab = eng.association()[0].copy()
            yr, cr, lr = eng.raw_map()
            x, m, cnt, K = eng.get_state()
            per.append((lab, yr, cr, lr, x, m, cnt, K))
        out[form] = per
    eng.set_debug(False)
    return out


def _compare_forms(out, what):
    for it, (a, b) in enumerate(zip(out["beams"], out["runs"]), 1):
        assert np.array_equal(a[0], b[0]), "%s sweep %d: labels differ on %d beams" % (what, it, int((a[0] != b[0]).sum()))
        assert a[3] == b[3] and np.array_equal(a[2], b[2]), "raw counters"
        assert a[7] == b[7] and np.array_equal(a[6], b[6]), "filtered counters"
        dr, dm, dx = np.abs(a[1] - b[1]).max(), np.abs(a[5] - b[5]).max(), np.abs(a[4] - b[4]).max()
        print("%s sweep %d: %d beams, labels exact; run form vs beam form: max|draw| %.2e max|dmap| %.2e max|dx| %.2e (poses bit-identical: %s)"
              % (what, it, a[0].size, dr, dm, dx, np.array_equal(a[4], b[4])))
        assert dr <= TOL and dm <= TOL and dx <= TOL


@pytest.mark.parametrize("which", ["dataset", "tiny", "S1"])
def test_run_form_equals_beam_form(which):
    """Labels of every beam, counters and K exact; raw map, map and poses <= 1e-9 (scripts/ICM_SLAM_tools.py:168-197)."""
    eng, map0, x_init, x0 = _engine_dataset() if which == "dataset" else _engine_synth(which)
    out = _sweep_both_forms(eng, map0, x_init, x0, sweeps=2, schedule="sequential" if which == "dataset" else "redblack")
    _compare_forms(out, which)
    n_runs, beam_by_beam = eng.run_counts()
    print("%s: %d runs per sweep, %d went beam by beam over 2 sweeps (%.2f %%)" % (which, n_runs, beam_by_beam, 100.0 * beam_by_beam / max(2 * n_runs, 1)))
    eng.close()


def test_crowded_and_distant_landmarks_take_the_beam_by_beam_path():
    """A map the bounding-circle test cannot settle: two thirds of the landmarks doubled 0.25 m beside themselves (the
    nearest candidate never wins by a run's diameter), the other third moved 0.92 m away (the beams of one trunk straddle
    the gate: some are gated out and create landmarks) and a crowd of six around a few trunks (more candidates than a grid
    record holds).
    The run form must then BE the per-beam rule: labels of every beam against the brute-force kernel (every beam against
    every landmark, scripts/ICM_SLAM_tools.py:168-172 literally) and against the beam-by-beam kernel."""
    from icmslam_hip import SweepEngine
    wl, cfg = _synthetic("tiny")
    rng = np.random.default_rng(7)
    m = wl.map_init.copy()
    K = m.shape[1]
    far = m[:, ::3] + np.array([[0.92], [0.0]])
    near = np.concatenate((m[:, 1::3], m[:, 2::3]), axis=1)
    v = rng.normal(0, 1, near.shape)
    twin = near + 0.25 * v / np.linalg.norm(v, axis=0)
    crowd = np.concatenate([m[:, [i]] + 0.3 * np.stack((np.cos(np.arange(6) * np.pi / 3), np.sin(np.arange(6) * np.pi / 3))) for i in (1, 20, 41)], axis=1)
    dense = np.concatenate((near, far, twin, crowd), axis=1)
    cfg.L = dense.shape[1] + 4096
    eng = SweepEngine(cfg)
    eng.upload(wl.scans, wl.odometry, wl.u, pose_major=True)
    labs = {}
    for form, brute in (("runs", False), ("beams", False), ("brute", True)):
        eng.set_assoc_form("beams" if form == "brute" else form)
        eng.set_brute_force(brute)
        eng.set_debug(True)
        eng.set_state(dense, wl.x_init, wl.x0)
        eng.sweep_device("redblack")
        labs[form] = (eng.association()[0].copy(), eng.raw_map(), eng.get_state())
    eng.set_brute_force(False)
    n_runs, bbb = eng.run_counts()
    print("crowded map: %d landmarks (was %d), %d runs, %d beam by beam (%.1f %%), %d beams gated out"
          % (dense.shape[1], K, n_runs, bbb, 100.0 * bbb / n_runs, int((labs["brute"][0] >= dense.shape[1]).sum())))
    assert bbb > n_runs // 4, "this map is meant to defeat the bounding-circle test"
    assert (labs["brute"][0] >= dense.shape[1]).any(), "... and to gate beams out (fresh ids)"
    for other in ("beams", "brute"):
        assert np.array_equal(labs["runs"][0], labs[other][0]), "labels differ from the %s kernel's on %d beams" % (other, int((labs["runs"][0] != labs[other][0]).sum()))
        (ya, ca, la), (yb, cb, lb) = labs["runs"][1], labs[other][1]
        assert la == lb and np.array_equal(ca, cb) and np.abs(ya - yb).max() <= TOL
        xa, xb = labs["runs"][2][0], labs[other][2][0]
        assert np.abs(xa - xb).max() <= TOL
    eng.close()


def test_run_form_against_the_reference_labels_of_sweep_1():
    """The run form against the reference's OWN labels and running-mean targets of every kept beam of sweep 1 on
    data_IJAC2018 (golden sweep1_perpose.npz, written by the imported reference: scripts/ICM_SLAM_tools.py:168-197)."""
    eng, map0, x_init, x0 = _engine_dataset()
    pp = gold("sweep1_perpose.npz")
    eng.set_assoc_form("runs")
    eng.set_debug(True)
    eng.set_state(map0, x_init, x0)
    eng.sweep_device("sequential")
    lab, tx, ty = eng.association()
    off = eng.kept_beams()[0]
    goff = pp["offsets"]
    nb = 0
    for i, t in enumerate(pp["t"]):
        sl = slice(off[t], off[t + 1])
        assert np.array_equal(lab[sl], pp["labels"][goff[i]:goff[i + 1]]), "labels of pose %d" % t
        tg = pp["targets"][goff[i]:goff[i + 1]]
        assert np.abs(tx[sl] - tg[:, 0]).max() <= TOL and np.abs(ty[sl] - tg[:, 1]).max() <= TOL
        nb += sl.stop - sl.start
    n_runs, bbb = eng.run_counts()
    print("data_IJAC2018 sweep 1: labels of %d beams exact against the reference's; %d of %d runs went beam by beam" % (nb, bbb, n_runs))
    eng.close()
