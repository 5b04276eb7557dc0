"""The two pipelines that turn per-pose entries into running-mean targets (include/icmslam.h,
icm_set_entry_path): hierarchical running sums (default; k_chunk_l1 .. k_rec_push, no sort)
and the sort-based one (k_compact, radix sort, k_lm_scan).  They add the same per-landmark
sums in a different order, so targets agree to ~1e-15 relative; the poses are compared at
1e-9 (a flipped simplex comparison would show up as a 1e-3 jump and is bounded like in
test_gpu_parity.py).
"""
import numpy as np
import pytest

from util import Cfg, dataset, gold

pytestmark = pytest.mark.gpu


def _run_real(mode, sweeps, schedule, debug=False):
    from icmslam_hip import SweepEngine
    zz, odo, u = dataset()
    init = gold("init_pass.npz")
    eng = SweepEngine(Cfg())
    eng.upload(zz, odo, u)
    eng.set_entry_path(mode)
    eng.set_debug(debug)
    x = init["x_init"].copy()
    mv, la = init["map_init"].copy(), int(init["landmarks_actuales"])
    paths, assoc = [], None
    for _ in range(sweeps):
        mo, co, K = eng.sweep(mv, x, odo[:, 0], la, schedule)
        paths.append(eng.entry_path())
        if debug:
            assoc = eng.association()
        mv, la = mo[:, :K].copy(), K
    eng.close()
    return x, mv, co, paths, assoc


@pytest.mark.parametrize("schedule", ["sequential", "redblack"])
def test_hierarchical_equals_sort_based_on_dataset(schedule):
    xs, ms, cs, ps, _ = _run_real("sort", 3, schedule)
    xh, mh, ch, ph, _ = _run_real("auto", 3, schedule)
    assert ps == ["sort"] * 3 and ph == ["hier"] * 3, "the default must be the hierarchical pipeline"
    assert ms.shape == mh.shape and np.array_equal(cs, ch)
    assert np.abs(ms - mh).max() <= 1e-12
    d = np.abs(xs - xh).max(axis=0)
    print("hier vs sort, %s: max|dx| %.3e, poses above 1e-9: %d" % (schedule, d.max(), int((d > 1e-9).sum())))
    assert d.max() <= 1e-9


def test_hierarchical_reproduces_reference_goldens():
    """Sweeps 1 and 2 of the reference (golden vectors) through the hierarchical pipeline."""
    x, mv, co, paths, _ = _run_real("hier", 2, "sequential")
    assert paths == ["hier", "hier"]
    g = gold("sweep02.npz")
    assert mv.shape[1] == int(g["landmarks_actuales"])
    assert np.abs(mv - g["mapa"]).max() <= 1e-9
    d = np.abs(x - g["x"]).max(axis=0)
    print("hier vs reference after sweep 2: max|dx| %.3e, poses above 1e-9: %d" % (d.max(), int((d > 1e-9).sum())))
    assert d.max() <= 1e-9


def test_association_dump_on_both_pipelines():
    """icm_set_debug + a forced hierarchical pipeline: labels and per-beam targets y[:, c]
    (scripts/ICM_ROS.py:150-152) equal to the sort-based pipeline's."""
    _, _, _, ps, a_s = _run_real("sort", 1, "sequential", debug=True)
    _, _, _, ph, a_h = _run_real("hier", 1, "sequential", debug=True)
    assert ps == ["sort"] and ph == ["hier"]
    assert np.array_equal(a_s[0], a_h[0])
    assert max(np.abs(a_s[1] - a_h[1]).max(), np.abs(a_s[2] - a_h[2]).max()) <= 1e-12
    # debug without forcing: the sort-based pipeline serves the dump
    _, _, _, pa, _ = _run_real("auto", 1, "sequential", debug=True)
    assert pa == ["sort"]


def test_hierarchical_on_synthetic_field_with_new_landmarks_and_turns():
    """Synthetic field (lanes + a turn where scans are empty, landmarks created during the
    sweep), several sweeps with the state resident on the device."""
    from ICM_SLAM_tools import ConfigICM
    from icmslam_hip import SweepEngine
    from icmslam_hip.synthetic import make_workload
    wl = make_workload(1900, 100, 180)
    cfg = ConfigICM(D=wl.config)
    out = {}
    for mode in ("sort", "hier"):
        eng = SweepEngine(cfg)
        eng.upload(wl.scans, wl.odometry, wl.u, pose_major=True)
        eng.set_entry_path(mode)
        eng.set_state(wl.map_init, wl.x_init, wl.x0)
        for _ in range(4):
            eng.sweep_device("redblack")
        assert eng.entry_path() == mode
        out[mode] = eng.get_state()
        eng.close()
    (xs, ms, cs, Ks), (xh, mh, ch, Kh) = out["sort"], out["hier"]
    assert Ks == Kh and np.array_equal(cs, ch)
    assert np.abs(ms - mh).max() <= 1e-11
    d = np.abs(xs - xh).max(axis=0)
    assert d.max() <= 1e-9


def test_dense_map_falls_back_to_sort_based_pipeline():
    """More than 64 distinct landmarks in one scan exceed the chunk tables of the hierarchical
    pipeline: the sweep must notice, run the sort-based pipeline instead and still match the
    CPU oracle."""
    from ICM_SLAM_tools import ConfigICM
    from icmslam_hip import SweepEngine
    from oracle import c_oracle as co
    from test_gpu_edge import _dense_ring_case
    lm, scans, x_true, u, cfgd = _dense_ring_case()
    cfg = ConfigICM(D=cfgd)
    odo = x_true.copy()
    eng = SweepEngine(cfg)
    eng.upload(scans, odo, u)
    x = x_true.copy()
    mo, cnt, K = eng.sweep(lm, x, x_true[:, 0], lm.shape[1], "redblack")
    assert eng.entry_path() == "sort"
    eng.close()
    keptc = co.prefilter(cfg, scans)
    xc = x_true.copy()
    mc, cntc, Kc, _ = co.sweep(cfg, keptc, u, odo, x_true[:, 0], lm, xc, lm.shape[1], "redblack")
    assert K == Kc and np.array_equal(cnt, cntc) and np.abs(mo[:, :K] - mc).max() <= 1e-9
    assert np.abs(x - xc).max() <= 1e-9


def test_many_landmarks_per_chunk_still_hierarchical_or_falls_back_consistently():
    """A field dense enough that 64 consecutive poses see close to the chunk table's capacity:
    whichever pipeline the sweep ends up on, the result equals the sort-based one."""
    from ICM_SLAM_tools import ConfigICM
    from icmslam_hip import SweepEngine
    from icmslam_hip.synthetic import make_workload
    wl = make_workload(1350, 400, 360)   # one lane through a 50 m field, ends inside it
    d = dict(wl.config)
    cfg = ConfigICM(D=d)
    out = {}
    for mode in ("sort", "auto"):
        eng = SweepEngine(cfg)
        eng.upload(wl.scans, wl.odometry, wl.u, pose_major=True)
        eng.set_entry_path(mode)
        eng.set_state(wl.map_init, wl.x_init, wl.x0)
        for _ in range(2):
            eng.sweep_device("redblack")
        out[mode] = (eng.get_state(), eng.entry_path())
        eng.close()
    (xs, ms, cs, Ks), _ = out["sort"]
    (xh, mh, ch, Kh), path = out["auto"]
    print("pipeline used:", path)
    assert Ks == Kh and np.array_equal(cs, ch) and np.abs(ms - mh).max() <= 1e-11
    dd = np.abs(xs - xh).max(axis=0)
    assert dd.max() <= 1e-9


@pytest.mark.parametrize("mode", ["sort", "hier"])
def test_new_landmarks_beyond_capacity_raise_index_error(mode):
    """Mapa.actualizar indexes cant_obs_i[L] when a sweep creates more landmarks than the map has
    room for (IndexError, scripts/ICM_SLAM_tools.py:191); both pipelines must refuse the same way
    and leave the handle usable."""
    from icmslam_hip import SweepEngine
    zz, odo, u = dataset()
    init = gold("init_pass.npz")
    la = int(init["landmarks_actuales"])
    eng = SweepEngine(Cfg(L=la + 5))          # sweep 1 of the dataset creates 67 landmarks
    eng.upload(zz, odo, u)
    eng.set_entry_path(mode)
    x = init["x_init"].copy()
    with pytest.raises(IndexError):
        eng.sweep(init["map_init"].copy(), x, odo[:, 0], la, "sequential")
    eng.close()
    # the same data with room to spare goes through
    eng = SweepEngine(Cfg(L=la + 100))
    eng.upload(zz, odo, u)
    eng.set_entry_path(mode)
    x = init["x_init"].copy()
    mo, co, K = eng.sweep(init["map_init"].copy(), x, odo[:, 0], la, "sequential")
    assert K == 11 and eng.entry_path() == mode
    eng.close()


def test_sweep_queued_without_a_host_look_refuses_and_recovers_like_the_careful_one():
    """The red-black sweep is queued whole (no host look at phase A's counts and flags in the middle; solves and
    Mapa.filtrar check the flags on the device).  When the sweep would create more landmarks than the map holds, the
    IndexError comes out at the end instead of the middle -- with the poses untouched and the handle usable: the same
    state then sweeps to the same result as a fresh handle."""
    from icmslam_hip import SweepEngine
    zz, odo, u = dataset()
    init = gold("init_pass.npz")
    la = int(init["landmarks_actuales"])
    eng = SweepEngine(Cfg(L=la + 5))          # sweep 1 of the dataset creates 67 landmarks
    eng.upload(zz, odo, u)
    eng.set_state(init["map_init"].copy(), init["x_init"].copy(), odo[:, 0], la)
    with pytest.raises(IndexError):
        eng.sweep_device("redblack")
    x_after, m_after, c_after, K_after = eng.get_state()
    assert np.array_equal(x_after, init["x_init"]) and K_after == la     # nothing was replaced
    eng.close()
    ref = SweepEngine(Cfg(L=la + 100))
    ref.upload(zz, odo, u)
    ref.set_state(init["map_init"].copy(), init["x_init"].copy(), odo[:, 0], la)
    ref.sweep_device("redblack")
    xr, mr, cr, Kr = ref.get_state()
    ref.set_colour_fusion(False)              # the careful path (host look in the middle) on the same state
    ref.set_state(init["map_init"].copy(), init["x_init"].copy(), odo[:, 0], la)
    ref.sweep_device("redblack")
    xc, mc, cc, Kc = ref.get_state()
    ref.close()
    assert Kr == Kc and np.array_equal(xr, xc) and np.array_equal(mr, mc) and np.array_equal(cr, cc)


def test_sequence_shorter_than_one_chunk():
    """Fewer poses than one 64-pose chunk (and not a multiple of the group size)."""
    from icmslam_hip import SweepEngine
    zz, odo, u = dataset()
    init = gold("init_pass.npz")
    T = 43
    out = {}
    for mode in ("sort", "hier"):
        eng = SweepEngine(Cfg(cota=5.0))
        eng.upload(zz[:, :T], odo[:, :T], u[:, :T])
        eng.set_entry_path(mode)
        x = np.ascontiguousarray(init["x_init"][:, :T]).copy()
        mo, co, K = eng.sweep(init["map_init"].copy(), x, odo[:, 0], int(init["landmarks_actuales"]), "redblack")
        assert eng.entry_path() == mode
        out[mode] = (x, mo[:, :K], co)
        eng.close()
    assert np.array_equal(out["sort"][2], out["hier"][2])
    assert np.abs(out["sort"][1] - out["hier"][1]).max() <= 1e-12
    assert np.abs(out["sort"][0] - out["hier"][0]).max() <= 1e-9
