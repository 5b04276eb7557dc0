"""GPU parity: the HIP sweep (through the C-ABI) against the golden vectors produced by the
real reference and against the CPU oracle, on data_IJAC2018.

Tolerances (BASELINE.json north_star: landmarks within 1e-4 m of the reference):
  * integer work (kept-beam indices, labels, counts, iteration counts): bit-exact;
  * filtrar_z rows: bit-exact (products of host-made cos/sin tables);
  * world points / running means / maps: 1e-9 m (device sin/cos differ from libm in the
    last ulp; the running mean is evaluated as sum/n instead of the reference's recurrence);
  * poses: 1e-9 m/rad on EVERY pose -- what is observed (bit-identical after sweeps 1 and 2,
    6.5e-14 after 30).  Nelder-Mead amplifies an energy difference only when it flips a simplex
    comparison (such a flip would move that pose by up to xtol = 1e-3); no flip occurs on any
    workload of this suite, and a regression that causes one fails here.
"""
import numpy as np
import pytest

from util import Cfg, dataset, gold

pytestmark = pytest.mark.gpu

POSE_TIGHT = 1e-9
MAP_TOL = 1e-9


@pytest.fixture(scope="module")
def engine():
    from icmslam_hip import SweepEngine
    zz, odo, u = dataset()
    eng = SweepEngine(Cfg())
    eng.upload(zz, odo, u)
    yield eng
    eng.close()


@pytest.fixture(scope="module")
def init_state():
    g = gold("init_pass.npz")
    return g["x_init"].copy(), g["map_init"].copy(), int(g["landmarks_actuales"])


def test_prefilter_matches_reference_filtrar_z(engine):
    g = gold("filtrar_z.npz")
    off, bk, d, bx, by = engine.kept_beams()
    assert np.array_equal(off, g["offsets"])
    rows = g["rows"]
    assert np.array_equal(d, rows[:, 0])
    assert np.array_equal(bk * np.pi / 180.0, rows[:, 1])
    assert np.array_equal(bx, rows[:, 2])
    assert np.array_equal(by, rows[:, 3])


def test_energy_and_single_solve_t100(engine, init_state):
    """SURVEY Appendix C unit pin: fun_xn at the start point and fmin's result for t=100."""
    s = gold("solve_t100.npz")
    x_init, _, _ = init_state
    _, odo, u = dataset()
    t = 100
    beams = s["beams"]
    args = (x_init[:, t - 1], x_init[:, t + 1], u[:, t - 1:t + 1], odo[:, t - 1:t + 2],
            beams[:, 2], beams[:, 3], s["targets"][:, 0], s["targets"][:, 1])
    f0 = engine.energy_one(True, s["start"], *args)
    assert abs(f0 - float(s["f_start"])) <= 1e-13
    out = engine.solve_one(True, *args)
    assert np.abs(out[:3] - s["xopt"]).max() <= POSE_TIGHT
    assert abs(out[3] - float(s["fopt"])) <= 1e-12
    assert int(out[4]) == int(s["nit"]) and int(out[5]) == int(s["nfev"])


def _pose_report(x, xref, what):
    d = np.abs(x - xref).max(axis=0)
    loose = int((d > POSE_TIGHT).sum())
    print("%s: max|dx| %.3e, poses above %.0e: %d of %d" % (what, d.max(), POSE_TIGHT, loose, d.size))
    return d, loose


@pytest.mark.parametrize("form", ["moments", "beam", "entry"])
def test_sweep1_sequential_matches_reference(engine, init_state, form):
    x_init, map_init, lact = init_state
    _, odo, _ = dataset()
    g = gold("sweep01.npz")
    x = x_init.copy()
    engine.set_debug(True)
    engine.set_energy_form(form)
    try:
        mo, co, K = engine.sweep(map_init, x, odo[:, 0], lact, "sequential")
    finally:
        engine.set_debug(False)
        engine.set_energy_form("moments")
    # phase A: labels of every kept beam and the running-mean targets of every solved pose
    pp = gold("sweep1_perpose.npz")
    lab, tx, ty = engine.association()
    off = engine.kept_beams()[0]
    goff = pp["offsets"]
    for i, t in enumerate(pp["t"]):
        sl = slice(off[t], off[t + 1])
        assert np.array_equal(lab[sl], pp["labels"][goff[i]:goff[i + 1]]), "labels of pose %d" % t
        tg = pp["targets"][goff[i]:goff[i + 1]]
        assert np.abs(tx[sl] - tg[:, 0]).max() <= MAP_TOL and np.abs(ty[sl] - tg[:, 1]).max() <= MAP_TOL
    # raw running map before Mapa.filtrar
    yr, cr, la = engine.raw_map()
    assert la == int(pp["filtrar_lact_in"])
    assert np.array_equal(cr[:la], pp["filtrar_cnt_in"])
    assert np.abs(yr[:, :la] - pp["filtrar_y_in"]).max() <= MAP_TOL
    # sweep outputs
    assert K == int(g["landmarks_actuales"])
    assert np.array_equal(co, g["cant_obs_i"])
    assert np.abs(mo[:, :K] - g["mapa"]).max() <= MAP_TOL
    d, loose = _pose_report(x, g["x"], "sweep 1 sequential vs reference")
    assert d.max() <= POSE_TIGHT and loose == 0


def test_two_sweeps_sequential(engine, init_state):
    x_init, map_init, lact = init_state
    _, odo, _ = dataset()
    x = x_init.copy()
    mv, la = map_init, lact
    for it in (1, 2):
        mo, co, K = engine.sweep(mv, x, odo[:, 0], la, "sequential")
        mv, la = mo[:, :K].copy(), K
    g = gold("sweep02.npz")
    assert K == int(g["landmarks_actuales"])
    assert np.abs(mv - g["mapa"]).max() <= MAP_TOL
    d, loose = _pose_report(x, g["x"], "sweep 2 sequential vs reference")
    assert d.max() <= POSE_TIGHT


def test_thirty_sweeps_device_resident(engine, init_state):
    """The reference driver loop (scripts/ICM_ROS.py:298-311) for N=30, state kept in HBM."""
    x_init, map_init, lact = init_state
    _, odo, _ = dataset()
    engine.set_state(map_init, x_init, odo[:, 0], lact)
    for _ in range(30):
        engine.sweep_device("sequential")
    x, mo, co, K = engine.get_state()
    g = gold("sweep30.npz")
    assert K == int(g["landmarks_actuales"])
    dm = np.abs(mo[:, :K] - g["mapa"]).max()
    d, loose = _pose_report(x, g["x"], "sweep 30 sequential vs reference")
    print("sweep 30 map max diff %.3e" % dm)
    assert dm <= MAP_TOL       # (north_star asks for 1e-4 m; observed 9e-15)
    assert d.max() <= POSE_TIGHT   # observed 6.5e-14


def test_brute_force_association_equals_grid(engine, init_state):
    x_init, map_init, lact = init_state
    _, odo, _ = dataset()
    x = x_init.copy()
    engine.set_debug(True)
    try:
        engine.sweep(map_init, x, odo[:, 0], lact, "sequential")
        lab_grid = engine.association()[0].copy()
        engine.set_brute_force(True)
        x2 = x_init.copy()
        engine.sweep(map_init, x2, odo[:, 0], lact, "sequential")
        lab_brute = engine.association()[0].copy()
    finally:
        engine.set_brute_force(False)
        engine.set_debug(False)
    assert np.array_equal(lab_grid, lab_brute)
    assert np.array_equal(x, x2)


def test_redblack_matches_oracle_redblack(engine, init_state):
    """The throughput schedule against the CPU oracle run in the same order."""
    from oracle import icm_oracle as o
    x_init, map_init, lact = init_state
    zz, odo, u = dataset()
    T = 400  # oracle cost: ~3 s
    cfg = Cfg(cota=20.0)  # 400 poses: keep the landmarks seen at least 20 times
    from icmslam_hip import SweepEngine
    eng = SweepEngine(cfg)
    eng.upload(zz[:, :T], odo[:, :T], u[:, :T])
    x = np.ascontiguousarray(x_init[:, :T])
    mo, co, K = eng.sweep(map_init, x, odo[:, 0], lact, "redblack")
    eng.close()
    ocfg = o.OracleConfig.from_config(cfg)
    st = o.MapState(ocfg, lact)
    xo = np.ascontiguousarray(x_init[:, :T])
    mref, xo = o.sweep(ocfg, st, zz[:, :T], u[:, :T], odo[:, :T], odo[:, 0], map_init.copy(), xo, schedule="redblack")
    d, loose = _pose_report(x, xo, "red-black vs oracle red-black (T=400)")
    assert d.max() <= POSE_TIGHT and loose == 0
    assert K == mref.shape[1]
    assert np.abs(mo[:, :K] - mref).max() <= MAP_TOL


def test_gpu_filtrar_equals_host_filtrar(engine, init_state):
    """Mapa.filtrar as the GPU kernel chain (k_fl_*) vs the host routine, on the real
    dataset and on a synthetic map with no merges: identical maps, counters and poses."""
    from ICM_SLAM_tools import ConfigICM
    from icmslam_hip import SweepEngine
    from icmslam_hip.synthetic import make_workload
    x_init, map_init, lact = init_state
    _, odo, _ = dataset()
    res = []
    for gpu in (True, False):
        engine.set_gpu_filtrar(gpu)
        engine.set_state(map_init, x_init, odo[:, 0], lact)
        for _ in range(3):
            engine.sweep_device("redblack")
        res.append(engine.get_state())
    engine.set_gpu_filtrar(True)
    for a, b in zip(res[0], res[1]):
        assert np.array_equal(a, b)
    wl = make_workload(1900, 100, 180)
    eng = SweepEngine(ConfigICM(D=wl.config))
    eng.upload(wl.scans, wl.odometry, wl.u, pose_major=True)
    res = []
    for gpu in (True, False):
        eng.set_gpu_filtrar(gpu)
        eng.set_state(wl.map_init, wl.x_init, wl.x0)
        for _ in range(3):
            eng.sweep_device("redblack")
        res.append(eng.get_state())
    eng.close()
    for a, b in zip(res[0], res[1]):
        assert np.array_equal(a, b)
    assert res[0][3] > 50


def _merge_case(seed, n=60, L=200):
    rng = np.random.default_rng(seed)
    pts = rng.uniform(-8, 8, (2, n))
    pts[:, 10:20] = pts[:, 0:10] + rng.normal(0, 0.3, (2, 10))   # pairs within the gate
    pts[:, 20:25] = pts[:, 0:5] + rng.normal(0, 0.3, (2, 5))     # triples
    cnt = rng.integers(1, 40, n).astype(float)
    y = np.zeros((2, L)); c = np.zeros(L)
    y[:, :n] = pts; c[:n] = cnt
    return y, c, n


@pytest.mark.parametrize("seed", range(8))
def test_device_filtrar_merges_like_the_oracle(seed):
    """Mapa.filtrar entirely on the GPU (reference scripts/ICM_SLAM_tools.py:204-265): random maps
    with pairs, triples and chains of landmarks closer than dist_thr and rarely-seen landmarks --
    prune, sequential label propagation, gap-closing renumbering and count-weighted means come out
    like the oracle's, without the host routine (path 1); coincident landmarks (odd seeds add a
    pair) are the documented hand-over to it (path 2)."""
    from icmslam_hip import SweepEngine
    from oracle import icm_oracle as o
    cfg = Cfg(L=200, cota=5.0)
    y, c, n = _merge_case(seed)
    if seed % 2:
        y[:, 30] = y[:, 31]
        c[30] = c[31] = 9.0
    eng = SweepEngine(cfg)
    yo, co, lo, path = eng.filtrar_device(y, c, n)
    eng.close()
    st = o.MapState(o.OracleConfig.from_config(cfg), n)
    st.cant_obs_i = c.copy()
    yr = o.filtrar(st, y.copy())
    assert path == (2 if seed % 2 else 1)
    assert lo == st.landmarks_actuales
    assert np.abs(yo - yr).max() <= 1e-12 and np.array_equal(co, st.cant_obs_i)


def test_device_filtrar_chains_large_components_and_edges():
    from icmslam_hip import SweepEngine
    from oracle import icm_oracle as o
    cfg = Cfg(L=64, cota=2.0)
    eng = SweepEngine(cfg)

    def both(y, c, n):
        st = o.MapState(o.OracleConfig.from_config(cfg), n)
        st.cant_obs_i = c.copy()
        yr = o.filtrar(st, y[:, :n].copy() if (c[:n] >= cfg.cota).all() else y.copy())
        yo, co, lo, path = eng.filtrar_device(y, c, n)
        assert lo == st.landmarks_actuales and np.abs(yo - yr[:, :cfg.L] if yr.shape[1] >= cfg.L else np.abs(yo[:, :yr.shape[1]] - yr)).max() <= 1e-12
        assert np.array_equal(co, st.cant_obs_i)
        return path

    # a chain of 6 landmarks 0.6 m apart (one component of 6: on the device), far singles around it
    y = np.zeros((2, 64)); c = np.zeros(64)
    y[0, :6] = 0.6 * np.arange(6); c[:6] = (3, 9, 4, 7, 5, 8)
    y[:, 6:10] = ((20, 30, 40, 50), (5, 5, 5, 5)); c[6:10] = (2, 1, 6, 6)      # index 7 is pruned
    assert both(y, c, 10) == 1
    # nine landmarks with shrinking gaps (each one's nearest neighbour is the next: ONE component of
    # 9 > 7 members; numpy sums 8+ terms pairwise) -> host routine
    y2 = np.zeros((2, 64)); c2 = np.zeros(64)
    y2[0, :9] = np.concatenate(([0.0], np.cumsum(0.9 - 0.05 * np.arange(8)))); c2[:9] = np.arange(3, 12)
    y2[:, 9] = (30, 30); c2[9] = 4
    assert both(y2, c2, 10) == 2
    # nothing to merge, nothing pruned
    y3 = np.zeros((2, 64)); c3 = np.zeros(64)
    y3[0, :5] = 3.0 * np.arange(5); c3[:5] = 4
    assert both(y3, c3, 5) == 0
    # nothing reaches cota: ValueError like the reference
    with pytest.raises(ValueError):
        eng.filtrar_device(y3, np.zeros(64), 5)
    eng.close()
