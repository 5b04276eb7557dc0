"""The CPU oracle against the golden vectors produced by the REAL reference
(tests/golden/make_golden.py): bit-for-bit, because the oracle restates the reference with
the same NumPy primitives.  This is what pins the oracle."""
import numpy as np
import pytest

from oracle import icm_oracle as o
from util import dataset, gold


@pytest.fixture(scope="module")
def cfg():
    return o.OracleConfig()


@pytest.fixture(scope="module")
def kept(cfg):
    zz, _, _ = dataset()
    return o.prefilter_all(zz, cfg)


def test_filtrar_z_bitwise(kept):
    g = gold("filtrar_z.npz")
    off, rows = g["offsets"], g["rows"]
    assert len(kept) == len(off) - 1
    for t, k in enumerate(kept):
        ref = rows[off[t]:off[t + 1]]
        got = k if k.ndim == 2 else np.zeros((0, 4))
        assert got.shape == ref.shape and np.array_equal(got, ref), "scan %d" % t


def test_known_answers_appendix_c(kept):
    """scan 0 keeps beams 61..65,146..148."""
    k0 = kept[0]
    assert np.array_equal(np.round(k0[:, 1] * 180 / np.pi).astype(int), [61, 62, 63, 64, 65, 146, 147, 148])
    assert np.allclose(k0[0], [5.1090001221, 1.0646508437, 2.476892409, 4.4684321905], atol=1e-9)


def test_energy_and_nelder_mead_t100(cfg):
    s = gold("solve_t100.npz")
    init = gold("init_pass.npz")
    _, odo, u = dataset()
    x_init = init["x_init"]
    t = 100
    f = lambda v: o.fun_xn(cfg, v, x_init[:, t - 1].reshape(3, 1), x_init[:, t + 1].reshape(3, 1), u, odo, t,  # noqa: E731
                           s["beams"][:, 0:2], s["targets"])
    assert o.h(cfg, s["start"], s["beams"][:, 0:2], s["targets"]) == float(s["h_start"]) == 0.169551072251446
    assert f(s["start"]) == float(s["f_start"])
    xo, fo, nit, nfev = o.nelder_mead(f, s["start"], full_output=True)
    assert np.array_equal(xo, s["xopt"]) and fo == float(s["fopt"])
    assert (nit, nfev) == (int(s["nit"]), int(s["nfev"])) == (26, 52)


def test_filtrar_pairs_from_sweep1(cfg):
    pp = gold("sweep1_perpose.npz")
    st = o.MapState(cfg, int(pp["filtrar_lact_in"]))
    st.cant_obs_i[:st.landmarks_actuales] = pp["filtrar_cnt_in"]
    y = np.zeros((2, cfg.L))
    y[:, :st.landmarks_actuales] = pp["filtrar_y_in"]
    out = o.filtrar(st, y)
    la = int(pp["filtrar_lact_out"])
    assert st.landmarks_actuales == la
    assert np.array_equal(out[:, :la], pp["filtrar_y_out"])
    assert np.array_equal(st.cant_obs_i[:la], pp["filtrar_cnt_out"])


@pytest.mark.slow
def test_two_sweeps_bitwise(cfg, kept):
    """Two full sweeps of the three-phase oracle == the reference's interleaved loop, bit for
    bit (poses, map, counters) -- about 30 s."""
    zz, odo, u = dataset()
    init = gold("init_pass.npz")
    st = o.MapState(cfg, int(init["landmarks_actuales"]))
    x = init["x_init"].copy()
    mv = init["map_init"].copy()
    for it in (1, 2):
        mv, x = o.sweep(cfg, st, zz, u, odo, odo[:, 0], mv, x, kept=kept)
        g = gold("sweep%02d.npz" % it)
        assert np.array_equal(x, g["x"]) and np.array_equal(mv, g["mapa"])
        assert np.array_equal(st.cant_obs_i, g["cant_obs_i"]) and st.landmarks_actuales == int(g["landmarks_actuales"])


def test_prefix_sweep_three_phase_equals_interleaved(cfg, kept):
    """SURVEY Appendix A.6 on a 150-pose prefix: phases A+B first, then the solves, is the
    same computation as the reference's interleaved loop."""
    zz, odo, u = dataset()
    init = gold("init_pass.npz")
    T = 150
    c2 = o.OracleConfig(cota=20.0)
    res = []
    for fn in (o.sweep, o.sweep_interleaved):
        st = o.MapState(c2, int(init["landmarks_actuales"]))
        x = np.ascontiguousarray(init["x_init"][:, :T]).copy()
        m, x = fn(c2, st, zz[:, :T], u[:, :T], odo[:, :T], odo[:, 0], init["map_init"].copy(), x, kept=kept[:T])
        res.append((m, x, st.cant_obs_i.copy()))
    assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])
    assert np.array_equal(res[0][2], res[1][2])


def test_redblack_differs_but_stays_close(cfg, kept):
    zz, odo, u = dataset()
    init = gold("init_pass.npz")
    T = 120
    c2 = o.OracleConfig(cota=20.0)
    out = {}
    for sch in ("sequential", "redblack"):
        st = o.MapState(c2, int(init["landmarks_actuales"]))
        x = np.ascontiguousarray(init["x_init"][:, :T]).copy()
        m, x = o.sweep(c2, st, zz[:, :T], u[:, :T], odo[:, :T], odo[:, 0], init["map_init"].copy(), x, schedule=sch, kept=kept[:T])
        out[sch] = (m, x)
    # the map of a sweep does not depend on the solve order
    assert np.array_equal(out["sequential"][0], out["redblack"][0])
    d = np.abs(out["sequential"][1] - out["redblack"][1]).max()
    assert 0 < d < 0.1


def test_init_pass_bitwise(cfg, kept):
    """The causal initialisation pass (reference scripts/ICM_ROS.py:57-119 incl. the first-scan
    clustering branch of Mapa.actualizar) against the reference's own result."""
    zz, odo, u = dataset()
    g = gold("init_pass.npz")
    x, m, st, c0, raw = o.init_pass(cfg, zz, u, odo, kept=kept)
    assert np.array_equal(c0, g["labels_scan0"])
    assert np.array_equal(x, g["x_init"]) and np.array_equal(m, g["map_init"])
    assert np.array_equal(st.cant_obs_i, g["cant_obs_i"]) and raw[2] == int(g["landmarks_raw"])
    assert np.array_equal(raw[0], g["y_raw"]) and np.array_equal(raw[1], g["cant_obs_raw"])


def test_clustering_restatement_equals_scipy():
    from scipy.cluster.hierarchy import fcluster, inconsistent, linkage
    from scipy.spatial.distance import pdist
    rng = np.random.default_rng(11)
    for _ in range(150):
        pts = np.concatenate([rng.normal(c, 0.15, (rng.integers(1, 6), 2)) for c in rng.uniform(-6, 6, (rng.integers(1, 6), 2))])
        if pts.shape[0] < 2:
            continue
        Z, Zs = o.single_linkage(pts), linkage(pdist(pts))
        assert np.array_equal(Z, Zs)
        assert np.array_equal(o.inconsistency(Z, 2), inconsistent(Zs, 2)[:, 3])
        assert np.array_equal(o.fcluster_inconsistent(Z, 1.0), fcluster(Zs, 1.0))


def test_c_oracle_matches_reference_goldens():
    """oracle/icm_oracle_c.c (the compiled restatement) against the same goldens: pre-filter rows
    bit-exact, the Appendix C solve, poses/map after sweeps 1 and 2 within 1e-9."""
    from oracle import c_oracle as co
    from util import Cfg
    cfgc = Cfg()
    zz, odo, u = dataset()
    keptc = co.prefilter(cfgc, zz)
    fz = gold("filtrar_z.npz")
    assert np.array_equal(keptc[0], fz["offsets"])
    for col, arr in ((0, keptc[2]), (1, keptc[3]), (2, keptc[4]), (3, keptc[5])):
        assert np.array_equal(arr, fz["rows"][:, col])
    s, init = gold("solve_t100.npz"), gold("init_pass.npz")
    xi = init["x_init"]
    out = co.solve_one(cfgc, 1, xi[:, 99], xi[:, 101], u[:, 99:101], odo[:, 99:102], s["beams"][:, 0:2], s["targets"])
    assert np.abs(out[:3] - s["xopt"]).max() <= 1e-12 and (out[4], out[5]) == (26, 52)
    x, mv, la = xi.copy(), init["map_init"].copy(), int(init["landmarks_actuales"])
    for it in (1, 2):
        mv, c, la, raw = co.sweep(cfgc, keptc, u, odo, odo[:, 0], mv, x, la, "sequential")
        g = gold("sweep%02d.npz" % it)
        assert la == int(g["landmarks_actuales"]) and np.array_equal(c, g["cant_obs_i"])
        assert np.abs(x - g["x"]).max() <= 1e-9 and np.abs(mv - g["mapa"]).max() <= 1e-9


def test_c_oracle_init_pass_matches_the_reference(cfg, kept):
    """The C restatement of the causal initialisation pass (scripts/ICM_ROS.py:102-119) from the map the first scan's
    clustering seeds, against the reference's own x_init / raw map / counters (init_pass.npz)."""
    from oracle import c_oracle as co
    from util import Cfg
    zz, odo, u = dataset()
    g = gold("init_pass.npz")
    cc = Cfg()
    st = o.MapState(cfg)
    y0 = np.zeros((2, cfg.L))
    y0, c0 = o.cluster_first_scan(st, y0, o.project_beams(odo[:, 0].copy(), kept[0][:, 2:4]))
    assert np.array_equal(c0, g["labels_scan0"])
    x, y, cnt, lact = co.init_pass(cc, co.prefilter(cc, zz), u, odo, y0, st.cant_obs_i, st.landmarks_actuales)
    assert lact == int(g["landmarks_raw"]) and np.array_equal(cnt, g["cant_obs_raw"])
    assert np.abs(y - g["y_raw"]).max() <= 1e-12
    d = np.abs(x - g["x_init"]).max()
    print("C init pass vs the reference: max|dx| %.2e" % d)
    assert d <= 1e-9


def test_c_oracle_equals_numpy_oracle_redblack(cfg, kept):
    from oracle import c_oracle as co
    from util import Cfg
    zz, odo, u = dataset()
    init = gold("init_pass.npz")
    T = 200
    cc = Cfg(cota=20.0)
    keptc = co.prefilter(cc, zz[:, :T])
    xc = np.ascontiguousarray(init["x_init"][:, :T]).copy()
    mc, cntc, Kc, _ = co.sweep(cc, keptc, u[:, :T], odo[:, :T], odo[:, 0], init["map_init"], xc, 11, "redblack")
    c2 = o.OracleConfig(cota=20.0)
    st = o.MapState(c2, 11)
    xn = np.ascontiguousarray(init["x_init"][:, :T]).copy()
    mn, xn = o.sweep(c2, st, zz[:, :T], u[:, :T], odo[:, :T], odo[:, 0], init["map_init"].copy(), xn, schedule="redblack", kept=kept[:T])
    assert Kc == mn.shape[1] and np.abs(mc - mn).max() <= 1e-12 and np.abs(xc - xn).max() <= 1e-9


def test_c_oracle_grid_association_and_threads_do_not_change_anything():
    """The C oracle's two accelerations are exact: the grid-accelerated association returns the
    brute-force cdist/argmin answer (labels, targets, map, poses bit-equal), and the OpenMP thread
    count does not enter the result."""
    from ICM_SLAM_tools import ConfigICM
    from icmslam_hip.synthetic import make_workload
    from oracle import c_oracle as co
    wl = make_workload(1400, 400, 360)
    cfg_s = ConfigICM(D=wl.config)
    kept = co.prefilter(cfg_s, wl.scans.T)
    res = []
    for grid, threads in ((False, 1), (True, 1), (True, 0)):
        co.set_grid(grid)
        co.set_threads(threads)
        x = wl.x_init.copy()
        a = {}
        mv, cnt, K, raw = co.sweep(cfg_s, kept, wl.u, wl.odometry, wl.x0, wl.map_init, x, wl.K, "redblack", assoc=a)
        res.append((x, mv, cnt, a["labels"].copy(), a["targets"].copy(), raw[0], raw[1]))
    co.set_grid(True)
    co.set_threads(0)
    assert (res[0][3] >= 0).all() and len(set(res[0][3])) >= 80      # the case really associates against many landmarks
    for other in res[1:]:
        for u, v in zip(res[0], other):
            assert np.array_equal(u, v)
