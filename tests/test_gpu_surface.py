"""The reference's class surface on the GPU: a driver written like scripts/example.py /
scripts/ICM_ROS.py:298-311 runs unchanged against icm-slam_amd/ and reproduces the reference's
own results (BASELINE.json configs[0]/[1]: data_IJAC2018, default config, N = 2)."""
import os
from copy import deepcopy as copy

import numpy as np
import pytest

from util import GOLD, gold

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def icm():
    from ICM_ROS import ICM_ROS
    from ICM_SLAM_tools import ConfigICM
    config = ConfigICM("config_default.yaml")          # unloadable by the reference itself
    m = ICM_ROS(config)
    m.load_data(os.path.join(GOLD, "data_IJAC2018.npz"))
    init = gold("init_pass.npz")
    m.set_initial_state(init["x_init"], init["map_init"], init["cant_obs_i"])
    return m


def test_driver_loop_reproduces_reference(icm):
    mapa_viejo = copy(icm.mapa_viejo)
    x = copy(icm.positions)
    assert icm.config.N == 2
    for it in range(icm.config.N):
        x_id = id(x)
        mapa_refinado, x = icm.iterations_process_offline(mapa_viejo, x)
        assert id(x) == x_id                      # updated in place AND returned
        g = gold("sweep%02d.npz" % (it + 1))
        assert mapa_refinado.shape == g["mapa"].shape
        assert np.abs(mapa_refinado - g["mapa"]).max() <= 1e-9
        assert np.abs(x - g["x"]).max() <= 1e-9
        assert np.array_equal(icm.mapa_obj.cant_obs_i, g["cant_obs_i"])
        assert icm.mapa_obj.landmarks_actuales == int(g["landmarks_actuales"])
        mapa_viejo = copy(mapa_refinado)
    assert np.array_equal(icm.mapa_viejo, gold("init_pass.npz")["map_init"])   # input map untouched


def test_non_contiguous_x_is_updated_in_place(icm):
    init = gold("init_pass.npz")
    big = np.zeros((3, 2 * init["x_init"].shape[1]))
    xv = big[:, ::2]
    xv[...] = init["x_init"]
    icm.mapa_obj.landmarks_actuales = init["map_init"].shape[1]
    m, xr = icm.iterations_process_offline(init["map_init"].copy(), xv)
    assert xr is xv and np.abs(xv - gold("sweep01.npz")["x"]).max() <= 1e-9


def test_model_functions_on_gpu(icm):
    """h / fun_xn / minimizar_xn with the reference's stashed-state calling convention
    (scripts/ICM_ROS.py:209-252), against the SURVEY Appendix C known answers."""
    s = gold("solve_t100.npz")
    init = gold("init_pass.npz")
    x = init["x_init"].copy()
    xo = icm.minimizar_xn(s["beams"][:, 0:2], s["targets"], x, 100)
    assert np.abs(xo - s["xopt"]).max() <= 1e-9
    assert abs(icm.fun_xn(s["start"]) - 0.17067461579800194) <= 1e-13
    assert abs(icm.h(s["start"], s["beams"][:, 0:2]) - 0.169551072251446) <= 1e-13
    g = icm.g(x[:, 99], icm.u[:, 99])
    assert g.shape == (3, 1)


def test_filtrar_z_helper_matches_reference(icm):
    from ICM_SLAM_tools import filtrar_z, tras_rot_z
    fz = gold("filtrar_z.npz")
    for t in (0, 100, 1832):
        rows = filtrar_z(icm.mediciones[:, t], icm.config)
        ref = fz["rows"][fz["offsets"][t]:fz["offsets"][t + 1]]
        assert rows.shape == ref.shape and np.array_equal(rows, ref)
    assert filtrar_z(np.full(181, 10.0), icm.config).shape == (0,)   # nothing in range
    z = filtrar_z(icm.mediciones[:, 0], icm.config)
    w = tras_rot_z(np.array([1.0, 2.0, 0.3]), z.copy())
    assert np.allclose(np.hypot(w[:, 2] - 1.0, w[:, 3] - 2.0), z[:, 0])


def test_errors_match_reference_types(icm):
    from ICM_ROS import ICM_ROS
    from ICM_SLAM_tools import ConfigICM
    init = gold("init_pass.npz")
    # label capacity exceeded -> IndexError (scripts/ICM_SLAM_tools.py:191)
    cfg = ConfigICM(D=dict(N=1, deltat=0.1, L=12, Q=[1, 1], R=[1, 1, 1], cte_odom=1.0, cota=300.0, dist_thr=1.0,
                           dist_thr_obs=1.0, rango_laser_max=10.0, radio=0.137))
    m = ICM_ROS(cfg)
    m.load_data(os.path.join(GOLD, "data_IJAC2018.npz"))
    m.set_initial_state(init["x_init"], init["map_init"])
    with pytest.raises(IndexError):
        m.iterations_process_offline(init["map_init"].copy(), init["x_init"].copy())
    # nothing reaches cota -> ValueError
    cfg2 = ConfigICM(D=dict(N=1, deltat=0.1, L=1000, Q=[1, 1], R=[1, 1, 1], cte_odom=1.0, cota=1e9, dist_thr=1.0,
                            dist_thr_obs=1.0, rango_laser_max=10.0, radio=0.137))
    m2 = ICM_ROS(cfg2)
    m2.load_data(os.path.join(GOLD, "data_IJAC2018.npz"))
    m2.set_initial_state(init["x_init"], init["map_init"])
    with pytest.raises(ValueError):
        m2.iterations_process_offline(init["map_init"].copy(), init["x_init"].copy())


def test_init_pass_matches_reference():
    """BASELINE configs[0] end to end: raw data -> initialisation pass -> N sweeps, all through
    the reference's class surface, against the reference's own init state and sweeps."""
    from ICM_ROS import ICM_ROS
    from ICM_SLAM_tools import ConfigICM
    m = ICM_ROS(ConfigICM("config_default.yaml"))
    m.load_data(os.path.join(GOLD, "data_IJAC2018.npz"))
    m.inicializar_offline()
    g = gold("init_pass.npz")
    assert m.mapa_viejo.shape == g["map_init"].shape
    assert np.abs(m.mapa_viejo - g["map_init"]).max() <= 1e-9
    d = np.abs(m.positions - g["x_init"]).max(axis=0)
    print("init pass vs reference: max|dx| %.3e, poses above 1e-9: %d" % (d.max(), int((d > 1e-9).sum())))
    assert d.max() <= 1e-9
    assert np.array_equal(m.mapa_obj.cant_obs_i, g["cant_obs_i"])
    mapa_viejo, x = copy(m.mapa_viejo), copy(m.positions)
    for it in range(m.config.N):
        mapa_refinado, x = m.iterations_process_offline(mapa_viejo, x)
        mapa_viejo = copy(mapa_refinado)
    g2 = gold("sweep02.npz")
    assert np.abs(mapa_viejo - g2["mapa"]).max() <= 1e-9
    assert np.abs(x - g2["x"]).max() <= 1e-9


def test_cluster_first_scan_matches_scipy():
    from scipy.cluster.hierarchy import fcluster, linkage
    from scipy.spatial.distance import pdist
    from icmslam_hip import cluster_first_scan
    rng = np.random.default_rng(5)
    for _ in range(50):
        pts = np.concatenate([rng.normal(c, 0.15, (rng.integers(1, 7), 2)) for c in rng.uniform(-6, 6, (rng.integers(1, 6), 2))])
        if pts.shape[0] < 2:
            continue
        assert np.array_equal(cluster_first_scan(pts, 1.0), fcluster(linkage(pdist(pts)), 1.0) - 1)


def test_message_stream_drives_the_same_pipeline():
    """Recorded sequence replayed as LaserScan / Odometry messages through the topic parsers
    (matlab2ros.replay -> Lidar / Odometria -> ICM_ROS.load_messages), then initialisation pass
    and one sweep: same result as handing the same 180-beam arrays over directly."""
    from ICM_ROS import ICM_ROS
    from ICM_SLAM_tools import ConfigICM
    from matlab2ros.replay import replay
    from sensors_definitions import Lidar, Odometria
    d = gold("data_IJAC2018.npz")
    T = 400
    cfg = ConfigICM("config_default.yaml")
    cfg.cota = 40.0
    lidar, odo = Lidar(config=cfg), Odometria(config=cfg)
    replay(d["observations"][:, :T], d["odometry"][:, :T], d["velocities"][:, :T], lidar.callback, odo.callback)
    a = ICM_ROS(cfg)
    a.load_messages(lidar, odo)
    a.inicializar_offline()
    ma, xa = a.iterations_process_offline(a.mapa_viejo.copy(), a.positions.copy())
    b = ICM_ROS(cfg)
    b.load_data(os.path.join(GOLD, "data_IJAC2018.npz"))
    b.mediciones = np.ascontiguousarray(b.mediciones[:180, :T])
    b.odometria = np.ascontiguousarray(b.odometria[:, :T])
    b.u = np.ascontiguousarray(b.u[:, :T])
    b.x0 = np.array([b.odometria[:, 0]]).T
    b.inicializar_offline()
    mb, xb = b.iterations_process_offline(b.mapa_viejo.copy(), b.positions.copy())
    assert ma.shape == mb.shape and ma.shape[1] > 0
    assert np.abs(ma - mb).max() <= 1e-9          # (yaw went through a quaternion: 1e-15 differences)
    dd = np.abs(xa - xb).max(axis=0)
    assert dd.max() <= 1e-9


def test_mapa_actualizar_replays_sweep_one_scan_by_scan():
    """`Mapa.actualizar` as a public method (reference scripts/ICM_SLAM_tools.py:128-201, called per
    sample by the online initialisation, scripts/ICM_ROS.py:114): replaying sweep 1 of the reference
    scan by scan -- filtrar_z rows, tras_rot_z with the previous-sweep pose, actualizar against
    mapa_viejo -- reproduces the reference's labels (exact) and gathered targets y[:, c] of every
    pose, and the raw map / counters it hands to Mapa.filtrar."""
    from ICM_SLAM_tools import ConfigICM, Mapa, tras_rot_z
    cfg = ConfigICM("config_default.yaml")
    init, pp, fz = gold("init_pass.npz"), gold("sweep1_perpose.npz"), gold("filtrar_z.npz")
    d = gold("data_IJAC2018.npz")
    x0 = d["odometry"][:, 0]
    m = Mapa(cfg)
    m.landmarks_actuales = int(init["landmarks_actuales"])
    m.clear_obs()
    y = np.zeros((2, cfg.L))
    rows, off = fz["rows"], fz["offsets"]
    where = {int(t): i for i, t in enumerate(pp["t"])}
    worst = 0.0
    for t in range(d["odometry"].shape[1]):
        z = rows[off[t]:off[t + 1]].copy()
        if z.shape[0] == 0:
            continue
        zt = tras_rot_z(x0 if t == 0 else init["x_init"][:, t], z)
        y_id = id(y)
        y, c = m.actualizar(y, init["map_init"], zt[:, 2:4])
        assert id(y) == y_id and c.shape == (z.shape[0],)
        if t in where:
            i = where[t]
            sl = slice(pp["offsets"][i], pp["offsets"][i + 1])
            assert np.array_equal(c, pp["labels"][sl]), "labels of pose %d" % t
            worst = max(worst, np.abs(y[:, c].T - pp["targets"][sl]).max())
    print("actualizar replay: max|dtarget| %.2e over %d scans" % (worst, len(where)))
    assert worst <= 1e-9
    la = int(pp["filtrar_lact_in"])
    assert m.landmarks_actuales == la
    assert np.array_equal(m.cant_obs_i[:la], pp["filtrar_cnt_in"])
    assert np.abs(y[:, :la] - pp["filtrar_y_in"]).max() <= 1e-9


def test_mapa_actualizar_first_scan_branch_and_errors():
    """landmarks_actuales == 0: the scan is clustered (single linkage at dist_thr) into the first
    landmarks -- the reference's labels for scan 0 of the dataset; then the error behaviour."""
    from ICM_SLAM_tools import ConfigICM, Mapa, tras_rot_z
    cfg = ConfigICM("config_default.yaml")
    init, fz, d = gold("init_pass.npz"), gold("filtrar_z.npz"), gold("data_IJAC2018.npz")
    z = fz["rows"][fz["offsets"][0]:fz["offsets"][1]].copy()
    zt = tras_rot_z(d["odometry"][:, 0], z)
    m = Mapa(cfg)
    y = np.zeros((2, cfg.L))
    y, c = m.actualizar(y, y, zt[:, 2:4])
    assert np.array_equal(c, init["labels_scan0"])
    ncl = int(c.max()) + 1
    assert m.landmarks_actuales == ncl
    for i in range(ncl):
        assert np.array_equal(y[:, i], np.mean(zt[c == i, 2:4], axis=0))
        assert m.cant_obs_i[i] == (c == i).sum()
    # a scan far from every landmark: ONE new landmark for all of its beams
    far = zt[:, 2:4] + 500.0
    y, c2 = m.actualizar(y, y.copy(), far)
    assert (c2 == ncl).all() and m.landmarks_actuales == ncl + 1 and m.cant_obs_i[ncl] == far.shape[0]
    assert np.abs(y[:, ncl] - far.sum(axis=0) / far.shape[0]).max() <= 1e-12
    # no room for a new landmark: IndexError like the reference (scripts/ICM_SLAM_tools.py:191)
    small = ConfigICM(D=dict(N=2, deltat=0.1, L=ncl + 1, Q=[1, 1], R=[1, 1, 1], cte_odom=1.0, cota=300, dist_thr=1.0,
                             dist_thr_obs=1, rango_laser_max=10.0, radio=0.137))
    m3 = Mapa(small)
    m3.landmarks_actuales = ncl + 1
    y3 = np.zeros((2, ncl + 1))
    with pytest.raises(IndexError):
        m3.actualizar(y3, y[:, :ncl + 1].copy(), far + 500.0)


def test_driver_loop_on_a_registered_pose_array_and_whatever_the_caller_does_to_it():
    """The drop-in call at resident speed (reference driver loop scripts/ICM_ROS.py:298-311, scripts/example.py:49-52): the
    pose array the caller hands back sweep after sweep is registered with the GPU runtime from its second sight on, the
    solves write the poses straight into it (no download) and a call that gets back the array the last call filled
    starts from the device's poses (no upload), checked against the array on the side.  Whatever the caller does in
    between -- edits the array in place, swaps in a copy, a reallocated or a non-contiguous array, goes back to the
    first one, hands in an edited map -- every call must return what a fresh engine returns for the same inputs."""
    from ICM_ROS import ICM_ROS
    from ICM_SLAM_tools import ConfigICM
    from icmslam_hip.synthetic import make_workload
    wl = make_workload(1900, 100, 180)
    cfg = ConfigICM(D=dict(wl.config, schedule="redblack"))

    def fresh_call(mapa, x):
        """the same call on a new object and new arrays: upload, sweep, download -- nothing carried over"""
        m = ICM_ROS(cfg)
        m.mediciones, m.odometria, m.u = wl.scans.T.copy(), wl.odometry.copy(), wl.u.copy()
        m.x0 = wl.x0.reshape(3, 1)
        m.set_initial_state(x, mapa)
        mr, xr = m.iterations_process_offline(np.array(mapa), np.array(x))
        out = mr, xr, m.mapa_obj.cant_obs_i.copy(), m.mapa_obj.landmarks_actuales
        m._engine.close()
        return out

    icm = ICM_ROS(cfg)
    icm.mediciones, icm.odometria, icm.u = wl.scans.T.copy(), wl.odometry.copy(), wl.u.copy()
    icm.x0 = wl.x0.reshape(3, 1)
    icm.set_initial_state(wl.x_init, wl.map_init)
    mapa, x = copy(icm.mapa_viejo), copy(icm.positions)
    first = x
    steps = ["same", "same", "same", "edit in place", "same", "copy", "same", "non-contiguous", "first again", "same",
             "edited map", "same", "same"]
    for k, what in enumerate(steps):
        if what == "edit in place":
            x[0, 700] += 0.05
            x[2, 1201] -= 0.01
        elif what == "copy":
            x = x.copy()
        elif what == "non-contiguous":
            big = np.zeros((3, 2 * x.shape[1]))
            big[:, ::2] = x
            x = big[:, ::2]
        elif what == "first again":
            first[...] = x
            x = first
        elif what == "edited map":
            mapa = mapa.copy()
            mapa[:, 3] += 0.02
        want = fresh_call(mapa, x)
        x_in = x
        mr, x = icm.iterations_process_offline(mapa, x)
        assert x is x_in, what
        assert np.array_equal(mr, want[0]) and np.array_equal(x, want[1]), (k, what)
        assert np.array_equal(icm.mapa_obj.cant_obs_i, want[2]) and icm.mapa_obj.landmarks_actuales == want[3], (k, what)
        mapa = copy(mr)
    n_fast, n_stale, n_mirror = icm._engine.dropin_counts()
    print("drop-in calls: %d of %d without an upload (%d of them started over: the array had been edited), %d mirrored"
          % (n_fast, len(steps), n_stale, n_mirror))
    assert n_fast >= 5 and n_stale == 1 and n_mirror >= 8
    icm._engine.close()


def test_dataset_driver_loop_in_reference_order_with_a_registered_array(icm):
    """The reference-order (sequential) sweeps of the dataset through a registered array (uploaded by DMA, written in
    place by the layout kernel): results as in test_driver_loop_reproduces_reference."""
    init = gold("init_pass.npz")
    icm.set_initial_state(init["x_init"], init["map_init"], init["cant_obs_i"])
    mapa_viejo, x = copy(icm.mapa_viejo), copy(icm.positions)
    for it in range(3):
        if it == 2:       # third call: back to the state of the first, in the same (by now registered) array
            x[...] = init["x_init"]
            mapa_viejo = copy(icm.mapa_viejo)
            icm.mapa_obj.landmarks_actuales = mapa_viejo.shape[1]
        mapa_refinado, x = icm.iterations_process_offline(mapa_viejo, x)
        g = gold("sweep%02d.npz" % (1 if it == 2 else it + 1))
        assert np.abs(mapa_refinado - g["mapa"]).max() <= 1e-9 and np.abs(x - g["x"]).max() <= 1e-9
        mapa_viejo = copy(mapa_refinado)


def test_in_place_edits_of_the_sequence_and_invalidate_sequence():
    """The reference re-reads `mediciones` on every call (scripts/ICM_ROS.py:142); here the sequence is uploaded and
    pre-filtered once and its identity is a sampled checksum: replacing the array or editing a sampled
    column re-uploads by itself, an edit of an unsampled column must be announced with invalidate_sequence() -- and is
    then honoured exactly (the result equals a fresh object's on the edited data)."""
    from ICM_ROS import ICM_ROS
    from ICM_SLAM_tools import ConfigICM
    init = gold("init_pass.npz")

    def fresh(med=None):
        m = ICM_ROS(ConfigICM("config_default.yaml"))
        m.load_data(os.path.join(GOLD, "data_IJAC2018.npz"))
        if med is not None:
            m.mediciones = med
        m.set_initial_state(init["x_init"], init["map_init"], init["cant_obs_i"])
        return m

    def sweep(m):
        m.set_initial_state(init["x_init"], init["map_init"], init["cant_obs_i"])
        mv, x = copy(m.mapa_viejo), copy(m.positions)
        return m.iterations_process_offline(mv, x)

    m = fresh()
    base_map, base_x = sweep(m)
    T = m.mediciones.shape[1]
    step = max(1, T // 16)
    unsampled = next(t for t in range(700, T) if t % step and t != T - 1)
    edited = m.mediciones.copy()
    edited[:, unsampled] = m.config.rango_laser_max        # that scan loses all its beams
    want_map, want_x = sweep(fresh(edited))
    assert np.abs(want_x - base_x).max() > 1e-6             # the edit matters
    # in place, unsampled column: not noticed (the documented difference) ...
    m.mediciones[:, unsampled] = m.config.rango_laser_max
    got_map, got_x = sweep(m)
    assert np.array_equal(got_x, base_x)
    # ... until announced
    m.invalidate_sequence()
    got_map, got_x = sweep(m)
    assert np.array_equal(got_x, want_x) and np.array_equal(got_map, want_map)
    # a sampled column (the last one is always sampled), in place: noticed by itself
    m2 = fresh()
    sweep(m2)
    m2.mediciones[:, 0] = m2.mediciones[:, 1]
    ed2 = m2.mediciones.copy()
    w_map, w_x = sweep(fresh(ed2))
    g_map, g_x = sweep(m2)
    assert np.array_equal(g_x, w_x) and np.array_equal(g_map, w_map)
    # a replaced array: noticed by itself
    m3 = fresh()
    sweep(m3)
    m3.mediciones = edited.copy()
    g_map, g_x = sweep(m3)
    assert np.array_equal(g_x, want_x)


def test_two_pose_sequence_through_the_drop_in_call_on_a_registered_array():
    """T = 2 (icm_upload's minimum): there is no even pose whose lane would mirror the pair (1, 2) into a registered host
    array, so the call must hand pose 1 back through the ordinary download -- call after call on the same array."""
    from ICM_SLAM_tools import ConfigICM
    from icmslam_hip import SweepEngine
    from icmslam_hip.synthetic import make_workload
    wl = make_workload(700, 64, 180)
    cfg = ConfigICM(D=wl.config)
    cfg.cota = 1.0
    sl = slice(0, 2)
    x = np.ascontiguousarray(wl.x_init[:, sl]).copy()
    eng = SweepEngine(cfg)
    eng.upload(wl.scans[sl], wl.odometry[:, sl], wl.u[:, sl], pose_major=True)
    ref = SweepEngine(cfg)
    ref.upload(wl.scans[sl], wl.odometry[:, sl], wl.u[:, sl], pose_major=True)
    m = wl.map_init.copy()
    K = m.shape[1]
    for call in range(4):
        xr = x.copy()
        mo_r, co_r, K_r = ref.sweep(m[:, :K], xr, wl.x0, K, "redblack")       # a fresh copy every call: never registered
        mo, co, K2 = eng.sweep(m[:, :K], x, wl.x0, K, "redblack")             # the same array every call: registered from the second
        assert K2 == K_r and np.array_equal(x, xr), "call %d" % call
        assert np.array_equal(mo, mo_r)
        assert not np.array_equal(x[:, 1], wl.x_init[:, 1])                   # pose 1 was solved and came back
        m, K = mo, K2
    eng.close()
    ref.close()


def test_serialised_streams_make_the_side_stream_wait_give_up_once():
    """With HIP_LAUNCH_BLOCKING=1 no two launches overlap: the one-wave kernel that polls for the raw map on the side stream
    (k_wait_word) can never see the word the main stream's next launch would set.  It must give up (bounded), that sweep's
    Mapa.filtrar then runs on the host, the handle switches to the stop event for good -- ONE give-up however many sweeps --
    and every sweep's result equals an ordinary run's."""
    import subprocess
    import sys
    code = r'''
import sys, zlib
sys.path.insert(0, "icm-slam_amd"); sys.path.insert(0, ".")
import numpy as np
from ICM_SLAM_tools import ConfigICM
from icmslam_hip import SweepEngine
from icmslam_hip.synthetic import WORKLOADS, make_workload
wl = make_workload(*WORKLOADS["tiny"])
eng = SweepEngine(ConfigICM(D=wl.config))
eng.upload(wl.scans, wl.odometry, wl.u, pose_major=True)
eng.set_state(wl.map_init, wl.x_init, wl.x0)
for _ in range(4):
    eng.sweep_device("redblack")
x, m, c, K = eng.get_state()
print("RESULT", eng.wait_giveups(), K, zlib.crc32(x.tobytes()), zlib.crc32(m.tobytes()))
eng.close()
'''
    root = os.path.dirname(GOLD.rstrip("/")).rsplit("/tests", 1)[0]
    outs = {}
    for blocking in ("0", "1"):
        env = dict(os.environ, HIP_LAUNCH_BLOCKING=blocking)
        p = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=600)
        line = [ln for ln in p.stdout.splitlines() if ln.startswith("RESULT")]
        assert p.returncode == 0 and line, p.stderr[-2000:]
        outs[blocking] = line[0].split()[1:]
    assert outs["0"][0] == "0", "an ordinary run never gives up"
    assert outs["1"][0] == "1", "serialised streams: one give-up, then the event path (got %s)" % outs["1"][0]
    assert outs["0"][1:] == outs["1"][1:], "same map and poses either way"
