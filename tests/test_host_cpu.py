"""CPU-only checks of the product's host side: the C-ABI library loads and exports every
symbol of include/icmslam.h, the host Mapa.filtrar matches the reference goldens and the
oracle, configs load, the synthetic generator is deterministic.  No GPU compute here."""
import os
import re

import numpy as np
import pytest

from util import Cfg, ROOT, dataset, gold


def test_library_exports_every_declared_symbol():
    from icmslam_hip import _lib
    lib = _lib.load()
    header = "".join(open(os.path.join(ROOT, "include", f)).read() for f in ("icmslam.h", "icmslam_tuning.h"))
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)        # declarations only, not the prose about them
    declared = set(re.findall(r"\b(icm_[a-z_0-9]+)\s*\(", header))
    declared -= {"icm_allgather_fn"}
    assert declared, "no declarations parsed"
    for name in sorted(declared):
        assert hasattr(lib, name), "libicmslam_hip.so lacks %s" % name
    assert declared == set(_lib.SIGNATURES), (declared ^ set(_lib.SIGNATURES))
    assert b"gfx950" in lib.icm_version()


def test_library_collectives_entry_points_without_a_gpu():
    """The entry points of the library-issued collectives refuse a null handle without resolving RCCL or touching a
    device (RCCL itself is only dlopen()ed when a communicator is asked for: tests/test_gpu_sharded.py)."""
    from icmslam_hip import _lib
    lib = _lib.load()
    assert lib.icm_sweep_sharded(None) != 0 and lib.icm_gather_poses(None) != 0
    assert lib.icm_comm_init(None, None, 0, 1) != 0 and lib.icm_comm_destroy(None) != 0


def test_staging_layout_is_consistent_for_every_shape_of_shard():
    """The one helper that sizes the staging area of phase A's entries, the sparse area behind it and the per-entry
    prefix arrays (staging_layout, icm_host.hpp; the round-2 out-of-bounds store came from this arithmetic being spread
    over four functions): for shards with far fewer kept beams than poses, none at all, one-chunk sequences and the S2
    shape, every place a kernel can address lies inside the arrays."""
    from icmslam_hip import _lib
    lib = _lib.load()
    SLACK, WAVE = 4, 64
    for nnz, nloc in ((0, 1), (0, 5000), (1, 1), (3, 100000), (37, 2), (64, 16), (1000, 16), (23_000_000, 100_000), (460_000_000, 2_000_000)):
        out = np.zeros(3, dtype=np.int64)
        assert lib.icm_staging_layout(nnz, nloc, out.ctypes.data_as(_lib._lp)) == 0
        sparse0, entries, stride = (int(v) for v in out)
        # packed area: pose t owns [plan[t] + SLACK t, plan[t+1] + SLACK (t+1)), plan <= nnz: its end for the last pose
        assert nnz + SLACK * nloc <= sparse0
        # sparse area: a pose that does not fit stages at sparse0 + (its first kept beam), at most one entry per beam,
        # and the wave-wide compaction store may run one wave past its last entry
        assert sparse0 + max(nnz, 1) + WAVE <= entries
        # the per-entry prefixes live at the same places; k_chunk_l1's branch-free stores reach one more wave
        assert entries + WAVE <= stride
        assert stride < 2 ** 31
    out = np.zeros(3, dtype=np.int64)
    assert lib.icm_staging_layout(1_200_000_000, 1_000_000, out.ctypes.data_as(_lib._lp)) == _lib.ICM_ERR_CAPACITY   # beyond 32-bit offsets
    assert lib.icm_staging_layout(-1, 10, out.ctypes.data_as(_lib._lp)) == _lib.ICM_ERR_CAPACITY


def test_shard_partition_cuts_at_even_poses():
    from icmslam_hip import _lib
    from icmslam_hip.sharded import partition, shard_block, world_fits
    lib = _lib.load()
    for T in (2, 3, 249, 600, 1833, 1900, 100_000, 800_000):
        for world in (1, 2, 3, 7, 8):
            assert shard_block(T, world) == lib.icm_shard_block(T, world)
            if not world_fits(T, world):      # (a trailing rank would be left without poses: refused)
                with pytest.raises(ValueError):
                    partition(T, world)
                continue
            blk, parts = partition(T, world)
            assert blk == shard_block(T, world) and blk % 2 == 0
            assert parts[0][0] == 0 and parts[-1][1] == T and all(b > a and a % 2 == 0 for a, b in parts)
            assert all(parts[i][1] == parts[i + 1][0] for i in range(world - 1))
            assert all(b - a == blk for a, b in parts if b < T)


def test_create_without_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from icmslam_hip import SweepEngine, IcmError
    with pytest.raises(IcmError, match="no CPU fallback"):
        SweepEngine(Cfg())


def test_host_filtrar_matches_reference_golden():
    from icmslam_hip import filtrar_map
    pp = gold("sweep1_perpose.npz")
    cfg = Cfg()
    la = int(pp["filtrar_lact_in"])
    y = np.zeros((2, cfg.L))
    c = np.zeros(cfg.L)
    y[:, :la] = pp["filtrar_y_in"]
    c[:la] = pp["filtrar_cnt_in"]
    yo, co, lo = filtrar_map(cfg, y, c, la)
    assert lo == int(pp["filtrar_lact_out"]) == 11
    assert np.array_equal(yo[:, :lo], pp["filtrar_y_out"])
    assert np.array_equal(co[:lo], pp["filtrar_cnt_out"])
    assert not yo[:, lo:].any() and not co[lo:].any()


@pytest.mark.parametrize("seed", range(6))
def test_host_filtrar_matches_oracle_on_merge_cases(seed):
    """Random maps with clusters closer than dist_thr (merges, chains of merges, coincident
    landmarks) and rarely-seen landmarks (pruned)."""
    from icmslam_hip import filtrar_map
    from oracle import icm_oracle as o
    rng = np.random.default_rng(seed)
    cfg = Cfg(L=200, cota=5.0)
    n = 60
    pts = rng.uniform(-8, 8, (2, n))
    pts[:, 10:20] = pts[:, 0:10] + rng.normal(0, 0.3, (2, 10))   # pairs within the gate
    pts[:, 20:25] = pts[:, 0:5] + rng.normal(0, 0.3, (2, 5))     # triples
    if seed % 2:
        pts[:, 30] = pts[:, 31]                                   # coincident pair
    cnt = rng.integers(1, 40, n).astype(float)
    y = np.zeros((2, cfg.L)); c = np.zeros(cfg.L)
    y[:, :n] = pts; c[:n] = cnt
    yo, co, lo = filtrar_map(cfg, y, c, n)
    st = o.MapState(o.OracleConfig.from_config(cfg), n)
    st.cant_obs_i = c.copy()
    yr = o.filtrar(st, y.copy())
    assert lo == st.landmarks_actuales
    assert np.allclose(yo, yr, rtol=0, atol=1e-12) and np.array_equal(co, st.cant_obs_i)


def test_host_filtrar_edge_cases():
    from icmslam_hip import filtrar_map
    cfg = Cfg(L=16, cota=3.0)
    y = np.zeros((2, 16)); c = np.zeros(16)
    with pytest.raises(ValueError):           # nothing reaches cota
        filtrar_map(cfg, y, c, 4)
    y[:, 0] = (1.0, 2.0); c[0] = 7            # a single survivor keeps its place (y*n/n)
    yo, co, lo = filtrar_map(cfg, y, c, 4)
    assert lo == 1 and np.allclose(yo[:, 0], (1.0, 2.0)) and co[0] == 7
    with pytest.raises(IndexError):
        filtrar_map(cfg, y, c, 17)


def test_config_default_yaml_loads_with_defaults():
    from ICM_SLAM_tools import ConfigICM
    c = ConfigICM("config_default.yaml")   # the reference's own ConfigICM raises KeyError here
    assert (c.N, c.L, c.cota, c.dist_thr, c.rango_laser_max, c.radio) == (2, 1000, 300.0, 1.0, 10.0, 0.137)
    assert c.Q.shape == (2, 2) and c.R.shape == (3, 3) and c.topic_laser and c.time == 275.0
    assert c.angle_increment is None and c.schedule == "sequential"
    r = ConfigICM("config_ros.yaml")
    assert r.N == 30 and r.file == "data_IJAC2018.mat"
    d = ConfigICM(D=dict(N=1, deltat=0.1, L=10, Q=[1, 2], R=[1, 2, 3], cte_odom=1.0, cota=5, dist_thr=1.0,
                         dist_thr_obs=1.0, rango_laser_max=10.0, radio=0.1, schedule="redblack"))
    assert d.Q[1, 1] == 2 and d.R[2, 2] == 3 and d.schedule == "redblack"


def test_subclass_overrides_are_refused():
    import ICM_ROS as M
    class Mine(M.ICM_ROS):
        def h(self, xt, zt):
            return 0.0
    m = Mine(Cfg())
    with pytest.raises(NotImplementedError, match="overrides"):
        m._check_overrides()
    M.ICM_ROS(Cfg())._check_overrides()


def test_synthetic_workload_is_deterministic_and_shardable():
    from icmslam_hip.synthetic import make_workload
    a = make_workload(300, 49, 180)
    b = make_workload(300, 49, 180)
    assert np.array_equal(a.scans, b.scans) and np.array_equal(a.x_init, b.x_init)
    lo = make_workload(300, 49, 180, t_begin=0, t_end=150)
    hi = make_workload(300, 49, 180, t_begin=150, t_end=300)
    assert lo.scans.shape == (150, 180) and hi.scans.shape == (150, 180)
    # same geometry AND the same range noise whoever generates a pose's scan (counter-based noise stream): an N-rank
    # job sweeps exactly the N = 1 job's inputs; the scan in front of a shard (its ghost pose) comes with it
    assert np.array_equal(lo.map_init, hi.map_init) and np.array_equal(lo.x_init, a.x_init)
    assert np.array_equal(lo.scans, a.scans[:150]) and np.array_equal(hi.scans, a.scans[150:])
    assert lo.ghost_scan is None and np.array_equal(hi.ghost_scan, a.scans[149])
    odd = make_workload(300, 49, 180, t_begin=77, t_end=201)
    assert np.array_equal(odd.scans, a.scans[77:201]) and np.array_equal(odd.ghost_scan, a.scans[76])
    hits = (a.scans < 10.0).sum(axis=1)
    assert hits.mean() > 10 and a.scans.min() > 0


def test_host_first_scan_clustering_equals_scipy():
    """icm_cluster_first_scan (host C++) == fcluster(linkage(pdist(.)), t) - 1."""
    from scipy.cluster.hierarchy import fcluster, linkage
    from scipy.spatial.distance import pdist
    from icmslam_hip import cluster_first_scan
    rng = np.random.default_rng(3)
    for _ in range(200):
        pts = np.concatenate([rng.normal(c, 0.2, (rng.integers(1, 8), 2)) for c in rng.uniform(-6, 6, (rng.integers(1, 6), 2))])
        if pts.shape[0] < 2:
            continue
        assert np.array_equal(cluster_first_scan(pts, 1.0), fcluster(linkage(pdist(pts)), 1.0) - 1)
    assert np.array_equal(cluster_first_scan(np.array([[1.0, 2.0]]), 1.0), [0])
    s = gold("init_pass.npz")
    assert set(s["labels_scan0"]) == {0, 1}


def test_message_replay_round_trip_through_the_topic_parsers():
    """matlab2ros.replay (counterpart of the reference's 10 Hz publisher) -> Lidar / Odometria
    callbacks -> ICM_ROS.load_messages reproduces what load_data() reads from the file: ranges
    inflated by the trunk radius and clipped, poses, twists.  (The LaserScan parser keeps 180 of
    the dataset's 181 beams, reference scripts/sensors_definitions.py:23-29.)"""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "icm-slam_amd"))
    from ICM_SLAM_tools import ConfigICM
    from ICM_ROS import ICM_ROS
    from sensors_definitions import Lidar, Odometria
    from matlab2ros.replay import replay, stamp_of
    d = gold("data_IJAC2018.npz")
    T = 300
    cfg = ConfigICM(os.path.join(ROOT, "icm-slam_amd", "config_default.yaml"))
    lidar, odo = Lidar(config=cfg), Odometria(config=cfg)
    n = replay(d["observations"][:, :T], d["odometry"][:, :T], d["velocities"][:, :T], lidar.callback, odo.callback)
    assert n == T and len(lidar.msgs) == T and len(odo.msgs) == T
    assert stamp_of(25) == {"secs": 2, "nsecs": 500000000}
    icm = ICM_ROS.__new__(ICM_ROS)      # (no GPU needed for the loaders)
    icm.config = cfg
    z, od, u = icm.load_messages(lidar, odo)
    zz, odo_ref, u_ref = dataset()
    assert z.shape == (180, T)
    assert np.array_equal(z, zz[:180, :T])
    assert np.array_equal(od[:2], odo_ref[:2, :T]) and np.array_equal(u, u_ref[:, :T])
    dyaw = np.abs(np.angle(np.exp(1j * (od[2] - odo_ref[2, :T]))))
    assert dyaw.max() <= 1e-14          # yaw through the quaternion and back


def test_replayed_messages_equal_the_reference_publishers():
    """Row f4 of the scope table: the LaserScan / Odometry messages matlab2ros/replay.py builds are, field by field,
    the ones the reference's publisher emits (createbag.py:37-121: mat2laser_scann, mat2odometry, Header.new_message;
    fixture made by tests/golden/make_golden_messages.py from the imported reference), and the arrays the topic
    parsers make of the reference's messages are bitwise the ones they make of ours."""
    import json
    from matlab2ros import replay
    from sensors_definitions import Lidar, Odometria
    fx = json.load(open(os.path.join(ROOT, "tests", "golden", "createbag_messages.json")))["samples"]
    d = gold("data_IJAC2018.npz")
    msgs = list(replay.messages(d["observations"], d["odometry"], d["velocities"]))
    assert sorted(int(k) for k in fx) == [0, 1, 100, 1832]
    cfg = Cfg()
    for k, ref in fx.items():
        scan, odom = msgs[int(k)]
        assert scan == ref["laser_scan"], "LaserScan of sample %s differs" % k
        assert odom == ref["odometry"], "Odometry of sample %s differs" % k
        assert set(scan) == set(ref["laser_scan"]) and set(odom["header"]) == set(ref["odometry"]["header"])
        parsed = []
        for pair in ((scan, odom), (ref["laser_scan"], ref["odometry"])):
            li, od = Lidar(config=cfg), Odometria(config=cfg)
            li.callback(pair[0])
            od.callback(pair[1])
            parsed.append((li.msgs[0], od.msgs[0]))
        (la, oa), (lb, ob) = parsed
        assert la["seq"] == lb["seq"] == int(k) and la["stamp"] == lb["stamp"] and np.array_equal(la["data"], lb["data"])
        assert oa["stamp"] == ob["stamp"] and np.array_equal(oa["data"]["odo"], ob["data"]["odo"]) and np.array_equal(oa["data"]["u"], ob["data"]["u"])
        # and the parsed sample is the dataset's (ranges prepared as scripts/sensors_definitions.py:22; the Lidar parser
        # keeps 180 of the 181 beams like the reference's, :23-29)
        zz = np.minimum(d["observations"][:, int(k)] + cfg.radio, cfg.rango_laser_max)
        assert np.array_equal(la["data"][:, 0], zz[:180])
        assert np.array_equal(oa["data"]["odo"][:2, 0], d["odometry"][:2, int(k)]) and np.array_equal(oa["data"]["u"][:, 0], d["velocities"][:, int(k)])
        assert abs(oa["data"]["odo"][2, 0] - np.arctan2(np.sin(d["odometry"][2, int(k)]), np.cos(d["odometry"][2, int(k)]))) < 1e-12
