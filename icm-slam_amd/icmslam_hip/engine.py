"""Host-side driver of the HIP sweep: owns one C-ABI handle (one GPU / rank).

The arrays keep the reference's layouts: `mediciones` (B,T) beam-major, `odometria`
(3,T), `u` (2,T), poses `x` (3,T), maps (2,K) (reference scripts/ICM_ROS.py:20-24,121).
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import IcmConfig, SCHEDULES, dptr, iptr, lptr


class IcmError(RuntimeError):
    pass


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def bearing_tables(B, angle_increment=None):
    """cos/sin of the beam bearings.  Default = the reference's hard-coded 1 degree per scan
    row, index*pi/180 (reference scripts/ICM_SLAM_tools.py:44,51)."""
    k = np.arange(B)
    ang = k * np.pi / 180.0 if angle_increment is None else k * float(angle_increment)
    return np.cos(ang), np.sin(ang), ang


def make_c_config(config):
    c = IcmConfig()
    c.deltat = float(config.deltat)
    c.Q[0], c.Q[1] = float(config.Q[0, 0]), float(config.Q[1, 1])
    c.R[0], c.R[1], c.R[2] = float(config.R[0, 0]), float(config.R[1, 1]), float(config.R[2, 2])
    c.cte_odom = float(config.cte_odom)
    c.cota = float(config.cota)
    c.dist_thr = float(config.dist_thr)
    c.rango_laser_max = float(config.rango_laser_max)
    c.L = int(config.L)
    return c


def filtrar_map(config, y, counts, lact):
    """Mapa.filtrar through the C-ABI host routine (no GPU involved).
    Returns (y_out (2,L), counts_out (L), lact_out)."""
    lib = _lib.load()
    cc = make_c_config(config)
    L = int(config.L)
    y = _f64(y)
    counts = _f64(counts)
    if y.shape != (2, L) or counts.shape != (L,):
        raise ValueError("filtrar: map must be (2,L) and counts (L,)")
    yo = np.zeros((2, L))
    co = np.zeros(L)
    lo = C.c_int64(0)
    rc = lib.icm_filtrar(C.byref(cc), dptr(y), dptr(counts), int(lact), dptr(yo), dptr(co), C.byref(lo))
    if rc:
        _raise(rc, lib.icm_last_error(None).decode())
    return yo, co, int(lo.value)


def cluster_first_scan(points, t):
    """Flat clusters of the first scan (`Mapa.actualizar` with no landmark yet, reference
    scripts/ICM_SLAM_tools.py:160-165): labels 0..ncl-1 of the (n,2) world points."""
    lib = _lib.load()
    pts = _f64(points)
    if pts.ndim != 2 or pts.shape[1] != 2:
        raise ValueError("points must be (n,2)")
    lab = np.zeros(pts.shape[0], dtype=np.int32)
    rc = lib.icm_cluster_first_scan(dptr(pts), pts.shape[0], float(t), iptr(lab))
    if rc:
        _raise(rc, lib.icm_last_error(None).decode())
    return lab


_PREFILTER_ENGINES = {}
_SCAN_ENGINES = {}


def _config_key(config, device):
    """Cache key of a helper engine: the VALUES the library is configured with (id() of a config object is recycled
    after garbage collection, and a config may be edited in place)."""
    c = make_c_config(config)
    return (c.deltat, c.Q[0], c.Q[1], c.R[0], c.R[1], c.R[2], c.cte_odom, c.cota, c.dist_thr, c.rango_laser_max, int(c.L),
            getattr(config, "angle_increment", None), int(device))


def actualizar_scan(config, mapa, mapa_referencia, obs, lact, cant_obs_i, device=0):
    """Mapa.actualizar for one scan through the C-ABI (`icm_associate`, reference
    scripts/ICM_SLAM_tools.py:128-201): `mapa` (2,L) float64 C-contiguous and `cant_obs_i` (L) are
    updated IN PLACE, `mapa_referencia` (2,K) is only read.  Returns (labels c (n,) int64, new
    landmarks_actuales)."""
    key = _config_key(config, device)
    eng = _SCAN_ENGINES.get(key)
    if eng is None:
        eng = _SCAN_ENGINES[key] = SweepEngine(config, device)
    L = int(config.L)
    if not (isinstance(mapa, np.ndarray) and mapa.dtype == np.float64 and mapa.flags.c_contiguous and mapa.shape == (2, L)):
        raise ValueError("actualizar: mapa must be a C-contiguous float64 (2,L) array (it is updated in place)")
    if not (isinstance(cant_obs_i, np.ndarray) and cant_obs_i.dtype == np.float64 and cant_obs_i.flags.c_contiguous and cant_obs_i.shape == (L,)):
        raise ValueError("actualizar: cant_obs_i must be a C-contiguous float64 (L,) array")
    o = _f64(obs)
    if o.ndim != 2 or (o.shape[0] and o.shape[1] != 2):
        raise ValueError("actualizar: obs must be (n,2) world points")
    n = o.shape[0]
    ref = _f64(mapa_referencia) if int(lact) > 0 else np.zeros((2, 0))
    c = np.zeros(max(n, 1), dtype=np.int64)
    la = C.c_int64(int(lact))
    eng._chk(eng.lib.icm_associate(eng.h, dptr(o), n, dptr(ref), ref.shape[1], dptr(mapa), dptr(cant_obs_i), C.byref(la), lptr(c)))
    return c[:n], int(la.value)


def prefilter_scans(config, scans, device=0):
    """filtrar_z for the columns of `scans` (B,n) on the GPU (reference
    scripts/ICM_SLAM_tools.py:22-58).  Returns a list of (n_t,4) arrays [d, ang, bx, by];
    an empty (0,) array where the reference returns `np.array([])`."""
    scans = np.asarray(scans, dtype=np.float64)
    if scans.ndim == 1:
        scans = scans[:, None]
    B, n = scans.shape
    key = _config_key(config, device)
    eng = _PREFILTER_ENGINES.get(key)
    if eng is None:
        eng = _PREFILTER_ENGINES[key] = SweepEngine(config, device)
    pad = max(2 - n, 0)
    if pad:
        scans = np.concatenate((scans, np.repeat(scans[:, -1:], pad, axis=1)), axis=1)
    T = scans.shape[1]
    eng.upload(scans, np.zeros((3, T)), np.zeros((2, T)))
    off, bk, d, bx, by = eng.kept_beams()
    _, _, ang = bearing_tables(B, getattr(config, "angle_increment", None))
    out = []
    for t in range(n):
        sl = slice(off[t], off[t + 1])
        if off[t + 1] == off[t]:
            out.append(np.array([]))
        else:
            out.append(np.stack((d[sl], ang[bk[sl]], bx[sl], by[sl]), axis=1))
    return out


def _raise(rc, msg):
    if rc == _lib.ICM_ERR_INDEX:
        raise IndexError(msg)
    if rc == _lib.ICM_ERR_EMPTY_MAP:
        raise ValueError(msg)
    if rc == _lib.ICM_ERR_UNSUPPORTED:
        raise NotImplementedError(msg)
    if rc == _lib.ICM_ERR_ARG:
        raise ValueError(msg)
    raise IcmError("icmslam_hip error %d: %s" % (rc, msg))


class SweepEngine:
    """One GPU worth of the offline ICM sweep."""

    def __init__(self, config, device=0):
        self.lib = _lib.load()
        self.config = config
        self.cconf = make_c_config(config)
        self.L = int(config.L)
        h = C.c_void_p()
        rc = self.lib.icm_create(C.byref(self.cconf), int(device), C.byref(h))
        if rc:
            _raise(rc, self.lib.icm_last_error(None).decode())
        self.h = h
        self.T = self.B = self.nloc = 0
        self.t_begin = 0
        self.nnz = 0
        self._last_x = self._pinned_x = None   # _pin_if_reused

    def close(self):
        if getattr(self, "h", None):
            self.lib.icm_destroy(self.h)   # (unregisters whatever icm_pin_host registered)
            self.h = None
        self._last_x = self._pinned_x = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        self.last_rc = rc
        if rc:
            _raise(rc, self.lib.icm_last_error(self.h).decode())

    # ---- sequence --------------------------------------------------------------------
    def upload(self, mediciones, odometria, u, t_begin=0, t_end=None, pose_major=False, ghost_scan=None):
        """Upload the sequence (or this rank's pose shard of the scans) and run the scan
        pre-filter once.  `mediciones` is (B,T) like the reference unless pose_major.
        ghost_scan (B,): the scan of pose t_begin - 1 -- ranks > 0 of a pose-sharded job solve that pose too
        (icm_upload_ghost_scan); taken from `mediciones` itself when that holds the whole sequence."""
        odometria = _f64(odometria)
        u = _f64(u)
        T = odometria.shape[1]
        t_end = T if t_end is None else int(t_end)
        if pose_major:
            scans = _f64(mediciones)
            B = scans.shape[1]
            if scans.shape[0] != t_end - t_begin:
                raise ValueError("pose-major scans must hold exactly the shard's poses")
        else:
            m = np.asarray(mediciones, dtype=np.float64)
            B = m.shape[0]
            if m.shape[1] != T:
                raise ValueError("mediciones must be (B,T)")
            scans = np.ascontiguousarray(m[:, t_begin:t_end].T)
            if ghost_scan is None and t_begin >= 2 and t_begin % 2 == 0:
                ghost_scan = m[:, t_begin - 1]
        if odometria.shape != (3, T) or u.shape != (2, T):
            raise ValueError("odometria must be (3,T) and u (2,T)")
        cosb, sinb, _ = bearing_tables(B, getattr(self.config, "angle_increment", None))
        self._chk(self.lib.icm_upload(self.h, dptr(scans), dptr(odometria), dptr(u), dptr(_f64(cosb)),
                                      dptr(_f64(sinb)), T, B, int(t_begin), t_end))
        if ghost_scan is not None:
            g = _f64(np.asarray(ghost_scan, dtype=np.float64).reshape(-1))
            if g.shape != (B,):
                raise ValueError("ghost_scan must hold B ranges")
            self._chk(self.lib.icm_upload_ghost_scan(self.h, dptr(g)))
        nnz = C.c_int64(0)
        self._chk(self.lib.icm_prefilter(self.h, C.byref(nnz)))
        self.T, self.B, self.t_begin, self.nloc, self.nnz = T, B, int(t_begin), t_end - int(t_begin), int(nnz.value)
        return self.nnz

    def kept_beams(self):
        """(offsets[nloc+1], beam_index, d, bx, by) of the cached filtrar_z output."""
        off = np.zeros(self.nloc + 1, dtype=np.int64)
        n = max(self.nnz, 1)
        bk = np.zeros(n, dtype=np.int32)
        d, bx, by = np.zeros(n), np.zeros(n), np.zeros(n)
        self._chk(self.lib.icm_get_kept(self.h, lptr(off), iptr(bk), dptr(d), dptr(bx), dptr(by)))
        k = self.nnz
        return off, bk[:k], d[:k], bx[:k], by[:k]

    # ---- initialisation pass ----------------------------------------------------------------
    def init_pass(self, x0):
        """The causal first pass over the uploaded sequence (reference inicializar_online,
        scripts/ICM_ROS.py:57-100, ROS-free): scan 0 is clustered into the first landmarks, then
        every pose is predicted, associated against the running map and solved (one-sided).
        Returns (x_init (3,T), y_raw (2,L), counts (L), landmarks_actuales) BEFORE Mapa.filtrar."""
        x0 = _f64(np.asarray(x0, dtype=np.float64).reshape(3))
        off, bk, d, bx, by = self.kept_beams()
        n0 = int(off[1] - off[0])
        if n0 == 0:
            raise ValueError("the first scan has no kept beam: nothing to seed the map with")
        ct, st = np.cos(x0[2] - np.pi / 2.0), np.sin(x0[2] - np.pi / 2.0)
        w = np.stack(((bx[:n0] * ct - by[:n0] * st) + x0[0], (bx[:n0] * st + by[:n0] * ct) + x0[1]), axis=1)
        c = cluster_first_scan(w, self.config.dist_thr)
        y = np.zeros((2, self.L))
        cnt = np.zeros(self.L)
        lact = int(c.max()) + 1
        for i in range(lact):          # cluster centres and sizes (scripts/ICM_SLAM_tools.py:163-165)
            y[:, i] = np.mean(w[c == i, :], axis=0)
            cnt[i] = np.sum(c == i)
        la = C.c_int64(lact)
        x = np.zeros((3, self.T))
        self._chk(self.lib.icm_init_pass(self.h, dptr(x0), dptr(y), dptr(cnt), C.byref(la), dptr(x)))
        return x, y, cnt, int(la.value), c

    # ---- one sweep through host arrays -------------------------------------------------
    def sweep(self, mapa_viejo, x, x0, lact, schedule="sequential"):
        """iterations_process_offline.  x (3,T) float64 C-contiguous is updated IN PLACE.
        Returns (map (2,L) zero padded, counts (L), K) or None if scan 0 has no beams."""
        if not (isinstance(x, np.ndarray) and x.dtype == np.float64 and x.flags.c_contiguous and x.shape == (3, self.T)):
            raise ValueError("x must be a C-contiguous float64 array of shape (3,T)")
        mv = _f64(mapa_viejo)
        K = mv.shape[1]
        x0v = _f64(np.asarray(x0, dtype=np.float64).reshape(3))
        mo = np.empty((2, self.L))     # (the library writes every element: zero padded like Mapa.filtrar's return)
        co = np.empty(self.L)
        ko = C.c_int64(0)
        self._pin_if_reused(x)
        self._chk(self.lib.icm_sweep(self.h, dptr(x), dptr(x0v), dptr(mv), K, int(lact), SCHEDULES[schedule],
                                     dptr(mo), dptr(co), C.byref(ko)))
        if ko.value < 0:
            return None
        return mo, co, int(ko.value)

    def _pin_if_reused(self, x):
        """The reference's driver loop hands the SAME pose array back sweep after sweep (it is updated in place,
        scripts/ICM_ROS.py:158,164,298-311).  An array seen in two consecutive calls is registered with the GPU runtime
        (icm_pin_host): from then on the library reads and writes it in place over PCIe instead of through two staged
        2.4 MB copies.  The engine keeps a reference to the registered array, so its memory cannot be freed and its
        address re-used while registered; one array at a time (a new one replaces it)."""
        if self._pinned_x is x:
            return
        if self._last_x is x:       # second call in a row with this very array
            if self._pinned_x is not None:
                self.lib.icm_unpin_host(self.h, C.c_void_p(self._pinned_x.ctypes.data))
                self._pinned_x = None
            if self.lib.icm_pin_host(self.h, C.c_void_p(x.ctypes.data), C.c_size_t(x.nbytes)) == 0:
                self._pinned_x = x
        self._last_x = x

    # ---- device-resident sweeps --------------------------------------------------------
    def set_state(self, mapa_viejo, x, x0, lact=None):
        mv = _f64(mapa_viejo)
        xx = _f64(x)
        x0v = _f64(np.asarray(x0, dtype=np.float64).reshape(3))
        lact = mv.shape[1] if lact is None else int(lact)
        self._chk(self.lib.icm_set_state(self.h, dptr(xx), dptr(x0v), dptr(mv), mv.shape[1], lact))

    def sweep_device(self, schedule="redblack"):
        self._chk(self.lib.icm_sweep_device(self.h, SCHEDULES[schedule]))

    def sweep_local(self):
        self._chk(self.lib.icm_sweep_local(self.h))

    def sweep_targets(self):
        self._chk(self.lib.icm_sweep_targets(self.h))

    def mark_failed(self, code):
        """This rank's phase A failed with `code` (< 0): put the code into its message so that the other ranks learn
        of it through the sweep's collective instead of waiting for a rank that left (icm_mark_failed)."""
        self._chk(self.lib.icm_mark_failed(self.h, int(code)))

    def failed_rank(self):
        """After the exchange, careful form: (rank, code) of the first rank that failed in phase A, or (-1, 0)."""
        r, c = C.c_int(-1), C.c_int(0)
        self._chk(self.lib.icm_failed_rank(self.h, C.byref(r), C.byref(c)))
        return int(r.value), int(c.value)

    def exchange_status(self):
        """After the exchange: (rank, code, retry) -- the first rank that failed in phase A with its code (-1, 0: none) and
        whether some rank reports flags of a sweep it had queued whole, i.e. everybody repeats the sweep
        (icm_exchange_status)."""
        r, c, q = C.c_int(-1), C.c_int(0), C.c_int(0)
        self._chk(self.lib.icm_exchange_status(self.h, C.byref(r), C.byref(c), C.byref(q)))
        return int(r.value), int(c.value), bool(q.value)

    def sweep_is_optimistic(self):
        """Whether the sweep sweep_local() started was queued whole (a request, set_optimistic, is not always granted)."""
        return self.lib.icm_get_optimistic(self.h) == 1

    def set_phase_timing(self, on=True):
        """Events at the phase boundaries of every sweep queued whole (diagnostics of a sharded job: icm_set_phase_timing)."""
        self._chk(self.lib.icm_set_phase_timing(self.h, 1 if on else 0))

    def phase_times(self):
        """ms per sweep of {local, exchange (+ waiting for the slowest rank), targets, solve, finish (host wait)} since
        set_phase_timing, and the number of sweeps."""
        out = np.zeros(5)
        n = C.c_int64(0)
        self._chk(self.lib.icm_get_phase_times(self.h, dptr(out), C.byref(n)))
        k = max(int(n.value), 1)
        return dict(zip(("local", "exchange", "targets", "solve", "finish_host_wait"), (out / k).round(4).tolist())), int(n.value)

    def set_fault(self, where):
        """Test hook: 1 = the next sweep_local fails like a HIP error, 2 = the next sweep_targets does -- behind the
        sweep's exchange (icm_set_fault)."""
        self._chk(self.lib.icm_set_fault(self.h, int(where)))

    def sweep_solve(self, schedule="redblack", colour=-1):
        self._chk(self.lib.icm_sweep_solve(self.h, SCHEDULES[schedule], int(colour)))

    def sweep_finish(self):
        """True = repeat the sweep's phase calls with set_optimistic(False) (a table overflowed on some rank of a sweep
        queued without a host look; nothing was replaced)."""
        rc = self.lib.icm_sweep_finish(self.h)
        if rc == _lib.RETRY_CAREFUL:
            return True
        self._chk(rc)
        return False

    def set_optimistic(self, on=True):
        """Phase calls only: queue the sweep without the host looking at phase A's flags in the middle
        (icm_set_optimistic); sweep_finish then says whether the sweep has to be repeated the careful way."""
        self._chk(self.lib.icm_set_optimistic(self.h, 1 if on else 0))

    def get_state(self):
        x = np.zeros((3, self.T))
        mo = np.zeros((2, self.L))
        co = np.zeros(self.L)
        ko = C.c_int64(0)
        self._chk(self.lib.icm_get_state(self.h, dptr(x), dptr(mo), dptr(co), C.byref(ko)))
        return x, mo, co, int(ko.value)

    # ---- sharding ------------------------------------------------------------------------
    def stats_stride(self):
        return int(self.lib.icm_stats_stride(self.h))

    def bind_exchange(self, stats_ptr, rank, world):
        self._chk(self.lib.icm_bind_exchange(self.h, C.c_void_p(stats_ptr), int(rank), int(world)))

    def bind_exchange_send(self, stats_send_ptr):
        self._chk(self.lib.icm_bind_exchange_send(self.h, C.c_void_p(stats_send_ptr)))

    def set_stream(self, stream_ptr):
        self._chk(self.lib.icm_set_stream(self.h, C.c_void_p(stream_ptr)))

    def bind_pose_buffer(self, ptr):
        self._chk(self.lib.icm_bind_pose_buffer(self.h, C.c_void_p(ptr)))

    def pose_buffer(self):
        return self.lib.icm_pose_buffer(self.h)

    # ---- collectives issued by the library (RCCL on the handle's stream) ---------------------
    @staticmethod
    def _torch_before_rccl():
        """Where a PyTorch-ROCm wheel is installed, import it BEFORE the library dlopen()s RCCL: the wheel's RCCL
        loaded ahead of the rest of the wheel tears down in the wrong order at interpreter exit (glibc reports a
        double free).  Processes without torch are not affected."""
        import sys
        if "torch" not in sys.modules and _lib._torch_lib("librccl.so"):
            import torch  # noqa: F401

    def comm_available(self):
        self._torch_before_rccl()
        return bool(self.lib.icm_comm_available())

    def comm_unique_id(self):
        """128-byte RCCL id (rank 0 makes it, every rank passes it to comm_init)."""
        self._torch_before_rccl()
        buf = (C.c_ubyte * 128)()
        self._chk(self.lib.icm_comm_unique_id(C.cast(buf, C.c_void_p)))
        return bytes(buf)

    def comm_init(self, id128, rank, world):
        self._torch_before_rccl()
        buf = (C.c_ubyte * 128).from_buffer_copy(bytes(id128))
        self._chk(self.lib.icm_comm_init(self.h, C.cast(buf, C.c_void_p), int(rank), int(world)))

    def comm_init_transport(self, rank, world, all_gather):
        """The library's sharded driver over a caller-supplied all-gather: all_gather(send_ptr, recv_ptr, count, stream_ptr)
        gathers `count` doubles of device memory from every rank (icm_comm_init_transport); exceptions become a failed
        sweep."""
        def tramp(send, recv, count, stream, user):
            try:
                all_gather(send, recv, int(count), stream)
                return 0
            except Exception:   # noqa: BLE001 (nothing may propagate through the C frames)
                import traceback
                traceback.print_exc()
                return 1
        self._transport_cb = _lib.ALLGATHER_FN(tramp)     # keep the trampoline alive as long as the handle
        self._chk(self.lib.icm_comm_init_transport(self.h, int(rank), int(world), self._transport_cb, None))

    def comm_destroy(self):
        self._chk(self.lib.icm_comm_destroy(self.h))

    def sweep_sharded(self):
        self._chk(self.lib.icm_sweep_sharded(self.h))

    def gather_poses(self):
        self._chk(self.lib.icm_gather_poses(self.h))

    def sharded_end(self):
        """The closing exchange of a sharded job driven by the library (icm_sharded_end): raises the error of a rank that
        failed behind the last sweep's exchange.  gather_poses() runs it first."""
        self._chk(self.lib.icm_sharded_end(self.h))

    # ---- inspection ----------------------------------------------------------------------
    def association(self):
        n = max(self.nnz, 1)
        lab = np.zeros(n, dtype=np.int32)
        tx, ty = np.zeros(n), np.zeros(n)
        self._chk(self.lib.icm_get_association(self.h, iptr(lab), dptr(tx), dptr(ty)))
        return lab[:self.nnz], tx[:self.nnz], ty[:self.nnz]

    def raw_map(self):
        y = np.zeros((2, self.L))
        c = np.zeros(self.L)
        la = C.c_int64(0)
        self._chk(self.lib.icm_get_raw_map(self.h, dptr(y), dptr(c), C.byref(la)))
        return y, c, int(la.value)

    def _one(self, energy, two_sided, x, x_ant, x_pos, u, odo, bx, by, tx, ty):
        odo = _f64(odo)
        u = _f64(np.asarray(u, dtype=np.float64).reshape(2, -1))
        bx, by, tx, ty = _f64(bx), _f64(by), _f64(tx), _f64(ty)
        xa = _f64(np.asarray(x_ant, dtype=np.float64).reshape(3))
        xp = _f64(np.asarray(x_pos, dtype=np.float64).reshape(3)) if x_pos is not None else None
        out = np.zeros(6)
        if energy:
            xv = _f64(np.asarray(x, dtype=np.float64).reshape(3))
            self._chk(self.lib.icm_energy_one(self.h, int(two_sided), dptr(xv), dptr(xa), dptr(xp), dptr(u), dptr(odo),
                                              odo.shape[1], dptr(bx), dptr(by), dptr(tx), dptr(ty), len(bx), dptr(out)))
            return float(out[0])
        self._chk(self.lib.icm_solve_one(self.h, int(two_sided), dptr(xa), dptr(xp), dptr(u), dptr(odo), odo.shape[1],
                                         dptr(bx), dptr(by), dptr(tx), dptr(ty), len(bx), dptr(out)))
        return out

    def solve_one(self, two_sided, x_ant, x_pos, u, odo, bx, by, tx, ty):
        """minimizar_xn (two_sided) / minimizar_x on the GPU -> [x, y, theta, f, nit, nfev]."""
        return self._one(False, two_sided, None, x_ant, x_pos, u, odo, bx, by, tx, ty)

    def energy_one(self, two_sided, x, x_ant, x_pos, u, odo, bx, by, tx, ty):
        """fun_xn (two_sided) / fun_x at pose x, evaluated on the GPU."""
        return self._one(True, two_sided, x, x_ant, x_pos, u, odo, bx, by, tx, ty)

    def set_debug(self, on=True):
        """Keep per-beam labels / targets of the next sweeps (for `association()`)."""
        self._chk(self.lib.icm_set_debug(self.h, int(bool(on))))

    def solve_diag(self):
        """(T,3) [final energy, NM iterations, function evaluations] per pose (debug on)."""
        out = np.zeros((self.T, 3))
        self._chk(self.lib.icm_get_solve_diag(self.h, dptr(out)))
        return out

    def set_energy_form(self, form):
        """0 / 'moments' (default), 1 / 'beam' (literal per-beam sum), 2 / 'entry'."""
        form = {"moments": 0, "beam": 1, "entry": 2}.get(form, form)
        self._chk(self.lib.icm_set_energy_form(self.h, int(form)))

    def set_solve_lanes(self, mode):
        """-1 automatic, 0 one lane per pose, 1 one quad per pose (latency form)."""
        self._chk(self.lib.icm_set_solve_lanes(self.h, int(mode)))

    def snapshot_state(self):
        """Keep a device-side copy of the current sweep state (poses, map, search structures)."""
        self._chk(self.lib.icm_snapshot_state(self.h))

    def restore_state(self):
        """Back to the snapshot (device-to-device copies, stream-ordered)."""
        self._chk(self.lib.icm_restore_state(self.h))

    def set_colour_fusion(self, on):
        """True (default): both colours of an unsharded red-black sweep in one launch."""
        self._chk(self.lib.icm_set_colour_fusion(self.h, int(bool(on))))

    def set_fold_mode(self, mode):
        """-1 automatic, 1 the one-launch solve evaluates the folded energy only and leaves poses outside its range to
        a second solve by the same wave, 0 the complete energy in the loop itself (icm_set_fold_mode)."""
        self._chk(self.lib.icm_set_fold_mode(self.h, int(mode)))

    def fixup_poses(self):
        """Poses solved a second time with the complete energy (an evaluation left the folded form's range), so far."""
        n = C.c_int64(0)
        self._chk(self.lib.icm_get_fixup_poses(self.h, C.byref(n)))
        return int(n.value)

    def dropin_counts(self):
        """icm_sweep calls on a registered pose array: (started without an upload, of those: started over because the
        caller had changed the array, poses written into the caller's array by the solves themselves)."""
        out = (C.c_int64 * 3)()
        self._chk(self.lib.icm_get_dropin_counts(self.h, out))
        return tuple(int(v) for v in out)

    def set_fused_spin_limit(self, polls):
        """Polls an even wave of the one-launch solve waits before deferring to the launch's last wave."""
        self._chk(self.lib.icm_set_fused_spin_limit(self.h, int(polls)))

    def fused_deferred(self):
        """Even waves that deferred so far (normally 0)."""
        n = C.c_int64(0)
        self._chk(self.lib.icm_get_fused_deferred(self.h, C.byref(n)))
        return int(n.value)

    def set_entry_path(self, mode):
        """-1 / 1 / 'hier': hierarchical running sums (default); 0 / 'sort': the sort-based pipeline."""
        mode = {"hier": 1, "sort": 0, "auto": -1}.get(mode, mode)
        self._chk(self.lib.icm_set_entry_path(self.h, int(mode)))

    def entry_path(self):
        """Pipeline the last sweep ran: 'hier' or 'sort'."""
        return "hier" if self.lib.icm_get_entry_path(self.h) == 1 else "sort"

    def set_gpu_filtrar(self, on):
        self._chk(self.lib.icm_set_gpu_filtrar(self.h, int(bool(on))))

    def last_filtrar_info(self):
        """(landmarks after the last sweep's Mapa.filtrar, where it ran: 0 GPU / 1 GPU with merges /
        2 host routine, landmarks that had a neighbour closer than dist_thr)."""
        a = np.zeros(3, dtype=np.int64)
        self._chk(self.lib.icm_last_filtrar_info(self.h, lptr(a)))
        return int(a[0]), int(a[1]), int(a[2])

    def filtrar_device(self, y, counts, lact):
        """Mapa.filtrar on the GPU for a caller-held (2,L) map -> (y_out (2,L), counts_out (L), lact, path)."""
        y, counts = _f64(y), _f64(counts)
        if y.shape != (2, self.L) or counts.shape != (self.L,):
            raise ValueError("filtrar: map must be (2,L) and counts (L,)")
        yo, co = np.zeros((2, self.L)), np.zeros(self.L)
        lo, path = C.c_int64(0), C.c_int(0)
        self._chk(self.lib.icm_filtrar_device(self.h, dptr(y), dptr(counts), int(lact), dptr(yo), dptr(co), C.byref(lo), C.byref(path)))
        return yo, co, int(lo.value), int(path.value)

    def set_brute_force(self, on):
        self._chk(self.lib.icm_set_brute_force(self.h, int(bool(on))))

    def wait_giveups(self):
        """Sweeps whose side-stream wait for the raw map gave up (serialised streams): 0 in a normal run."""
        n = C.c_int64(0)
        self._chk(self.lib.icm_get_wait_giveups(self.h, C.byref(n)))
        return int(n.value)

    def set_assoc_form(self, form):
        """What phase A associates: 1 / 'runs' (default) geometric runs of each scan's kept beams, settled by the
        bounding-circle test, beam by beam where it does not settle; 0 / 'beams' every kept beam on its own."""
        form = {"runs": 1, "beams": 0}.get(form, form)
        self._chk(self.lib.icm_set_assoc_form(self.h, int(form)))

    def runs(self):
        """The geometric runs of the shard's kept beams: (offsets by pose (nloc+1), centre (n,2), sum of body points (n,2),
        radius (n) float32, beams (n), first beam's offset among the pose's kept beams (n))."""
        n = self.run_counts()[0]
        off = np.zeros(self.nloc + 1, dtype=np.int64)
        c, sb = np.zeros((max(n, 1), 2)), np.zeros((max(n, 1), 2))
        r = np.zeros(max(n, 1), dtype=np.float32)
        k, f = np.zeros(max(n, 1), dtype=np.int32), np.zeros(max(n, 1), dtype=np.int32)
        self._chk(self.lib.icm_get_runs(self.h, lptr(off), dptr(c), dptr(sb), r.ctypes.data_as(C.POINTER(C.c_float)), iptr(k), iptr(f)))
        return off, c[:n], sb[:n], r[:n], k[:n], f[:n]

    def run_counts(self):
        """(runs of the uploaded shard, runs that went beam by beam so far)."""
        a = np.zeros(2, dtype=np.int64)
        self._chk(self.lib.icm_get_run_counts(self.h, lptr(a)))
        return int(a[0]), int(a[1])

    def enable_timing(self, on=True):
        self._chk(self.lib.icm_enable_timing(self.h, int(bool(on))))
        self._chk(self.lib.icm_reset_timing(self.h))

    def kernel_times(self):
        out = {}
        for i in range(self.lib.icm_kernel_count(self.h)):
            name = C.c_char_p()
            ms = C.c_double(0)
            n = C.c_int64(0)
            self._chk(self.lib.icm_kernel_time(self.h, i, C.byref(name), C.byref(ms), C.byref(n)))
            out[name.value.decode()] = (ms.value, n.value)
        return out

    def flop_per_eval(self):
        """FP64 flops of one energy evaluation (fma = 2), counted from the built kernel's ISA."""
        return int(self.lib.icm_flop_per_eval())

    def last_stats(self):
        a = np.zeros(4, dtype=np.int64)
        self._chk(self.lib.icm_last_stats(self.h, lptr(a)))
        return dict(kept_beams=int(a[0]), entries=int(a[1]), new_landmarks=int(a[2]), labels=int(a[3]))
