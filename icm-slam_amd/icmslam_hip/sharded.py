"""Pose-sharded ICM sweep: one process per GPU, ONE collective per sweep (torch.distributed, backend
"nccl" = RCCL over xGMI; or the C library's own RCCL call, `LibrarySweep`).

Partition: contiguous pose blocks of `blk` = ceil(T / world) rounded up to an EVEN number of poses
(`icm_shard_block`); rank r owns [r*blk, min((r+1)*blk, T)) -- every rank must own at least one pose
(icm_upload refuses an empty shard).  Scans live only on their owner, plus ONE extra scan per rank > 0:
that of the pose in front of its block, the rank's *ghost pose*.  Odometry, velocities, the pose array
and the landmark table are replicated (they are KB..MB).

Per sweep (the loop being sharded is scripts/ICM_ROS.py:141-158):
  1. local phase A and per-landmark sufficient statistics (sum x, sum y, n)  [no comm]
  2. ONE all-gather of the [3L+16] message of every rank: the statistics give every rank the total
     (new map) and the exclusive prefix over lower ranks (state of each running mean at its first
     pose); the 16-double header carries the rank's new-landmark count, its overflow flags and its
     first / last / last-but-one pose as the PREVIOUS sweep left them -- the old values the
     neighbours' boundary solves read
  3. targets; the ghost pose's entries and moments                            [no comm]
  4. both colours of the shard in ONE launch.  A red-black sweep solves the odd poses from old even
     neighbours, then the even poses from new odd neighbours; shards start at even poses, so the only
     value a shard would need from another rank in mid-sweep is the new value of the odd pose in front
     of it -- and that one it computes itself (the ghost pose), from the same beams and neighbour values
     as its owner and from targets that equal the owner's up to the rounding of differently associated
     sums (~1e-16 relative: include/icmslam.h, icm_upload_ghost_scan).  No halo exchange.
  5. Mapa.filtrar, replicated (deterministic) on every rank

The pose blocks themselves are gathered only when the caller asks for the state (`get_state`), not
per sweep.  The payload is tiny (240 KB per rank at L = 10k), so the collective is latency-bound;
xGMI link bandwidth is irrelevant here.
"""
import numpy as np


def shard_block(T, world):
    """Poses per rank: ceil(T / world) rounded up to an even number (include/icmslam.h icm_shard_block)."""
    blk = (T + world - 1) // world
    return blk + (blk & 1)


def world_fits(T, world):
    """Every one of `world` even-sized blocks holds at least one pose."""
    return world == 1 or (world - 1) * shard_block(T, world) < T


def usable_world(T, world):
    """The largest world size <= `world` that world_fits (not every smaller one does: T = 21 splits 6 or 11 ways, not 7..10)."""
    while world > 1 and not world_fits(T, world):
        world -= 1
    return world


def partition(T, world):
    """(block, [(a_r, b_r)]) -- blocks of shard_block(T, world) poses.  Rounding the block up to an even number can leave
    trailing ranks without poses (T = 21, world = 7: blocks of 4 cover the sequence with six ranks): refused here, with
    the largest usable world size, rather than failing later in icm_upload."""
    blk = shard_block(T, world)
    if not world_fits(T, world):
        raise ValueError("a %d-pose sequence cannot be split into %d blocks of %d poses (every shard starts at an even pose "
                         "and holds at least one): use %d ranks" % (T, world, blk, usable_world(T, world)))
    return blk, [(min(r * blk, T), min((r + 1) * blk, T)) for r in range(world)]


def ghost_scan(scans_pose_major_full, a):
    """The scan a rank whose block starts at pose `a` uploads beside its own: pose a - 1 (None for rank 0)."""
    return None if a < 2 else scans_pose_major_full[a - 1]


class ShardedSweep:
    """Drives one rank.  `engine` is a SweepEngine (or anything with the same phase API);
    `comm` does the collective: TorchComm (torch.distributed) or NoComm (ranks that share
    the statistics buffer inside one process; their pose arrays stay private, like real replicas)."""

    def __init__(self, engine, rank, world, T, comm=None, stats=None):
        import torch
        self.torch = torch
        self.eng, self.rank, self.world, self.T = engine, rank, world, T
        self.blk, self.parts = partition(T, world)
        self.stride = engine.stats_stride()
        dev = getattr(engine, "exchange_device", "cuda")
        if stats is None:
            stats = torch.zeros(world * self.stride, dtype=torch.float64, device=dev)
        poses = torch.zeros(world * self.blk * 3, dtype=torch.float64, device=dev)
        self.stats, self.poses = stats, poses
        self.comm = comm if comm is not None else TorchComm()
        self.native = False
        if hasattr(engine, "bind_tensors"):      # test doubles work on the tensors directly
            engine.bind_tensors(stats, poses, rank, world)
        else:
            if dev == "cuda":
                engine.set_stream(torch.cuda.current_stream().cuda_stream)
            engine.bind_exchange(stats.data_ptr(), rank, world)
            engine.bind_pose_buffer(poses.data_ptr())
            if not isinstance(self.comm, NoComm):
                # a real collective: the library writes its message into a send buffer, so that the exchange
                # is one collective call and nothing else (all-gather input and output must not alias)
                self.stats_send = torch.zeros(self.stride, dtype=torch.float64, device=dev)
                engine.bind_exchange_send(self.stats_send.data_ptr())
                self.native = True
        self.own = self.parts[rank]

    def set_state(self, mapa_viejo, x, x0, lact=None):
        self.eng.set_state(mapa_viejo, x, x0, lact)

    def sweep(self, schedule="redblack"):
        if schedule != "redblack":
            raise NotImplementedError("only the red-black schedule shards (the reference order is one chain)")
        e = self.eng
        # Queued whole first (no host look at phase A's flags in the middle of the sweep: on the short shards of a
        # strong-scaled job that round trip is a quarter of the sweep).  Every rank's flags travel in the header of its
        # statistics, so if ANY rank's tables overflowed all ranks leave their state alone, all see it at the end, and
        # all repeat the sweep the careful way.
        can = callable(getattr(e, "set_optimistic", None)) and not isinstance(self.comm, NoComm)   # (shared buffers: see run_virtual_ranks)
        for attempt in (0, 1):
            if can:
                e.set_optimistic(attempt == 0)
            # Failing together: a rank whose phase A fails on its own (labels beyond L, a no-beam last pose, a table too
            # small at its largest size, a device error) still sends its message, with the error code in the header, and
            # every rank raises after the exchange.
            err = None
            try:
                e.sweep_local()
            except (IndexError, RuntimeError, ValueError) as ex:
                if not hasattr(e, "mark_failed") or getattr(e, "last_rc", 0) >= 0:
                    raise
                err = ex
                e.mark_failed(e.last_rc)
            self.comm.gather_stats(self)            # the sweep's one collective
            status = getattr(e, "exchange_status", None)
            whole = can and attempt == 0 and err is None and (e.sweep_is_optimistic() if hasattr(e, "sweep_is_optimistic") else True)
            if status is not None and not whole:
                # this rank's sweep was not queued whole: it looks at every header now.  A rank that reports flags of a
                # sweep IT had queued whole will repeat the sweep with everybody of its kind -- so must this one.
                fr, code, retry = status()
                if err is not None:
                    raise err
                if fr >= 0:
                    from .engine import _raise
                    _raise(code, "sharded sweep: rank %d failed in phase A" % fr)
                if retry and attempt == 0:
                    continue
            elif err is not None:
                raise err
            try:
                e.sweep_targets()
                e.sweep_solve("redblack", -1)       # both colours, ghost pose included, one launch
                again = e.sweep_finish()
            except (IndexError, RuntimeError, ValueError):
                # Failing together BEHIND the exchange: the peers' phases ran and they are on their way to the next exchange
                # (the next sweep's, or the closing one of end() / get_state()).  Meet them there with the code in the
                # header -- a farewell message -- so that they raise too instead of waiting for a rank that has left.
                if hasattr(e, "mark_failed") and getattr(e, "last_rc", 0) < 0 and not isinstance(self.comm, NoComm):
                    try:
                        e.mark_failed(e.last_rc)
                        self.comm.gather_stats(self)
                    except Exception:
                        pass
                raise
            if not again:
                break
            if status is not None:
                # before the repeated sweep's collective: a rank that FAILED in this one has left with its error and will
                # not join another (its code reads as "flags set" to the ranks that queued the sweep whole)
                fr, code, _ = status()
                if fr >= 0:
                    from .engine import _raise
                    _raise(code, "sharded sweep: rank %d failed in phase A" % fr)
        if can:
            e.set_optimistic(False)

    def end(self):
        """The closing exchange: every rank calls it once after its last sweep (get_state does).  One more all-gather of
        the statistics message with a clean header; a rank that failed behind the last sweep's exchange delivers its
        error here, and every rank raises it."""
        e = self.eng
        if isinstance(self.comm, NoComm) or not hasattr(e, "mark_failed") or not hasattr(e, "exchange_status"):
            return
        e.mark_failed(0)
        self.comm.gather_stats(self)
        fr, code, _ = e.exchange_status()
        if fr >= 0:
            from .engine import _raise
            _raise(code, "sharded job: rank %d failed behind the last sweep's exchange" % fr)

    def get_state(self):
        """(x, map, counts, K) of the whole sequence: the closing exchange, then the pose blocks are gathered."""
        self.end()
        self.comm.all_gather(self.poses, self.rank, self.blk * 3)
        return self.eng.get_state()


class LibrarySweep:
    """One rank of a pose-sharded sweep whose collective the C library issues itself (RCCL
    `ncclAllGather` on the handle's stream: `icm_comm_init` / `icm_sweep_sharded`, include/icmslam.h) --
    a sweep is ONE C call, nothing of it runs through Python or torch.distributed.  The 128-byte
    communicator id travels once, by `bcast(bytes_or_None) -> bytes` (rank 0 passes the id, the others
    None); default: a broadcast on the default torch.distributed group, whatever its backend.
    `transport(send_ptr, recv_ptr, count, stream_ptr)`: carry the all-gathers yourself instead of RCCL
    (`icm_comm_init_transport`: MPI, a test harness hopping through host memory, ...)."""

    def __init__(self, engine, rank, world, T, bcast=None, transport=None):
        self.eng, self.rank, self.world, self.T = engine, rank, world, T
        self.blk, self.parts = partition(T, world)
        self.own = self.parts[rank]
        if transport is not None:
            engine.comm_init_transport(rank, world, transport)
            return
        if bcast is None:
            bcast = _torch_bcast
        uid = bcast(engine.comm_unique_id() if rank == 0 else None)
        engine.comm_init(uid, rank, world)

    def set_state(self, mapa_viejo, x, x0, lact=None):
        self.eng.set_state(mapa_viejo, x, x0, lact)

    def sweep(self, schedule="redblack"):
        if schedule != "redblack":
            raise NotImplementedError("only the red-black schedule shards (the reference order is one chain)")
        self.eng.sweep_sharded()

    def end(self):
        """The closing exchange (icm_sharded_end); get_state() runs it itself."""
        self.eng.sharded_end()

    def get_state(self):
        self.eng.gather_poses()
        return self.eng.get_state()

    def close(self):
        self.eng.comm_destroy()


def _torch_bcast(payload):
    import torch
    import torch.distributed as dist
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.zeros(128, dtype=torch.uint8, device=dev)
    if payload is not None:
        t.copy_(torch.frombuffer(bytearray(payload), dtype=torch.uint8))
    dist.broadcast(t, src=0)
    return bytes(t.cpu().numpy().tobytes())


class TorchComm:
    """all_gather_into_tensor on the default process group (NCCL/RCCL on GPUs, gloo on CPU)."""

    def __init__(self, group=None):
        import torch.distributed as dist
        self.dist, self.group = dist, group

    def all_gather(self, buf, rank, count):
        mine = buf[rank * count:(rank + 1) * count].clone()
        self.dist.all_gather_into_tensor(buf, mine, group=self.group)

    def gather_stats(self, sw):
        if sw.native:
            self.dist.all_gather_into_tensor(sw.stats, sw.stats_send, group=self.group)
        else:
            self.all_gather(sw.stats, sw.rank, sw.stride)


class NoComm:
    """Ranks living in one process on one GPU and bound to the SAME statistics buffer: every rank's
    message is already visible to the others (run_virtual_ranks orders the phases)."""

    def all_gather(self, buf, rank, count):
        pass

    def gather_stats(self, sw):
        pass


def run_virtual_ranks(runs, sweeps, schedule="redblack"):
    """Lock-step execution of several in-process ranks (`runs`: their ShardedSweep objects, built with
    comm=NoComm() and the same `stats` tensor): the exact phase order of ShardedSweep.sweep with the
    collective replaced by shared memory.  Each rank keeps its own pose array, like a real replica;
    at the end every rank's block is copied into every other rank's array (the pose all-gather of
    get_state)."""
    # (always the careful form: ranks that share a buffer inside one process are ordered by the host's look at phase A
    # in the middle of the sweep -- a sweep queued whole relies on the collective for that)
    engines = [r.eng for r in runs]
    torch = runs[0].torch
    for _ in range(sweeps):
        for e in engines:
            e.sweep_local()
        torch.cuda.synchronize()       # every rank's message is in the shared buffer
        for e in engines:
            e.sweep_targets()
        for e in engines:
            e.sweep_solve(schedule, -1)
        for e in engines:
            e.sweep_finish()
    torch.cuda.synchronize()
    n = runs[0].blk * 3
    for src in runs:
        blk = src.poses[src.rank * n:(src.rank + 1) * n]
        for dst in runs:
            if dst is not src:
                dst.poses[src.rank * n:(src.rank + 1) * n].copy_(blk)
    torch.cuda.synchronize()
