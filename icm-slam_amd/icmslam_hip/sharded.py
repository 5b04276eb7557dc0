"""Pose-sharded ICM sweep: one process per GPU, torch.distributed (backend "nccl" = RCCL over
xGMI) for the two exchanges a sweep needs.

Partition: contiguous pose blocks of `blk = ceil(T / world)` poses; rank r owns poses
[r*blk, min((r+1)*blk, T)) -- every rank must own at least one pose (icm_upload refuses an empty
shard).  Scans live only on their owner; odometry, velocities, the pose
array and the landmark table are replicated (they are KB..MB).

Per sweep:
  1. local phase A and per-landmark sufficient statistics (sum x, sum y, n)  [no comm]
  2. ONE all-gather of the [3L+8] statistics: every rank gets the total (new map) and the
     exclusive prefix over lower ranks (state of each running mean at its first pose)
  3. targets, then the odd poses of the shard                                 [no comm]
  4. halo exchange: a solve reads only the poses t-1 and t+1, so a shard needs one pose from
     each neighbour -- every rank contributes its first and last pose (48 B) to one tiny
     all-gather and copies the two it needs next to its block
  5. the even poses; their boundary values travel in the header of the NEXT sweep's statistics
     message (step 2), so a sweep costs two collectives
  6. Mapa.filtrar, replicated (deterministic) on every rank

The pose blocks themselves are gathered only when the caller asks for the state
(`get_state`), not per sweep.  The payloads are tiny (240 KB of statistics per rank at
L = 10k, 48 B of halo), so the collectives are latency-bound; xGMI link bandwidth is
irrelevant here.
"""
import numpy as np


def partition(T, world):
    blk = (T + world - 1) // world
    return blk, [(min(r * blk, T), min((r + 1) * blk, T)) for r in range(world)]


class ShardedSweep:
    """Drives one rank.  `engine` is a SweepEngine (or anything with the same phase API);
    `comm` does the collectives: TorchComm (torch.distributed) or NoComm (ranks that share
    the exchange buffers inside one process)."""

    def __init__(self, engine, rank, world, T, comm=None, buffers=None):
        import torch
        self.torch = torch
        self.eng, self.rank, self.world, self.T = engine, rank, world, T
        self.blk, self.parts = partition(T, world)
        self.stride = engine.stats_stride()
        dev = getattr(engine, "exchange_device", "cuda")
        if buffers is None:
            stats = torch.zeros(world * self.stride, dtype=torch.float64, device=dev)
            poses = torch.zeros(world * self.blk * 3, dtype=torch.float64, device=dev)
        else:
            stats, poses = buffers
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
                # real collectives: the library writes the send-side buffers (its statistics; its
                # first and last pose after every half sweep) and unpacks the neighbours' poses,
                # so that each exchange is one collective call and one C call
                self.stats_send = torch.zeros(self.stride, dtype=torch.float64, device=dev)
                self.halo_send = torch.zeros(6, dtype=torch.float64, device=dev)
                self.halo_recv = torch.zeros(world * 6, dtype=torch.float64, device=dev)
                engine.bind_exchange_send(self.stats_send.data_ptr(), self.halo_send.data_ptr(), self.halo_recv.data_ptr())
                self.native = True
        # halo bookkeeping: rows (first pose, last pose) of every rank; which of them this rank
        # needs (the last pose of the rank below, the first pose of the rank above; only trailing
        # ranks can be empty, and an empty rank needs nothing)
        a, b = self.parts[rank]
        self.own = (a, b)
        P = self.poses.view(-1, 3)
        idev = P.device
        self.halo_all = torch.zeros(world * 2, 3, dtype=torch.float64, device=idev)
        last = max(b - 1, a) if b > a else 0
        self.edge_idx = torch.tensor([a if b > a else 0, last], dtype=torch.long, device=idev)
        src, dst = [], []
        if b > a and a > 0:
            src.append(2 * (rank - 1) + 1)
            dst.append(a - 1)
        if b > a and b < T:
            src.append(2 * (rank + 1))
            dst.append(b)
        self.halo_src = torch.tensor(src, dtype=torch.long, device=idev)
        self.halo_dst = torch.tensor(dst, dtype=torch.long, device=idev)

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
        can = hasattr(e, "set_optimistic") and not isinstance(self.comm, NoComm)   # (shared buffers: see run_virtual_ranks)
        for attempt in (0, 1):
            if can:
                e.set_optimistic(attempt == 0)
            e.sweep_local()
            self.comm.gather_stats(self)
            e.sweep_targets()
            if self.world == 1:        # no neighbour, no halo: both colours in the one-launch solve
                e.sweep_solve("redblack", -1)
            else:
                e.sweep_solve("redblack", 1)
                self.comm.halo(self)
                e.sweep_solve("redblack", 0)
                if not self.native:      # (native: the even poses' boundary values ride in the next statistics message)
                    self.comm.halo(self)
            if not e.sweep_finish():
                break
        if can:
            e.set_optimistic(False)

    def get_state(self):
        """(x, map, counts, K) of the whole sequence: gathers the pose blocks first."""
        self.comm.all_gather(self.poses, self.rank, self.blk * 3)
        return self.eng.get_state()


class LibrarySweep:
    """One rank of a pose-sharded sweep whose collectives the C library issues itself (RCCL
    `ncclAllGather` on the handle's stream: `icm_comm_init` / `icm_sweep_sharded`, include/icmslam.h) --
    a sweep is ONE C call, nothing of it runs through Python or torch.distributed.  The 128-byte
    communicator id travels once, by `bcast(bytes_or_None) -> bytes` (rank 0 passes the id, the others
    None); default: a broadcast on the default torch.distributed group, whatever its backend."""

    def __init__(self, engine, rank, world, T, bcast=None):
        self.eng, self.rank, self.world, self.T = engine, rank, world, T
        self.blk, self.parts = partition(T, world)
        self.own = self.parts[rank]
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

    def halo(self, sw):
        if sw.native:   # packed by icm_sweep_solve, unpacked by icm_halo_unpack
            self.dist.all_gather_into_tensor(sw.halo_recv, sw.halo_send, group=self.group)
            sw.eng.halo_unpack()
            return
        P = sw.poses.view(-1, 3)
        edges = P.index_select(0, sw.edge_idx)                       # (2,3): my first and last pose
        self.dist.all_gather_into_tensor(sw.halo_all, edges, group=self.group)
        if sw.halo_dst.numel():
            P.index_copy_(0, sw.halo_dst, sw.halo_all.index_select(0, sw.halo_src))


class NoComm:
    """Ranks living in one process on one GPU and bound to the SAME buffers: every rank's
    slot is already visible to the others."""

    def all_gather(self, buf, rank, count):
        pass

    def gather_stats(self, sw):
        pass

    def halo(self, sw):
        pass


def run_virtual_ranks(engines, sweeps, schedule="redblack"):
    """Lock-step execution of several in-process ranks (one GPU, shared buffers): the exact
    phase order of ShardedSweep.sweep with the collectives replaced by shared memory."""
    # (always the careful form: ranks that share buffers inside one process are ordered by the host's look at phase A
    # in the middle of the sweep -- a sweep queued whole relies on the collective for that)
    for _ in range(sweeps):
        for e in engines:
            e.sweep_local()
        for e in engines:
            e.sweep_targets()
        for colour in (1, 0):
            for e in engines:
                e.sweep_solve(schedule, colour)
        for e in engines:
            e.sweep_finish()
