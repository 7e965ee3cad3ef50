"""Batch sharding over the GPUs of one node (one process per GPU).

The sweep shards over the training batch: every series is an independent contributor to
the bond gradient (src/Training/loss_functions.jl:353-369 is a plain sum over series), so each
rank keeps N/G series (class-sorted locally), their encodings and their LE/RE rows; the only
exchange is one RCCL all-reduce of the (C x (d chi)^2 + loss) buffer per optimiser step, issued
inside libmpstime_hip.so on the engine's stream.  This module only splits the data and
distributes the ncclUniqueId over the host-side process group (torch.distributed).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib as L
from .encodings import EncodedTimeSeriesSet


def shard_bounds(count: int, rank: int, world: int):
    """Contiguous, near-equal split of ``count`` series of one class."""
    return (count * rank) // world, (count * (rank + 1)) // world


def split_encoded(ets: EncodedTimeSeriesSet, rank: int, world: int):
    """Local part of a class-sorted set: every class is split across ranks so each shard stays
    class-sorted and class-balanced.  Returns (local set, global per-class counts)."""
    if len(ets) == 0:
        return ets, np.zeros(0, dtype=np.int64)
    counts = np.asarray(ets.class_distribution, dtype=np.int64)
    # every rank must hold at least one series: a rank with an empty shard would return from the engine before the
    # collective and leave the others waiting.  The split is a pure function of (counts, world), so every rank reaches
    # the same verdict and raises together.
    for r in range(world):
        if sum(shard_bounds(int(n), r, world)[1] - shard_bounds(int(n), r, world)[0] for n in counts) == 0:
            raise ValueError(f"cannot shard {int(counts.sum())} series with class counts {counts.tolist()} over {world} ranks: "
                             f"rank {r} would hold no series")
    starts = np.concatenate([[0], np.cumsum(counts)])
    idx = []
    for c, n in enumerate(counts):
        lo, hi = shard_bounds(int(n), rank, world)
        idx.append(np.arange(starts[c] + lo, starts[c] + hi))
    idx = np.concatenate(idx) if idx else np.zeros(0, dtype=np.int64)
    local_counts = np.array([shard_bounds(int(n), rank, world)[1] - shard_bounds(int(n), rank, world)[0] for n in counts],
                            dtype=np.int64)
    local = EncodedTimeSeriesSet(ets.phi[idx], ets.labels[idx], ets.label_index[idx],
                                 ets.original_data[idx] if ets.original_data.size else ets.original_data, local_counts)
    return local, counts


class Shard:
    """rank/world + the torch.distributed process group used to hand out the ncclUniqueId (``attach``) and to gather
    the inbox handles of the one-shot all-reduce (``attach_oneshot``).  ``rccl=False`` skips the RCCL communicator:
    the one-shot path alone then carries every sum over ranks."""

    def __init__(self, rank: int, world: int, group=None, rccl: bool = True, oneshot: bool = False):
        self.rank, self.world, self.group = rank, world, group
        self.rccl, self.oneshot = rccl, oneshot

    def split(self, ets):
        return split_encoded(ets, self.rank, self.world)

    def attach(self, eng):
        """Create the RCCL communicator inside the engine (mpst_comm_init)."""
        if self.world == 1 or not self.rccl:
            return
        import torch
        import torch.distributed as dist
        lib = L.load()
        uid = (C.c_uint8 * 128)()
        if self.rank == 0:
            rc = lib.mpst_comm_unique_id(uid)
            if rc:
                raise L.MPSTError(rc, (lib.mpst_last_error(None) or b"").decode())
        dev = "cuda" if dist.get_backend(self.group) == "nccl" else "cpu"
        t = torch.tensor(list(uid), dtype=torch.uint8, device=dev)
        dist.broadcast(t, src=0, group=self.group)
        uid = (C.c_uint8 * 128)(*t.cpu().tolist())
        eng._chk(lib.mpst_comm_init(eng.ctx, uid, self.world, self.rank))

    def attach_oneshot(self, eng):
        """Export this rank's inbox, gather every rank's handle over the host-side process group and map the peers
        (mpst_comm_ipc_export / mpst_comm_ipc_attach).  Call after set_options / set_dataset / set_mps."""
        if self.world == 1:
            return
        import torch
        import torch.distributed as dist
        lib = L.load()
        h = (C.c_uint8 * 64)()
        eng._chk(lib.mpst_comm_ipc_export(eng.ctx, self.world, self.rank, h))
        dev = "cuda" if dist.get_backend(self.group) == "nccl" else "cpu"
        mine = torch.tensor(list(h), dtype=torch.uint8, device=dev)
        allh = [torch.zeros(64, dtype=torch.uint8, device=dev) for _ in range(self.world)]
        dist.all_gather(allh, mine, group=self.group)
        flat = (C.c_uint8 * (64 * self.world))(*[int(x) for t in allh for x in t.cpu().tolist()])
        eng._chk(lib.mpst_comm_ipc_attach(eng.ctx, flat))
        dist.barrier(group=self.group)       # nobody pushes before every rank has mapped every inbox

    def select(self, eng, oneshot: bool):
        eng._chk(L.load().mpst_comm_select(eng.ctx, int(bool(oneshot))))
