"""Multi-GPU layer: one process per GPU, patches/slices sharded, weights broadcast once.

The forward pass has no exchange step -- every patch is independent given the (replicated, ~4 MB)
weights -- so the only collective is one broadcast of the weight blob from the source rank at load
time (RCCL over xGMI when the process group is "nccl"; "gloo" in the CPU tests).  Outputs stay on
the GPU that produced them; ``gather_outputs`` exists for callers that want them on one rank.

The reference is single-process; this is the scale-out the north star asks for.
"""

from __future__ import annotations

import ctypes as C
import os

import numpy as np


def shard_range(n_items: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous block partition: the first ``n % world`` ranks get one extra item."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError(f"bad rank/world {rank}/{world}")
    base, extra = divmod(int(n_items), world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def _layout(sd: dict):
    keys = sorted(sd.keys())
    return [(k, tuple(sd[k].shape)) for k in keys]


def broadcast_state_dict(sd: dict | None, src: int = 0, device=None, group=None) -> dict:
    """Broadcast a ``{key: float32 ndarray}`` state_dict from ``src`` to every rank.

    Metadata (keys, shapes) travels as a Python object; the payload as ONE flat float32 tensor --
    a single collective of ~4 MB rather than one per tensor (xGMI broadcast is latency-bound at
    this size).  ``device``: torch device of the staging tensor (a CUDA device for nccl/RCCL,
    None/cpu for gloo).
    """
    import torch
    import torch.distributed as dist

    rank = dist.get_rank(group)
    meta = [_layout(sd)] if rank == src else [None]
    dist.broadcast_object_list(meta, src=src, group=group)
    layout = meta[0]
    total = int(sum(int(np.prod(s, dtype=np.int64)) for _, s in layout))
    dev = device if device is not None else torch.device("cpu")
    if rank == src:
        flat = np.concatenate([np.ascontiguousarray(sd[k], dtype=np.float32).reshape(-1) for k, _ in layout])
        buf = torch.from_numpy(flat).to(dev)
    else:
        buf = torch.empty(total, dtype=torch.float32, device=dev)
    dist.broadcast(buf, src=src, group=group)
    host = buf.cpu().numpy()
    out, off = {}, 0
    for k, shp in layout:
        n = int(np.prod(shp, dtype=np.int64))
        out[k] = host[off:off + n].reshape(shp).copy()
        off += n
    return out


def gather_outputs(local: np.ndarray, n_total: int, dst: int = 0, group=None):
    """Collect per-rank output blocks (block partition of axis 0, see :func:`shard_range`) on ``dst``."""
    import torch.distributed as dist

    world, rank = dist.get_world_size(group), dist.get_rank(group)
    parts = [None] * world if rank == dst else None
    dist.gather_object(np.ascontiguousarray(local), parts, dst=dst, group=group)
    if rank != dst:
        return None
    out = np.concatenate(parts, axis=0)
    assert out.shape[0] == n_total, (out.shape, n_total)
    return out


def sharded_forward(model, tiles: np.ndarray, group=None, gather: bool = True):
    """Evaluate ``model`` on this rank's contiguous block of ``tiles`` (B, O, O).

    Returns the full (B, S, S) array on rank 0 (None elsewhere) when ``gather`` is set, otherwise
    the local block and its (lo, hi) range.  ``model`` is any callable tiles -> outputs, so the CPU
    tests can exercise the partition/gather logic with a stand-in.
    """
    import torch.distributed as dist

    world, rank = dist.get_world_size(group), dist.get_rank(group)
    lo, hi = shard_range(tiles.shape[0], rank, world)
    local = np.asarray(model(tiles[lo:hi]))
    if not gather:
        return local, (lo, hi)
    return gather_outputs(local, tiles.shape[0], 0, group)


# ---- torch-free path: RCCL through the C ABI (include/msiren.h, "multi-GPU") -----------------------------------

class RcclGroup:
    """This process's rank in an RCCL communicator owned by ``libmsiren`` -- no torch.distributed, one HIP runtime.

    ``RcclGroup(model)`` reads RANK / WORLD_SIZE / MASTER_* (torchrun's contract, also what
    :func:`mri_inr_amd.launch.spawn_ranks` sets), ships rank 0's 128-byte unique id to the other ranks
    (:func:`mri_inr_amd.launch.exchange_from_rank0`) and joins the communicator with the model's handle.
    With WORLD_SIZE == 1 (or unset) nothing is initialised and every collective is the identity.
    """

    def __init__(self, model, env=None):
        from . import _lib
        from .launch import exchange_from_rank0

        env = os.environ if env is None else env
        self.rank, self.world = int(env.get("RANK", "0")), int(env.get("WORLD_SIZE", "1"))
        self.model = model
        model._ensure_handle()
        self._lib, self._h = model._lib, model._h
        if self.world > 1:
            # Rank 0 ALWAYS serves a status blob -- the unique id, or an error marker when it could not make one
            # (librccl missing, ...): every rank then leaves this constructor the same way and at once, instead of the
            # peers waiting out the rendezvous timeout for an id that will never come.
            payload = None
            if self.rank == 0:
                uid = C.create_string_buffer(_lib.COMM_ID_BYTES)
                rc = self._lib.msiren_comm_unique_id(uid, _lib.COMM_ID_BYTES)
                payload = b"OK:" + bytes(uid.raw) if rc == 0 else b"ER:" + _lib.last_error().encode("utf-8", "replace")
            blob = exchange_from_rank0(payload, env=env)
            if blob[:3] != b"OK:":
                raise _lib.MsirenError("rank 0 could not create the RCCL unique id: " + blob[3:].decode("utf-8", "replace"))
            blob = blob[3:]
            _lib.check(self._lib.msiren_comm_init_rank(self._h, blob, len(blob), self.world, self.rank))

    def info(self) -> tuple[int, int]:
        """(ranks, rank) of the communicator as the LIBRARY sees it (msiren_comm_info): (1, 0) without one."""
        n, r = C.c_int32(), C.c_int32()
        self._lib.msiren_comm_info(self._h, C.byref(n), C.byref(r))
        return int(n.value), int(r.value)

    def broadcast_weights(self, src: int = 0, state_dict=None):
        """load_state_dict on ``src`` only (pass the state_dict there, None elsewhere); every rank ends up with the
        source's weights, committed, and with its ``state_dict()`` mirror in step."""
        from . import _lib

        if self.rank == src:
            if state_dict is not None:
                self.model.load_state_dict(state_dict)
            self.model._push_tensors()
        if self.world == 1:
            self.model._ensure_committed()
            return
        _lib.check(self._lib.msiren_broadcast_weights(self._h, src))
        self.model._committed = True
        if self.rank != src:
            self.model._pull_tensors()  # keep state_dict() in step with what the device now holds

    def barrier(self):
        from . import _lib

        _lib.check(self._lib.msiren_comm_barrier(self._h))

    def max(self, value: float) -> float:
        from . import _lib

        v = (C.c_double * 1)(float(value))
        _lib.check(self._lib.msiren_comm_allreduce_max_f64(self._h, v, 1))
        return float(v[0])

    def min(self, value: float) -> float:
        return -self.max(-float(value))

    def max_array(self, values) -> list:
        """Element-wise MAX over the ranks of a short list of floats (one msiren_comm_allreduce_max_f64)."""
        from . import _lib

        v = (C.c_double * len(values))(*[float(x) for x in values])
        _lib.check(self._lib.msiren_comm_allreduce_max_f64(self._h, v, len(values)))
        return [float(x) for x in v]

    def destroy(self):
        from . import _lib

        _lib.check(self._lib.msiren_comm_destroy(self._h))
