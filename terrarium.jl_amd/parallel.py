"""Multi-GPU: block-sharding of the laterally independent columns over the
ranks of one node (one process per GPU, `torch.distributed`; backend "nccl" is
RCCL over xGMI on ROCm), and the global diagnostic reductions.

There is NO collective on the step path: columns never exchange data (the
reference's only cross-column code is the x-periodic halo copy, which nothing
reads).  Collectives appear only where the path has a real
exchange: diagnostics.  Each `global_reduce` packs its per-row partials into one
small tensor and issues ONE all-reduce; the messages are a few hundred bytes, so
they are latency-bound and ring/link bandwidth is irrelevant.
"""
from typing import Tuple

import numpy as np


def shard_range(num_columns: int, world_size: int, rank: int) -> Tuple[int, int]:
    """Contiguous block [lo, hi) of the ring-ordered columns owned by `rank`:
    blocks of B = ceil(Nh / P) columns, device d owns [d*B, min((d+1)*B, Nh))."""
    assert 0 <= rank < world_size
    block = -(-num_columns // world_size)
    lo = min(rank * block, num_columns)
    hi = min(lo + block, num_columns)
    return lo, hi


def shard_columns(array: np.ndarray, world_size: int, rank: int) -> np.ndarray:
    """Slice the trailing (column) axis of a host array to this rank's block."""
    lo, hi = shard_range(array.shape[-1], world_size, rank)
    return array[..., lo:hi]


def _dist():
    import torch.distributed as dist
    return dist if dist.is_available() and dist.is_initialized() else None


def _device():
    import torch
    dist = _dist()
    if dist is not None and dist.get_backend() == "nccl":
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


_OPS = {"sum": "SUM", "min": "MIN", "max": "MAX", "hasnan": "MAX", "volume_integral_z": "SUM"}


def combine(local: np.ndarray, op: str) -> np.ndarray:
    """All-reduce per-rank partial results (`DeviceState.reduce` output) across ranks."""
    import torch
    dist = _dist()
    local = np.asarray(local, dtype=np.float64)
    if dist is None or dist.get_world_size() == 1:
        return local
    if op in ("min", "max"):
        # Base.minimum / maximum: a NaN on any rank must reach every rank.  The backends' MIN / MAX do not promise that, so
        # the values travel NaN-free next to a flag that travels with the same operator (the scheme of trm_reduce_global).
        packed, _ = pack_nan_flags(local, op)
        t = torch.from_numpy(packed).to(_device())
        dist.all_reduce(t, op=getattr(dist.ReduceOp, _OPS[op]))
        return unpack_nan_flags(t.cpu().numpy())
    t = torch.from_numpy(local.copy()).to(_device())
    dist.all_reduce(t, op=getattr(dist.ReduceOp, _OPS[op]))
    return t.cpu().numpy()


def pack_nan_flags(values: np.ndarray, op: str):
    """[values with NaN replaced by the operator's neutral element | flags]: flag = -1 (min) / +1 (max) where the value was
    NaN, else 0 -- reducing both halves with the same operator leaves a non-zero flag wherever any rank held a NaN."""
    values = np.asarray(values, dtype=np.float64)
    nan = np.isnan(values)
    sentinel, yes = (np.inf, -1.0) if op == "min" else (-np.inf, 1.0)
    return np.concatenate([np.where(nan, sentinel, values), np.where(nan, yes, 0.0)]), nan


def unpack_nan_flags(packed: np.ndarray) -> np.ndarray:
    n = packed.size // 2
    return np.where(packed[n:] != 0.0, np.nan, packed[:n])


def init_library_communicator(state):
    """Gives `state`'s context an RCCL communicator over the ranks of the torch.distributed group (trm_comm_init): the
    128-byte id travels through the existing process group, everything after that happens inside the library."""
    import torch
    dist = _dist()
    world, rank = (dist.get_world_size(), dist.get_rank()) if dist is not None else (1, 0)
    uid = bytearray(state.comm_unique_id() if rank == 0 else bytes(128))
    if world > 1:
        t = torch.frombuffer(uid, dtype=torch.uint8).clone().to(_device())
        dist.broadcast(t, src=0)
        uid = bytearray(t.cpu().numpy().tobytes())
    state.comm_init(rank, world, bytes(uid))


def global_reduce(state, field: str, op: str) -> np.ndarray:
    """Global diagnostic over all ranks' columns.  With a library communicator (init_library_communicator): local GPU
    reduction + one RCCL all-reduce inside libterrarium_hip.so (trm_reduce_global); otherwise the local reduction
    (trm_reduce) combined by one torch.distributed all-reduce."""
    if hasattr(state, "comm_world") and state.comm_world() > 0:
        return state.reduce_global(field, op)
    return combine(state.reduce(field, op), op)


def global_status(local_flags: int) -> int:
    """OR of the status flag words of all ranks (NaN seen / composition out of range)."""
    import torch
    dist = _dist()
    if dist is None or dist.get_world_size() == 1:
        return int(local_flags)
    bits = torch.tensor([(local_flags >> b) & 1 for b in range(8)], dtype=torch.int32, device=_device())
    dist.all_reduce(bits, op=dist.ReduceOp.MAX)
    return int(sum(int(v) << b for b, v in enumerate(bits.cpu().tolist())))


def gather_columns(local: np.ndarray, num_columns: int) -> np.ndarray:
    """Gather a column-sharded host array [..., Nh_local] to the full ring-ordered array on every rank
    (output path only; e.g. a 2-D field at N145 is 0.46 MB)."""
    import torch
    dist = _dist()
    if dist is None or dist.get_world_size() == 1:
        return np.asarray(local)
    world = dist.get_world_size()
    block = -(-num_columns // world)
    lead = local.shape[:-1]
    padded = np.zeros(lead + (block,), dtype=local.dtype)
    padded[..., : local.shape[-1]] = local
    t = torch.from_numpy(padded).to(_device())
    parts = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(parts, t)
    full = np.concatenate([p.cpu().numpy() for p in parts], axis=-1)
    return full[..., :num_columns]
