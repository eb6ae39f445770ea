"""Multi-GPU sharding of the hot path: one process per GPU, torch.distributed (backend
"nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests).

Units that shard:
  * whole sequences            -> no collective at all (bench.py: one sequence per rank, weak scaling)
  * BA points                  -> each rank sums the raw S,b of a contiguous point range; ONE all-reduce(sum) of
                                  D*D + D doubles per BA iteration (10.7 KB at W=6, 29.3 KB at W=10: latency bound)
  * RANSAC hypotheses          -> each rank scores a contiguous range of iterations; ONE all-reduce(max) of a packed
                                  64-bit key (count << 32 | ~iteration): ties resolve to the LOWEST iteration, which
                                  is the reference's strict '>' (T:673)
The reductions are tiny, so they are issued as single collectives (no bucketing); rank-ordered summation of the BA
partials rounds differently from the sequential reference, hence this mode is held to 1e-9 relative agreement
(single-GPU ordered mode stays the bit-exact parity reference).  Hypothesis sharding is exact: counts are integers.

The product path is native: libsfmx issues the collectives itself (csrc/hip/comm.hip: sfmx_comm_*, sfmx_ba_step_sharded;
csrc/host/pipeline.cpp: find_E_ransac_gpu / GpuBundleAdjuster with a communicator).  The functions below that take
torch tensors are the same index arithmetic and reductions in torch form; tests/test_dist_cpu.py runs them over gloo
with world size 2 (no GPU needed), make_comms() / ba_step_sharded() are the glue to the native path.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist


def shard_range(n: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous, order-preserving split of range(n): the first (n % world) ranks get one extra item."""
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_ba_points(X, obs_ptr, obs_li, obs_uv, rank: int, world: int):
    """Point range [lo,hi) of this rank with its CSR re-based to 0 (points keep the reference's order)."""
    X = np.asarray(X)
    obs_ptr = np.asarray(obs_ptr)
    lo, hi = shard_range(len(X), rank, world)
    o0, o1 = int(obs_ptr[lo]), int(obs_ptr[hi])
    return (np.ascontiguousarray(X[lo:hi]), np.ascontiguousarray(obs_ptr[lo:hi + 1] - o0, dtype=np.int32),
            np.ascontiguousarray(np.asarray(obs_li)[o0:o1], dtype=np.int32), np.ascontiguousarray(np.asarray(obs_uv)[o0:o1]))


def allreduce_normal_equations(S: torch.Tensor, b: torch.Tensor, group=None):
    """Sum the per-rank partial S (DxD) and b (D) in place: one collective on a packed buffer."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return S, b
    D = b.numel()
    buf = torch.cat([S.reshape(-1), b.reshape(-1)])
    dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
    S.copy_(buf[:D * D].reshape(S.shape))
    b.copy_(buf[D * D:].reshape(b.shape))
    return S, b


def damp_and_gauge(S: torch.Tensor, b: torch.Tensor, lam: float):
    """T:1064-1071 on the reduced system."""
    D = b.numel()
    idx = torch.arange(D, device=S.device)
    S[idx, idx] += lam
    S[idx[:6], idx[:6]] += 1e9
    b[:6] = 0.0
    return S, b


def pack_best(count: int, iteration: int) -> int:
    return (int(count) << 32) | (0x7FFFFFFF - int(iteration))


def unpack_best(key: int) -> tuple[int, int]:
    return int(key) >> 32, 0x7FFFFFFF - (int(key) & 0xFFFFFFFF)


def allreduce_best_hypothesis(count: int, iteration: int, device="cpu", group=None) -> tuple[int, int]:
    """Global (best_count, best_iter) from per-rank winners; lowest iteration wins ties."""
    t = torch.tensor([pack_best(count, iteration)], dtype=torch.int64, device=device)
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return unpack_best(int(t.item()))


def make_comms(count: int, device_index: int, group=None):
    """`count` native RCCL communicators (capi.Comm) over the ranks of an initialised torch.distributed group: rank 0
    creates the unique ids, torch.distributed carries them (the only thing torch does on this path; the collectives of
    the data path are issued by libsfmx itself, in HBM, on its own streams).  World size 1 needs no RCCL at all."""
    from . import capi
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    comms = []
    for _ in range(count):
        uid = None
        if world > 1:
            box = [capi.comm_unique_id() if rank == 0 else None]
            dist.broadcast_object_list(box, src=0, group=group)
            uid = box[0]
        comms.append(capi.Comm(device_index, uid, rank, world))
    return comms


def ba_step_sharded(ctx, prob, poses_wc, fx, fy, cx, cy, huber, lam, comm=None):
    """One BA iteration with this rank's point shard resident in `prob` (native path: partial build, RCCL all-reduce of
    S | b in HBM on the context's stream, damping + gauge and the dense solve on the device).  Returns (status, dx)."""
    return prob.step_sharded(comm, poses_wc, fx, fy, cx, cy, huber, lam)
