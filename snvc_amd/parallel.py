"""Multi-GPU execution of the path: one process per GPU, batch sharding, RCCL only where the path
has a real exchange step.

The reference's only parallelism is ``torch.nn.DataParallel`` over the RoI / pair batch
(tools/inference_agnostic.py:472): a single process that re-broadcasts the parameters on every
forward, scatters inputs on dim 0 and gathers outputs on GPU 0.  On MI355X the units (stereo
pairs, RoI crops) are independent, so:

  * inference  -- each rank owns a contiguous slice of dim 0 (`shard_range`), parameters are
    placed once, and there is NO data-path collective; `gather_outputs` optionally collects the
    small results (ncf / occupancy / coordinates, <= 1 MB per crop) on every rank;
  * training   -- the one exchange step is the gradient all-reduce after backward
    (`all_reduce_gradients`): gradients are packed into ONE flat fp32 bucket so that a single
    RCCL call moves them (the 3D stack has ~0.6-2.3 M parameters = 2.6-9 MB, which is
    latency-bound on xGMI, so one large message beats many small ones).  BatchNorm statistics
    stay per-rank, matching DataParallel (no SyncBN in the reference).

``torch.distributed`` is initialised by the launcher (`python -m torch.distributed.run`); the
backend is "nccl" (= RCCL on ROCm) on GPUs and "gloo" in the CPU tests.
"""
from typing import Iterable, List, Sequence, Tuple

import torch
import torch.distributed as dist


def world() -> Tuple[int, int]:
    """(rank, world_size); (0, 1) when torch.distributed is not initialised."""
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_range(n_items: int, rank: int, world_size: int) -> Tuple[int, int]:
    """Contiguous [lo, hi) slice of dim 0 owned by `rank`; the first n_items % world ranks get one
    extra item (the same split torch's scatter uses for DataParallel)."""
    if n_items < 0 or world_size < 1 or not (0 <= rank < world_size):
        raise ValueError("bad shard request")
    base, extra = divmod(n_items, world_size)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard(tensors: Sequence[torch.Tensor], rank: int = None, world_size: int = None) -> List[torch.Tensor]:
    """Slices every tensor's dim 0 to this rank's shard (views, no copies)."""
    r, w = world()
    rank = r if rank is None else rank
    world_size = w if world_size is None else world_size
    n = tensors[0].shape[0]
    for t in tensors:
        if t.shape[0] != n:
            raise ValueError("all tensors must share dim 0")
    lo, hi = shard_range(n, rank, world_size)
    return [t[lo:hi] for t in tensors]


def gather_outputs(local: torch.Tensor, n_items: int) -> torch.Tensor:
    """All-gathers per-rank results back into dim-0 order (uneven shards are padded to the
    largest shard for the collective and trimmed afterwards)."""
    rank, w = world()
    if w == 1:
        return local
    sizes = [shard_range(n_items, r, w)[1] - shard_range(n_items, r, w)[0] for r in range(w)]
    cap = max(sizes)
    buf = local.new_zeros((cap,) + tuple(local.shape[1:]))
    buf[: local.shape[0]] = local
    parts = [torch.empty_like(buf) for _ in range(w)]
    dist.all_gather(parts, buf.contiguous())
    return torch.cat([p[:s] for p, s in zip(parts, sizes)], dim=0)


def all_reduce_gradients(params: Iterable[torch.nn.Parameter], average: bool = True) -> int:
    """Sums (averages) .grad of every trainable parameter across ranks with ONE flat-bucket all-reduce.
    The bucket covers EVERY ``requires_grad`` parameter in iteration order -- a parameter whose ``.grad`` is
    None on this rank (an unused head, a skipped branch) contributes zeros and receives the reduced
    value -- so the bucket has the same length and layout on all ranks whatever each rank's graph touched.
    Returns the number of bytes moved per rank (0 when single-process)."""
    rank, w = world()
    plist = [p for p in params if p.requires_grad]
    if w == 1 or not plist:
        return 0
    flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1).float() for p in plist])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    if average:
        flat /= w
    off = 0
    for p in plist:
        n = p.numel()
        g = flat[off:off + n].view_as(p).to(p.dtype)
        if p.grad is None:
            p.grad = g.clone()
        else:
            p.grad.copy_(g)
        off += n
    return flat.numel() * 4


def broadcast_parameters(module: torch.nn.Module, src: int = 0) -> None:
    """One-time parameter/buffer placement (DataParallel repeats this on every forward).  The broadcast
    writes into the parameters' storage without going through autograd, so the packed-weight caches of the
    HIP layers are dropped explicitly afterwards."""
    _, w = world()
    if w == 1:
        return
    with torch.no_grad():
        for t in list(module.parameters()) + list(module.buffers()):
            dist.broadcast(t.detach(), src=src)
    from .models.submodule import invalidate_plans
    invalidate_plans(module)
