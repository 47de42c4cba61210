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


RS_AG_MIN_BYTES = 8 << 20      # buckets from this size on are reduced as reduce-scatter + all-gather (algorithm="auto")


def _reduce_flat(flat: torch.Tensor, w: int, average: bool, algorithm: str) -> str:
    """Sum (average) a flat fp32 bucket across ranks in place; returns the algorithm used.

    "all_reduce": one RCCL all-reduce (latency-bound sizes: the 3D stack's 2.5 MB).
    "rs_ag": reduce-scatter + all-gather over a bucket padded to a multiple of the world size -- every rank reduces and
    scales only ITS 1/w of the bucket, and both halves are bandwidth-optimal collectives that RCCL spreads over all the
    xGMI links of a rank (the 133.5 MB full-model case).  gloo (CPU tests) has no reduce-scatter."""
    if algorithm == "auto":
        algorithm = "rs_ag" if (flat.numel() * 4 >= RS_AG_MIN_BYTES and dist.get_backend() == "nccl") else "all_reduce"
    if algorithm == "rs_ag":
        n = flat.numel()
        per = -(-n // w)
        buf = flat if per * w == n else torch.cat([flat, flat.new_zeros(per * w - n)])
        mine = torch.empty(per, dtype=flat.dtype, device=flat.device)
        dist.reduce_scatter_tensor(mine, buf, op=dist.ReduceOp.SUM)
        if average:
            mine /= w
        dist.all_gather_into_tensor(buf, mine)
        if buf is not flat:
            flat.copy_(buf[:n])
    elif algorithm == "all_reduce":
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        if average:
            flat /= w
    else:
        raise ValueError(f"unknown gradient reduction algorithm {algorithm!r}")
    return algorithm


def all_reduce_gradients(params: Iterable[torch.nn.Parameter], average: bool = True, force: bool = False,
                         algorithm: str = "auto") -> int:
    """Sums (averages) .grad of every trainable parameter across ranks with ONE flat-bucket collective.
    The bucket covers EVERY ``requires_grad`` parameter in iteration order -- a parameter whose ``.grad`` is
    None on this rank (an unused head, a skipped branch) contributes zeros -- followed by one "has a gradient" flag
    per parameter, so the bucket has the same length and layout on all ranks whatever each rank's graph touched.
    A parameter that received a gradient on SOME rank gets the reduced value everywhere; one that has none on ANY
    rank keeps ``.grad = None`` (as DDP and a single process leave it: no weight decay / momentum on it).
    ``force``: run the collective even in a one-rank group (exercises RCCL on a single GPU; the values do not change).
    Returns the number of gradient bytes moved per rank (0 when nothing was exchanged)."""
    rank, w = world()
    plist = [p for p in params if p.requires_grad]
    if not plist or (w == 1 and not (force and dist.is_available() and dist.is_initialized())):
        return 0
    dev = plist[0].device
    mask = torch.tensor([0.0 if p.grad is None else 1.0 for p in plist], dtype=torch.float32, device=dev)
    flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1).float() for p in plist] + [mask])
    nmask = len(plist)
    _reduce_flat(flat, w, False, algorithm)
    # Who received a gradient matters only to a rank that is MISSING one (an unused head, a skipped branch): it has to learn
    # whether some other rank had it.  A rank whose parameters all carry gradients -- every step of an ordinary training run --
    # takes the reduced values as they are and never waits for the device (r4 read the flags back on every step: a sync).
    if all(p.grad is not None for p in plist):
        has = [1.0] * nmask
    else:
        has = flat[-nmask:].tolist()      # one small device -> host copy, only on ranks with a missing gradient
    if average:
        flat[:-nmask] /= w
    off = 0
    for p, h in zip(plist, has):
        n = p.numel()
        if h > 0.0:
            g = flat[off:off + n].view_as(p).to(p.dtype)
            if p.grad is None:
                p.grad = g.clone()
            else:
                p.grad.copy_(g)
        off += n
    return (flat.numel() - nmask) * 4


def all_reduce_bucket(nbytes: int, device, algorithm: str = "auto", reps: int = 5):
    """Times the gradient collective on a synthetic flat bucket of ``nbytes`` (2.6 MB for the 3D stack,
    133.5 MB for the full model).  Returns (microseconds per call, algorithm used); needs an initialised group."""
    rank, w = world()
    flat = torch.ones(max(nbytes // 4, 1), dtype=torch.float32, device=device)
    used = _reduce_flat(flat, w, True, algorithm)       # warm-up (communicator set-up)
    if flat.is_cuda:
        torch.cuda.synchronize(device)
    import time
    t0 = time.perf_counter()
    for _ in range(reps):
        _reduce_flat(flat, w, True, algorithm)
    if flat.is_cuda:
        torch.cuda.synchronize(device)
    dt = (time.perf_counter() - t0) / reps
    if not torch.allclose(flat[:4], torch.ones(4, device=flat.device)):
        raise RuntimeError("gradient collective changed an all-ones bucket under averaging")
    return 1e6 * dt, used


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
