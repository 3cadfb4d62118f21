"""Multi-GPU layer: one process per GPU (torch.distributed, backend "nccl" = RCCL on ROCm).

The path shards by independent audio streams: each rank owns a contiguous block
of segments, weights (52 kB) are replicated, and nothing is exchanged on the data path.  The only
collective is ONE all-reduce of four fp64 scalars for the loss aggregate that code/test-model.py
computes at :386-398 (mean over segments of the per-segment loss) -- 32 bytes, latency-bound.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / MASTER_* if WORLD_SIZE > 1 (or, with
    NTM_DIST_FORCE_INIT=1, also for a single rank: every collective of the N-rank path then really runs on the
    backend -- how the RCCL calls are exercised on a one-GPU box).  Returns (rank, world_size, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if backend is None:
        # NTM_DIST_BACKEND=gloo lets several ranks share one GPU (a dev box with a single MI355X): it exercises the
        # N>1 control flow; real multi-GPU runs use nccl (= RCCL), one rank per GPU
        backend = os.environ.get("NTM_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
    if backend != "nccl" and torch.cuda.is_available() and torch.cuda.device_count() > 0:
        local %= torch.cuda.device_count()
    force = os.environ.get("NTM_DIST_FORCE_INIT") == "1"
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def shard_range(n, rank, world):
    """Contiguous block [lo, hi) of n streams owned by `rank` (sizes differ by at most 1)."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def reduce_loss_sums(per_segment_loss, err_sums=None, group=None):
    """Aggregate a rank's per-segment losses into the job-wide result.

    per_segment_loss: (n_local,) tensor; err_sums: optional (n_local, 2) [sum e^2, sum t^2].
    Returns dict(mean_segment_loss, segments, sum_err2, sum_tgt2): one SUM all-reduce of 4 fp64."""
    dev = per_segment_loss.device
    v = torch.zeros(4, dtype=torch.float64, device=dev)
    v[0] = per_segment_loss.double().sum()
    v[1] = per_segment_loss.numel()
    if err_sums is not None:
        v[2:4] = err_sums.double().sum(dim=0)
    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(v, op=dist.ReduceOp.SUM, group=group)
    v = v.cpu()
    n = int(v[1].item())
    return {"mean_segment_loss": float(v[0] / max(n, 1)), "segments": n,
            "sum_err2": float(v[2]), "sum_tgt2": float(v[3])}


def local_loss_sums(per_segment_loss, err_sums=None):
    """This rank's 4 fp64 scalars [sum of per-segment losses, segment count, sum err^2, sum tgt^2] as a device
    tensor on the CURRENT stream -- no collective, no host synchronisation (bench.py keeps one per timed step
    and reduces them together with reduce_many)."""
    dev = per_segment_loss.device
    parts = [per_segment_loss.double().sum().reshape(1),
             torch.full((1,), float(per_segment_loss.numel()), dtype=torch.float64, device=dev),
             err_sums.double().sum(dim=0) if err_sums is not None else torch.zeros(2, dtype=torch.float64, device=dev)]
    return torch.cat(parts)


def reduce_many(local_vectors, group=None):
    """Job-wide results for a list of local_loss_sums vectors: ONE SUM all-reduce of the stacked (K,4) tensor on the
    current stream, then one fetch.  -> list of the dicts reduce_loss_sums returns."""
    if not local_vectors:
        return []
    V = torch.stack(local_vectors)
    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(V, op=dist.ReduceOp.SUM, group=group)
    V = V.cpu()
    out = []
    for v in V:
        n = int(v[1].item())
        out.append({"mean_segment_loss": float(v[0] / max(n, 1)), "segments": n,
                    "sum_err2": float(v[2]), "sum_tgt2": float(v[3])})
    return out


def max_over_ranks(value, device):
    """MAX all-reduce of one python float (used for the benchmark's elapsed time)."""
    t = torch.tensor([value], dtype=torch.float64, device=device)
    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def barrier():
    if dist.is_available() and dist.is_initialized():
        dist.barrier()
