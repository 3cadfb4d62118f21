"""CPU, world_size 2 over gloo: the N>1 control flow (stream sharding + the one all-reduce of the
loss scalars) gives the same job-wide ESR as a single process.  The per-segment sums come from the
oracle here because the data path itself needs a HIP device."""
import os
import socket
import sys

import numpy as np
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, y, t, skip, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import ntm_amd
    import oracle
    from ntm_amd import distributed as D
    r, w, _ = D.init_from_env("gloo")
    assert (r, w) == (rank, world)
    lo, hi = D.shard_range(y.shape[0], rank, world)
    s = torch.from_numpy(oracle.esr_sums(y[lo:hi], t[lo:hi], skip))
    n = y.shape[1] - skip
    per_seg = (s[:, 0] / n) / (s[:, 1] / n + ntm_amd.model.ESR_EPS)
    res = D.reduce_loss_sums(per_seg, s)
    # the stacked form bench.py uses: K steps' local scalars, ONE all-reduce
    many = D.reduce_many([D.local_loss_sums(per_seg, s), D.local_loss_sums(per_seg, s)])
    assert many[0] == res and many[1] == res
    tmax = D.max_over_ranks(float(rank + 1), torch.device("cpu"))
    D.barrier()
    q.put((rank, res, tmax))
    torch.distributed.destroy_process_group()


def test_two_rank_loss_reduction_matches_single_process():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle
    rng = np.random.default_rng(42)
    B, T, skip = 7, 3000, 1024                      # odd B: ranks get 4 and 3 segments
    t = rng.standard_normal((B, T)).astype(np.float32)
    y = (t + 0.2 * rng.standard_normal((B, T))).astype(np.float32)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, y, t, skip, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = float(np.mean(oracle.esr_per_segment(y, t, skip)))
    sums = oracle.esr_sums(y, t, skip).sum(0)
    for rank, res, tmax in got:
        assert res["segments"] == B
        assert abs(res["mean_segment_loss"] - want) < 1e-12
        assert abs(res["sum_err2"] - sums[0]) < 1e-9 * sums[0]
        assert abs(res["sum_tgt2"] - sums[1]) < 1e-9 * sums[1]
        assert tmax == 2.0


def test_local_sums_and_reduce_many_single_process():
    """local_loss_sums + reduce_many (used by bench.py to keep the loss leg off the launch path) == the one-call form."""
    import torch
    from ntm_amd import distributed as D
    per = torch.tensor([0.1, 0.2, 0.4], dtype=torch.float64)
    sums = torch.tensor([[1.0, 2.0], [3.0, 4.0], [5.0, 6.0]], dtype=torch.float64)
    a = D.reduce_loss_sums(per, sums)
    b = D.reduce_many([D.local_loss_sums(per, sums), D.local_loss_sums(2 * per, None)])
    assert a == b[0] and abs(a["mean_segment_loss"] - 0.7 / 3) < 1e-15 and a["sum_err2"] == 9.0 and a["sum_tgt2"] == 12.0
    assert abs(b[1]["mean_segment_loss"] - 1.4 / 3) < 1e-15 and b[1]["sum_err2"] == 0.0 and D.reduce_many([]) == []
