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


def _run_bench(args, env_extra=None):
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=300)
    lines = [json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith("{")]
    return r, lines


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus N` with no WORLD_SIZE in the environment (how the driver runs it): the parent spawns N
    rank processes, relays rank 0's single JSON line, returns 0 -- here with the data path switched off
    (--launch-check, CPU, gloo); the GPU variant of this test runs the real step on the GPU box."""
    r, lines = _run_bench(["--gpus", "2", "--launch-check"])
    assert r.returncode == 0, r.stderr[-1500:]
    assert lines == [{"launch_check": True, "world": 2, "segments_total": 8192, "rank_sum": 1, "ranks": 2, "scaling": "weak"}]
    # strong scaling: 10 segments over 3 ranks = 4 + 3 + 3
    r, lines = _run_bench(["--gpus", "3", "--launch-check", "--scaling", "strong", "--total-batch", "10"])
    assert r.returncode == 0 and lines[0]["segments_total"] == 10 and lines[0]["ranks"] == 3


def test_bench_launcher_propagates_a_failing_rank():
    r, lines = _run_bench(["--gpus", "2", "--launch-check", "--fail-rank", "1"])
    assert r.returncode != 0 and "rank 1 exited with status 3" in r.stderr


def test_bench_launcher_blames_the_culprit_not_its_victims():
    """A rank that dies BEFORE the collective leaves its peers inside the all-reduce, where they abort (signal) or
    raise (status 1) or hang until the launcher terminates them: whatever they do, the launcher names the rank that
    exited by itself with a positive status first, lists the others, and returns non-zero well inside the timeout."""
    import time
    t0 = time.monotonic()
    r, lines = _run_bench(["--gpus", "4", "--launch-check", "--fail-rank", "2", "--fail-early"])
    assert r.returncode != 0 and "rank 2 exited with status 3" in r.stderr, r.stderr[-1500:]
    assert time.monotonic() - t0 < 120
    assert not lines                                      # nobody printed a result line


def test_bench_under_an_external_launcher_keeps_working():
    """RANK / WORLD_SIZE given by a launcher (torch.distributed.run style): bench.py must NOT spawn again."""
    import subprocess
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launch-check"],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=300) for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert '"world": 2' in outs[0][0] and "{" not in outs[1][0]        # only rank 0 prints the JSON line
