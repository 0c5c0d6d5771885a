"""The multi-rank path on CPU (gloo, world_size 2): per-rank contiguous slices tile the batch
and the optional result gather restores rank order.  The solver itself needs a GPU, so each
rank's "solve" here is the C oracle on its slice; what is under test is the sharding and the
collective, which are the same code on RCCL."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import lm6_indices, load_golden


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, total, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import pnp_solver_test_amd as pnp
        from oracle import c_oracle, pnp_oracle as po
        g = load_golden("stress_qeif_n6")
        uv = g["uv15"][:total, lm6_indices(g), :].astype(np.float64)
        P6 = po.pattern_array(po.PATTERN_ALEXANDER, po.LM6_KEYS)
        lo, hi = pnp.shard_bounds(total, rank, world)
        r = c_oracle.qeif_batch(uv[lo:hi], P6, po.K_DEFAULT)
        local = dict(R=torch.from_numpy(r["R"].astype(np.float32)), t=torch.from_numpy(r["t"].astype(np.float32)),
                     iters=torch.from_numpy(r["iters"]))
        full = pnp.gather_results(local)
        rooted = pnp.gather_results(local, dst=1)        # the same slices on one rank only
        if rank == 1:
            assert all(torch.equal(rooted[k], full[k]) for k in full)
        else:
            assert rooted is None
        # whole-job time is the slowest rank's, as bench.py reports it
        tmax = torch.tensor([1.0 + rank], dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        assert tmax.item() == float(world)
        if rank == 0:
            np.savez(out_path, R=full["R"].numpy(), t=full["t"].numpy(), iters=full["iters"].numpy())
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(120)
@pytest.mark.parametrize("total", [2000, 1999])          # equal slices, and slices that differ by one
def test_two_rank_slices_and_gather(tmp_path, total):
    from oracle import c_oracle, pnp_oracle as po
    world = 2
    out_path = str(tmp_path / "gathered.npz")
    mp.spawn(_worker, args=(world, _free_port(), total, out_path), nprocs=world, join=True)
    got = np.load(out_path)
    g = load_golden("stress_qeif_n6")
    uv = g["uv15"][:total, lm6_indices(g), :].astype(np.float64)
    ref = c_oracle.qeif_batch(uv, po.pattern_array(po.PATTERN_ALEXANDER, po.LM6_KEYS), po.K_DEFAULT)
    assert got["R"].shape == (total, 3, 3)
    assert np.array_equal(got["R"], ref["R"].astype(np.float32))
    assert np.array_equal(got["t"], ref["t"].astype(np.float32))
    assert np.array_equal(got["iters"], ref["iters"])


@pytest.mark.timeout(180)
def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no RANK in the environment starts two ranks itself (a child
    `torch.distributed.run`), relays rank 0's JSON line as the last line of stdout and exits with
    the child's status.  --launcher-check runs the rank plumbing on gloo without the solver."""
    import json
    import subprocess
    import sys

    from conftest import ROOT
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    proc = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launcher-check",
                           "--steps", "3", "--batch", "1001"], env=env, stdout=subprocess.PIPE,
                          stderr=subprocess.PIPE, text=True, timeout=170)
    assert proc.returncode == 0, proc.stderr[-2000:]
    line = json.loads(proc.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["launcher_check"] is True
    assert line["slice_sizes"] == [1001, 1001]           # weak scaling: --batch problems per rank
    assert line["slowest_rank_offset_s"] == 1            # the max over ranks, not rank 0's own time


@pytest.mark.timeout(180)
def test_bench_launcher_passes_a_failure_on():
    """Without a GPU the real two-rank bench cannot run: the launcher must hand the ranks' failure
    to its caller as a non-zero status instead of printing a result."""
    import subprocess
    import sys

    import torch
    from conftest import ROOT
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: the ranks would run")
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    proc = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2"], env=env,
                          stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=170)
    assert proc.returncode != 0
    assert '"metric"' not in proc.stdout
