"""The N-rank leg of bench.py on the one GPU a test box has: a fresh child process runs
`python -m torch.distributed.run --nproc-per-node 1 bench.py --gpus 1 ...`, i.e. the code an 8-GPU
scaling run executes on every rank - torch.distributed over RCCL (backend "nccl"), the barriers
around the timed region, the max over ranks, the per-rank times, the four variants of the optional
result exchange (torch.distributed and the C ABI's pnp_gather_results, all-gather and gather to a
root) and the deadline that guards them - with world size 1.  Its value must agree with a plain
`python bench.py --gpus 1` run in the same test.

The children are started with subprocess from this process (which has touched the GPU) and are
never a replacement of it."""
import json
import os
import socket
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

BENCH = os.path.join(ROOT, "bench.py")
COMMON = ["--gpus", "1", "--steps", "20", "--warmup", "5", "--no-cpu-baseline", "--no-other-configs"]


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _clean_env():
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "PNP_HIP_LIB")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return env


def _line(proc):
    assert proc.returncode == 0, (proc.stdout[-1500:], proc.stderr[-3000:])
    lines = [ln for ln in proc.stdout.splitlines() if ln.lstrip().startswith("{") and '"metric"' in ln]
    assert len(lines) == 1, proc.stdout[-3000:]
    return json.loads(lines[0])


@pytest.mark.timeout(420)
def test_one_rank_under_torch_distributed_run_matches_the_plain_run():
    plain = _line(subprocess.run([sys.executable, BENCH] + COMMON, env=_clean_env(), capture_output=True, text=True,
                                 timeout=200))
    ranked = _line(subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1",
                                   "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), BENCH] + COMMON,
                                  env=_clean_env(), capture_output=True, text=True, timeout=200))
    assert plain["rccl_ranks"] == 0 and ranked["rccl_ranks"] == 1 and ranked["n_gpus"] == 1
    for line in (plain, ranked):
        assert line["roofline"]["kernel"] == "pnp::qeif_rows6_kernel<float>"      # from the library, not a literal
        assert line["roofline"]["kernel_ms_samples"] >= 8
        assert line["roofline"]["launch"] == dict(grid=(1 << 20) // 64, block=64, lds_bytes=0)
        assert line["config"]["all_outputs_finite"] is True
    gather = ranked["with_result_gather"]
    for variant in ("all_gather", "to_rank_0_only", "c_abi_all_gather", "c_abi_to_rank_0_only"):
        assert variant in gather and "error" not in gather[variant], gather
        assert gather[variant]["solves_per_s"] > 0
    assert "error" not in gather and "c_abi_error" not in gather
    assert ranked["per_rank"]["slowest_rank"] == 0 and len(ranked["per_rank"]["kernel_ms"]) == 1
    assert abs(ranked["per_rank"]["kernel_ms"][0] - ranked["roofline"]["kernel_ms"]) < 1e-3
    # the same kernel on the same box within the same minute: the two legs must agree
    assert abs(ranked["value"] / plain["value"] - 1.0) < 0.10, (ranked["value"], plain["value"])


@pytest.mark.timeout(300)
def test_more_ranks_than_devices_is_refused_with_a_reason():
    import torch
    n = torch.cuda.device_count() + 1
    proc = subprocess.run([sys.executable, BENCH, "--gpus", str(n), "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
                           "--no-other-configs"], env=_clean_env(), capture_output=True, text=True, timeout=280)
    assert proc.returncode != 0
    assert '"metric"' not in proc.stdout
    assert "only %d" % torch.cuda.device_count() in proc.stderr and "visible" in proc.stderr
