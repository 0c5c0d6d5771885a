"""The optional result exchange over RCCL: through the C ABI (pnp_comm_*, pnp_gather_results)
with one rank on one GPU, and - where at least two GPUs are visible - a two-rank sharded solve
gathered both through torch.distributed (backend nccl = RCCL) and through the C ABI, against
the oracle."""
import os
import socket

import numpy as np
import pytest

from conftest import lm6_indices, load_golden

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_gather_through_the_c_abi_with_one_rank():
    """world = 1: RCCL is opened, a communicator made, and both forms of the exchange (all-gather,
    gather to a root) hand back this rank's own arrays."""
    import torch

    import pnp_solver_test_amd as pnp
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    comm = pnp.RcclComm(0, 1, pnp.RcclComm.unique_id())
    try:
        local = dict(R=torch.arange(5 * 9, dtype=torch.float32, device=dev).reshape(5, 3, 3),
                     iters=torch.arange(5, dtype=torch.int32, device=dev))
        for dst in (None, 0):
            full = pnp.gather_results(local, dst=dst, comm=comm)
            torch.cuda.synchronize()
            assert all(torch.equal(full[k], local[k]) for k in local)
        lib = pnp._lib.load()
        assert lib.pnp_gather_results(local["R"].data_ptr(), None, 16, 3, 0, 1, comm._comm, None) == -1
        assert b"root" in lib.pnp_last_error()
    finally:
        comm.close()


def _worker(rank, world, port, total, out_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch
    import torch.distributed as dist

    import pnp_solver_test_amd as pnp
    from oracle import pnp_oracle as po
    dev = torch.device("cuda", rank)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    comm = None
    try:
        g = load_golden("stress_qeif_n6")
        uv = np.ascontiguousarray(g["uv15"][:total, lm6_indices(g), :])
        P6 = po.pattern_array(po.PATTERN_ALEXANDER, po.LM6_KEYS).astype(np.float32)
        lo, hi = pnp.shard_bounds(total, rank, world)
        out = pnp.solve_pnp_batch(torch.from_numpy(uv[lo:hi]).to(dev), torch.from_numpy(P6).to(dev), po.K_DEFAULT)
        local = {k: out[k] for k in ("R", "t", "iters")}
        full = pnp.gather_results(local)                                  # torch.distributed, all-gather
        rooted = pnp.gather_results(local, dst=1)                         # ... gather to rank 1
        comm = pnp.RcclComm.from_torch_group()
        full_c = pnp.gather_results(local, comm=comm)                     # C ABI, all-gather
        rooted_c = pnp.gather_results(local, dst=0, comm=comm)            # C ABI, gather to rank 0
        torch.cuda.synchronize()
        assert all(torch.equal(full[k], full_c[k]) for k in full)
        assert (rooted is not None) == (rank == 1) and (rooted_c is not None) == (rank == 0)
        if rank == 1:
            assert all(torch.equal(rooted[k], full[k]) for k in full)
        if rank == 0:
            assert all(torch.equal(rooted_c[k], full[k]) for k in full)
            np.savez(out_path, **{k: v.cpu().numpy() for k, v in full.items()})
        dist.barrier()
    finally:
        if comm is not None:
            comm.close()
        dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("total", [4096, 4097])              # equal slices, and slices that differ by one
def test_two_gpus_shard_and_gather(tmp_path, total):
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    import torch.multiprocessing as mp

    from oracle import c_oracle, pnp_oracle as po
    out_path = str(tmp_path / "gathered.npz")
    mp.spawn(_worker, args=(2, _free_port(), total, out_path), nprocs=2, join=True)
    got = np.load(out_path)
    g = load_golden("stress_qeif_n6")
    uv = g["uv15"][:total, lm6_indices(g), :].astype(np.float64)
    P6 = po.pattern_array(po.PATTERN_ALEXANDER, po.LM6_KEYS).astype(np.float32).astype(np.float64)
    ref = c_oracle.qeif_batch(uv, P6, po.K_DEFAULT)
    assert got["R"].shape == (total, 3, 3)
    assert np.array_equal(got["iters"], ref["iters"])
    assert np.abs(got["R"] - ref["R"]).max() <= 5e-7 and np.abs(got["t"] - ref["t"]).max() <= 1e-6
