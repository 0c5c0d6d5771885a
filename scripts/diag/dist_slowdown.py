"""Does bringing up RCCL in the process change the solver kernel's time?  Times the headline
launch before process-group init, after init, after the first collective, and after destroy."""
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import pnp_solver_test_amd as pnp  # noqa: E402
from pnp_solver_test_amd import synth  # noqa: E402

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
K = synth.K_DEFAULT
B = 1 << 20
uv_h, _ = synth.stress_batch(B, seed=42, pattern=synth.PATTERN6, K=K, is_quantized=True)
uv = torch.from_numpy(uv_h).to(dev)
xyz = torch.from_numpy(np.ascontiguousarray(np.broadcast_to(synth.PATTERN6.astype(np.float32), (B, 6, 3)))).to(dev)


def timed(tag, reps=100):
    for _ in range(400):
        pnp.solve_pnp_batch(uv, xyz, K, want_rpy=False)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        pnp.solve_pnp_batch(uv, xyz, K, want_rpy=False)
    b.record()
    torch.cuda.synchronize()
    print("%-40s %.4f ms" % (tag, a.elapsed_time(b) / reps), flush=True)


timed("before any process group")
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29544", RANK="0", WORLD_SIZE="1")
dist.init_process_group("nccl", device_id=dev)
timed("after init_process_group(nccl)")
dist.barrier()
torch.cuda.synchronize()
timed("after the first barrier")
t = torch.ones(4, device=dev)
dist.all_reduce(t)
torch.cuda.synchronize()
timed("after an all_reduce")
dist.destroy_process_group()
timed("after destroy_process_group")
