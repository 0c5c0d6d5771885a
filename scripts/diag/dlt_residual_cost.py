import sys
import numpy as np, torch
sys.path.insert(0, ".")
import pnp_solver_test_amd as pnp
from pnp_solver_test_amd import synth
dev = torch.device("cuda:0"); K = synth.K_DEFAULT; B = 1 << 20
uv_h, _ = synth.stress_batch(B, seed=42, pattern=synth.PATTERN6, K=K, is_quantized=True)
uv = torch.from_numpy(uv_h).to(dev)
xyz = torch.from_numpy(np.ascontiguousarray(np.broadcast_to(synth.PATTERN6.astype(np.float32), (B, 6, 3)))).to(dev)
def timed(fn, reps=200, warm=600):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
for r in range(2):
    for wr in (True, False):
        print("want_res", wr, round(timed(lambda: pnp.solve_pnp_batch(uv, xyz, K, method="dlt", want_rpy=False, want_res=wr)), 4), "ms", flush=True)
