#!/usr/bin/env python3
"""Lanes-per-problem sweep, host-pointer (PCIe-inclusive) batch time and single-call latency.
Interleaved rounds in one process; prints one JSON document."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pnp_solver_test_amd as pnp  # noqa: E402
from pnp_solver_test_amd import _lib, synth  # noqa: E402

dev = torch.device("cuda:0")
lib = _lib.load()
K = synth.K_DEFAULT


def kernel_ms(fn, rounds=5):
    fn()
    ts = []
    for _ in range(rounds):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return float(np.median(ts))


out = {}
B = 1 << 20
uv_h, _ = synth.stress_batch(B, seed=42)
uv = torch.from_numpy(uv_h).to(dev)
P6 = torch.from_numpy(synth.PATTERN6.astype(np.float32)).to(dev)
xyz = P6[None].expand(B, 6, 3).contiguous()
B68 = 1 << 18
P68 = synth.face68()
uv68 = torch.from_numpy(synth.stress_batch(B68, seed=46, pattern=P68, chunk=1 << 15)[0]).to(dev)
P68d = torch.from_numpy(P68.astype(np.float32)).to(dev)
own68 = P68d[None].expand(B68, 68, 3).contiguous()
sweep = {"qeif_n6_b1m_per_problem_xyz": {}, "qeif_n6_b1m_shared": {}, "dlt_n6_b1m_per_problem_xyz": {},
         "qeif_n68_b256k_shared_point_form": {}, "qeif_n68_b256k_per_problem_xyz_point_form": {}}
for lanes in (1, 2, 4, 8, 16, 32, 64):
    lib.pnp_set_lanes_per_problem(lanes)
    nb = B if lanes <= 8 else B >> 3          # keep the slow mappings short; rates are per problem
    sweep["qeif_n6_b1m_per_problem_xyz"][lanes] = nb / kernel_ms(lambda: pnp.solve_pnp_batch(uv[:nb], xyz[:nb], K, want_rpy=False)) * 1e3
    sweep["qeif_n6_b1m_shared"][lanes] = nb / kernel_ms(lambda: pnp.solve_pnp_batch(uv[:nb], P6, K, want_rpy=False)) * 1e3
    sweep["dlt_n6_b1m_per_problem_xyz"][lanes] = nb / kernel_ms(lambda: pnp.solve_pnp_batch(uv[:nb], xyz[:nb], K, method="dlt", want_rpy=False)) * 1e3
    nb68 = B68 if lanes <= 16 else B68 >> 2
    # a forced lane count may take up to 160 KB of LDS per workgroup: what ran is read back
    for key, pat, stride in (("qeif_n68_b256k_shared_point_form", P68d, 0),
                             ("qeif_n68_b256k_per_problem_xyz_point_form", own68, 204)):
        x = pat if stride == 0 else pat[:nb68]
        rate = nb68 / kernel_ms(lambda: pnp.solve_pnp_batch(uv68[:nb68], x, K, want_rpy=False)) * 1e3
        ran = _lib.last_launch()
        sweep[key]["asked %d" % lanes] = dict(solves_per_s=rate, ran_lanes=lib.pnp_get_lanes_per_problem(nb68, 68, stride),
                                              kernel=ran["kernel"], lds_bytes=ran["lds_bytes"])
lib.pnp_set_lanes_per_problem(0)
out["solves_per_s_by_lanes_per_problem"] = sweep

# host-pointer path: PCIe-inclusive
P6h = synth.PATTERN6.astype(np.float32)
xyz_h = np.ascontiguousarray(np.broadcast_to(P6h, (B, 6, 3)))
pnp.solve_pnp_batch(uv_h, xyz_h, K, want_rpy=False)
t0 = time.perf_counter()
for _ in range(3):
    pnp.solve_pnp_batch(uv_h, xyz_h, K, want_rpy=False)
dt = (time.perf_counter() - t0) / 3
out["host_pointer_path_b1m"] = dict(ms_per_batch=dt * 1e3, solves_per_s=B / dt,
                                    note="pageable NumPy buffers: H2D 126 MB + kernel + D2H 59 MB, synchronous")

# single-call latency through the drop-in class
s = pnp.PNP_SOLVER(K, [synth.get_golden_pattern("Alexander")])
R = s.get_rotation_matrix_from_Euler(15.0, 45.0, -15.0, is_degree=True)
img = s.perspective_projection_golden_landmarks(R, np.array([[0.35], [0.35], [0.5]]), is_quantized=True)
for _ in range(20):
    s.solve_pnp(img)
t0 = time.perf_counter()
for _ in range(500):
    s.solve_pnp(img)
out["solve_pnp_single_call_us"] = (time.perf_counter() - t0) / 500 * 1e6
print(json.dumps(out, indent=1))
