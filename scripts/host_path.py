#!/usr/bin/env python3
"""The host-pointer path (NumPy arrays in pageable memory) on the bench workload: wall-clock
per 1M-problem call, bytes moved over PCIe per second, and equality with the device-pointer
path on the same inputs."""
import json
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import pnp_solver_test_amd as pnp  # noqa: E402
from pnp_solver_test_amd import synth  # noqa: E402

dev = torch.device("cuda:0")
K = synth.K_DEFAULT
B = 1 << 20
uv, _ = synth.stress_batch(B, seed=42, pattern=synth.PATTERN6, K=K, is_quantized=True)
P6 = synth.PATTERN6.astype(np.float32)
own = np.ascontiguousarray(np.broadcast_to(P6, (B, 6, 3)))


def wall(fn, reps=8):
    fn()
    fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    return (time.perf_counter() - t0) / reps


report = {}
for tag, xyz, method, nbytes in (("qeif_own_xyz", own, "qeif", 120 + 48 + 8 + 12), ("qeif_shared_xyz", P6, "qeif", 48 + 48 + 8 + 12),
                                 ("dlt_own_xyz", own, "dlt", 120 + 48 + 4 + 12)):
    s = wall(lambda: pnp.solve_pnp_batch(uv, xyz, K, method=method))
    host = pnp.solve_pnp_batch(uv, xyz, K, method=method)
    d = pnp.solve_pnp_batch(torch.from_numpy(uv).to(dev), torch.from_numpy(xyz).to(dev), K, method=method)
    same = all(np.array_equal(host[k], d[k].cpu().numpy(), equal_nan=True) for k in host)
    report[tag] = dict(ms_per_call=round(s * 1e3, 3), solves_per_s=round(B / s, 1), pcie_GBps=round(B * nbytes / s / 1e9, 2),
                       equals_device_path=bool(same))
    print(tag, report[tag], flush=True)
# the C ABI alone, with result arrays that exist already (no fresh pages to fault in)
from pnp_solver_test_amd import _lib  # noqa: E402
lib = _lib.load()
R, t = np.zeros((B, 9), np.float32), np.zeros((B, 3), np.float32)
res, it = np.zeros(B, np.float32), np.zeros(B, np.int32)
Kc = _lib.k_array(K)
for tag, xyz, stride, nbytes in (("c_abi_qeif_own_xyz", own, 18, 120 + 48 + 8), ("c_abi_qeif_shared_xyz", P6, 0, 48 + 48 + 8)):
    s = wall(lambda: _lib.check(lib.pnp_qeif_batch_f32(uv.ctypes.data, xyz.ctypes.data, stride, Kc, B, 6, None, R.ctypes.data,
                                                       t.ctypes.data, res.ctypes.data, it.ctypes.data, None, 0, None)))
    report[tag] = dict(ms_per_call=round(s * 1e3, 3), solves_per_s=round(B / s, 1), pcie_GBps=round(B * nbytes / s / 1e9, 2))
    print(tag, report[tag], flush=True)
# the same with every array in page-locked memory (pinned torch tensors): no staging copies
def locked(a):
    t = torch.empty(a.shape, dtype=torch.from_numpy(a).dtype, pin_memory=True)
    t.copy_(torch.from_numpy(a))
    return t
uv_l, own_l, P6_l = locked(uv), locked(own), locked(P6)
R_l, t_l, res_l, it_l = locked(R), locked(t), locked(res), locked(it)
for tag, xyz_l, stride, nbytes in (("c_abi_qeif_own_xyz_page_locked", own_l, 18, 120 + 48 + 8), ("c_abi_qeif_shared_xyz_page_locked", P6_l, 0, 48 + 48 + 8)):
    s = wall(lambda: _lib.check(lib.pnp_qeif_batch_f32(uv_l.data_ptr(), xyz_l.data_ptr(), stride, Kc, B, 6, None, R_l.data_ptr(),
                                                       t_l.data_ptr(), res_l.data_ptr(), it_l.data_ptr(), None, 0, None)))
    report[tag] = dict(ms_per_call=round(s * 1e3, 3), solves_per_s=round(B / s, 1), pcie_GBps=round(B * nbytes / s / 1e9, 2))
    print(tag, report[tag], flush=True)
small = uv[:1000]
s = wall(lambda: pnp.solve_pnp_batch(small, P6, K), reps=200)
report["qeif_1000_problems"] = dict(us_per_call=round(s * 1e6, 1))
one = uv[:1]
s = wall(lambda: pnp.solve_pnp_batch(one, P6, K), reps=500)
report["qeif_1_problem"] = dict(us_per_call=round(s * 1e6, 1))
print(json.dumps(report))
