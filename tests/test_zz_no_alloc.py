"""pnp_qeif_batch_noalloc_f32: the host-pointer QEIF call that stages through a static device
array instead of hipMalloc'ed buffers.  Same kernels as pnp_qeif_batch_f32, so every output must
be bit-identical to that entry point's on the same host arrays; the oracle bounds both."""
import ctypes

import numpy as np
import pytest

from conftest import lm6_indices, rel_t
from oracle import c_oracle, pnp_oracle as po
from pnp_solver_test_amd import _lib

pytestmark = pytest.mark.gpu

K = po.K_DEFAULT
P6f = po.pattern_array(po.PATTERN_ALEXANDER, po.LM6_KEYS).astype(np.float32)


def _outputs(B):
    return dict(R=np.full((B, 9), np.nan, np.float32), t=np.full((B, 3), np.nan, np.float32),
                res=np.full(B, np.nan, np.float32), iters=np.full(B, -1, np.int32),
                rpy=np.full((B, 3), np.nan, np.float32))


def _call(fn, uv, xyz, stride, o, *tail):
    B, N = uv.shape[0], uv.shape[1]
    return fn(uv.ctypes.data, xyz.ctypes.data, stride, _lib.k_array(K), B, N, None, o["R"].ctypes.data,
              o["t"].ctypes.data, o["res"].ctypes.data, o["iters"].ctypes.data, o["rpy"].ctypes.data, *tail)


def _both(lib, uv, xyz, stride):
    a, b = _outputs(len(uv)), _outputs(len(uv))
    assert _call(lib.pnp_qeif_batch_noalloc_f32, uv, xyz, stride, a, None) == 0, lib.pnp_last_error()
    assert _call(lib.pnp_qeif_batch_f32, uv, xyz, stride, b, _lib.PTR_HOST, None) == 0, lib.pnp_last_error()
    for k in a:
        assert np.array_equal(a[k], b[k], equal_nan=True), k
    return a


def test_noalloc_entry_point_equals_the_allocating_one(golden):
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    lib = _lib.load()
    assert lib.pnp_init(0) == 0, lib.pnp_last_error()
    g = golden("stress_qeif_n6")
    uv = np.ascontiguousarray(g["uv15"][:, lm6_indices(g), :], dtype=np.float32)

    # first call: 10 000 problems with a pattern of their own each (the larger staging layout)
    own = np.ascontiguousarray(np.broadcast_to(P6f, (len(uv), 6, 3)))
    a = _both(lib, uv, own, 18)
    ref = c_oracle.qeif_batch(uv.astype(np.float64), P6f.astype(np.float64), K)
    assert np.array_equal(a["iters"], ref["iters"])
    assert np.abs(a["R"].reshape(-1, 3, 3) - ref["R"]).max() <= 5e-7
    assert rel_t(a["t"], ref["t"]).max() <= 1e-6

    # second call, smaller and with a shared pattern: the array still holds the first call's bytes
    small = np.ascontiguousarray(uv[137:137 + 333])
    b = _both(lib, small, P6f, 0)
    assert np.array_equal(b["iters"], ref["iters"][137:137 + 333])
    assert np.abs(b["R"].reshape(-1, 3, 3) - ref["R"][137:137 + 333]).max() <= 5e-7

    # optional outputs may be NULL, an empty batch is fine
    o = _outputs(333)
    rc = lib.pnp_qeif_batch_noalloc_f32(small.ctypes.data, P6f.ctypes.data, 0, _lib.k_array(K), 333, 6, None,
                                         o["R"].ctypes.data, o["t"].ctypes.data, None, None, None, None)
    assert rc == 0 and np.array_equal(o["R"], b["R"]) and np.array_equal(o["t"], b["t"])
    assert lib.pnp_qeif_batch_noalloc_f32(None, None, 0, _lib.k_array(K), 0, 6, None, None, None, None, None, None,
                                           None) == 0


def test_noalloc_entry_point_rejects_what_does_not_fit():
    lib = _lib.load()
    per_problem = (2 * 6 + 3 * 6 + 9 + 3 + 1 + 3) * 4 + 4
    B = _lib.NOALLOC_SCRATCH_BYTES // per_problem + 1          # one more than could ever fit
    uv = np.zeros((B, 6, 2), np.float32)
    own = np.zeros((B, 6, 3), np.float32)
    o = _outputs(B)
    rc = _call(lib.pnp_qeif_batch_noalloc_f32, uv, own, 18, o, None)
    assert rc == -1
    msg = lib.pnp_last_error().decode()
    assert "static buffer" in msg and "split the batch" in msg
    assert np.isnan(o["R"]).all() and (o["iters"] == -1).all()  # nothing was written
    # the largest batch the message names does fit
    fits = int(msg.split("at most ")[1].split(" problems")[0])
    assert 0 < fits < B
    uv2 = np.ascontiguousarray(np.broadcast_to(np.array([[170., 130.], [150., 130.], [168., 150.], [152., 150.],
                                                         [158., 143.], [160., 165.]], np.float32), (fits, 6, 2)))
    o2 = _outputs(fits)
    assert _call(lib.pnp_qeif_batch_noalloc_f32, uv2, np.ascontiguousarray(own[:fits] + P6f), 18, o2, None) == 0, \
        lib.pnp_last_error()
    assert np.isfinite(o2["R"]).all() and (o2["R"] == o2["R"][0]).all()
