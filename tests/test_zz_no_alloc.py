"""pnp_qeif_batch_noalloc_f32: the host-pointer QEIF call that stages through a static device
array instead of buffers the library allocates.  Every output must be bit-identical to
pnp_qeif_batch_f32's on the same host arrays; the oracle bounds both.  The two need not run the same
kernel: the static array is device memory, so large six-point batches are solved there by lanes
that read their own rows, while the host-pointer path launches the LDS-staged kernel on host
memory - the last test holds the two to the same bits through both entry points."""
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


def test_noalloc_entry_point_rejects_what_does_not_fit(golden):
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
    g = golden("stress_qeif_n6")
    uv6 = np.ascontiguousarray(g["uv15"][:, lm6_indices(g), :], dtype=np.float32)
    uv2 = np.ascontiguousarray(np.tile(uv6, (fits // len(uv6) + 1, 1, 1))[:fits])
    own2 = np.ascontiguousarray(own[:fits] + P6f)
    o2, o3 = _outputs(fits), _outputs(fits)
    assert _call(lib.pnp_qeif_batch_noalloc_f32, uv2, own2, 18, o2, None) == 0, lib.pnp_last_error()
    # a batch of this size has one lane per problem: on the static device array that is the kernel
    # whose lanes read their own rows ...
    assert _lib.last_launch()["kernel"] == "pnp::qeif_rows6_kernel<float>"
    # ... and through the allocating entry point, on host memory, the LDS-staged one: same bits
    assert _call(lib.pnp_qeif_batch_f32, uv2, own2, 18, o3, _lib.PTR_HOST, None) == 0, lib.pnp_last_error()
    assert _lib.last_launch()["kernel"] == "pnp::qeif_kernel<1, 6, float, false>"
    for k in o2:
        assert np.array_equal(o2[k], o3[k], equal_nan=True), k
    ref = c_oracle.qeif_batch(uv6.astype(np.float64), P6f.astype(np.float64), K)
    reps = fits // len(uv6)
    assert np.array_equal(o2["iters"][:reps * len(uv6)], np.tile(ref["iters"], reps))
    assert np.abs(o2["R"][:len(uv6)].reshape(-1, 3, 3) - ref["R"]).max() <= 5e-7
