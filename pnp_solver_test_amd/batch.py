"""Batched entry points over the C ABI: many independent PnP problems per call.

`solve_pnp_batch` is the batched counterpart of `PNP_SOLVER.solve_pnp`
(reference: scripts/PNP_SOLVER_LIB.py:144-203).  Inputs may be NumPy arrays (host memory:
the library stages them through device buffers) or torch tensors on a HIP device (device
memory: zero-copy, enqueued on torch's current stream).  Either way the arithmetic runs in
the HIP kernels of libpnp_hip.so; there is no CPU path.
"""
from __future__ import annotations

import ctypes

import numpy as np

from . import _lib

METHODS = ("qeif", "dlt", "lm12")


def _is_torch(x) -> bool:
    return type(x).__module__.split(".")[0] == "torch"


def make_params(**overrides) -> _lib.QeifParams:
    """QEIF tunables: reference defaults (PNP_SOLVER_LIB.py:2791-2794, :2836, :2845-2846,
    :2857, :2863, :2952) with keyword overrides, e.g. make_params(max_it=5, rel_tol=0.0)."""
    p = _lib.default_params()
    for k, v in overrides.items():
        if not hasattr(p, k):
            raise TypeError("unknown QEIF parameter %r" % k)
        setattr(p, k, v)
    return p


def _shapes(uv_shape, xyz_shape):
    if len(uv_shape) != 3 or uv_shape[2] != 2:
        raise ValueError("uv must have shape (B, N, 2), got %s" % (tuple(uv_shape),))
    B, N = int(uv_shape[0]), int(uv_shape[1])
    if tuple(xyz_shape) == (N, 3):
        return B, N, 0, 1
    if tuple(xyz_shape) == (B, N, 3):
        return B, N, 3 * N, 1
    raise ValueError("xyz must have shape (N, 3) or (B, N, 3) with B=%d, N=%d; got %s"
                     % (B, N, tuple(xyz_shape)))


def _np_dtype(uv, dtype):
    """Element type of a call: an explicit `dtype`, else float64 if `uv` already is float64
    (NumPy array or torch tensor), else float32."""
    if dtype is None:
        name = str(getattr(uv, "dtype", "")).replace("torch.", "")
        return np.dtype(np.float64 if name == "float64" else np.float32)
    dtype = np.dtype(str(dtype).replace("torch.", "") if type(dtype).__module__ == "torch" else dtype)
    if dtype not in (np.dtype(np.float32), np.dtype(np.float64)):
        raise TypeError("dtype must be float32 or float64, got %s" % dtype)
    return dtype


def solve_pnp_batch(uv, xyz, K, *, method="qeif", params=None, num_it=None, want_rpy=True,
                    patterns=None, lm_lambda=1e-5, dtype=None, want_res=True, out=None):
    """Solve B independent PnP problems.

    uv       (B, N, 2) pixel coordinates, float32 or float64
    xyz      (N, 3) pattern shared by all problems, or (B, N, 3) one pattern per problem
    K        (3, 3) camera matrix (float64, host)
    method   "qeif": quaternion information filter, the reference's live solver
             (solve_pnp_QEIF_single_pattern, :2771-3025);
             "dlt": iterated linear DLT (solve_pnp_single_pattern, :205-430)
             "lm12": the 12-state Levenberg-Marquardt solver (solve_pnp_LM_single_pattern,
             :2567-2769); it diverges on a sizeable share of off-axis poses, as the original does
    params   QEIF tunables from make_params(); None = reference defaults
    num_it   DLT rounds (reference: 3) / LM iterations (reference: 14; pass num_it=14)
    lm_lambda LM damping (reference: 1e-5)
    patterns optional (P, N, 3): QEIF against P shared patterns, keeping per problem the first
             with the smallest residual (the pattern loop of solve_pnp, :166-199); `xyz` is
             then ignored and the result gains "pattern_id"
    want_res False leaves out res_norm (the DLT kernel then skips the residual of its last round)
    dtype    element type of the arrays handed to the kernels and of the results: float32
             (half the bytes) or float64 (the reference's dtype; agrees with it to float64
             round-off).  None follows `uv`: float64 if it is float64, float32 otherwise.
             The arithmetic is float64 either way.
    out      the dict an earlier call with the same shapes, dtype, options and kind of arrays
             returned: its arrays are written again and returned, nothing is allocated (a fresh
             60 MB of NumPy results for a million problems costs about a millisecond of page
             faults).  ValueError if an array is missing or does not fit.

    Returns a dict: R (B,3,3), t (B,3), t3 (B,) (a view of t[:, 2]), res_norm (B,), rpy_deg (B,3) = roll, yaw, pitch
    in the reference's tuple order (if want_rpy), iters (B,) int32 (QEIF only).  Arrays are of
    the same kind as `uv`: NumPy, torch on the same device, or - for a host tensor in pinned
    memory - pinned host tensors (no copies then: the kernels work on that memory over the bus; a
    pinned view that is not contiguous is copied to pinned memory first).  Without `out=` every
    call allocates its result arrays - for pinned host tensors that is one hipHostMalloc per array
    and call - so repeated calls, pinned ones above all, should pass `out=`.
    """
    if method not in METHODS:
        raise ValueError("method must be one of %s" % (METHODS,))
    if patterns is not None and method != "qeif":
        raise ValueError("multiple patterns are supported by method='qeif' only")
    if num_it is None:
        num_it = 14 if method == "lm12" else 3
    lib = _lib.load()
    Kc = _lib.k_array(K)
    extra = (num_it, float(lm_lambda))
    ft = _np_dtype(uv, dtype)
    if _is_torch(uv):
        return _solve_torch(lib, uv, xyz, Kc, method, params, extra, want_rpy, patterns, ft, want_res, out)
    return _solve_numpy(lib, uv, xyz, Kc, method, params, extra, want_rpy, patterns, ft, want_res, out)


def _result_array(out, key, shape, make, fits):
    """out[key] if the caller handed in results to write again, else a new array from make()."""
    if out is None:
        return make()
    a = out.get(key)
    if a is None or tuple(a.shape) != tuple(shape) or not fits(a):
        raise ValueError("out[%r] is missing or is not the %s array this call writes" % (key, tuple(shape)))
    return a


def _solve_numpy(lib, uv, xyz, Kc, method, params, num_it, want_rpy, patterns, ft, want_res=True, reuse=None):
    uv = np.ascontiguousarray(uv, dtype=ft)
    src = patterns if patterns is not None else xyz
    src = np.ascontiguousarray(src, dtype=ft)
    if patterns is not None:
        if src.ndim != 3 or src.shape[1:] != (uv.shape[1], 3):
            raise ValueError("patterns must have shape (P, N, 3)")
        B, N, stride, n_pat = int(uv.shape[0]), int(uv.shape[1]), 0, int(src.shape[0])
    else:
        B, N, stride, n_pat = _shapes(uv.shape, src.shape)
    def result(key, shape, dt=ft):
        return _result_array(reuse, key, shape, lambda: np.empty(shape, dt),
                             lambda a: isinstance(a, np.ndarray) and a.dtype == dt and a.flags.c_contiguous
                             and a.flags.writeable)

    out = dict(R=result("R", (B, 3, 3)), t=result("t", (B, 3)))
    if want_res:
        out["res_norm"] = result("res_norm", (B,))
    rpy = result("rpy_deg", (B, 3)) if want_rpy else None
    iters = result("iters", (B,), np.dtype(np.int32)) if method == "qeif" else None
    pid = result("pattern_id", (B,), np.dtype(np.int32)) if patterns is not None else None

    def ptr(a):
        return None if a is None else a.ctypes.data

    _call(lib, method, ptr(uv), ptr(src), stride, n_pat, Kc, B, N, params, num_it, ptr(out["R"]),
          ptr(out["t"]), ptr(out.get("res_norm")), ptr(iters), ptr(rpy), ptr(pid), _lib.PTR_HOST, None,
          patterns is not None, ft)
    # t3 is a view of t, as on the torch path (a strided copy of a million values costs 0.4 ms);
    # a reused result dict keeps the view object it already holds
    t3 = out["t"][:, 2]
    held = None if reuse is None else reuse.get("t3")
    if (isinstance(held, np.ndarray) and held.shape == t3.shape and held.strides == t3.strides
            and held.__array_interface__["data"][0] == t3.__array_interface__["data"][0]):
        t3 = held
    out["t3"] = t3
    if rpy is not None:
        out["rpy_deg"] = rpy
    if iters is not None:
        out["iters"] = iters
    if pid is not None:
        out["pattern_id"] = pid
    return out


def _solve_torch(lib, uv, xyz, Kc, method, params, num_it, want_rpy, patterns, ft, want_res=True, reuse=None):
    import torch
    tt = torch.float64 if ft == np.dtype(np.float64) else torch.float32

    # Tensors on a HIP device, or host tensors in page-locked memory (pin_memory=True): the latter
    # go in as host pointers, the results come back as pinned host tensors, and with every array
    # page-locked the kernels work on that memory over the bus, without copies.
    on_host = not uv.is_cuda
    if on_host and not uv.is_pinned():
        raise ValueError("torch inputs must live on a HIP device or in pinned host memory (got an unpinned "
                         "%s tensor); pass NumPy arrays for pageable host memory" % uv.device)
    src = patterns if patterns is not None else xyz
    if not _is_torch(src):
        src = torch.as_tensor(np.ascontiguousarray(src, dtype=ft))
        src = src.pin_memory() if on_host else src.to(uv.device)
    elif on_host and not (src.device.type == "cpu" and src.is_pinned()):
        raise ValueError("with uv in pinned host memory, xyz must be a pinned host tensor or a NumPy array")
    if uv.dtype != tt or src.dtype != tt:
        raise TypeError("uv and xyz must both be %s tensors (got %s and %s)" % (tt, uv.dtype, src.dtype))
    if src.device != uv.device:
        raise ValueError("uv and xyz must be on the same device")
    uv = uv.contiguous()
    src = src.contiguous()
    if on_host:
        # .contiguous() of a pinned view that is not contiguous hands back a pageable copy, which
        # would send the call down the chunked path: pin the copy, so that what was promised for
        # pinned input (kernels working on that memory, no staging) holds for views as well
        uv = uv if uv.is_pinned() else uv.pin_memory()
        src = src if src.is_pinned() else src.pin_memory()
    if patterns is not None:
        if src.dim() != 3 or tuple(src.shape[1:]) != (uv.shape[1], 3):
            raise ValueError("patterns must have shape (P, N, 3)")
        B, N, stride, n_pat = int(uv.shape[0]), int(uv.shape[1]), 0, int(src.shape[0])
    else:
        B, N, stride, n_pat = _shapes(tuple(uv.shape), tuple(src.shape))
    dev = uv.device
    where = dict(pin_memory=True) if on_host else dict(device=dev)
    with (torch.cuda.device(torch.cuda.current_device()) if on_host else torch.cuda.device(dev)):
        def result(key, shape, dt=tt):
            return _result_array(reuse, key, shape, lambda: torch.empty(shape, dtype=dt, **where),
                                 lambda a: _is_torch(a) and a.dtype == dt and a.is_contiguous() and
                                 (a.device.type == "cpu" and a.is_pinned() if on_host else a.device == dev))

        out = dict(R=result("R", (B, 3, 3)), t=result("t", (B, 3)))
        if want_res:
            out["res_norm"] = result("res_norm", (B,))
        rpy = result("rpy_deg", (B, 3)) if want_rpy else None
        iters = result("iters", (B,), torch.int32) if method == "qeif" else None
        pid = result("pattern_id", (B,), torch.int32) if patterns is not None else None
        stream = torch.cuda.current_stream().cuda_stream

        def ptr(a):
            return None if a is None else a.data_ptr()

        _call(lib, method, ptr(uv), ptr(src), stride, n_pat, Kc, B, N, params, num_it, ptr(out["R"]),
              ptr(out["t"]), ptr(out.get("res_norm")), ptr(iters), ptr(rpy), ptr(pid),
              _lib.PTR_HOST if on_host else _lib.PTR_DEVICE, stream, patterns is not None, ft)
    out["t3"] = out["t"][:, 2]
    if rpy is not None:
        out["rpy_deg"] = rpy
    if iters is not None:
        out["iters"] = iters
    if pid is not None:
        out["pattern_id"] = pid
    return out


def _call(lib, method, uv, xyz, stride, n_pat, Kc, B, N, params, num_it, R, t, res, iters, rpy, pid,
          kind, stream, multi, ft):
    pref = ctypes.byref(params) if params is not None else None
    num_it, lm_lambda = num_it
    sfx = "f64" if ft == np.dtype(np.float64) else "f32"
    if method == "dlt":
        rc = getattr(lib, "pnp_dlt_batch_" + sfx)(uv, xyz, stride, Kc, B, N, int(num_it), R, t, res, rpy, kind,
                                                  stream)
    elif method == "lm12":
        rc = getattr(lib, "pnp_lm12_batch_" + sfx)(uv, xyz, stride, Kc, B, N, lm_lambda, int(num_it), R, t, res,
                                                   rpy, kind, stream)
    elif multi:
        rc = getattr(lib, "pnp_qeif_multi_batch_" + sfx)(uv, xyz, n_pat, Kc, B, N, pref, R, t, res, iters, rpy,
                                                         pid, kind, stream)
    else:
        rc = getattr(lib, "pnp_qeif_batch_" + sfx)(uv, xyz, stride, Kc, B, N, pref, R, t, res, iters, rpy, kind,
                                                   stream)
    _lib.check(rc)


def euler_deg_batch(R):
    """(B,3,3) or (B,9) rotation matrices -> (B,3) roll, yaw, pitch in degrees
    (get_Euler_from_rotation_matrix(..., is_degree=True), :4474-4517), on the device.
    float64 input gives float64 output; anything else is handled as float32."""
    lib = _lib.load()
    ft = _np_dtype(R, None)
    fn = lib.pnp_euler_deg_f64 if ft == np.dtype(np.float64) else lib.pnp_euler_deg_f32
    if _is_torch(R):
        import torch
        if not R.is_cuda or R.dtype not in (torch.float32, torch.float64):
            raise TypeError("R must be a float32 or float64 tensor on a HIP device")
        R = R.contiguous()
        B = R.numel() // 9
        with torch.cuda.device(R.device):
            out = torch.empty((B, 3), dtype=R.dtype, device=R.device)
            _lib.check(fn(R.data_ptr(), B, out.data_ptr(), _lib.PTR_DEVICE,
                          torch.cuda.current_stream(R.device).cuda_stream))
        return out
    R = np.ascontiguousarray(R, dtype=ft)
    B = R.size // 9
    out = np.empty((B, 3), ft)
    _lib.check(fn(R.ctypes.data, B, out.ctypes.data, _lib.PTR_HOST, None))
    return out


def shard_bounds(total: int, rank: int, world_size: int):
    """Contiguous slice [lo, hi) of `total` problems owned by `rank`: sizes differ by at most
    one and the slices tile [0, total) in rank order.  Problems are independent, so a
    multi-GPU run is this slice per rank with no exchange on the data path."""
    if not (0 <= rank < world_size):
        raise ValueError("rank %d outside [0, %d)" % (rank, world_size))
    base, extra = divmod(int(total), int(world_size))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


class RcclComm:
    """A communicator of libpnp_hip.so's own over RCCL (pnp_comm_init), for the optional result
    exchange through the C ABI, pnp_gather_results: no torch.distributed on the data path.  One
    per process; the process's HIP device must be current when it is created and used."""

    def __init__(self, rank: int, world: int, unique_id: bytes):
        lib = _lib.load()
        if len(unique_id) != _lib.COMM_ID_BYTES:
            raise ValueError("unique_id must be %d bytes" % _lib.COMM_ID_BYTES)
        self.rank, self.world = int(rank), int(world)
        self._comm = ctypes.c_void_p()
        buf = ctypes.create_string_buffer(unique_id, _lib.COMM_ID_BYTES)
        _lib.check(lib.pnp_comm_init(buf, self.rank, self.world, ctypes.byref(self._comm)))

    @staticmethod
    def unique_id() -> bytes:
        """On ONE rank; hand the bytes to the others (file, MPI, torch.distributed broadcast)."""
        buf = ctypes.create_string_buffer(_lib.COMM_ID_BYTES)
        _lib.check(_lib.load().pnp_comm_unique_id(buf))
        return buf.raw

    @classmethod
    def from_torch_group(cls, group=None):
        """Rank 0 of a torch.distributed group draws the id and broadcasts it; every rank joins."""
        import torch.distributed as dist
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        box = [cls.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        return cls(rank, world, box[0])

    def gather(self, val, dst=None):
        """val: a contiguous torch tensor on this process's HIP device, same shape on every rank.
        Returns the ranks' tensors concatenated along dim 0, on every rank (dst None) or on rank
        dst only (None elsewhere).  Enqueued on torch's current stream."""
        import torch
        if not val.is_cuda:
            raise ValueError("RcclComm.gather takes tensors on a HIP device")
        val = val.contiguous()
        receives = dst is None or dst == self.rank
        out = torch.empty((self.world * val.shape[0],) + tuple(val.shape[1:]), dtype=val.dtype,
                          device=val.device) if receives else None
        with torch.cuda.device(val.device):
            _lib.check(_lib.load().pnp_gather_results(
                val.data_ptr(), out.data_ptr() if receives else None, val.numel() * val.element_size(),
                -1 if dst is None else int(dst), self.rank, self.world, self._comm,
                torch.cuda.current_stream(val.device).cuda_stream))
        return out

    def close(self):
        if self._comm:
            _lib.check(_lib.load().pnp_comm_destroy(self._comm))
            self._comm = ctypes.c_void_p()


def _gather_through_comm(local: dict, comm: RcclComm, dst):
    import torch
    first = next(iter(local.values()))
    counts = comm.gather(torch.tensor([first.shape[0]], dtype=torch.int64, device=first.device)).tolist()
    n_max = max(counts)
    out = {}
    for key, val in local.items():
        val = val.contiguous()
        if val.shape[0] < n_max:
            pad = torch.zeros((n_max - val.shape[0],) + tuple(val.shape[1:]), dtype=val.dtype, device=val.device)
            val = torch.cat([val, pad], dim=0)
        buf = comm.gather(val, dst)
        if buf is None:
            continue
        out[key] = buf if min(counts) == n_max else torch.cat(
            [buf[r * n_max:r * n_max + counts[r]] for r in range(comm.world)], dim=0)
    return out if (dst is None or dst == comm.rank) else None


def gather_results(local: dict, group=None, dst=None, comm=None):
    """Optional collective after a sharded solve: every rank's result arrays (torch tensors) in
    rank order, on every rank (all-gather, the default) or on rank `dst` only (gather; the other
    ranks get None).  Gathering to one rank moves world-1 slices over world-1 different xGMI
    links at once instead of world-1 slices into every rank.  Slices may differ in length
    (shard_bounds hands out sizes that differ by one): shorter ones are padded for the collective
    and trimmed afterwards.  Uses torch.distributed, i.e. RCCL over xGMI for HIP tensors and gloo
    for CPU tensors - or, with comm= an RcclComm, the C ABI's pnp_gather_results (device tensors
    only, nothing of torch.distributed involved)."""
    if comm is not None:
        return _gather_through_comm(local, comm, dst)
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    first = next(iter(local.values()))
    n_local = torch.tensor([first.shape[0]], dtype=torch.int64, device=first.device)
    counts = torch.empty(world, dtype=torch.int64, device=first.device)
    dist.all_gather_into_tensor(counts, n_local, group=group)
    counts = counts.tolist()
    n_max = max(counts)
    out = {}
    for key, val in local.items():
        val = val.contiguous()
        if val.shape[0] < n_max:
            pad = torch.zeros((n_max - val.shape[0],) + tuple(val.shape[1:]), dtype=val.dtype, device=val.device)
            val = torch.cat([val, pad], dim=0)
        if dst is None:
            buf = torch.empty((world * n_max,) + tuple(val.shape[1:]), dtype=val.dtype, device=val.device)
            dist.all_gather_into_tensor(buf, val, group=group)
        else:
            buf = None
            parts = None
            if rank == dst:
                buf = torch.empty((world * n_max,) + tuple(val.shape[1:]), dtype=val.dtype, device=val.device)
                parts = list(buf.chunk(world, dim=0))
            dist.gather(val, gather_list=parts, dst=dist.get_global_rank(group, dst) if group is not None else dst,
                        group=group)
            if rank != dst:
                continue
        if min(counts) == n_max:
            out[key] = buf
        else:
            out[key] = torch.cat([buf[r * n_max:r * n_max + counts[r]] for r in range(world)], dim=0)
    return out if (dst is None or rank == dst) else None


def project_batch(roll, yaw, pitch, t, xyz, K, quantize_q=0.0, device=None, dtype=np.float32):
    """Pixels of a pattern under B poses, on the device: the generator half of the reference's
    drivers (get_rotation_matrix_from_Euler :4442-4472 with angles in degrees, then
    perspective_projection :4532-4557).  roll, yaw, pitch (B,), t (B,3), xyz (N,3) or (B,N,3),
    all float64; quantize_q = 1.0 reproduces is_quantized=True.  Returns uv (B,N,2) of `dtype`
    (float32 or float64): a NumPy array, or a torch tensor on `device` if one is given."""
    lib = _lib.load()
    ft = _np_dtype(None, dtype)
    is64 = int(ft == np.dtype(np.float64))
    rpy = np.ascontiguousarray(np.stack([roll, yaw, pitch], axis=1), dtype=np.float64)
    t = np.ascontiguousarray(t, dtype=np.float64)
    xyz = np.ascontiguousarray(xyz, dtype=np.float64)
    B = rpy.shape[0]
    if t.shape != (B, 3):
        raise ValueError("t must have shape (B, 3)")
    if xyz.ndim == 2 and xyz.shape[1] == 3:
        N, stride = xyz.shape[0], 0
    elif xyz.ndim == 3 and xyz.shape[0] == B and xyz.shape[2] == 3:
        N, stride = xyz.shape[1], 3 * xyz.shape[1]
    else:
        raise ValueError("xyz must have shape (N, 3) or (B, N, 3)")
    Kc = _lib.k_array(K)
    if device is None:
        uv = np.empty((B, N, 2), ft)
        _lib.check(lib.pnp_project_batch_f64(rpy.ctypes.data, t.ctypes.data, xyz.ctypes.data, stride, Kc, B, N,
                                             float(quantize_q), uv.ctypes.data, is64, _lib.PTR_HOST, None))
        return uv
    import torch
    dev = torch.device(device)
    with torch.cuda.device(dev):
        d_rpy, d_t, d_xyz = (torch.from_numpy(a).to(dev) for a in (rpy, t, xyz))
        uv = torch.empty((B, N, 2), dtype=torch.float64 if is64 else torch.float32, device=dev)
        _lib.check(lib.pnp_project_batch_f64(d_rpy.data_ptr(), d_t.data_ptr(), d_xyz.data_ptr(), stride, Kc, B, N,
                                             float(quantize_q), uv.data_ptr(), is64, _lib.PTR_DEVICE,
                                             torch.cuda.current_stream(dev).cuda_stream))
        torch.cuda.current_stream(dev).synchronize()   # the float64 inputs above are temporaries
    return uv


def compare_batch(R, t, rpy_gt_deg, distance_gt_cm, xyz, K, uv_obs=None, bound=(10.0, 10.0, 10.0, 10.0),
                  normalize=True):
    """Comparison epilogue of the reference's stress driver on B solved samples, on the device:
    pass flags (TEST_TOOLBOX.check_if_the_sample_passed, :55-62) and the landmark error sets of
    compare_result_and_generate_result_dict (:312-343, :453-463), with the Euler angles of the
    estimate computed in the same kernel.

    R (B,3,3), t (B,3)   solver outputs, float32 or float64, NumPy or torch on a HIP device
    rpy_gt_deg (B,3)     ground-truth roll, yaw, pitch in degrees;  distance_gt_cm (B,) depth in cm
    xyz (M,3)            the golden pattern;  uv_obs (B,M,2) the solved pixels of all M points, same
                         kind and dtype as R (None: the ground-truth projections stand in)
    Returns a dict of the same kind as R: drpy_err (B,4) = estimate - truth for depth[cm], roll,
    pitch, yaw; passed (B,4) bool; lm_err_avg / lm_err_max (B,3) for observed-vs-truth,
    reprojection-vs-observed, reprojection-vs-truth (scaled by the true depth in metres if
    `normalize`); lm_err_key (B,3) index of the worst landmark (-1 if all distances are 0)."""
    lib = _lib.load()
    ft = _np_dtype(R, None)
    fn = lib.pnp_compare_batch_f64 if ft == np.dtype(np.float64) else lib.pnp_compare_batch_f32
    Kc = _lib.k_array(K)
    bd = (ctypes.c_double * 4)(*[float(v) for v in bound])
    rpy = np.ascontiguousarray(rpy_gt_deg, dtype=np.float64)
    dist = np.ascontiguousarray(distance_gt_cm, dtype=np.float64)
    xyz = np.ascontiguousarray(xyz, dtype=np.float64)
    B, M = int(dist.shape[0]), int(xyz.shape[0])
    if rpy.shape != (B, 3) or xyz.shape != (M, 3):
        raise ValueError("rpy_gt_deg must be (B, 3) and xyz (M, 3)")
    bits = np.arange(4, dtype=np.int32)
    if _is_torch(R):
        import torch
        dev = R.device
        if not R.is_cuda or t.dtype != R.dtype or (uv_obs is not None and uv_obs.dtype != R.dtype):
            raise TypeError("R, t and uv_obs must be tensors of one dtype on a HIP device")
        if R.numel() != B * 9 or t.numel() != B * 3 or (uv_obs is not None and uv_obs.numel() != B * M * 2):
            raise ValueError("R, t, uv_obs do not match B = %d, M = %d" % (B, M))
        R, t = R.contiguous(), t.contiguous()
        uv_obs = None if uv_obs is None else uv_obs.contiguous()
        with torch.cuda.device(dev):
            d_rpy, d_dist, d_xyz = (torch.from_numpy(a).to(dev) for a in (rpy, dist, xyz))
            err = torch.empty((B, 4), dtype=R.dtype, device=dev)
            lm = torch.empty((B, 6), dtype=R.dtype, device=dev)
            mask = torch.empty(B, dtype=torch.int32, device=dev)
            key = torch.empty((B, 3), dtype=torch.int32, device=dev)
            stream = torch.cuda.current_stream(dev)
            _lib.check(fn(R.data_ptr(), t.data_ptr(), None if uv_obs is None else uv_obs.data_ptr(),
                          d_rpy.data_ptr(), d_dist.data_ptr(), d_xyz.data_ptr(), Kc, B, M, bd, int(bool(normalize)),
                          err.data_ptr(), mask.data_ptr(), lm.data_ptr(), key.data_ptr(), _lib.PTR_DEVICE,
                          stream.cuda_stream))
            stream.synchronize()                       # the float64 uploads above are temporaries
        passed = (mask[:, None] >> torch.from_numpy(bits).to(dev)[None, :]) & 1
        return dict(drpy_err=err, passed=passed.bool(), lm_err_avg=lm[:, 0::2], lm_err_max=lm[:, 1::2],
                    lm_err_key=key)
    R = np.ascontiguousarray(R, dtype=ft)
    t = np.ascontiguousarray(t, dtype=ft)
    uv_obs = None if uv_obs is None else np.ascontiguousarray(uv_obs, dtype=ft)
    if R.size != B * 9 or t.size != B * 3 or (uv_obs is not None and uv_obs.size != B * M * 2):
        raise ValueError("R, t, uv_obs do not match B = %d, M = %d" % (B, M))
    err, lm = np.empty((B, 4), ft), np.empty((B, 6), ft)
    mask, key = np.empty(B, np.int32), np.empty((B, 3), np.int32)
    _lib.check(fn(R.ctypes.data, t.ctypes.data, None if uv_obs is None else uv_obs.ctypes.data, rpy.ctypes.data,
                  dist.ctypes.data, xyz.ctypes.data, Kc, B, M, bd, int(bool(normalize)), err.ctypes.data,
                  mask.ctypes.data, lm.ctypes.data, key.ctypes.data, _lib.PTR_HOST, None))
    passed = ((mask[:, None] >> bits[None, :]) & 1).astype(bool)
    return dict(drpy_err=err, passed=passed, lm_err_avg=lm[:, 0::2], lm_err_max=lm[:, 1::2], lm_err_key=key)
