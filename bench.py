#!/usr/bin/env python3
"""Throughput of the batched PnP hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]

One "step" is one pass of the hot path over one batch of synthetic problems that already sit
in HBM.  The default workload is the configuration the headline metric is quoted on:
"Batched PnP + LM refine, N=6, batch=1M fp32" = the quaternion information-filter solver
(QEIF, the reference's live solve_pnp path) with its reference defaults on 1,048,576
six-point problems per GPU, every problem carrying its own 3-D points (uv 8N + xyz 12N bytes
in, R 36 + t 12 bytes out = 168 algorithmic bytes per solve).  With N > 1 ranks every rank
solves its own 1M-problem slice - weak scaling, no collective on the data path - and the
reported value is all ranks' solves over the slowest rank's time.  The ranks are either started
from outside (`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...`: RANK is
set and this process is one of them) or, for a plain `python bench.py --gpus N`, by this process
itself: before it imports torch or touches a device it starts that same torch.distributed.run
command as a child, relays the child's output and exits with its status.

Rank 0 prints ONE JSON line.  Besides the contract fields it carries
  roofline      algorithmic HBM bytes of the dominant kernel / its mean launch time (HIP events
                on the launch stream inside the timed region) against the 8 TB/s HBM peak
  roofline_valu the same launches priced in float64 flops against the vector-FP64 peak, since
                this kernel is bound by the vector ALU, not by HBM
  roofline_issue the same launches priced in issue slots: vector instructions per wavefront (from
                the counter passes) x 4 cycles x wavefronts per SIMD against the cycles of the launch
The kernel names in the line are what the library says it dispatched (pnp_last_kernel_name), not
literals; `traffic` comes from the counter passes under profiles/ (traffic_source says which file,
which commit produced it, and whether it was taken on the kernel that ran here).
  cpu_baseline  the NumPy per-problem restatement of the reference (oracle port) timed on one
                host core over a bounded sample of the same workload
  other_configs short measurements of the remaining baseline configurations (not the headline)
"""
import argparse
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E, spec
FP64_VECTOR_PEAK_TF = 78.6     # MI355X vector FP64, spec (half the 157.3 TF vector-FP32 rate)
BATCH = 1 << 20
N_POINTS = 6


MIN_SAMPLES = 8       # at least this many timed launches carry an event pair of their own
SIMDS = 1024          # 256 compute units x 4
ISSUE_CLOCK_GHZ = 2.4 # MI355X peak engine clock: one vector instruction of a wavefront takes 4 cycles of its SIMD's issue port
SPINUP_MS = 60.0     # every measurement is preceded by at least this much untimed GPU work
GATHER_DEADLINE_S = 120   # N > 1: the optional result exchange may take this long, then the line goes out without it


def spin_up(torch, fn, min_ms=SPINUP_MS):
    """Clock spin-up, part of set-up and reported as such: an idle MI355X needs some tens of
    milliseconds of work before it holds its compute clock.  Launches `fn` in groups of 16 until
    `min_ms` of wall-clock have passed with the device busy; returns the number of launches.  The
    headline and every variant beside it use this one rule."""
    n = 0
    t0 = time.perf_counter()
    while True:
        for _ in range(16):
            fn()
        n += 16
        torch.cuda.synchronize()
        if (time.perf_counter() - t0) * 1e3 >= min_ms:
            return n


def qeif_flops(n_points, iters_mean):
    """float64 flops the point-form kernel executes per solve (fma = 2, multiply / add /
    reciprocal seed = 1), counted from its instruction listing.  Per iteration: 64 fma + 6 single
    operations per point (shared Jacobian products, residual from the Jacobian row, sums straight
    into the information matrix) = 134 flops; 501 flops for the rest (N = 6 loop body: 582 fma,
    104 multiplies, 25 adds, 12 reciprocal seeds): one packed 6x6 symmetric sweep with the
    21-entry rescaling of the predict step, one LDL^T solve, state update and stopping rule.  The
    first iteration starts from q = (1,0,0,0) with a diagonal information matrix: 97 flops per
    point and no sweep."""
    rest = 134.0 * n_points + 501.0
    first = 97.0 * n_points + 225.0
    return max(iters_mean - 1.0, 0.0) * rest + first


def qeif_moment_flops(n_points, iters_mean):
    """float64 flops (fma = 2) per solve of the moment form: 49 fused ops per point once, then
    per iteration ~650 for H^T H, H^T r and |r|^2 from the 40 sums, and ~280 for the predict
    sweep, the LDL^T solve, the update and the stopping rule."""
    return 2.0 * (49.0 * n_points + iters_mean * (650.0 + 280.0))


def dlt_flops(n_points, num_it=3):
    return 2.0 * (num_it * (30.0 * n_points + 110.0) + 20.0 * n_points)


class CounterTable:
    """profiles/hbm_traffic.json: what the rocprofv3 --pmc passes measured per launch (HBM bytes,
    vector instructions per wavefront), with the kernel each row was taken on and the commit the
    passes ran at.  A row is only handed out for the kernel it was measured on."""

    def __init__(self):
        self.path = os.path.join("profiles", "hbm_traffic.json")
        self.table = {}
        full = os.path.join(ROOT, self.path)
        if os.path.exists(full):
            with open(full) as f:
                self.table = json.load(f)

    def row(self, label, kernel):
        """(traffic bytes per launch | None, vector instructions per wavefront | None, source dict)"""
        measured_on = (self.table.get("kernels") or {}).get(label)
        src = dict(file=self.path, passes=self.table.get("source"), commit=self.table.get("commit"), row=label,
                   measured_on_kernel=measured_on, this_run_kernel=kernel)
        if measured_on is not None and kernel and measured_on != kernel:
            src["note"] = "the counters were taken on another kernel than the one this run dispatched: not reported"
            return None, None, src
        return (self.table.get(label + "_bytes_per_launch"), self.table.get(label + "_valu_insts_per_wave"), src)


def issue_roofline(insts_per_wave, waves, kernel_ms):
    """Issue-slot view of a launch: every vector instruction of a wavefront occupies its SIMD's
    issue port for 4 cycles; achieved = cycles of instruction issue per SIMD, peak = the cycles the
    launch lasted at the peak engine clock."""
    if not insts_per_wave:
        return None
    issue = insts_per_wave * 4.0 * waves / SIMDS
    avail = kernel_ms * 1e-3 * ISSUE_CLOCK_GHZ * 1e9
    return dict(bound="valu_issue", achieved=round(issue), peak=round(avail), unit="cycles per SIMD",
                frac=round(issue / avail, 4), valu_insts_per_wave=insts_per_wave, wavefronts=waves,
                clock_GHz=ISSUE_CLOCK_GHZ)


def time_launches(torch, fn, steps, warmup=8):
    """Mean per-launch device time (ms): one HIP event pair on the current (= launch) stream
    around `steps` back-to-back launches, after the common spin-up and `warmup` more launches."""
    spin_up(torch, fn)
    for _ in range(warmup):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(steps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / steps


def cpu_baseline(uv, xyz, K):
    """The reference's algorithm restated in NumPy (oracle/pnp_oracle.py), one problem at a
    time on one core, on the first problems of this very workload, for about 10 s."""
    from oracle import pnp_oracle as po
    t0 = time.perf_counter()
    n = 0
    while n < len(uv) and (time.perf_counter() - t0 < 10.0 or n < 64):
        po.qeif_solve(uv[n].astype(np.float64), xyz[n].astype(np.float64), K)
        n += 1
    dt = time.perf_counter() - t0
    return dict(value=round(n / dt, 1), unit="solves/s", cores=1, kind="port",
                sample="first %d problems of the workload, NumPy float64 per-problem loop "
                       "(oracle/pnp_oracle.qeif_solve), %.1f s" % (n, dt),
                host_cpus=os.cpu_count(), host=host_description())


def host_description():
    """CPU model and NumPy / BLAS versions of the box the CPU baselines ran on."""
    model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    blas = "unknown"
    try:
        cfg = np.show_config(mode="dicts")
        dep = cfg.get("Build Dependencies", {}).get("blas", {})
        blas = "%s %s" % (dep.get("name", "?"), dep.get("version", "?"))
    except Exception:
        pass
    return "%s; numpy %s; %s" % (model, np.__version__, blas)


def cpu_baseline_c_threads(uv, xyz, K, threads):
    """The scalar C restatement on `threads` host threads, one contiguous slice each (the C call
    releases the interpreter lock; problems are independent)."""
    from concurrent.futures import ThreadPoolExecutor

    from oracle import c_oracle
    n = min(len(uv), 1 << 20)
    u64, x64 = uv[:n].astype(np.float64), xyz[:n].astype(np.float64)
    bounds = [(i * n // threads, (i + 1) * n // threads) for i in range(threads)]
    c_oracle.qeif_batch(u64[:64], x64[:64], K)
    t0 = time.perf_counter()
    with ThreadPoolExecutor(threads) as ex:
        list(ex.map(lambda b: c_oracle.qeif_batch(u64[b[0]:b[1]], x64[b[0]:b[1]], K), bounds))
    dt = time.perf_counter() - t0
    return dict(value=round(n / dt, 1), unit="solves/s", cores=threads, kind="port",
                sample="first %d problems in %d slices, scalar C float64 (oracle/pnp_oracle.c), one thread per "
                       "slice, %.2f s" % (n, threads, dt))


def cpu_baseline_numpy_batched(uv, xyz, K):
    """The same recurrence in array operations over many problems at once (what a NumPy user
    would write instead of the per-problem loop), chunks of 32 768 problems for about 5 s."""
    from oracle import pnp_oracle as po
    chunk, n = 1 << 15, 0
    po.qeif_solve_batch(uv[:256].astype(np.float64), xyz[:256].astype(np.float64), K)
    t0 = time.perf_counter()
    while n < len(uv) and (time.perf_counter() - t0 < 5.0 or n == 0):
        m = min(chunk, len(uv) - n)
        po.qeif_solve_batch(uv[n:n + m].astype(np.float64), xyz[n:n + m].astype(np.float64), K)
        n += m
    dt = time.perf_counter() - t0
    return dict(value=round(n / dt, 1), unit="solves/s", cores=1, kind="port",
                sample="first %d problems in chunks of %d, NumPy float64 array operations over the batch "
                       "(oracle/pnp_oracle.qeif_solve_batch), one process, %.1f s" % (n, chunk, dt))


def cpu_baseline_c(uv, xyz, K):
    """Same sample idea with the scalar C restatement (oracle/pnp_oracle.c), one core."""
    from oracle import c_oracle
    n = min(len(uv), 1 << 17)
    t0 = time.perf_counter()
    c_oracle.qeif_batch(uv[:n].astype(np.float64), xyz[:n].astype(np.float64), K)
    dt = time.perf_counter() - t0
    return dict(value=round(n / dt, 1), unit="solves/s", cores=1, kind="port",
                sample="first %d problems, scalar C float64 (oracle/pnp_oracle.c), %.1f s" % (n, dt))


def free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def launch_ranks(n_ranks, argv):
    """Starts the N ranks of `python bench.py --gpus N` as children of this process:
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py <argv>`.  Called
    before this process has imported torch or made any HIP call, and it never replaces itself by
    another program: it waits for the child, relays its output (rank 0's JSON line is the last
    line of stdout) and returns the child's exit status."""
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # RCCL needs dmabuf IPC on this driver
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % n_ranks,
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + list(argv)
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    sys.stderr.write(proc.stderr)
    lines = [ln for ln in proc.stdout.splitlines() if ln.strip()]
    json_lines = [ln for ln in lines if ln.lstrip().startswith("{") and '"metric"' in ln]
    for ln in lines:
        if ln not in json_lines:
            print(ln, file=sys.stderr)
    if proc.returncode != 0:
        # the one-line reason a rank gave for leaving (SystemExit messages start with "bench.py:")
        reasons = [ln.strip() for ln in proc.stderr.splitlines() if ln.strip().startswith("bench.py:")]
        print("bench.py --gpus %d failed (exit %d): %s" % (n_ranks, proc.returncode,
                                                           reasons[0][len("bench.py:"):].strip() if reasons else
                                                           "see the ranks' output above"), file=sys.stderr)
        return proc.returncode
    if json_lines:
        print(json_lines[-1], flush=True)
        return 0
    print("bench.py: the ranks exited without printing a result line", file=sys.stderr)
    return 1


def launcher_check(args):
    """The rank plumbing of an N-rank run without the solver, on the gloo backend (no GPU
    needed): rendezvous from the environment torch.distributed.run sets, this rank's slice of the
    job, the barriers around the timed region and the max-over-ranks time.  Rank 0 prints a line
    of the usual shape with "launcher_check": true and no value."""
    import torch
    import torch.distributed as dist

    import pnp_solver_test_amd as pnp
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    dist.init_process_group("gloo")
    lo, hi = pnp.shard_bounds(world * args.batch, rank, world)
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        pass
    dist.barrier()
    tmax = torch.tensor([time.perf_counter() - t0 + rank], dtype=torch.float64)
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    sizes = torch.zeros(world, dtype=torch.int64)
    sizes[rank] = hi - lo
    dist.all_reduce(sizes)
    if rank == 0:
        print(json.dumps({"metric": "PnP solves/sec (N=6, LM 5 iters)", "value": None, "unit": "solves/s",
                          "n_gpus": world, "rccl_ranks": 0, "backend": "gloo", "launcher_check": True,
                          "steps": args.steps, "warmup": args.warmup, "slice_sizes": sizes.tolist(),
                          "slowest_rank_offset_s": int(tmax.item())}), flush=True)
    dist.barrier()
    dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--batch", type=int, default=BATCH, help="problems per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true")
    ap.add_argument("--no-c-abi-gather", action="store_true",
                    help="N > 1: time the result exchange through torch.distributed only")
    ap.add_argument("--launcher-check", action="store_true",
                    help="run only the N-rank plumbing on the gloo backend (no GPU, no solver)")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")

    if args.gpus > 1 and "RANK" not in os.environ:
        # nothing in this process has touched torch or HIP yet
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))
    if args.launcher_check:
        if "RANK" not in os.environ:
            os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1",
                              MASTER_PORT=str(free_port()))
        return launcher_check(args)

    import torch
    import torch.distributed as dist

    import pnp_solver_test_amd as pnp
    from pnp_solver_test_amd import synth

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py: no HIP device is visible; the solver has no CPU path")
    if local_rank >= torch.cuda.device_count():
        raise SystemExit("bench.py: rank %d wants device %d but only %d are visible" % (rank, local_rank, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    distributed = "RANK" in os.environ          # one of the ranks of torch.distributed.run, any world size
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", device_id=dev)

    K = synth.K_DEFAULT
    B = args.batch
    # this rank's slice of the (world*B)-problem job: own seed stream, own device buffers
    lo, hi = pnp.shard_bounds(world * B, rank, world)
    uv_h, _ = synth.stress_batch(hi - lo, seed=42 + rank, pattern=synth.PATTERN6, K=K, is_quantized=True)
    xyz_h = np.ascontiguousarray(np.broadcast_to(synth.PATTERN6.astype(np.float32), (hi - lo, N_POINTS, 3)))
    uv = torch.from_numpy(uv_h).to(dev)
    xyz = torch.from_numpy(xyz_h).to(dev)

    def step():
        return pnp.solve_pnp_batch(uv, xyz, K, method="qeif", want_rpy=False)

    def barrier():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        out = step()
    barrier()
    # The opening barrier of an N-rank run (a collective, on first use the communicator's set-up,
    # and a host round trip) leaves the device idle long enough for its clock to drop, so the
    # spin-up comes after it: it ends in a bare synchronize, and the clock starts right there.
    spinup_steps = spin_up(torch, step)
    # One HIP event pair on the launch stream around the K launches: its span / K is the mean
    # launch duration, gaps between launches included.  (A pair around every launch puts two
    # markers between consecutive kernels and costs the loop 3 % of wall-clock, so only every
    # sample_every-th launch carries a pair of its own - at least MIN_SAMPLES of them whatever
    # --steps is - for the minimum and the median.)
    ev_a, ev_b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    samples = []
    sample_every = max(1, args.steps // MIN_SAMPLES)
    t0 = time.perf_counter()
    ev_a.record()
    for i in range(args.steps):
        if i % sample_every == sample_every - 1:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = step()
            e1.record()
            samples.append((e0, e1))
        else:
            out = step()
    ev_b.record()
    enqueue_s = time.perf_counter() - t0        # host time to queue the K launches
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0          # this rank's K steps, start to completion
    launched = pnp._lib.last_launch()           # what the last of those steps dispatched, from the library
    barrier()                                   # every rank is done; the job's time is the slowest rank's
    kernel_ms = ev_a.elapsed_time(ev_b) / args.steps
    sampled = sorted(a.elapsed_time(b) for a, b in samples)
    per_rank = None
    if distributed:
        mine = torch.tensor([elapsed, kernel_ms], dtype=torch.float64, device=dev)
        everyone = torch.empty((world, 2), dtype=torch.float64, device=dev)
        dist.all_gather_into_tensor(everyone, mine)
        everyone = everyone.cpu()
        per_rank = dict(elapsed_ms_per_step=[round(float(v) / args.steps * 1e3, 4) for v in everyone[:, 0]],
                        kernel_ms=[round(float(v), 4) for v in everyone[:, 1]],
                        slowest_rank=int(everyone[:, 0].argmax().item()))
        elapsed = float(everyone[:, 0].max().item())
    iters_mean = float(out["iters"].float().mean().item())
    ok = bool(torch.isfinite(out["R"]).all().item())

    line = None
    if rank == 0:
        total = world * B * args.steps
        value = total / elapsed
        bytes_per_solve = 20 * N_POINTS + 48
        achieved = B * bytes_per_solve / (kernel_ms * 1e-3) / 1e9
        flops = qeif_flops(N_POINTS, iters_mean)
        counters = CounterTable()
        traffic, insts, traffic_source = counters.row("qeif_n6_b1m_per_problem_xyz", launched["kernel"])
        waves = -(-B // 64)
        line = {
            "metric": "PnP solves/sec (N=6, LM 5 iters)",
            "value": round(value, 1),
            "unit": "solves/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "rccl_ranks": world if distributed else 0,
            "spinup_steps": spinup_steps,
            "spinup_rule": ">= %.0f ms of untimed launches before every measurement" % SPINUP_MS,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "host_enqueue_ms_per_step": round(enqueue_s / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "io_dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": "BASELINE configs[2]: batched PnP + LM refine = QEIF solver, reference defaults "
                            "(data-dependent early exit, cap 14), N=6, %d problems per GPU, float32 I/O, "
                            "per-problem 3-D points" % B,
                "batch_per_gpu": B,
                "n_points": N_POINTS,
                "mean_iterations": round(iters_mean, 3),
                "inputs": "random-stress pose distribution, seed 42+rank, pixel-quantised",
                "sharding": "contiguous per-rank slices, no data-path collective",
                "all_outputs_finite": ok,
            },
            "roofline": {
                "bound": "hbm",
                "achieved": round(achieved, 2),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 5),
                "traffic": traffic,
                "traffic_source": traffic_source,
                "kernel": launched["kernel"],
                "kernel_source": "pnp_last_kernel_name() after the last timed step",
                "launch": {k: launched[k] for k in ("grid", "block", "lds_bytes")},
                "kernel_ms": round(kernel_ms, 4),
                "kernel_ms_min": round(sampled[0], 4) if sampled else None,
                "kernel_ms_median": round(sampled[len(sampled) // 2], 4) if sampled else None,
                "kernel_ms_samples": len(sampled),
                "bytes_per_solve": bytes_per_solve,
                "note": "168 B and ~%.0f float64 flops per solve (%.0f flop/B): the kernel is bound by the "
                        "vector ALU, see roofline_valu" % (flops, flops / bytes_per_solve),
            },
            "roofline_valu": {
                "bound": "valu_fp64",
                "achieved": round(B * flops / (kernel_ms * 1e-3) / 1e12, 3),
                "peak": FP64_VECTOR_PEAK_TF,
                "unit": "TFLOP/s",
                "frac": round(B * flops / (kernel_ms * 1e-3) / 1e12 / FP64_VECTOR_PEAK_TF, 4),
                "flops_per_solve": round(flops, 1),
            },
            "roofline_issue": issue_roofline(insts, waves, kernel_ms),
            "library": {"version": pnp.version(), "path": os.path.relpath(pnp._lib.LIB_PATH, ROOT)},
        }
        if per_rank is not None:
            line["per_rank"] = per_rank
        if world == 1 and not args.no_other_configs:
            other = {}
            n_small = max(5, min(50, args.steps // 2))
            # every variant gets the same spin-up as the headline (time_launches): the host work
            # between two variants leaves the GPU idle long enough for its clock to drop
            P6 = torch.from_numpy(synth.PATTERN6.astype(np.float32)).to(dev)
            ms = time_launches(torch, lambda: pnp.solve_pnp_batch(uv, P6, K, want_rpy=False), n_small)
            other["qeif_n6_shared_pattern"] = dict(solves_per_s=round(B / ms * 1e3, 1), kernel_ms=round(ms, 4),
                                                   bytes_per_solve=8 * N_POINTS + 48)
            p5 = pnp.make_params(max_it=5, rel_tol=0.0)
            ms = time_launches(torch, lambda: pnp.solve_pnp_batch(uv, xyz, K, params=p5, want_rpy=False), n_small)
            other["qeif_n6_fixed_5_iterations"] = dict(solves_per_s=round(B / ms * 1e3, 1), kernel_ms=round(ms, 4))
            ms = time_launches(torch, lambda: pnp.solve_pnp_batch(uv, xyz, K, want_rpy=True), n_small)
            other["qeif_n6_with_euler_output"] = dict(solves_per_s=round(B / ms * 1e3, 1), kernel_ms=round(ms, 4))
            ms = time_launches(torch, lambda: pnp.solve_pnp_batch(uv, xyz, K, method="dlt", want_rpy=False), n_small)
            ran = pnp._lib.last_launch()
            gbs = B * bytes_per_solve / (ms * 1e-3) / 1e9
            tr, ins, tsrc = counters.row("dlt_n6_b1m_per_problem_xyz", ran["kernel"])
            other["configs[1]_dlt_n6_b1m"] = dict(
                solves_per_s=round(B / ms * 1e3, 1), kernel_ms=round(ms, 4),
                achieved_GBps=round(gbs, 1), hbm_frac=round(gbs / HBM_PEAK_GBS, 4),
                valu_fp64_TFLOPs=round(B * dlt_flops(6) / (ms * 1e-3) / 1e12, 2),
                roofline=dict(bound="hbm", achieved=round(gbs, 2), peak=HBM_PEAK_GBS, unit="GB/s",
                              frac=round(gbs / HBM_PEAK_GBS, 5), traffic=tr, traffic_source=tsrc,
                              kernel=ran["kernel"], launch={k: ran[k] for k in ("grid", "block", "lds_bytes")},
                              kernel_ms=round(ms, 4), bytes_per_solve=bytes_per_solve),
                roofline_issue=issue_roofline(ins, waves, ms),
                achievable_hbm=dict(GBps=6300.0, frac=round(gbs / 6300.0, 4),
                                    what="against the 6.3 TB/s a streaming read-and-write kernel reaches on this chip"))
            # ... and with the optional residual left out: exactly the 168 bytes (uv, xyz in; R, t out)
            ms = time_launches(torch, lambda: pnp.solve_pnp_batch(uv, xyz, K, method="dlt", want_rpy=False, want_res=False),
                               n_small)
            gbs = B * bytes_per_solve / (ms * 1e-3) / 1e9
            other["configs[1]_dlt_n6_b1m_pose_only"] = dict(
                solves_per_s=round(B / ms * 1e3, 1), kernel_ms=round(ms, 4), achieved_GBps=round(gbs, 1),
                hbm_frac=round(gbs / HBM_PEAK_GBS, 4),
                what="res_out = NULL: the kernel skips the residual of its last round (the reference returns it, so the entry above is the drop-in's launch)")
            B_lm = 1 << 18
            ms = time_launches(torch, lambda: pnp.solve_pnp_batch(uv[:B_lm], xyz[:B_lm], K, method="lm12", want_rpy=False),
                               10)
            other["lm12_n6_b256k (secondary solver)"] = dict(solves_per_s=round(B_lm / ms * 1e3, 1), kernel_ms=round(ms, 4))
            B68 = 1 << 18
            P68 = synth.face68()
            uv68_h, _ = synth.stress_batch(B68, seed=46, pattern=P68, K=K, is_quantized=True, chunk=1 << 15)
            uv68 = torch.from_numpy(uv68_h).to(dev)
            P68d = torch.from_numpy(P68.astype(np.float32)).to(dev)
            lib = pnp._lib.load()
            o68 = pnp.solve_pnp_batch(uv68, P68d, K, want_rpy=False)
            it68 = float(o68["iters"].float().mean().item())
            own68 = P68d[None].expand(B68, 68, 3).contiguous()
            for tag, xyz68, nbytes in (("shared_pattern", P68d, 8 * 68 + 48), ("per_problem_pattern", own68, 20 * 68 + 48)):
                ms = time_launches(torch, lambda: pnp.solve_pnp_batch(uv68, xyz68, K, want_rpy=False), n_small)
                ran = pnp._lib.last_launch()
                gbs = B68 * nbytes / (ms * 1e-3) / 1e9
                tr, ins, tsrc = counters.row("qeif_n68_b256k_%s" % ("shared_pattern" if tag == "shared_pattern" else "per_problem_xyz"),
                                             ran["kernel"])
                other["configs[3]_qeif_n68_b256k_" + tag] = dict(
                    solves_per_s=round(B68 / ms * 1e3, 1), kernel_ms=round(ms, 4), mean_iterations=round(it68, 3),
                    bytes_per_solve=nbytes, achieved_GBps=round(gbs, 1), hbm_frac=round(gbs / HBM_PEAK_GBS, 4),
                    form="moments" if lib.pnp_get_qeif_form(B68, 68) == pnp._lib.QEIF_FORM_MOMENTS else "points",
                    valu_fp64_TFLOPs=round(B68 * qeif_moment_flops(68, it68) / (ms * 1e-3) / 1e12, 2),
                    roofline=dict(bound="hbm", achieved=round(gbs, 2), peak=HBM_PEAK_GBS, unit="GB/s",
                                  frac=round(gbs / HBM_PEAK_GBS, 5), traffic=tr, traffic_source=tsrc,
                                  kernel=ran["kernel"], launch={k: ran[k] for k in ("grid", "block", "lds_bytes")},
                                  kernel_ms=round(ms, 4), bytes_per_solve=nbytes),
                    roofline_issue=issue_roofline(ins, -(-B68 // 64), ms))
            lib.pnp_set_qeif_form(pnp._lib.QEIF_FORM_POINTS)        # the point-by-point form, for comparison
            try:
                ms = time_launches(torch, lambda: pnp.solve_pnp_batch(uv68, P68d, K, want_rpy=False), 10)
            finally:
                lib.pnp_set_qeif_form(pnp._lib.QEIF_FORM_AUTO)
            other["configs[3]_qeif_n68_b256k_shared_pattern_point_form"] = dict(
                solves_per_s=round(B68 / ms * 1e3, 1), kernel_ms=round(ms, 4),
                lanes_per_problem=lib.pnp_get_lanes_per_problem(B68, 68, 0),
                valu_fp64_TFLOPs=round(B68 * qeif_flops(68, it68) / (ms * 1e-3) / 1e12, 2))
            del own68
            uv64, xyz64 = uv.double(), xyz.double()          # the reference's own dtype end to end
            ms = time_launches(torch, lambda: pnp.solve_pnp_batch(uv64, xyz64, K, want_rpy=False), n_small)
            other["qeif_n6_b1m_float64_io"] = dict(solves_per_s=round(B / ms * 1e3, 1), kernel_ms=round(ms, 4),
                                                   bytes_per_solve=2 * bytes_per_solve)
            ms = time_launches(torch, lambda: pnp.solve_pnp_batch(uv64, xyz64, K, method="dlt", want_rpy=False),
                               n_small)
            gbs = B * (2 * bytes_per_solve) / (ms * 1e-3) / 1e9
            other["dlt_n6_b1m_float64_io"] = dict(solves_per_s=round(B / ms * 1e3, 1), kernel_ms=round(ms, 4),
                                                  achieved_GBps=round(gbs, 1), hbm_frac=round(gbs / HBM_PEAK_GBS, 4))
            del uv64, xyz64
            # the host-pointer path: the same 1M problems handed over as NumPy arrays in pageable
            # memory, copies over PCIe included (never the headline)
            def host_call():
                return pnp.solve_pnp_batch(uv_h, xyz_h, K, want_rpy=False)

            def wall_ms(fn, reps=12):
                """Wall-clock of `reps` calls after two untimed ones: (mean, best, median) in ms."""
                fn()
                fn()
                ts = []
                for _ in range(reps):
                    t0 = time.perf_counter()
                    fn()
                    ts.append((time.perf_counter() - t0) * 1e3)
                ts.sort()
                return sum(ts) / len(ts), ts[0], ts[len(ts) // 2]

            hs_mean, hs_best, hs_med = wall_ms(host_call)
            held = host_call()                       # ... and with the result arrays of an earlier call (out=)
            re_mean, re_best, re_med = wall_ms(lambda: pnp.solve_pnp_batch(uv_h, xyz_h, K, want_rpy=False, out=held))
            del held
            other["qeif_n6_b1m_host_pointers_pcie_inclusive"] = dict(
                solves_per_s=round(B / (hs_mean * 1e-3), 1), ms_per_call=round(hs_mean, 3),
                ms_per_call_best=round(hs_best, 3), ms_per_call_median=round(hs_med, 3),
                pcie_GBps=round(B * (bytes_per_solve + 8) / (hs_mean * 1e-3) / 1e9, 2),
                ms_per_call_results_reused=round(re_mean, 3), ms_per_call_results_reused_best=round(re_best, 3),
                solves_per_s_results_reused=round(B / (re_mean * 1e-3), 1),
                calls_timed=12,
                host_lane_threads=lib.pnp_host_lane_threads(local_rank),
                what="NumPy arrays in, NumPy arrays out: chunks through page-locked buffers on the library's standing host threads, kernels launched on those buffers; ms_per_call is the mean over the timed calls, each of which faults in 59 MB of fresh result arrays")
            # ... and as host tensors in page-locked memory: the kernels work on them over the bus
            uv_p, xyz_p = torch.from_numpy(uv_h).pin_memory(), torch.from_numpy(xyz_h).pin_memory()

            def pinned_call():
                return pnp.solve_pnp_batch(uv_p, xyz_p, K, want_rpy=False)
            pinned_call()
            pinned_call()
            t0 = time.perf_counter()
            for _ in range(5):
                pinned_call()
            hs = (time.perf_counter() - t0) / 5
            other["qeif_n6_b1m_page_locked_host_tensors_pcie_inclusive"] = dict(
                solves_per_s=round(B / hs, 1), ms_per_call=round(hs * 1e3, 3),
                pcie_GBps=round(B * (bytes_per_solve + 8) / hs / 1e9, 2),
                what="pinned host tensors in, pinned host tensors out: one launch that reads and writes host memory")
            del uv_p, xyz_p
            line["other_configs"] = other
        if world == 1 and not args.no_cpu_baseline:   # last: the GPU idles (and clocks down) meanwhile
            line["cpu_baseline"] = cpu_baseline(uv_h, xyz_h, K)
            line["cpu_baseline_numpy_batched"] = cpu_baseline_numpy_batched(uv_h, xyz_h, K)
            line["cpu_baseline_c"] = cpu_baseline_c(uv_h, xyz_h, K)
            threads = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(16, os.cpu_count() or 1)
            line["cpu_baseline_c_all_cores"] = cpu_baseline_c_threads(uv_h, xyz_h, K, threads)

    # The optional exchange of results (R, t: 48 B per problem), reported beside the headline and
    # never part of it.  It runs after everything the line needs has been measured, each variant
    # on its own (one that fails is reported as an error string), and under a deadline: if a
    # collective never returns, rank 0 prints the line without it and every rank leaves.
    if distributed:
        printed = []

        def bail():
            if rank == 0 and not printed:
                line["with_result_gather"] = dict(gather, error="did not finish within %d s" % GATHER_DEADLINE_S)
                print(json.dumps(line), flush=True)
            os._exit(0)

        gather = {}
        if rank == 0:
            # should a rank die inside a collective (no exception, no time-out to catch), what was
            # measured is at least in the log; standard output still carries one line only
            print("bench.py: headline before the optional exchange: " + json.dumps(line), file=sys.stderr, flush=True)
        watchdog = threading.Timer(GATHER_DEADLINE_S, bail)
        watchdog.daemon = True
        watchdog.start()
        g_steps = max(5, args.steps // 10)

        def timed_exchange(exchange):
            exchange(out)
            barrier()
            t0 = time.perf_counter()
            for _ in range(g_steps):
                exchange(step())
            barrier()
            gmax = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
            dist.all_reduce(gmax, op=dist.ReduceOp.MAX)
            return dict(solves_per_s=round(world * B * g_steps / float(gmax.item()), 1), steps=g_steps,
                        ms_per_step=round(float(gmax.item()) / g_steps * 1e3, 4))

        variants = [("all_gather", "solve + all_gather_into_tensor of R and t (RCCL through torch.distributed)",
                     lambda o: pnp.gather_results({"R": o["R"], "t": o["t"]})),
                    ("to_rank_0_only", "solve + gather of R and t to rank 0 (RCCL through torch.distributed)",
                     lambda o: pnp.gather_results({"R": o["R"], "t": o["t"]}, dst=0))]
        comm = None
        if not args.no_c_abi_gather:
            try:
                comm = pnp.RcclComm.from_torch_group()
                variants += [("c_abi_all_gather", "the same through pnp_gather_results on a communicator of the library's own",
                              lambda o: pnp.gather_results({"R": o["R"], "t": o["t"]}, comm=comm)),
                             ("c_abi_to_rank_0_only", "pnp_gather_results with a root",
                              lambda o: pnp.gather_results({"R": o["R"], "t": o["t"]}, dst=0, comm=comm))]
            except Exception as e:          # noqa: BLE001 - reported, the headline stands
                gather["c_abi_error"] = "%s: %s" % (type(e).__name__, e)
        for tag, what, exchange in variants:
            try:
                gather[tag] = dict(timed_exchange(exchange), what=what)
            except Exception as e:          # noqa: BLE001
                gather[tag] = dict(error="%s: %s" % (type(e).__name__, e))
                break                        # the ranks may no longer agree on what comes next
        if comm is not None:
            try:
                comm.close()
            except Exception:               # noqa: BLE001
                pass
        gather["gathered_bytes_per_rank_per_step"] = world * B * 48
        if rank == 0:
            line["with_result_gather"] = gather
            print(json.dumps(line), flush=True)
            printed.append(True)
        # still under the deadline: a rank that gave up on an exchange must not wait here for good
        dist.barrier()
        dist.destroy_process_group()
        watchdog.cancel()
    elif rank == 0:
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
