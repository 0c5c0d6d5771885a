#!/usr/bin/env python3
"""Condenses the rocprofv3 --pmc passes of scripts/gpu_pmc.sh (gpurun_out/pmc/*) into
profiles/<tag>_pmc_summary.json and the per-launch HBM traffic that bench.py reports.

HBM bytes per launch = read bytes + WRITE_SIZE.  Read bytes come from the read requests counted by
size (TCC_EA0_RDREQ_32B / _64B / _128B, pass "rdreq"): 32 n32 + 64 n64 + 128 n128.  FETCH_SIZE, the
derived counter, tallies a 128-byte request as 64 bytes on gfx950 (its expression takes the
128-byte share from TCC_BUBBLE, which stays 0), so where the rdreq pass is missing 2 * FETCH_SIZE
stands in (right for kernels that ask for whole lines only).  WRITE_SIZE is exact for streaming
stores.  Both are checked here against the known input size of each launch."""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
src = os.path.join(ROOT, "gpurun_out", "pmc")

# dispatch order of scripts/pmc_probe.py: 3 launches each
LAUNCHES = [("qeif_n6_b1m_per_problem_xyz", "qeif_rows6_kernel<", 1 << 20, 6, True),
            ("dlt_n6_b1m_per_problem_xyz", "dlt_rows6_kernel<", 1 << 20, 6, True),
            ("qeif_n6_b1m_shared_pattern", "qeif_rows6_kernel<", 1 << 20, 6, False),
            ("qeif_n68_b256k_shared_pattern", "qeif_moment_lines_kernel<float, true>", 1 << 18, 68, False),
            ("qeif_n68_b256k_per_problem_xyz", "qeif_moment_lines_kernel<float, false>", 1 << 18, 68, True)]


KERNEL_SEEN = {}   # label -> kernel name the counter rows carry (without "void" and the argument list)


def load(name):
    files = glob.glob(os.path.join(src, name, "*counter_collection.csv"))
    rows = list(csv.DictReader(open(files[0]))) if files else []
    rows = [r for r in rows if "pnp::" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    return rows


def per_launch(name):
    """{launch label: {counter: mean value}}, dispatches taken in probe order."""
    rows = load(name)
    by_disp = collections.OrderedDict()
    for r in rows:
        by_disp.setdefault(int(r["Dispatch_Id"]), {"kernel": r["Kernel_Name"]})[r["Counter_Name"]] = float(r["Counter_Value"])
        by_disp[int(r["Dispatch_Id"])]["dur_ns"] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    disp = list(by_disp.values())
    out = {}
    for i, (label, kern, B, N, own) in enumerate(LAUNCHES):
        chunk = disp[3 * i:3 * i + 3]
        if len(chunk) < 3 or any(kern not in d["kernel"] for d in chunk):
            continue
        keys = [k for k in chunk[0] if k != "kernel"]
        out[label] = {k: sum(d[k] for d in chunk) / len(chunk) for k in keys}
        KERNEL_SEEN[label] = chunk[0]["kernel"].split("(")[0].replace("void ", "").strip()
    return out


summary = {}
for p in ("fetch", "rdreq", "write", "sq1", "sq2", "grbm"):
    for label, vals in per_launch(p).items():
        summary.setdefault(label, {}).update({k: v for k, v in vals.items() if k != "dur_ns"})
        summary[label].setdefault("profiled_kernel_us", {})[p] = round(vals["dur_ns"] / 1e3, 1)

traffic = {}
for label, kern, B, N, own in LAUNCHES:
    s = summary.get(label)
    if not s or "FETCH_SIZE" not in s or "WRITE_SIZE" not in s:
        continue
    in_bytes = B * (8 * N + (12 * N if own else 0))
    s["input_bytes"] = in_bytes
    s["fetch_bytes_2x"] = 2 * s["FETCH_SIZE"] * 1024
    s["write_bytes"] = s["WRITE_SIZE"] * 1024
    s["hbm_bytes_per_launch"] = s["fetch_bytes_2x"] + s["write_bytes"]
    s["fetch_2x_over_input"] = round(s["fetch_bytes_2x"] / in_bytes, 4)
    # FETCH_SIZE tallies a 128-byte request as 64 bytes (hence the factor 2) and a 64-byte one as
    # 64: where short requests occur it over-counts, and it was once "calibrated" on probes of the
    # kernels' access shapes, which shared the kernels' re-reads and hid them.  The request counts
    # by size below are what the traffic figure is made of; 2 x FETCH_SIZE stays beside it.
    if all(k in s for k in ("TCC_EA0_RDREQ_32B_sum", "TCC_EA0_RDREQ_64B_sum", "TCC_EA0_RDREQ_128B_sum")):
        n32, n64, n128 = s["TCC_EA0_RDREQ_32B_sum"], s["TCC_EA0_RDREQ_64B_sum"], s["TCC_EA0_RDREQ_128B_sum"]
        s["read_bytes_by_request_size"] = 32 * n32 + 64 * n64 + 128 * n128
        s["read_requests"] = dict(n32=n32, n64=n64, n128=n128, all=s.get("TCC_EA0_RDREQ_sum"))
        s["read_bytes_by_request_size_over_input"] = round(s["read_bytes_by_request_size"] / in_bytes, 4)
        s["hbm_bytes_per_launch_by_2x_fetch_size"] = s["hbm_bytes_per_launch"]
        s["hbm_bytes_per_launch"] = s["read_bytes_by_request_size"] + s["write_bytes"]
    if "SQ_ACTIVE_INST_VALU" in s and "GRBM_GUI_ACTIVE" in s:
        # SQ_* activity counters tick in quad-cycles; GRBM_GUI_ACTIVE sums the 8 XCDs
        s["valu_busy_frac"] = round(4 * s["SQ_ACTIVE_INST_VALU"] / 1024 / (s["GRBM_GUI_ACTIVE"] / 8), 4)
        s["valu_insts_per_wave"] = round(s["SQ_INSTS_VALU"] / s["SQ_WAVES"], 1)
    traffic[label + "_bytes_per_launch"] = round(s["hbm_bytes_per_launch"])
    if "valu_insts_per_wave" in s:
        traffic[label + "_valu_insts_per_wave"] = s["valu_insts_per_wave"]

os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
json.dump(summary, open(os.path.join(ROOT, "profiles", tag + "_pmc_summary.json"), "w"), indent=1, sort_keys=True)
traffic["qeif_n6_b1m_bytes_per_launch"] = traffic.get("qeif_n6_b1m_per_problem_xyz_bytes_per_launch")
traffic["source"] = ("profiles/%s_pmc_summary.json (rocprofv3 --pmc, separate passes: read requests by size "
                     "TCC_EA0_RDREQ_32B/_64B/_128B, WRITE_SIZE, FETCH_SIZE, SQ_*, GRBM)" % tag)
traffic["kernels"] = dict(KERNEL_SEEN)
import subprocess
try:
    traffic["commit"] = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True,
                                       timeout=30).stdout.strip() or None
    dirty = subprocess.run(["git", "-C", ROOT, "status", "--porcelain", "--", "pnp_solver_test_amd", "include"],
                           capture_output=True, text=True, timeout=30).stdout.strip()
    if dirty:
        traffic["commit"] += " + uncommitted changes to the library sources"
except Exception:
    traffic["commit"] = None
json.dump(traffic, open(os.path.join(ROOT, "profiles", "hbm_traffic.json"), "w"), indent=1, sort_keys=True)
for label, s in summary.items():
    print(label, {k: s[k] for k in ("hbm_bytes_per_launch", "read_bytes_by_request_size_over_input", "read_requests", "fetch_2x_over_input",
                                    "read_bytes_over_input", "valu_busy_frac", "valu_insts_per_wave") if k in s})
