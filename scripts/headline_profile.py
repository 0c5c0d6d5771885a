#!/usr/bin/env python3
"""Reduces a rocprofv3 kernel trace of `bench.py --no-other-configs` to the figures that the
bench line's roofline is checked against: the mean duration of the headline kernel over the
launches of the timed region (the last `steps` dispatches of that kernel) and over all of its
launches (spin-up and warm-up included, as the --stats table averages them).

usage: headline_profile.py <kernel_trace.csv> <bench.log> <out.json> [kernel-name-prefix]
"""
import csv
import json
import sys


def main():
    trace, bench_log, out = sys.argv[1:4]
    prefix = sys.argv[4] if len(sys.argv) > 4 else "void pnp::qeif_rows6_kernel<float>"
    line = None
    for text in open(bench_log):
        text = text.strip()
        if text.startswith("{") and '"metric"' in text:
            line = json.loads(text)
    if line is None:
        sys.exit("no bench line in %s" % bench_log)
    durations = []
    with open(trace, newline="") as fh:
        for row in csv.DictReader(fh):
            if row["Kernel_Name"].startswith(prefix):
                durations.append((int(row["Start_Timestamp"]), int(row["End_Timestamp"]) - int(row["Start_Timestamp"])))
    durations.sort()
    ns = [d for _, d in durations]
    steps = int(line["steps"])
    if len(ns) < steps:
        sys.exit("only %d dispatches of %s in the trace, bench timed %d" % (len(ns), prefix, steps))
    timed = ns[-steps:]
    summary = {
        "command": "rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline --no-other-configs",
        "kernel": prefix,
        "dispatches": len(ns),
        "all_dispatches_mean_ms": sum(ns) / len(ns) / 1e6,
        "timed_region_dispatches": steps,
        "timed_region_mean_ms": sum(timed) / steps / 1e6,
        "timed_region_min_ms": min(timed) / 1e6,
        "timed_region_max_ms": max(timed) / 1e6,
        "bench_line_kernel_ms_same_run": line["roofline"]["kernel_ms"],
        "bench_line_ms_per_step_same_run": line["ms_per_step"],
        "bench_line_value_same_run": line["value"],
    }
    summary["trace_over_hip_events"] = summary["timed_region_mean_ms"] / summary["bench_line_kernel_ms_same_run"]
    with open(out, "w") as fh:
        json.dump(summary, fh, indent=1)
    print(json.dumps(summary))


if __name__ == "__main__":
    main()
