#!/bin/bash
# FETCH_SIZE calibration for the moment kernel's access shape (scripts/probes/fetch_calibration.hip).
set -u
export GRAFT_REPO_ROOT="${GRAFT_REPO_ROOT:-$PWD}"
OUT="$GRAFT_REPO_ROOT/gpurun_out/fetch_calibration"
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT" -o c -- "$GRAFT_REPO_ROOT/scripts/probes/fetch_calibration" > "$OUT/run.log" 2>&1
echo "exit $?"
python3 - "$OUT" <<'PY'
import csv, glob, json, sys, collections
out = sys.argv[1]
f = glob.glob(out + "/**/*counter_collection.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == "FETCH_SIZE" and "calib" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Dispatch_Id"]))
B, N = 1 << 18, 68
known = {"calib_rows<16>": B * 2 * N * 4, "calib_rows<32>": B * 2 * N * 4, "calib_rows<24>": B * 3 * N * 4, "calib_wide": B * 2 * N * 4,
         "calib_lines": B * 5 * N * 4}
res = collections.OrderedDict()
seen = collections.Counter()
for r in rows:
    name = next(k for k in known if k in r["Kernel_Name"])
    seen[name] += 1
    tag = name
    if name == "calib_rows<16>" and seen[name] % 2 == 0:
        tag = "calib_rows<16> (no eviction between launches)"
    res.setdefault(tag, []).append(float(r["Counter_Value"]) * 1024 / known[name])
summary = {k: dict(fetch_size_over_bytes=[round(x, 4) for x in v], mean=round(sum(v) / len(v), 4)) for k, v in res.items()}
json.dump(summary, open(out + "/summary.json", "w"), indent=1)
print(json.dumps(summary, indent=1))
PY
