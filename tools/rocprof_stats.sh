#!/bin/bash
# Per-kernel time of a command on the GPU box: rocprofv3 --kernel-trace --stats, keeping only the
# small summary CSVs under gpurun_out/<tag>/ (the raw trace is far above gpurun's 64 MiB limit).
#   tools/rocprof_stats.sh <tag> <command...>
set -e
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT" /tmp/prof_$TAG
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o $TAG -- "$@" > "$OUT/cmd.log" 2>&1 || true
find /tmp/prof_$TAG -name "*stats*.csv" -exec cp {} "$OUT/" \;
# domain-level numbers + the 60 hottest kernels in a compact text form
python3 - "$OUT" <<'PY'
import csv, glob, sys
out = sys.argv[1]
for f in glob.glob(out + "/*kernel_stats.csv"):
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    with open(out + "/top_kernels.txt", "w") as fh:
        fh.write(f"total kernel time {tot/1e6:.2f} ms over {sum(int(r['Calls']) for r in rows)} launches, {len(rows)} distinct kernels\n")
        for r in rows[:70]:
            fh.write(f"{float(r['TotalDurationNs'])/1e6:10.3f} ms {100*float(r['TotalDurationNs'])/tot:5.1f}% calls {int(r['Calls']):6d} avg {float(r['AverageNs'])/1e3:9.1f} us  {r['Name'][:150]}\n")
PY
tail -3 "$OUT/cmd.log"
head -40 "$OUT/top_kernels.txt"
