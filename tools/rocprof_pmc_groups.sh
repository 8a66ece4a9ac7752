#!/bin/bash
# Per-kernel averages of one group of PMC counters over a command (one rocprofv3 pass):
#   tools/rocprof_pmc_groups.sh <tag> "<CTR1 CTR2 ...>" <command...>
set -e
TAG=$1; GRP=$2; shift; shift
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pg_$TAG
rocprofv3 --pmc $GRP --kernel-trace --output-format csv -d /tmp/pg_$TAG -o c -- "$@" > "$OUT/log.txt" 2>&1 || true
python3 - /tmp/pg_$TAG "$OUT/per_kernel.txt" <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:110]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[(k, r["Counter_Name"])] += 1
names = sorted({c for v in agg.values() for c in v})
rows = []
for k, v in agg.items():
    n = max(cnt[(k, c)] for c in names if (k, c) in cnt)
    rows.append((sum(v.values()), k, n, [v.get(c, 0.0) / max(cnt[(k, c)], 1) for c in names]))
rows.sort(reverse=True)
with open(sys.argv[2], "w") as fh:
    fh.write("dispatches  " + "  ".join(f"{c:>28s}" for c in names) + "  kernel\n")
    for _, k, n, vals in rows[:80]:
        fh.write(f"{n:10d}  " + "  ".join(f"{x:28.1f}" for x in vals) + f"  {k}\n")
print(open(sys.argv[2]).read()[:6000])
PY
