#!/bin/bash
# HBM traffic counters of a command, one rocprofv3 pass per counter (FETCH_SIZE and WRITE_SIZE do not
# fit one pass on gfx950), keeping a per-kernel average in gpurun_out/<tag>/pmc_<counter>.txt
#   tools/rocprof_pmc.sh <tag> <command...>
set -e
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for CTR in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$TAG_$CTR
  rocprofv3 --pmc $CTR --kernel-trace --output-format csv -d /tmp/pmc_${TAG}_$CTR -o $CTR -- "$@" > "$OUT/pmc_$CTR.log" 2>&1 || true
  python3 - "$OUT" "$CTR" /tmp/pmc_${TAG}_$CTR <<'PY'
import csv, glob, sys, collections
out, ctr, d = sys.argv[1:4]
agg = collections.defaultdict(lambda: [0, 0.0])
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r.get("Counter_Name") == ctr:
            a = agg[r["Kernel_Name"][:120]]; a[0] += 1; a[1] += float(r["Counter_Value"])
with open(f"{out}/pmc_{ctr}.txt", "w") as fh:
    for k, (n, s) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        fh.write(f"{s/n:14.1f} avg {ctr} per dispatch over {n:5d} dispatches  {k}\n")
print(open(f"{out}/pmc_{ctr}.txt").read()[:1500])
PY
done
