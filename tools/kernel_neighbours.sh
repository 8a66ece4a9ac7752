#!/bin/bash
# Which kernels run right before / after a given kernel (default: the blit copy kernel) in a steady-state step:
# names the producer of anonymous runtime copies.   tools/kernel_neighbours.sh <tag> <pattern> <command...>
set -e
TAG=$1; PAT=$2; shift; shift
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT" /tmp/prof_$TAG
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_$TAG -o $TAG -- "$@" > "$OUT/cmd.log" 2>&1 || true
python3 - "$PAT" "$OUT" $(find /tmp/prof_$TAG -name "*kernel_trace.csv" | head -1) <<'PY'
import csv, sys, collections
pat, out, f = sys.argv[1], sys.argv[2], sys.argv[3]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
short = lambda n: n.replace("void ", "").replace("dfine::", "").replace("at::native::", "")[:70]
# last third of the trace = steady state
lo = len(names) * 2 // 3
pairs = collections.Counter()
for i in range(max(lo, 1), len(names) - 1):
    if pat in names[i]:
        j = i - 1
        while j > 0 and pat in names[j]:
            j -= 1
        k = i + 1
        while k < len(names) - 1 and pat in names[k]:
            k += 1
        pairs[(short(names[j]), short(names[k]))] += 1
with open(out + "/neighbours.txt", "w") as fh:
    for (a, b), n in pairs.most_common(60):
        fh.write(f"{n:5d}  after [{a}]  before [{b}]\n")
print(open(out + "/neighbours.txt").read())
PY
