#!/bin/bash
# Same-box A/B of two BUILDS of the tree (kernel code cannot be toggled in-process like the module flags of tools/ab_step.py,
# and boxes differ by 1 - 3 %): exports <commit> with its own built library into tools/probe/_old (git-ignored, travels with the
# gpurun snapshot), to be alternated with the working tree on ONE box:
#     bash tools/ab_trees.sh <commit>                      (here: builds tools/probe/_old)
#     gpurun -- 'bash tools/ab_trees.sh run [rounds]'      (GPU box: old / head alternated, 60 timed steps each)
# Round 6 found a +0.33 ms regression this way that every single-kernel bench had called an improvement.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
if [ "$1" = "run" ]; then
    run() { (cd "$1" && python bench.py --steps 60 --warmup 10 --cpu-steps 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$2', d['value'], d['ms_per_step'], d['median_ms_per_step'])"); }
    for i in $(seq 1 ${2:-3}); do run "$ROOT/tools/probe/_old" old; run "$ROOT" head; done
    exit 0
fi
C=${1:?commit}
T=$(mktemp -d)
git -C "$ROOT" archive "$C" custom_d_fine_amd bench.py include | tar -x -C "$T"
(cd "$T" && python -c "
import sys; sys.path.insert(0, '$T')
from custom_d_fine_amd.csrc import build
build.build(verbose=False)")
rm -rf "$ROOT/tools/probe/_old" && mkdir -p "$ROOT/tools/probe/_old/profiles"
cp -r "$T/custom_d_fine_amd" "$T/bench.py" "$ROOT/tools/probe/_old/"
rm -rf "$ROOT/tools/probe/_old/custom_d_fine_amd/csrc/build"
cp "$ROOT"/profiles/r06_conv_pmc.json "$ROOT"/profiles/r06_roofline_from_profile.json "$ROOT/tools/probe/_old/profiles/" 2>/dev/null || true
rm -rf "$T"
du -sh "$ROOT/tools/probe/_old"
