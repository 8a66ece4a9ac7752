"""Recomputes the bench line's `roofline` (dense-conv implicit-GEMM family against the bf16 MFMA peak) from a rocprofv3
`--kernel-trace --stats` summary of the SAME command, i.e. from kernel time in the graph-replay two-stream mode that produces
`value`:

    python tools/roofline_from_stats.py profiles/r05_bench_kernel_stats.csv profiles/r05_bench_line.json \
        > profiles/r05_roofline_from_profile.json

family ms per step = sum of TotalDurationNs of the family's kernels / traced steps; traced steps = launches of a kernel that runs
exactly once per backward pass (stem_pool_bwd*: warm-up and graph-capture passes are traced too and counted the same way);
algorithmic FLOPs per step = the bench line's `roofline.algorithmic_tflop_per_step` (sum of 2 B HW Cin Cout KS^2 over the
launches); frac = FLOPs / time / 2.5 PFLOP/s.  bench.py attaches the result as `roofline.profile` (measured_in_run: false)."""
import csv
import json
import re
import sys

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import FAMILY_GROUPS, KERNEL_GROUPS, MFMA_BF16_PEAK_TFS as PEAK_TFS  # noqa: E402  (the SAME name patterns the bench line uses)

GROUPS = [(n, p) for n, p in KERNEL_GROUPS if n in FAMILY_GROUPS]
OTHER = [(n, p) for n, p in KERNEL_GROUPS if n not in FAMILY_GROUPS]


def main(stats_csv, bench_json):
    rows = list(csv.DictReader(open(stats_csv)))
    line = json.load(open(bench_json))
    steps = sum(int(r["Calls"]) for r in rows if "stem_pool_bwd" in r["Name"])
    if steps <= 0:
        raise SystemExit("no stem_pool_bwd launches in the trace: cannot count the traced steps")

    def group_ms(pattern):
        rx = re.compile(pattern)
        sel = [r for r in rows if rx.search(r["Name"])]
        return sum(float(r["TotalDurationNs"]) for r in sel) / 1e6 / steps, sum(int(r["Calls"]) for r in sel) / steps

    groups = {}
    for name, pat in GROUPS:
        ms, n = group_ms(pat)
        groups[name] = {"ms_per_step": round(ms, 3), "launches_per_step": round(n, 1)}
    others = {}
    for name, pat in OTHER:
        ms, n = group_ms(pat)
        others[name] = {"ms_per_step": round(ms, 3), "launches_per_step": round(n, 1)}
    fam_ms = sum(g["ms_per_step"] for g in groups.values())
    tflop = line["roofline"]["algorithmic_tflop_per_step"]
    tf = tflop / (fam_ms * 1e-3)
    all_ms = sum(float(r["TotalDurationNs"]) for r in rows) / 1e6 / steps
    out = {"mode": "graph replay, two streams (rocprofv3 --kernel-trace --stats of the bench command)",
           "stats_csv": stats_csv, "traced_steps": steps, "family_ms_per_step": round(fam_ms, 3),
           "algorithmic_tflop_per_step": tflop, "achieved": round(tf, 1), "peak": PEAK_TFS, "unit": "TFLOP/s",
           "frac": round(tf / PEAK_TFS, 4), "groups": groups, "other_kernel_families": others,
           "all_kernels_ms_per_step": round(all_ms, 3), "bench_ms_per_step": line.get("median_ms_per_step")}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
