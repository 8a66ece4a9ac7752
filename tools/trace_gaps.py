"""Steady-state device timeline from a rocprofv3 --kernel-trace CSV (low tracing overhead: the host runs at its real pace):
per train step, the span, the busy time of the main queue, and where that queue idles - is the eager decoder / criterion
stretch paced by the host?   python tools/trace_gaps.py gpurun_out/r05_stats/r05_stats_kernel_trace.csv"""
import csv, sys, collections
rows = []
with open(sys.argv[1]) as fh:
    for r in csv.DictReader(fh):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Queue_Id"]))
rows.sort()
marks = [i for i, r in enumerate(rows) if "stem_conv_s2_vec" in r[2]]          # first convolution of a step's forward pass
print(f"{len(rows)} dispatches, {len(marks)} steps")
steps = [(rows[marks[i]:marks[i + 1]]) for i in range(len(marks) - 1)]
steps = steps[len(steps) // 2: len(steps) // 2 + 20]                             # 20 steps from the middle of the run
agg = collections.defaultdict(list)
for st in steps:
    t0 = st[0][0]
    qcount = collections.Counter(r[3] for r in st)
    main = qcount.most_common(1)[0][0]
    mq = [r for r in st if r[3] == main]
    span = (st[-1][1] - t0) / 1e6
    busy = sum(e - s for s, e, _, _ in mq) / 1e6
    # phases by marker kernels of the main queue
    def first(pat, after=0):
        return next((r[0] for r in mq if pat in r[2] and r[0] >= after), None)
    t_dec = first("maps_to_tokens")                       # end of the forward graph (encoder maps -> decoder memory)
    t_bwd = first("tokens_to_maps", t_dec or 0)           # start of the backbone / encoder backward
    agg["span"].append(span); agg["busy"].append(busy)
    if t_dec and t_bwd:
        win = [r for r in mq if t_dec <= r[0] < t_bwd]
        agg["window"].append((t_bwd - t_dec) / 1e6)
        agg["window_busy"].append(sum(e - s for s, e, _, _ in win) / 1e6)
        agg["window_launches"].append(len(win))
        gaps = sorted(((win[i + 1][0] - win[i][1]) / 1e3, win[i + 1][2][:60]) for i in range(len(win) - 1))
        agg["gap50"].append(sum(1 for g, _ in gaps if g > 50)); agg["gapsum"].append(sum(g for g, _ in gaps if g > 5) / 1e3)
        agg["fwd"].append((t_dec - t0) / 1e6); agg["bwd"].append((st[-1][1] - t_bwd) / 1e6)
med = lambda v: sorted(v)[len(v) // 2]
for k, v in agg.items():
    print(f"{k:16s} median {med(v):8.3f}   min {min(v):8.3f}  max {max(v):8.3f}")
