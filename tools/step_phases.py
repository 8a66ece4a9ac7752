"""Phases of one train step from tools/step_events.py's JSON lines: forward graph / decoder + criterion stretch / backward + tail on
the main stream (length, launches, busy time), the side stream's busy time, the smallest launch-to-launch pitch (the per-launch
floor of dependent kernels on one queue) and the stretch's kernels by total time.   python tools/step_phases.py EVENTS.jsonl"""
import collections, json, re, sys
ev = [json.loads(l) for l in open(sys.argv[1])]
streams = collections.Counter(e[2] for e in ev)
main_id = streams.most_common(1)[0][0]
main = [e for e in ev if e[2] == main_id]
side = [e for e in ev if e[2] != main_id]
T = max(e[0] + e[1] for e in ev)
f0 = next(e[0] for e in main if "stem_conv_s2_vec" in e[3])
d0 = next(e[0] for e in main if e[0] > f0 and ("maps_tokens" in e[3] or "topk" in e[3]))
b0 = next(e[0] for e in main if e[0] > d0 and re.search(r"bn2?_.*bwd|bn_bwd|bn_one_bwd", e[3]))
print(f"step span {T / 1e3:.2f} ms, {len(main)} launches on the main stream, {len(side)} on the side stream ({sum(e[1] for e in side) / 1e3:.2f} ms busy)")
for name, lo, hi in (("forward graph (backbone + encoder)", f0, d0), ("decoder + criterion + decoder backward (eager)", d0, b0), ("backbone / encoder backward graphs + optimizer tail", b0, T)):
    sel = [e for e in main if lo <= e[0] < hi]
    d = sorted(e[1] for e in sel)
    print(f"  {name:52s} {(hi - lo) / 1e3:6.2f} ms  {len(sel):4d} launches  busy {sum(d) / 1e3:6.2f} ms  shortest launches {d[0]:.1f} / {d[len(d) // 20]:.1f} us (min / 5th percentile)")
sel = [e for e in main if d0 <= e[0] < b0]
agg = collections.defaultdict(lambda: [0, 0.0])
for e in sel:
    n = re.sub(r"\(.*", "", e[3]).replace("void ", "")[:100]
    agg[n][0] += 1
    agg[n][1] += e[1]
print("decoder + criterion stretch by kernel (a launch's duration here = end of the previous launch -> its own end: launches are back to back):")
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f"  {t:8.1f} us {c:4d} x {t / c:7.1f}  {n}")
tiny = [e for e in sel if e[1] < 9.0]
print(f"  launches under 9 us: {len(tiny)} of {len(sel)}, {sum(e[1] for e in tiny) / 1e3:.2f} ms")
