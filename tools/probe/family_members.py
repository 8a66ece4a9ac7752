import os, sys, collections
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools")
os.environ.pop("STEP_PROFILE_TOP", None)
import step_profile as sp
rows = []
for k in sp.prof.key_averages():
    fam = next((f for pat, f in sp.FAMILIES if pat in k.key), "other")
    if fam in ("HIP other", "dwconv", "msda", "ATen other", "ATen copy/cast", "stem", "ln"):
        rows.append((k.device_time_total / 3e3, k.count // 3, fam, k.key[:130]))
for r in sorted(rows, reverse=True)[:70]:
    print(f"{r[0]:7.3f} {r[1]:4d} {r[2]:14s} {r[3]}")
