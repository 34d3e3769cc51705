#!/usr/bin/env python3
"""Per-kernel picture of the device-resident Gym step from a rocprofv3 kernel trace of tools/device_loop_bench.py:
   tools/device_loop_trace.py <dir with *_kernel_trace.csv>   -> per kernel: calls, mean / median / p90 duration; consume-kernel histogram"""
import csv, glob, sys
import numpy as np

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
by = {}
for r in rows:
    k = r["Kernel_Name"]
    for tag in ("k_consume_staged", "k_refill_staged", "k_sfm_step", "k_collision_reward_wave"):
        if tag in k:
            by.setdefault(tag, []).append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
for tag, v in by.items():
    d = np.array([e - s for s, e in v]) / 1e3
    print(f"{tag:26s} calls {len(d):5d}  mean {d.mean():7.1f}  median {np.median(d):7.1f}  p90 {np.percentile(d, 90):7.1f}  max {d.max():7.1f} us")
c = np.array([e - s for s, e in sorted(by.get("k_consume_staged", []))]) / 1e3
if len(c):
    print("consume: share of calls above 20 us:", float((c > 20).mean()))
# overlap: how often does a step kernel run while a refill pass is in flight, and how long is it then?
ref = sorted(by.get("k_refill_staged", []))
st = sorted(by.get("k_sfm_step", []))
if ref and st:
    import bisect
    starts = [s for s, _ in ref]
    inside, outside = [], []
    for s, e in st:
        i = bisect.bisect_right(starts, e) - 1
        hit = any(rs < e and re > s for rs, re in ref[max(0, i - 2): i + 1])
        (inside if hit else outside).append((e - s) / 1e3)
    print(f"step kernel beside a refill pass: {len(inside)} calls mean {np.mean(inside) if inside else 0:.1f} us; alone: {len(outside)} calls mean {np.mean(outside) if outside else 0:.1f} us")
