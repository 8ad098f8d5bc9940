#!/usr/bin/env python3
"""Busy time per kernel and idle time between kernels from a rocprofv3 --kernel-trace CSV (last `frac` of the trace = steady state).
usage: tools/trace_gaps.py <kernel_trace.csv> [frac=0.6] [n_long=0]   (n_long: also list the longest idle intervals with the kernels around them)"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.6
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
ev = ev[int(len(ev) * (1 - frac)):]
span = ev[-1][1] - ev[0][0]
busy = collections.defaultdict(lambda: [0, 0])
gap_hist = collections.Counter(); gaps = 0; big = 0
n_long = int(sys.argv[3]) if len(sys.argv) > 3 else 0
longs = []; last_name = ""
short = lambda k: k.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:48]
last_end = ev[0][0]
for s, e, k in ev:
    if s > last_end and s - last_end > 200_000: longs.append((s - last_end, (last_end - ev[0][0]) / 1e6, last_name, short(k)))
    last_name = short(k)
    name = k.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:70]
    busy[name][0] += e - s; busy[name][1] += 1
    if s > last_end:
        g = s - last_end
        if g > 200_000: big += g
        else: gaps += g; gap_hist[min(g // 2000 * 2, 30)] += 1
    last_end = max(last_end, e)
tot_busy = sum(v[0] for v in busy.values())
print(f"span {span / 1e6:.2f} ms; kernels busy {tot_busy / 1e6:.2f} ms ({100 * tot_busy / span:.1f} %); gaps < 200 us {gaps / 1e6:.2f} ms ({100 * gaps / span:.1f} %); "
      f"long idle (host sync) {big / 1e6:.2f} ms ({100 * big / span:.1f} %); {len(ev)} launches")
durs = collections.defaultdict(list)
for s_, e_, k_ in ev: durs[k_.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:70]].append((e_ - s_) / 1e3)
for k, (ns, c) in sorted(busy.items(), key=lambda kv: -kv[1][0])[:24]:
    d = sorted(durs[k])
    print(f"  {k:70s} {c:6d} x {ns / c / 1e3:8.1f} us = {ns / 1e6:8.2f} ms ({100 * ns / span:5.1f} %)  min {d[0]:.1f} med {d[len(d) // 2]:.1f} max {d[-1]:.1f}")
print("gap histogram (us bucket: count):", " ".join(f"{b}:{c}" for b, c in sorted(gap_hist.items())))
if n_long:
    print(f"{len(longs)} idle intervals > 200 us; the longest (ms idle, at ms of the window, after kernel -> before kernel):")
    for g, at, a, b in sorted(longs, reverse=True)[:n_long]: print(f"  {g / 1e6:7.2f} at {at:8.1f}  {a} -> {b}")
