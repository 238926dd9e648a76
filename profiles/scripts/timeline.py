"""Overlap of the kernels of a bench run from a rocprofv3 --kernel-trace csv: per kernel the
average duration, and over the steady part of the run the fraction of the time 0, 1, 2, 3+
kernels were in flight, plus the time per step during which only one given kernel ran.
usage: python profiles/scripts/timeline.py <dir with *_kernel_trace.csv> [first step] [last step]
(steps are counted by the launches of the z-buffer kernel)"""
import collections
import csv
import glob
import sys

rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("mvf::", "")[:40]
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name, r.get("Queue_Id", "")))
rows.sort()
zb = [r for r in rows if r[2].startswith("tile_zbuffer")]
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 20
stop = int(sys.argv[3]) if len(sys.argv) > 3 else len(zb) - 5
t0, t1 = zb[skip][0], zb[stop][0]
steps = stop - skip
sel = [r for r in rows if r[0] >= t0 and r[1] <= t1]
ev = []
for s, e, n, q in sel:
    ev.append((s, 1, n))
    ev.append((e, -1, n))
ev.sort()
active = collections.Counter()
conc = collections.Counter()
alone = collections.Counter()
last = t0
for t, d, n in ev:
    k = sum(active.values())
    conc[min(k, 4)] += t - last
    if k == 1:
        alone[[a for a, c in active.items() if c][0]] += t - last
    last = t
    active[n] += d
total = t1 - t0
print(f"{steps} steps, {total / steps / 1e3:.1f} us per step")
print("kernels in flight:", {k: round(v / total, 3) for k, v in sorted(conc.items())})
dur = collections.defaultdict(list)
for s, e, n, q in sel:
    dur[n].append(e - s)
print(f"{'kernel':42s} {'avg us':>8s} {'per step':>9s} {'alone us/step':>14s}")
for n, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    print(f"{n:42s} {sum(v) / len(v) / 1e3:8.1f} {sum(v) / steps / 1e3:9.1f} {alone[n] / steps / 1e3:14.1f}")
print("queues:", collections.Counter(q for _, _, _, q in sel))
