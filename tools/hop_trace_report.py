"""Timeline of the last tokens of a chain run from a rocprofv3 --kernel-trace --memory-copy-trace directory: per k6_ring launch its start,
end, and everything (copies, other kernels) between it and the next k6_ring launch. usage: hop_trace_report.py <dir>"""
import csv, glob, sys

d = sys.argv[1]
ev = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K " + r["Kernel_Name"][:60]))
for f in glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C " + r.get("Direction", "copy")))
ev.sort()
ring = [i for i, e in enumerate(ev) if "k6_ring" in e[2] or "k47_persist" in e[2]]
if len(ring) < 12:
    print("too few persistent launches in the trace:", len(ring)); sys.exit(0)
lo = ring[-12]
t0 = ev[lo][0]
prev_end = None
gaps = []
for i in range(lo, len(ev)):
    s, e, n = ev[i]
    gap = "" if prev_end is None else f"  gap {((s - prev_end) / 1e3):7.2f} us"
    print(f"{(s - t0) / 1e3:10.2f} {(e - t0) / 1e3:10.2f}  dur {((e - s) / 1e3):8.2f} us{gap}  {n}")
    prev_end = e
durs = [(ev[i][1] - ev[i][0]) / 1e3 for i in ring[-12:]]
span = (ev[ring[-1]][1] - ev[ring[-12]][0]) / 1e3
print(f"\nlast 12 persistent launches: sum of durations {sum(durs):.1f} us, wall {span:.1f} us, between launches {(span - sum(durs)) / 11:.2f} us each")
