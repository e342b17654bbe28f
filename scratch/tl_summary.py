"""Per-queue / per-kernel busy time of a timeline.txt (scratch/timeline.py output): python scratch/tl_summary.py <timeline.txt>"""
import sys, re, collections
agg = collections.defaultdict(lambda: [0, 0.0])
qbusy = collections.defaultdict(float)
end = 0.0
for line in open(sys.argv[1]):
    m = re.match(r"\s*([\d.]+)\s+([\d.]+)\s+q(\S+)\s+(\S+)", line)
    if not m: continue
    t, d, q, nm = float(m.group(1)), float(m.group(2)), m.group(3), m.group(4)
    agg[(q, nm)][0] += 1; agg[(q, nm)][1] += d; qbusy[q] += d; end = max(end, t + d)
print("end %.1f us" % end)
for q in sorted(qbusy): print("queue %s busy %.1f us (%.0f %%)" % (q, qbusy[q], 100 * qbusy[q] / end))
for (q, nm), (cnt, tot) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
    print("q%-3s %-30s calls %5d total %9.1f us avg %8.1f" % (q, nm, cnt, tot, tot / cnt))
