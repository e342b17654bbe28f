"""Print the kernel timeline of the last fit in a rocprofv3 kernel-trace CSV: t_start (us), dur (us), queue, kernel."""
import csv, sys, re, glob
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
ks = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), re.sub(r"\(.*", "", r["Kernel_Name"])) for r in rows]
ks.sort()
# the last kbuild marks the start of the last fit
idx = max(i for i, k in enumerate(ks) if "kbuild_kernel" in k[3])
# ... of which the K build is launched in two parts (head columns first), after the upload of y and the padding kernel
while idx > 0 and ("kbuild_kernel" in ks[idx - 1][3] or "fill_pad" in ks[idx - 1][3] or "copyBuffer" in ks[idx - 1][3] or "fillBuffer" in ks[idx - 1][3] or "set_flag" in ks[idx - 1][3] or "upload_pad" in ks[idx - 1][3]) \
        and ks[idx][0] - ks[idx - 1][1] < 100000:
    idx -= 1
t0 = ks[idx][0]
lo = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
hi = float(sys.argv[3]) if len(sys.argv) > 3 else 1e9
short = lambda s: re.sub(r"^void |<.*", "", s)[:28]
qs = sorted(set(k[2] for k in ks[idx:]))
for s, e, q, nm in ks[idx:]:
    ts = (s - t0) / 1e3
    if lo <= ts <= hi:
        print("%9.1f %8.1f  q%s %s%s" % (ts, (e - s) / 1e3, q, "    " * qs.index(q), short(nm)))
print("total %.1f us" % ((max(k[1] for k in ks[idx:]) - t0) / 1e3))
