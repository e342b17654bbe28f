"""Summarise a rocprofv3 sqlite (rocpd) kernel trace: per-kernel count / total / avg / min / max (us)."""
import sqlite3, sys, re
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
name_col = 'name' if 'name' in cols else [c for c in cols if 'name' in c][0]
rows = cur.execute("select %s, start, end from kernels" % name_col).fetchall()
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
agg = {}
for nm, s, e in rows:
    nm = re.sub(r'\(.*', '', nm)
    a = agg.setdefault(nm, [0, 0, 1e30, 0])
    d = (e - s) / 1e3
    a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
tot = sum(a[1] for a in agg.values())
print("%-70s %7s %12s %10s %10s %10s %6s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "%"))
for nm, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%-70s %7d %12.1f %10.2f %10.2f %10.2f %6.1f" % (nm[:70], a[0], a[1], a[1] / a[0], a[2], a[3], 100 * a[1] / tot))
