"""dev tool: per-kernel stats from a rocprofv3 rocpd sqlite db -> text table (and optional csv)."""
import sqlite3, sys, re
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
name_c = "name" if "name" in cols else cols[0]
rows = cur.execute(f"select {name_c}, start, end from kernels").fetchall()
agg = {}
for n, s, e in rows:
    n = re.sub(r"^void ", "", n); n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"\(.*$", "", n)[:90]
    a = agg.setdefault(n, [0, 0.0, 1e30, 0.0]); d = (e - s) / 1e3
    a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
tot = sum(a[1] for a in agg.values())
div = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0   # number of steps to normalise by
lines = [f"{'kernel':90s} {'calls':>7s} {'total_us':>11s} {'avg_us':>9s} {'min_us':>8s} {'max_us':>8s} {'%':>6s}"]
for n, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    lines.append(f"{n:90s} {a[0]:7d} {a[1]:11.1f} {a[1]/a[0]:9.2f} {a[2]:8.2f} {a[3]:8.2f} {100*a[1]/tot:6.2f}")
lines.append(f"TOTAL kernel time {tot:.1f} us over {len(rows)} dispatches" + (f"; per step ({div:g} steps): {tot/div:.1f} us" if div != 1 else ""))
print("\n".join(lines))
