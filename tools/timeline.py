"""dev tool: per-stream timeline of one of the last training steps (the median one by wall time) in a rocprofv3 rocpd db (kernel-trace):
wall, busy union, per-stream busy, overlap, and the main stream's longest kernels / idle gaps."""
import sqlite3, sys, re
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
print("columns:", cols)
import os
sc = os.environ.get("TL_COL", "queue_id")
rows = cur.execute(f"select name, start, end, {sc} from kernels order by start").fetchall()
# a step ends with adamw_kernel: take the kernels between the last two adamw launches
ad = [i for i, r in enumerate(rows) if "adamw_kernel" in r[0]]
# ... of the last five steps the one with the median wall time (the very last one of a trace is sometimes stretched by the tracer's flush)
cands = []
for k in range(2, min(7, len(ad) + 1)):
    st = rows[ad[-k] + 1: ad[-k + 1] + 1]
    cands.append((max(r[2] for r in st) - st[0][1], k))
cands.sort()
kk = cands[len(cands) // 2][1]
lo, hi = ad[-kk] + 1, ad[-kk + 1] + 1
step = rows[lo:hi]
print(f"(step -{kk - 1} of the trace: the median wall of the last {len(cands)})")
t0, t1 = step[0][1], max(r[2] for r in step)
print(f"step: {len(step)} kernels, wall {(t1 - t0) / 1e3:.1f} us (stream column: {sc})")
def union(iv):
    iv = sorted(iv); tot = 0; cs, ce = iv[0]
    for s, e in iv[1:]:
        if s > ce: tot += ce - cs; cs, ce = s, e
        else: ce = max(ce, e)
    return tot + ce - cs
print(f"busy union {union([(r[1], r[2]) for r in step]) / 1e3:.1f} us; sum of kernel time {sum(r[2] - r[1] for r in step) / 1e3:.1f} us")
streams = {}
for r in step: streams.setdefault(r[3], []).append(r)
for sid, rs in sorted(streams.items(), key=lambda kv: -len(kv[1])):
    busy = sum(r[2] - r[1] for r in rs)
    span = (max(r[2] for r in rs) - min(r[1] for r in rs))
    gaps = sorted(((rs[i + 1][1] - rs[i][2]) / 1e3 for i in range(len(rs) - 1)), reverse=True)
    print(f" stream {sid}: {len(rs)} kernels, busy {busy / 1e3:.1f} us, span {span / 1e3:.1f} us, first at +{(rs[0][1] - t0) / 1e3:.1f}, "
          f"last end +{(max(r[2] for r in rs) - t0) / 1e3:.1f}, median gap {gaps[len(gaps) // 2] if gaps else 0:.2f} us, "
          f"sum gaps {sum(g for g in gaps if g > 0):.1f} us, top gaps {[round(g, 1) for g in gaps[:6]]}")
if len(sys.argv) > 2:
    for r in step:
        n = re.sub(r"\(anonymous namespace\)::|^void ", "", r[0])[:60]
        print(f"  q{r[3]} +{(r[1] - t0) / 1e3:8.1f} {(r[2] - r[1]) / 1e3:7.1f} {n}")
# gap positions of the last few steps
for k in range(2, min(6, len(ad))):
    st = rows[ad[-k] + 1: ad[-k + 1] + 1]
    base = st[0][1]
    gaps = sorted(((st[i + 1][1] - max(r[2] for r in st[:i + 1][-8:])) / 1e3, i, (st[i][2] - base) / 1e3) for i in range(len(st) - 1))
    print(f"step -{k - 1}: wall {(max(r[2] for r in st) - base) / 1e3:.1f} us; top gaps (us, after kernel #, at +us):",
          [(round(g, 1), i, round(t, 0)) for g, i, t in gaps[-5:]])
