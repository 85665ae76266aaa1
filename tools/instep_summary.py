"""Per-family IN-STEP kernel durations from a rocprofv3 kernel trace (rocpd sqlite db) of bench.py: the steps between
consecutive adamw_kernel launches, the first `skip` of them dropped (warm-up / capture).  Writes the json bench.py reads
for `roofline.*.in_step` (stamped with the hash of the kernel sources, like profiles/pmc_traffic.json).
usage: python tools/instep_summary.py trace.db out.json [skip=8]"""
import json, os, re, sqlite3, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import FAMILIES, kernel_source_stamp

db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
skip = int(sys.argv[3]) if len(sys.argv) > 3 else 8
cols = [r[1] for r in cur.execute("pragma table_info(kernels)").fetchall()]
qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else "0")
rows = cur.execute(f"select name, start, end, {qcol} from kernels order by start").fetchall()
clean = lambda n: re.sub(r"\(.*$", "", re.sub(r"\(anonymous namespace\)::|^void ", "", n))
rows = [(clean(n), s, e, q) for n, s, e, q in rows]
ad = [i for i, r in enumerate(rows) if r[0].startswith("adamw_kernel")]
steps = [rows[ad[k] + 1: ad[k + 1] + 1] for k in range(skip, len(ad) - 1)]
FAM = FAMILIES
fams = {}
for fam, pre in FAM.items():
    n = sum(1 for st in steps for r in st if r[0].startswith(pre))
    if not n:
        continue
    us = sum((r[2] - r[1]) / 1e3 for st in steps for r in st if r[0].startswith(pre))
    fams[fam] = {"launches_per_step": n / len(steps), "mean_launch_us": us / n, "us_per_step": us / len(steps)}
wall = [(max(r[2] for r in st) - st[0][1]) / 1e3 for st in steps]
# when the SIDE queue's first kernel starts in the traced step, measured from the head kernel of the backward chain (tail_bwd_*): the
# un-traced twin of this number is tools/step_stamps.py's "side group begins #0" - "chain backward begins" (profiles/step_stamps.json);
# the difference is what the tracer does to the thing it measures (round-5 review, weak #4b)
side = []
for st in steps:
    chainq = st[0][3]
    bwd0 = next((r[1] for r in st if r[0].startswith("tail_bwd")), None)
    s0 = next((r[1] for r in st if r[3] != chainq), None)
    if bwd0 is not None and s0 is not None:
        side.append((s0 - bwd0) / 1e3)
json.dump({"source": "rocprofv3 --kernel-trace of bench.py --steps 20 --warmup 5; kernels between consecutive adamw launches, "
                     f"first {skip} steps dropped; durations under the tracer (the traced step is ~15 % longer than the untraced one)",
           "source_stamp": kernel_source_stamp(), "steps": len(steps), "traced_step_us": sum(wall) / len(wall),
           "traced_side_queue_start_after_backward_begins_us": (sum(side) / len(side)) if side else None,
           "families": fams}, open(sys.argv[2], "w"), indent=1)
print(json.dumps(fams, indent=1))
