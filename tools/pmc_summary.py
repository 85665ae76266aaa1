"""dev tool: per-kernel mean FETCH_SIZE / WRITE_SIZE (KB -> MB) from two rocprofv3 --pmc csv passes."""
import csv, sys, re, collections
def load(path, counter):
    agg = collections.defaultdict(lambda: [0, 0.0])
    with open(path) as f:
        for r in csv.DictReader(f):
            if r.get("Counter_Name") != counter: continue
            n = re.sub(r"\(anonymous namespace\)::", "", re.sub(r"^void ", "", r["Kernel_Name"]))
            n = re.sub(r"\(.*$", "", n)[:60]
            a = agg[n]; a[0] += 1; a[1] += float(r["Counter_Value"])
    return agg
f = load(sys.argv[1], "FETCH_SIZE"); w = load(sys.argv[2], "WRITE_SIZE")
print(f"{'kernel':60s} {'calls':>6s} {'fetch_MB/call(x2 corr)':>24s} {'write_MB/call':>14s}")
rows = []
for k in f:
    fm = f[k][1] / f[k][0] / 1024.0 * 2.0     # KB -> MB; gfx950 FETCH_SIZE counts 64 B per 128-B request: x2
    wm = w.get(k, [1, 0.0])[1] / max(1, w.get(k, [1, 0.0])[0]) / 1024.0
    rows.append((f[k][0] * (fm + wm), k, f[k][0], fm, wm))
for _, k, n, fm, wm in sorted(rows, reverse=True)[:25]:
    print(f"{k:60s} {n:6d} {fm:24.2f} {wm:14.2f}")
tot_f = sum(v[1] for v in f.values()) / 1024.0 * 2.0; tot_w = sum(v[1] for v in w.values()) / 1024.0
print(f"TOTAL over run: fetch {tot_f:.0f} MB (x2 corrected), write {tot_w:.0f} MB")
# per-family aggregates for bench.py's roofline.traffic (optional 3rd argument: output json), stamped with the hash of
# the kernel sources they were measured on (bench.kernel_source_stamp): a stale file is not reported
if len(sys.argv) > 3:
    import json, os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import FAMILIES, kernel_source_stamp
    FAM = FAMILIES
    fams = {}
    for fam, pre in FAM.items():
        n = sum(v[0] for k, v in f.items() if k.startswith(pre))
        if not n:
            continue
        gf = sum(v[1] for k, v in f.items() if k.startswith(pre)) * 1024.0 * 2.0
        gw = sum(v[1] for k, v in w.items() if k.startswith(pre)) * 1024.0
        fams[fam] = {"launches_counted": n, "fetch_bytes_per_launch": gf / n, "write_bytes_per_launch": gw / n,
                     "hbm_bytes_per_launch": (gf + gw) / n}
    json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), bench.py --steps 3 --warmup 2; "
                         "FETCH_SIZE x2 (gfx950 counts 64 B per 128-B request, MI355X_MICROARCH.md HBM section); "
                         "WRITE_SIZE uncorrected; calibration of both on a known-bytes copy: profiles/pmc_calibration.txt",
               "source_stamp": kernel_source_stamp(), "families": fams}, open(sys.argv[3], "w"), indent=1)
