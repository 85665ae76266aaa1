"""MFMA utilisation from hardware counters, per kernel family: one rocprofv3 pass
    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -- python bench.py ...
SQ_VALU_MFMA_BUSY_CYCLES counts the cycles a SIMD's MFMA pipe is busy, summed over all 1024 SIMDs (MI355X_MICROARCH.md: 32 per
32x32x16 bf16 MFMA, i.e. 16 per 16x16x32 -- checked here: wgrad_group_kernel 16.6 M cycles per launch = 16.05 GFLOP / 16384 FLOP x 16 + the
row-sum MFMAs).  GRBM_GUI_ACTIVE comes back summed over the 8 XCDs on this part (per launch ~ 8 x duration x clock), so the gfx94x
`MfmaUtil` formula (ROCm 7.2 ships no gfx950 derived-counter section) is used with GUI_ACTIVE / 8:
    util = MFMA busy cycles / (1024 SIMDs x GRBM_GUI_ACTIVE / 8)
for the launch as it ran under the counter pass (kernels serialised: isolated durations).  bench.py re-prices the same busy cycles with
the IN-STEP duration of the family: busy / (1024 x t x 2.4 GHz).  Writes the json bench.py reads for `roofline.*.mfma_util_counter`, stamped like the other counter summaries.
usage: python tools/mfma_summary.py counter_collection.csv [out.json]"""
import collections, csv, json, os, re, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import FAMILIES, kernel_source_stamp

SIMDS, XCDS = 256 * 4, 8
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(set)
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        n = re.sub(r"\(anonymous namespace\)::", "", re.sub(r"^void ", "", r["Kernel_Name"]))
        n = re.sub(r"\(.*$", "", n)[:60]
        agg[n][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[n].add(r.get("Dispatch_Id", r.get("Correlation_Id")))
print(f"{'kernel':60s} {'calls':>6s} {'mfma_busy/launch':>18s} {'gui_active/launch':>18s} {'mfma_util':>10s} {'sq_busy/launch':>16s}")
rows = []
for k, c in agg.items():
    n = max(1, len(cnt[k]))
    mf, gui, sq = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / n, c.get("GRBM_GUI_ACTIVE", 0.0) / n, c.get("SQ_BUSY_CYCLES", 0.0) / n
    rows.append((mf * n, k, n, mf, gui, sq))
for _, k, n, mf, gui, sq in sorted(rows, reverse=True)[:25]:
    print(f"{k:60s} {n:6d} {mf:18.0f} {gui:18.0f} {(mf * XCDS / (gui * SIMDS) if gui else 0):10.4f} {sq:16.0f}")
if len(sys.argv) > 2:
    fams = {}
    for fam, pre in FAMILIES.items():
        ks = [k for k in agg if k.startswith(pre)]
        n = sum(len(cnt[k]) for k in ks)
        if not n:
            continue
        mf = sum(agg[k].get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) for k in ks)
        gui = sum(agg[k].get("GRBM_GUI_ACTIVE", 0.0) for k in ks)
        fams[fam] = {"launches_counted": n, "mfma_busy_cycles_per_launch": mf / n, "gui_active_cycles_per_launch": gui / n,
                     "mfma_util": mf * XCDS / (gui * SIMDS) if gui else 0.0}
    json.dump({"source": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE (own pass), bench.py --steps 3 --warmup 2; "
                         "util = MFMA busy cycles / (1024 SIMDs x GUI-active cycles / 8 XCDs): the launch as it ran in the (serialising) counter pass",
               "source_stamp": kernel_source_stamp(), "families": fams}, open(sys.argv[2], "w"), indent=1)
