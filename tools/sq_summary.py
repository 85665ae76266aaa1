"""Where the waves of each kernel spend their cycles, from SQ counters (three rocprofv3 passes over the same bench command, merged):
    pass 1  SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY
    pass 2  SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS
    pass 3  SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM SQ_INSTS_LDS
Per kernel (all launches of the pass summed, then divided by the launch count): wave-cycles (sum over waves of resident cycles),
the share of them spent waiting for anything / for an instruction's operands (memory, LDS, export counters), the share in which an
instruction of the wave was executing (any / vector ALU / vector memory / LDS), and instruction counts per wave-kilocycle.
usage: python tools/sq_summary.py pass1.csv pass2.csv pass3.csv"""
import collections, csv, re, sys
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(set))
for path in sys.argv[1:]:
    with open(path) as f:
        for r in csv.DictReader(f):
            n = re.sub(r"\(anonymous namespace\)::", "", re.sub(r"^void ", "", r["Kernel_Name"]))
            n = re.sub(r"\(.*$", "", n)[:52]
            agg[n][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[n][r["Counter_Name"]].add(r.get("Dispatch_Id", r.get("Correlation_Id")))
def per(k, c):
    n = max(1, len(cnt[k][c]))
    return agg[k].get(c, 0.0) / n
print(f"{'kernel':52s} {'calls':>5s} {'wave-kcyc':>10s} | {'wait any':>8s} {'wait inst':>9s} | {'act any':>7s} {'valu':>6s} {'vmem':>6s} {'lds':>6s} | per wave-kcycle: {'VALU':>6s} {'MFMA':>6s} {'VMEM':>6s} {'LDS':>6s}")
rows = []
for k in agg:
    wc = per(k, "SQ_WAVE_CYCLES")
    if wc <= 0:
        continue
    rows.append((agg[k]["SQ_WAVE_CYCLES"], k, wc))
for _, k, wc in sorted(rows, reverse=True)[:28]:
    f = lambda c: per(k, c) / wc
    # (SQ_WAVE_CYCLES and the WAIT / ACTIVE counters tick once per 4 cycles per wave on this part; the ratios are what is read)
    print(f"{k:52s} {len(cnt[k]['SQ_WAVE_CYCLES']):5d} {wc / 1e3:10.1f} | {f('SQ_WAIT_ANY'):8.2f} {f('SQ_WAIT_INST_ANY'):9.2f} | "
          f"{f('SQ_ACTIVE_INST_ANY'):7.2f} {f('SQ_ACTIVE_INST_VALU'):6.2f} {f('SQ_ACTIVE_INST_VMEM'):6.2f} {f('SQ_ACTIVE_INST_LDS'):6.2f} | "
          f"{'':16s} {1e3 * f('SQ_INSTS_VALU'):6.1f} {1e3 * f('SQ_INSTS_MFMA'):6.1f} {1e3 * f('SQ_INSTS_VMEM'):6.1f} {1e3 * f('SQ_INSTS_LDS'):6.1f}")
