"""Per-kernel averages of the SQ counters of one or more rocprofv3 --pmc passes (each with GRBM_GUI_ACTIVE), divided by the SIMD-cycles
of the launch: value / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs).  usage: python tools/pmc_summary.py <pass_dir> [<pass_dir> ...]"""
import collections
import csv
import glob
import sys

print("# rocprofv3 --pmc passes, values per SIMD-cycle (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs); SQ_VALU_MFMA_BUSY_CYCLES per SIMD-cycle = share of")
print("# time the matrix pipe is busy (fp32 32x32x2: 64 cycles per MFMA; bf16 32x32x16: 32); counters slow the kernels by ~10-15 %")
for d in sys.argv[1:]:
    files = glob.glob(d + "/**/*_counter_collection.csv", recursive=True)
    if not files:
        print(f"# {d}: no counter file")
        continue
    agg = collections.OrderedDict()
    for r in csv.DictReader(open(files[0])):
        k = r["Kernel_Name"].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").split("(")[0]
        if not any(s in k for s in ("tile", "wide", "wino", "attn", "igemm", "wgrad", "thin", "conv1")):
            continue
        agg.setdefault(k, collections.defaultdict(list))[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(f"## {d.rstrip('/').split('/')[-1]}")
    for k, dd in agg.items():
        m = {c: sum(v) / len(v) for c, v in dd.items()}
        n = len(next(iter(dd.values())))
        cyc = m.pop("GRBM_GUI_ACTIVE") / 8.0
        print(f"{k:52s} launches {n:4d} cycles {cyc:10.0f}  " + "  ".join(f"{c} {v / (cyc * 1024):7.4f}" for c, v in sorted(m.items())))
