#!/bin/bash
# MFMA-busy share and effective clock of the contraction kernels (own rocprofv3 --pmc pass; no other trace domains).
# usage (GPU box): bash tools/pmc_mfma.sh <tag> [microbench shapes...]   env MB_PREC=bf16 MB_PACK=1 for the bf16 kernels
set -e
R=$GRAFT_REPO_ROOT; TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_ACTIVE_INST_LDS --output-format csv -d $R/gpurun_out/pmc_$TAG -- python3 $R/tools/microbench.py "$@" > $R/gpurun_out/pmc_$TAG.log 2>&1
cd $R
python3 - <<PY
import csv, glob, collections
f = glob.glob("gpurun_out/pmc_$TAG/**/*_counter_collection.csv", recursive=True)[0]
kt = glob.glob("gpurun_out/pmc_$TAG/**/*_kernel_trace.csv", recursive=True)[0]
dur = collections.defaultdict(list)
for r in csv.DictReader(open(kt)):
    dur[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
agg = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"]
    if "tile" not in k: continue
    agg.setdefault(k, collections.defaultdict(list))[r["Counter_Name"]].append(float(r["Counter_Value"]))
out = []
for k, d in agg.items():
    m = {c: sum(v) / len(v) for c, v in d.items()}
    ns = sum(dur[k]) / len(dur[k])
    cyc = m["GRBM_GUI_ACTIVE"] / 8.0            # summed over the 8 XCDs
    name = k.replace("void (anonymous namespace)::", "").split("(")[0]
    out.append(f"{name:50s} {ns/1e3:8.1f} us  clock {cyc/ns:5.2f} GHz  MFMA busy {m['SQ_VALU_MFMA_BUSY_CYCLES']/(cyc*1024):5.3f}  "
               f"waves waiting {m['SQ_WAIT_ANY']/m['SQ_WAVE_CYCLES']:5.3f}")
open("gpurun_out/pmc_${TAG}_summary.txt", "w").write("\n".join(out) + "\n")
print("\n".join(out))
PY
