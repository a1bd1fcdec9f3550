#!/bin/bash
# MFMA-busy share and effective clock of the bf16 image kernels (own rocprofv3 --pmc pass; no other trace domains).
# usage (GPU box): bash tools/pmc_mfma_bf16.sh <tag> [microbench_bf16 shapes...]
set -e
R=$GRAFT_REPO_ROOT; TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_ACTIVE_INST_LDS --output-format csv -d $R/gpurun_out/pmc_$TAG -- python3 $R/tools/${MB:-microbench_bf16.py} "$@" > $R/gpurun_out/pmc_$TAG.log 2>&1
cd $R
python3 - <<PY
import csv, glob, collections
f = glob.glob("gpurun_out/pmc_$TAG/**/*_counter_collection.csv", recursive=True)[0]
kt = glob.glob("gpurun_out/pmc_$TAG/**/*_kernel_trace.csv", recursive=True)[0]
dur = collections.defaultdict(list)
for r in csv.DictReader(open(kt)):
    dur[(r["Kernel_Name"], r["Grid_Size_X"] if "Grid_Size_X" in r else "")].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
agg = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"]
    if "tile" not in k and "wide" not in k and "wino" not in k: continue
    key = (k, r.get("Grid_Size", r.get("Grid_Size_X", "")), r.get("LDS_Block_Size", ""), r.get("Dispatch_Id", "0"))
    agg.setdefault(key, collections.defaultdict(list))[r["Counter_Name"]].append(float(r["Counter_Value"]))
# one line per kernel instantiation: averages over its dispatches
per = collections.OrderedDict()
for (k, gs, lds, did), d in agg.items():
    per.setdefault(k, []).append({c: sum(v) / len(v) for c, v in d.items()})
durk = collections.defaultdict(list)
for (k, _), v in dur.items():
    durk[k] += v
out = []
for k, lst in per.items():
    m = {c: sum(x[c] for x in lst) / len(lst) for c in lst[0]}
    ns = sum(durk[k]) / max(1, len(durk[k]))
    cyc = m["GRBM_GUI_ACTIVE"] / 8.0            # summed over the 8 XCDs
    name = k.replace("void (anonymous namespace)::", "").split("(")[0]
    out.append(f"{name:52s} {ns/1e3:8.1f} us  clock {cyc/ns:5.2f} GHz  MFMA busy {m['SQ_VALU_MFMA_BUSY_CYCLES']/(cyc*1024):5.3f}  "
               f"waves waiting {m['SQ_WAIT_ANY']/m['SQ_WAVE_CYCLES']:5.3f}  dispatches {len(lst)}")
open("gpurun_out/pmc_${TAG}_summary.txt", "w").write("\n".join(out) + "\n")
print("\n".join(out))
PY
