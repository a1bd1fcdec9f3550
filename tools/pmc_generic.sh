#!/bin/bash
# Any set of SQ counters for the kernels of a microbench, averaged per kernel instantiation and divided by the SIMD-cycles of the
# launch (own rocprofv3 --pmc pass).  usage (GPU box): PMC="SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS" bash tools/pmc_generic.sh <tag> <script.py> [args...]
set -e
R=$GRAFT_REPO_ROOT; TAG=$1; SCRIPT=$2; shift; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE $PMC --output-format csv -d $R/gpurun_out/pmcg_$TAG -- python3 $R/tools/$SCRIPT "$@" > $R/gpurun_out/pmcg_$TAG.log 2>&1
cd $R
python3 - <<PY
import csv, glob, collections
f = glob.glob("gpurun_out/pmcg_$TAG/**/*_counter_collection.csv", recursive=True)[0]
agg = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].replace("void (anonymous namespace)::", "").split("(")[0]
    if not any(s in k for s in ("tile", "wide", "wino", "attn", "igemm", "wgrad")): continue
    agg.setdefault(k, collections.defaultdict(list))[r["Counter_Name"]].append(float(r["Counter_Value"]))
out = []
for k, d in agg.items():
    m = {c: sum(v) / len(v) for c, v in d.items()}
    cyc = m.pop("GRBM_GUI_ACTIVE") / 8.0
    out.append(f"{k:48s} cycles {cyc:10.0f}  " + "  ".join(f"{c} {v / (cyc * 1024):7.4f}/SIMD-cycle" for c, v in m.items()))
open("gpurun_out/pmcg_${TAG}_summary.txt", "w").write("\n".join(out) + "\n")
print("\n".join(out))
PY
