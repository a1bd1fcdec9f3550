#!/bin/bash
# Round profile set (GPU box): rocprofv3 kernel stats of the fp32 and bf16 bench, HBM-traffic PMC passes (separate runs,
# FETCH_SIZE / WRITE_SIZE only), the dead-weight scan.  usage: bash tools/profile_round.sh r04 <commit>
set -x
R=$GRAFT_REPO_ROOT; TAG=${1:-r04}; COMMIT=${2:-unrecorded}; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_f32 -o k -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-profile --no-extra-legs > $O/stats_f32.json 2> $O/stats_f32.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_bf16 -o k -- python3 $R/bench.py --dtype bf16 --steps 4 --warmup 2 --no-cpu-baseline --no-profile --no-extra-legs > $O/stats_bf16.json 2> $O/stats_bf16.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_dead -o k -- python3 $R/tools/dead_scan_bench.py > $O/dead_scan.json 2> $O/dead_scan.err
# BASELINE configs[2] per-GPU shape, configs[3] (512^2, nudge in the loop) and configs[4] (1024^2, decoder checkpointed, blockwise attention)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_bf16_b32 -o k -- python3 $R/bench.py --dtype bf16 --batch 32 --steps 3 --warmup 2 --no-cpu-baseline --no-profile --no-extra-legs > $O/stats_bf16_b32.json 2> $O/stats_bf16_b32.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_bf16_512 -o k -- python3 $R/bench.py --dtype bf16 --res 512 --batch 8 --nudge-interval 100 --steps 3 --warmup 2 --no-cpu-baseline --no-profile --no-extra-legs > $O/stats_bf16_512.json 2> $O/stats_bf16_512.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_bf16_1024 -o k -- python3 $R/bench.py --dtype bf16 --res 1024 --batch 2 --checkpoint-decoder --steps 3 --warmup 2 --no-cpu-baseline --no-profile --no-extra-legs > $O/stats_bf16_1024.json 2> $O/stats_bf16_1024.err
for P in f32 bf16; do
  for C in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/pmc_${P}_$C -- python3 $R/bench.py --dtype $P --steps 1 --warmup 1 --no-cpu-baseline --no-profile --no-extra-legs > $O/pmc_${P}_$C.json 2> $O/pmc_${P}_$C.err
  done
done
# matrix-pipe / issue counters of the fp32 Winograd kernels (microbench, own passes: 8 SQ slots each) and of the bf16 image kernels
for SET in A B; do
  if [ $SET = A ]; then C="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; else C="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS"; fi
  MB_ONLY=fwd rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE $C --output-format csv -d $O/pmc_wino4_$SET -- python3 $R/tools/microbench_wino4.py c128 c512 > $O/pmc_wino4_$SET.log 2>&1
done
rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_ACTIVE_INST_LDS --output-format csv -d $O/pmc_mfma_bf16 -- python3 $R/tools/microbench_bf16.py > $O/pmc_mfma_bf16.log 2>&1
cd $R
python3 tools/pmc_summary.py $O/pmc_wino4_A $O/pmc_wino4_B > $O/pmc_wino4.txt
python3 tools/pmc_summary.py $O/pmc_mfma_bf16 > $O/pmc_mfma_bf16.txt
python3 tools/hbm_traffic.py $O/pmc_f32_FETCH_SIZE $O/pmc_f32_WRITE_SIZE $O/hbm_traffic_f32.json $COMMIT
python3 tools/hbm_traffic.py $O/pmc_bf16_FETCH_SIZE $O/pmc_bf16_WRITE_SIZE $O/hbm_traffic_bf16.json $COMMIT
# keep the merged-back volume small: the raw counter CSVs are large
rm -rf $O/pmc_*_SIZE/*/ $O/pmc_wino4_*/*/ $O/pmc_mfma_bf16/*/ 2>/dev/null; find $O -name "*_kernel_trace.csv" -delete; find $O -name "*_agent_info.csv" -delete
ls -la $O
