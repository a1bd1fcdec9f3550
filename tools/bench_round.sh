#!/bin/bash
# The round's bench lines (GPU box): fp32 metric line with cpu_baseline, bf16 configurations, tracker A/B.  usage: bash tools/bench_round.sh r03
R=$GRAFT_REPO_ROOT; TAG=${1:-r03}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
python bench.py > $O/bench_f32.json 2> $O/bench_f32.err
python bench.py --no-profile --no-cpu-baseline > $O/bench_f32_noprof.json 2>> $O/bench_f32.err
python bench.py --no-profile --no-cpu-baseline --no-tracking > $O/bench_f32_notrack.json 2>> $O/bench_f32.err
python bench.py --dtype bf16 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_bf16.json 2> $O/bench_bf16.err
python bench.py --dtype bf16 --steps 10 --warmup 3 --no-cpu-baseline --no-profile > $O/bench_bf16_noprof.json 2>> $O/bench_bf16.err
python bench.py --dtype bf16 --steps 10 --warmup 3 --no-cpu-baseline --no-profile --no-tracking > $O/bench_bf16_notrack.json 2>> $O/bench_bf16.err
python bench.py --dtype bf16 --batch 32 --steps 8 --warmup 3 --no-cpu-baseline > $O/bench_bf16_b32.json 2>> $O/bench_bf16.err
python bench.py --dtype bf16 --batch 32 --steps 8 --warmup 3 --no-cpu-baseline --no-profile > $O/bench_bf16_b32_noprof.json 2>> $O/bench_bf16.err
python bench.py --dtype bf16 --batch 32 --steps 8 --warmup 3 --no-cpu-baseline --no-profile --act-fp32 > $O/bench_bf16_b32_actfp32.json 2>> $O/bench_bf16.err
python bench.py --dtype bf16 --res 512 --batch 8 --nudge-interval 100 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_bf16_512.json 2>> $O/bench_bf16.err
python bench.py --dtype bf16 --res 1024 --batch 2 --checkpoint-decoder --steps 4 --warmup 2 --no-cpu-baseline > $O/bench_bf16_1024.json 2>> $O/bench_bf16.err
python bench.py --res 512 --batch 8 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_f32_512.json 2>> $O/bench_f32.err
for f in $O/bench_*.json; do python -c "
import json,sys;d=json.load(open('$f'));print('$f'.split('/')[-1], d['value'], d['ms_per_step'])"; done
