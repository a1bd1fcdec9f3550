#!/bin/bash
# The round's bench lines (GPU box): the default line (fp32 metric + tracker A/B + bf16 configs[2] leg + cpu_baseline), the other
# BASELINE configurations, A/B switches.  usage: bash tools/bench_round.sh r04
R=$GRAFT_REPO_ROOT; TAG=${1:-r04}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
python bench.py --steps 20 --warmup 5 > $O/bench_f32.json 2> $O/bench_f32.err
VAEHIP_NO_WINO4=1 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra-legs > $O/bench_f32_no_wino4.json 2>> $O/bench_f32.err
python bench.py --dtype bf16 --steps 10 --warmup 3 --no-cpu-baseline --no-extra-legs > $O/bench_bf16.json 2> $O/bench_bf16.err
python bench.py --dtype bf16 --batch 32 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_bf16_b32.json 2>> $O/bench_bf16.err
VAEHIP_NO_THIN_MFMA=1 python bench.py --dtype bf16 --batch 32 --steps 10 --warmup 3 --no-cpu-baseline --no-extra-legs --no-profile > $O/bench_bf16_b32_no_thin.json 2>> $O/bench_bf16.err
python bench.py --dtype bf16 --res 512 --batch 8 --nudge-interval 100 --steps 10 --warmup 3 --no-cpu-baseline --no-extra-legs > $O/bench_bf16_512.json 2>> $O/bench_bf16.err
python bench.py --dtype bf16 --res 1024 --batch 2 --checkpoint-decoder --steps 4 --warmup 2 --no-cpu-baseline --no-extra-legs > $O/bench_bf16_1024.json 2>> $O/bench_bf16.err
python bench.py --res 512 --batch 8 --steps 3 --warmup 1 --no-cpu-baseline --no-extra-legs > $O/bench_f32_512.json 2>> $O/bench_f32.err
for f in $O/bench_*.json; do python -c "
import json,sys;d=json.loads(open('$f').read().strip().splitlines()[-1]);print('$f'.split('/')[-1], d['value'], d['ms_per_step'], d.get('ms_per_step_median'))"; done
