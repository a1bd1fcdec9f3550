"""print the per-kernel table of a bench.py JSON line: python tools/show_bench.py <file> [top_n]"""
import json
import sys

lines = [l for l in open(sys.argv[1]) if l.startswith("{")]
d = json.loads(lines[-1])
top = int(sys.argv[2]) if len(sys.argv) > 2 else 14
print(f"{d['value']} {d['unit']}  {d['ms_per_step']} ms/step  contractions {d['roofline']['contraction_ms_per_step']} ms "
      f"at {d['roofline']['all_contraction_kernels_tflops']} TFLOP/s")
for k, v in sorted(d["kernels"].items(), key=lambda kv: -kv[1]["ms_per_step"])[:top]:
    print(f"{k:58s} {v['launches_per_step']:6.1f} {v['ms_per_step']:8.2f} ms {v['tflops']:7.1f} TF")
