"""`bench.py --gpus N` as the driver may call it WITHOUT a launcher: the script starts its own ranks through
torch.distributed.run and rank 0 prints one JSON line.  Here (no GPU) the ranks only form the process group on gloo and
all-reduce one number (--rendezvous-only); the same launcher code starts the RCCL ranks on a GPU node."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_self_launches_its_ranks():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--dist-backend", "gloo",
                        "--rendezvous-only"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout  # only rank 0 prints
    out = json.loads(lines[0])
    assert out == {"rendezvous": "ok", "world": 2, "backend": "gloo", "allreduce_sum": 3.0}


def test_a_failing_child_rank_fails_the_bench():
    """rank 1 exits non-zero after the collective: the self-launching parent must pass a non-zero status on (the driver reads
    it), and a rank other than 0 never prints a result line"""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dist-backend", "gloo", "--rendezvous-only",
                        "--rendezvous-fail-rank", "1"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode != 0, r.stdout[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) <= 1 and all(json.loads(ln).get("rendezvous") == "ok" for ln in lines), r.stdout  # rank 0's line at most
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dist-backend", "gloo", "--rendezvous-only",
                        "--rendezvous-fail-rank", "0"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode != 0 and not [ln for ln in r.stdout.splitlines() if ln.startswith("{")], r.stdout  # rank 1 prints nothing


def test_bench_rejects_a_world_that_does_not_match():
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rendezvous-only"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stderr + r.stdout)


def test_workload_labels_name_the_right_baseline_config():
    sys.path.insert(0, ROOT)
    import bench
    assert bench.workload_label("f32", 256, 16, False, 0, True).startswith("BASELINE configs[1]")
    assert bench.workload_label("bf16", 256, 32, False, 0, True).startswith("BASELINE configs[2]")
    assert bench.workload_label("bf16", 512, 8, False, 100, True).startswith("BASELINE configs[3]")
    assert "nudge every 100" in bench.workload_label("bf16", 512, 8, False, 100, True)
    assert bench.workload_label("bf16", 1024, 2, True, 0, True).startswith("BASELINE configs[4]")
    assert bench.workload_label("f32", 512, 8, False, 0, True).startswith("off-baseline")
