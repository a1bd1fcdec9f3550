"""Cost of train.py's control-plane agreement (all_ranks_ok: one int32 MIN all-reduce on a gloo group per batch), measured on
the host: python tools/ctrl_allreduce_cost.py [world]  (spawns `world` processes on 127.0.0.1, CPU only)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vae-channel-dynamics_amd", "src"))


def worker(rank, world, port):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import train
    for _ in range(20):
        train.all_ranks_ok(True, world)
    dist.barrier()
    t0 = time.perf_counter()
    n = 500
    for _ in range(n):
        train.all_ranks_ok(True, world)
    dt = (time.perf_counter() - t0) / n
    if rank == 0:
        print(f"world {world}: {dt * 1e6:.0f} us per all_ranks_ok call = {100 * dt / 0.189:.3f} % of a 189 ms step")
    dist.destroy_process_group()


if __name__ == "__main__":
    import torch.multiprocessing as mp
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    mp.spawn(worker, args=(world, 29731), nprocs=world)
