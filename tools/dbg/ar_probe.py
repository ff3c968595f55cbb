"""Functional probe of the peer-to-peer all-reduce: W processes on ONE GPU (IPC within the device)."""
import os, sys, time
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def worker(rank, world, port):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import logging
    logging.basicConfig(level=logging.INFO)
    from minivllm_amd.custom_allreduce import maybe_create
    t0 = time.time()
    ar = maybe_create(rank, world, torch.device("cuda", 0))
    print(rank, "setup", "OK" if ar is not None else "DISABLED", f"{time.time() - t0:.2f}s", flush=True)
    if ar is not None:
        torch.manual_seed(rank)
        x = torch.randn(256, 2048, device="cuda").bfloat16()
        ref = x.float().clone()
        dist.all_reduce(ref)
        y = ar.all_reduce(x.clone())
        torch.cuda.synchronize()
        print(rank, "random data max err vs fp32 sum:", (y.float() - ref).abs().max().item(), flush=True)
        # timing (all ranks on one GPU: functional only)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        dist.barrier(); e0.record()
        for _ in range(50):
            ar.all_reduce(x)
        e1.record(); torch.cuda.synchronize()
        print(rank, "us per call (shared GPU):", e0.elapsed_time(e1) * 1e3 / 50, "status", ar.status(), flush=True)
        ar.close()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    w = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    mp.spawn(worker, args=(w, 29544), nprocs=w, join=True)
