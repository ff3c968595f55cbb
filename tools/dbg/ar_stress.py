"""Stress the peer-to-peer all-reduce validation: W ranks on one GPU, N rounds of the full validation."""
import logging, os, sys, time
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def worker(rank, world, port, rounds):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    logging.basicConfig(level=logging.WARNING)
    from minivllm_amd.custom_allreduce import maybe_create
    ar = maybe_create(rank, world, torch.device("cuda", 0))
    print(rank, "setup", ar is not None, flush=True)
    bad = 0
    if ar is not None:
        t0 = time.time()
        for i in range(rounds):
            ok = ar._validate()
            bad += not ok
            if not ok:
                print(rank, "round", i, "FAILED status", ar.status(), flush=True)
        print(rank, f"{rounds} rounds, {bad} failed, {time.time() - t0:.1f}s, calls {ar.calls}", flush=True)
        ar.close()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    w, r = int(sys.argv[1]), int(sys.argv[2])
    mp.spawn(worker, args=(w, 29577, r), nprocs=w, join=True)
