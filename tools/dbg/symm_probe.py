"""Does torch's symmetric-memory all-reduce run on this ROCm build?  2 ranks on ONE GPU (functional probe)."""
import os, sys, time
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def worker(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import torch.distributed._symmetric_memory as symm_mem
    try:
        group = dist.group.WORLD
        try:
            symm_mem.enable_symm_mem_for_group(group.group_name)
        except Exception as e:
            print(rank, "enable_symm_mem_for_group:", repr(e), flush=True)
        buf = symm_mem.empty(1 << 20, dtype=torch.bfloat16, device="cuda:0")
        hdl = symm_mem.rendezvous(buf, group.group_name)
        print(rank, "rendezvous ok", type(hdl), flush=True)
        buf.fill_(rank + 1)
        torch.cuda.synchronize(); dist.barrier()
        for name in ("one_shot_all_reduce", "two_shot_all_reduce_"):
            try:
                buf.fill_(rank + 1)
                torch.cuda.synchronize(); dist.barrier()
                out = getattr(torch.ops.symm_mem, name)(buf, "sum", group.group_name)
                torch.cuda.synchronize()
                print(rank, name, "->", out[:4].float().tolist(), flush=True)
                dist.barrier()
            except Exception as e:
                print(rank, name, "FAILED:", repr(e)[:300], flush=True)
    except Exception as e:
        print(rank, "FAILED:", repr(e)[:500], flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    mp.spawn(worker, args=(2, 29533), nprocs=2, join=True)
