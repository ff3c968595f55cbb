"""RCCL as far as ONE GPU allows (round-2 verdict item 4): a world-size-1 "nccl" process group.

What this exercises that no gloo / shared-GPU test does: the RCCL communicator bootstrap exactly as
``DistributedManager._setup_distributed_backend`` does it for the "nccl" backend (``device_id=``, the extra gloo
control group, ``_validate_setup`` on a device tensor), ``dist.all_reduce`` / ``all_gather_into_tensor`` as RCCL
kernels on the device, and — the question the gathered-logits graph family depends on — whether those collectives can
be CAPTURED into a hipGraph next to this library's kernels and replayed (reference call sites:
minivllm/engine/distributed_manager.py:88-141, models/layers/linear.py:314-319, layers/embed_head.py:219-229).
Runs in a spawned process: the process group is process-global state."""
import json
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, port, out_path):
    import sys
    from types import SimpleNamespace

    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch.distributed as dist

    from minivllm_amd import hip_ops
    from minivllm_amd.engine.distributed_manager import DistributedManager

    res = {}
    torch.set_grad_enabled(False)
    dm = DistributedManager(SimpleNamespace(tensor_parallel_size=1), 0)
    dm.world_size = 1
    os.environ["MINIVLLM_TP_BACKEND"] = "nccl"
    dm._setup_distributed_backend()                 # init_process_group("nccl", device_id=cuda:0) + gloo control group
    dm._validate_setup()                            # all-reduce of a device tensor through RCCL
    res["backend"] = dist.get_backend()
    res["control_group_backend"] = dist.get_backend(dm.control_group)
    box = [("generate", [1, 2, 3])]
    dist.broadcast_object_list(box, src=0, group=dm.control_group)
    res["control_broadcast_ok"] = box[0] == ("generate", [1, 2, 3])

    dev = torch.device("cuda", 0)
    g = torch.Generator(device="cpu").manual_seed(0)
    M, K, N, V = 256, 2048, 2048, 151936 // 8
    x = (torch.randn(M, K, generator=g) * 0.5).bfloat16().to(dev)
    w = (torch.randn(N, K, generator=g) * 0.02).bfloat16().to(dev)
    wp = hip_ops.pack_weight(w)
    res0 = (torch.randn(M, N, generator=g) * 0.5).bfloat16().to(dev)
    nw = torch.ones(N, dtype=torch.bfloat16, device=dev)
    logits_local = torch.randn(M, V, generator=g).bfloat16().to(dev)
    gathered = torch.empty(1 * M, V, dtype=torch.bfloat16, device=dev)

    def step(residual):
        y = hip_ops.linear_mid_m(x, wp, N)          # row-parallel projection (this library's kernel)
        dist.all_reduce(y)                          # C1: the RCCL all-reduce behind it (linear.py:314-319)
        normed, new_res = hip_ops.add_rmsnorm(y, residual, nw, 1e-6, inplace=True)
        dist.all_gather_into_tensor(gathered, logits_local)   # C3 on the logits shape (embed_head.py:219-229)
        return normed, new_res

    # eager reference: three steps from the same starting residual
    r = res0.clone()
    eager = []
    for _ in range(3):
        n_, r = step(r)
        eager.append((n_.clone(), r.clone()))
    torch.cuda.synchronize()
    res["eager_all_gather_ok"] = bool(torch.equal(gathered, logits_local))

    captured = False
    err = ""
    try:
        static_res = res0.clone()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):               # warm-up on a side stream, as capture wants
            step(static_res.clone())
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        gathered.zero_()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            out_n, out_r = step(static_res)         # residual updated in place: replay i continues from replay i-1
        captured = True
        ok = True
        for i in range(3):
            graph.replay()
            torch.cuda.synchronize()
            ok &= bool(torch.equal(out_n, eager[i][0]) and torch.equal(out_r, eager[i][1]))
        res["graph_replays_equal_eager"] = ok
        res["graph_all_gather_ok"] = bool(torch.equal(gathered, logits_local))
    except Exception as e:  # recorded, not raised: whether RCCL captures is the finding
        err = f"{type(e).__name__}: {e}"
    res["rccl_in_graph_capture"] = captured
    res["capture_error"] = err[:500]

    # the gather-free nucleus sequence (parallel_state.tp_nucleus_tokens): its MAX all-reduce of fp32 and its two SUM
    # all-reduces of int64 histograms are RCCL calls of kinds the step above does not contain -- eager, then captured
    from minivllm_amd import parallel_state

    parallel_state.init_tensor_parallel(0, 1)
    B = 64
    ll = logits_local[:B].contiguous()
    temps = torch.full((B,), 0.8, device=dev)
    temps[::7] = 0.0
    top_ps = torch.full((B,), 0.9, device=dev)
    min_ps = torch.zeros(B, device=dev)
    min_ps[1::5] = 0.05
    rng = torch.zeros(1, dtype=torch.int64, device=dev)

    def nuc_step():
        return parallel_state.tp_nucleus_tokens(
            ll, lambda m: hip_ops.sample(m, temps, None, None, None, seed=3, offset=0, offset_dev=rng, return_stats=True),
            temps, top_ps, min_ps, 3, 0, rng)

    try:
        want = []
        for i in range(3):
            rng.fill_(i + 1)
            want.append(nuc_step().clone())
        torch.cuda.synchronize()
        _, mask = hip_ops.sample(ll, temps, torch.full((B,), -1, dtype=torch.int64, device=dev), top_ps, min_ps,
                                 seed=1, offset=1, return_mask=True)
        res["nucleus_tokens_inside_the_kept_set"] = bool(all(
            mask[torch.arange(B, device=dev), w].bool().all() for w in want))
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            nuc_step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        graph2 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph2):
            toks = nuc_step()
        ok = True
        for i in range(3):
            rng.fill_(i + 1)
            graph2.replay()
            torch.cuda.synchronize()
            ok &= bool(torch.equal(toks, want[i]))
        res["nucleus_graph_replays_equal_eager"] = ok
        res["nucleus_error"] = ""
    except Exception as e:
        res["nucleus_graph_replays_equal_eager"] = False
        res["nucleus_error"] = f"{type(e).__name__}: {e}"[:500]
    parallel_state.reset_tensor_parallel()
    with open(out_path, "w") as f:
        json.dump(res, f)
    try:
        dist.barrier()
        dist.destroy_process_group()
    except Exception:
        pass


@pytest.mark.gpu
def test_world1_nccl_bootstrap_and_collectives_in_a_graph(tmp_path):
    out = str(tmp_path / "rccl.json")
    mp.spawn(_worker, args=(_free_port(), out), nprocs=1, join=True)
    with open(out) as f:
        res = json.load(f)
    print("RCCL world-1 findings:", json.dumps(res))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "rccl_world1.json"), "w") as f:
        json.dump(res, f, indent=1)
    assert res["backend"] == "nccl" and res["control_group_backend"] == "gloo" and res["control_broadcast_ok"]
    assert res["eager_all_gather_ok"]
    # the finding DESIGN.md §5 records; the gathered-logits graph family relies on it
    assert res["rccl_in_graph_capture"], res["capture_error"]
    assert res["graph_replays_equal_eager"] and res["graph_all_gather_ok"]
    # the nucleus-floor family (MAX of fp32, SUM of int64 through RCCL) captures and replays too
    assert res["nucleus_error"] == "" and res["nucleus_tokens_inside_the_kept_set"]
    assert res["nucleus_graph_replays_equal_eager"]
