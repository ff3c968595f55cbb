"""bench.py host logic that needs no GPU: the N > 1 self-launch command, the usable-core count."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402


def test_multi_gpu_bench_launches_itself_under_torch_distributed_run():
    """`python3 bench.py --gpus N` (the driver's command) must become `python -m torch.distributed.run
    --nproc-per-node N ... bench.py --gpus N ...` with the same arguments, rendezvous on 127.0.0.1."""
    args = bench.parse(["--gpus", "4", "--steps", "7", "--warmup", "3", "--auto-kv", "--no-cpu-baseline"])
    cmd = bench.launch_command(args, port=29555)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29555"
    script = cmd.index(os.path.join(ROOT, "bench.py"))
    tail = cmd[script + 1:]
    again = bench.parse(tail)                                  # the child parses to the same request
    assert (again.gpus, again.steps, again.warmup, again.auto_kv, again.no_cpu_baseline) == (4, 7, 3, True, True)
    assert again.model == args.model and again.batch == args.batch and again.ctx == args.ctx
    # a free port is chosen when none is given
    port = int(bench.launch_command(args)[bench.launch_command(args).index("--master-port") + 1])
    assert 1024 < port < 65536


def test_usable_cores_is_positive_and_bounded():
    n = bench.usable_cores()
    assert 1 <= n <= (os.cpu_count() or 1)
