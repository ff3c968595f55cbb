import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/tools")
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench_kernels as bk
for _ in range(2):
    bk.bench_decode(256, 16, 8, 128, 1024)
for rep in range(2):
    for fl in (0, 32 << 8, 64 << 8, 16 << 8):
        bk.bench_decode(256, 2, 1, 128, 1024, layers=8, flags=fl)     # Qwen3-1.7B at TP=8: vector kernel, one kv head
        bk.bench_decode(256, 4, 2, 128, 1024, layers=8, flags=fl)     # TP=4
        bk.bench_decode(256, 8, 1, 128, 1024, layers=8, flags=fl)     # Qwen2-72B at TP=8: matrix-core kernel
