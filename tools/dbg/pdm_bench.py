import sys, os
sys.path.insert(0, "/root/repo/tools"); sys.path.insert(0, "/root/repo")
import bench_kernels as bk
bk.bench_decode(256, 28, 4, 128, 1024)
bk.bench_decode(256, 64, 8, 128, 1024)
bk.bench_decode(256, 8, 1, 128, 1024)
bk.bench_decode(256, 28, 4, 128, 4096, layers=2)
bk.bench_decode(32, 28, 4, 128, 1024)
