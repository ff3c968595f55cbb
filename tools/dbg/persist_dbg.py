"""Debug driver of the persistent decode launch (csrc/decode_persistent.hip) at the bench's B = 1 / 8 cells: graph steps, status
words, step times, and the phase stamps of layer 1 (flags bit 8) with two weight chunks per wave in flight and with none —
the table of profiles/r04_persistent_decode.md."""
import json, os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench

def run(eager, batch, layers=None):
    args = bench.parse(["--batch", str(batch), "--steps", "16", "--warmup", "4", "--no-cpu-baseline", "--no-secondary"] + (["--eager"] if eager else []))
    cfgm = dict(bench.MODELS["qwen3-1.7b"])
    if layers: cfgm["num_hidden_layers"] = layers
    bench.MODELS["dbg"] = cfgm
    args.model = "dbg"
    with tempfile.TemporaryDirectory() as wd:
        eng = bench.build_engine(args, wd, "dbg")
        bench.fill_prompts(eng, args, "dbg")
        inner = eng.model_runner.inference_executor.model.model
        times = []
        for i in range(12):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            try:
                eng.step()
            except RuntimeError as e:
                print("  step", i, "RAISED:", str(e)[:200])
                torch.cuda.synchronize()
                st = inner._persist
                print("   device status:", [hex(int(v) & 0xffffffff) for v in st["status"].cpu()], "host:", [hex(int(v) & 0xffffffff) for v in st["status_host"]])
                for key, ws in st["ws"].items():
                    bar = ws[:3584].view(torch.int32).cpu()
                    names = dict(census=(0, 256), flat=(256, 288), xcd=(288, 544), top=(544, 576), gen=(576, 832), giveup=(832, 864))
                    print("   ws", key, {k: [hex(int(v) & 0xffffffff) for v in bar[a:b][::32]] for k, (a, b) in names.items()})
                break
            torch.cuda.synchronize(); times.append((time.perf_counter() - t0) * 1e3)
        global LAST
        LAST = [round(v, 2) for v in inner.persistent_stamps()]
        print(f"eager={eager} B={batch} layers={cfgm['num_hidden_layers']}: status {inner.persistent_status()} ms/step {[round(t,3) for t in times]}")
        eng.exit()

NAMES = ["bar A", "norm1", "qkv", "bar B", "attention", "bar C", "combine", "bar D", "o_proj", "bar E", "norm2", "gate_up", "bar F", "down"]
from minivllm_amd.glue import qwen
qwen.PERSISTENT = True          # (off by default: MV_PERSISTENT=1)
for pre in (2, 0xff):
    qwen.PERSISTENT_FLAGS = 0x100 | pre
    for b in (1, 8):
        run(False, b, 28)
        print("   prefetch chunks", pre, "B", b, "us per phase (workgroup 1, layer 1):", dict(zip(NAMES, LAST)), "sum", round(sum(LAST), 2))
