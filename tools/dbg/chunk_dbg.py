import os, sys, json
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import pathlib, tempfile
import test_gpu_e2e as T
from minivllm_amd.engine.scheduler import Scheduler
from minivllm_amd.engine.sequence import Sequence
from minivllm_amd.sampling_params import SamplingParams
z = T._load(os.path.join(ROOT, "tests", "golden"), "qwen3")
print(bytes(z["config_json"]).decode())
for dtype in (torch.float32, torch.bfloat16):
    for budget in (1024, 64):
        tmp = pathlib.Path(tempfile.mkdtemp())
        cfg, _, ex = T._build(tmp, z, dtype)
        cfg.max_num_batched_tokens, cfg.enable_chunked_prefill = budget, True
        sched = Scheduler(cfg)
        sp = SamplingParams(temperature=0.0, max_tokens=64, ignore_eos=True)
        seqs = [Sequence(p, sp, block_size=64) for p in T._prompts(z)]
        for s in seqs: sched.add(s)
        forced = z["forced_tokens"]; rows = {}
        while sched.waiting or sched.prefilling is not None:
            batch, is_prefill = sched.schedule()
            logits, _ = ex.execute_batch(batch, prefill=True)
            logits = logits.float().cpu()
            toks = []
            for i, s in enumerate(batch):
                if s.prefill_end is None: rows[seqs.index(s)] = logits[i]
                toks.append(int(forced[0][seqs.index(s)]))
            sched.postprocess(batch, toks)
        got = torch.stack([rows[i] for i in range(3)])
        want = torch.from_numpy(z["f32_prefill_logits"])
        print(dtype, "budget", budget, "per-prompt max err", (got - want).abs().max(dim=1).values.tolist(), "scale", want.abs().max().item())
