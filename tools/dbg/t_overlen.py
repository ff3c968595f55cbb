import sys, json, os, tempfile
sys.path.insert(0, "/root/repo")
import torch
from minivllm_amd import LLM, Config, SamplingParams
import bench
d = tempfile.mkdtemp()
cfgm = dict(bench.MODELS["qwen3-0.6b"]); cfgm.update(num_hidden_layers=2, hidden_size=256, intermediate_size=512, num_attention_heads=4, num_key_value_heads=2, vocab_size=1024)
json.dump(cfgm, open(os.path.join(d, "config.json"), "w"))
for eager in (True, False):
    llm = LLM(Config(d, dtype="bfloat16", max_model_len=128, max_num_batched_tokens=256, max_num_seqs=4, num_kvcache_blocks=16, load_format="dummy", enforce_eager=eager))
    try:
        outs = llm.generate([[5] * 120, [7] * 10], SamplingParams(temperature=0.0, max_tokens=40, ignore_eos=True), use_tqdm=False)
        print("eager" if eager else "graph", [len(o["token_ids"]) for o in outs])
    except Exception as e:
        print("eager" if eager else "graph", "EXC", type(e).__name__, str(e)[:200])
    llm.exit()
