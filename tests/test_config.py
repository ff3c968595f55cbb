"""Config validation mirrors minivllm/config.py:130-264 (same fields, defaults and error cases)."""
import json

import pytest

from minivllm_amd import Config


@pytest.fixture
def model_dir(tmp_path):
    d = tmp_path / "m"
    d.mkdir()
    (d / "config.json").write_text(json.dumps({"model_type": "llama", "hidden_size": 768, "num_hidden_layers": 12,
                                               "num_attention_heads": 12, "max_position_embeddings": 4096}))
    return str(d)


def test_defaults(model_dir):
    c = Config(model_dir)
    assert (c.max_num_batched_tokens, c.max_num_seqs, c.max_model_len) == (16384, 512, 4096)
    assert (c.device_memory_utilization, c.tensor_parallel_size, c.enforce_eager) == (0.9, 1, False)
    assert (c.kvcache_block_size, c.num_kvcache_blocks, c.eos, c.dtype, c.seed) == (64, -1, -1, "auto", None)
    assert c.hf_config.hidden_size == 768
    c.gpu_memory_utilization = 0.5
    assert c.device_memory_utilization == 0.5 and "max_num_seqs=512" in repr(c)


@pytest.mark.parametrize("kw", [
    dict(kvcache_block_size=48), dict(tensor_parallel_size=0), dict(tensor_parallel_size=9),
    dict(device_memory_utilization=0.05), dict(device_memory_utilization=1.5), dict(dtype="int8"),
    dict(eos=-2), dict(num_kvcache_blocks=0), dict(num_kvcache_blocks=-5), dict(max_num_seqs=0),
    dict(max_model_len=0), dict(max_num_batched_tokens=0),
    dict(max_num_batched_tokens=1024, max_model_len=2048),     # must fit a full context in one batch
    dict(load_format="gguf"),
])
def test_rejects(model_dir, kw):
    with pytest.raises(ValueError):
        Config(model_dir, **kw)


def test_bad_path_and_len_clamp(tmp_path, model_dir):
    with pytest.raises(ValueError):
        Config(str(tmp_path / "missing"))
    short = tmp_path / "s"
    short.mkdir()
    (short / "config.json").write_text(json.dumps({"model_type": "llama", "max_position_embeddings": 128}))
    with pytest.warns(UserWarning):
        c = Config(str(short), max_model_len=4096, max_num_batched_tokens=2048)
    assert c.max_model_len == 128
    assert Config(model_dir, kvcache_block_size=256, tensor_parallel_size=8).kvcache_block_size == 256
