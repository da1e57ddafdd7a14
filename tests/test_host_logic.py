"""CPU tests of the host-side logic that needs no device: config parsing, device map, auto-split, synthetic checkpoint
layout, safetensors round trip."""
import json
import os

import numpy as np
import pytest
import torch

from exllama_amd import synth
from exllama_amd.model import ExLlama, ExLlamaConfig, ExLlamaDeviceMap, _layer_dtype_size, _skip_key


def test_config_from_json_and_dict(tmp_path):
    d = synth.config_dict(synth.LLAMA_7B)
    p = tmp_path / "config.json"
    p.write_text(json.dumps(d))
    for src in (str(p), d):
        c = ExLlamaConfig(src)
        assert (c.hidden_size, c.intermediate_size, c.num_hidden_layers, c.num_attention_heads) == (4096, 11008, 32, 32)
        assert c.head_dim == 128 and c.num_key_value_groups == 1 and c.rotary_embedding_base == 10000.0
        # reference defaults (model.py:83-103)
        assert (c.max_seq_len, c.max_input_len, c.matmul_recons_thd, c.fused_mlp_thd, c.sdp_thd, c.fused_attn) == (2048, 2048, 8, 2, 8, True)
    c = ExLlamaConfig({**d, "num_key_value_heads": 8, "rope_theta": 1e6})
    assert c.num_key_value_groups == 4 and c.rotary_embedding_base == 1e6
    c.alpha_value = 2.0
    c.calculate_rotary_embedding_base()
    assert abs(c.rotary_embedding_base - 1e6 * 2.0 ** (128 / 126)) < 1e-3
    c.set_auto_map("17.2,24")
    assert c.auto_map == [17.2, 24.0]
    c.set_auto_map(None)
    assert c.auto_map is None


def test_device_map():
    m = ExLlamaDeviceMap(4)
    m.layers = ["cuda:0", "cuda:0", "cuda:1", "cuda:1"]
    m.norm = m.lm_head = "cuda:1"
    assert m.map("model.layers.2.mlp.up_proj.qweight") == "cuda:1"
    assert m.map("lm_head.weight") == "cuda:1" and m.map("model.embed_tokens.weight") == "cuda:0"
    assert m.get_layers_devs() == ["cuda:0", "cuda:1"]
    with pytest.raises(ValueError):
        m.map("something.else")


def test_auto_split_greedy_fill():
    """Greedy layer placement under per-device GB budgets (reference: model.py:770-801)."""
    cfg = ExLlamaConfig(synth.config_dict(synth.LLAMA_7B))
    m = ExLlama.__new__(ExLlama)
    m.config = cfg
    layer = 105_000_000
    cfg.auto_map = [1.0, 1.0, 10.0]
    m._auto_split({"decoder": layer, "norm": 8192, "head": 262_144_000})
    per_dev = int(1024 ** 3 // layer)
    assert cfg.device_map.layers[:per_dev] == ["cuda:0"] * per_dev
    assert cfg.device_map.layers[per_dev:2 * per_dev] == ["cuda:1"] * per_dev
    assert set(cfg.device_map.layers[2 * per_dev:]) == {"cuda:2"}
    assert cfg.device_map.norm == "cuda:2" and cfg.device_map.lm_head == "cuda:2"
    cfg.auto_map = [0.5]
    with pytest.raises(ValueError, match="too large"):
        m._auto_split({"decoder": layer, "norm": 8192, "head": 262_144_000})


def test_key_filters():
    assert _skip_key("model.layers.0.self_attn.q_proj.bias") and _skip_key("model.layers.0.self_attn.rotary_emb.inv_freq")
    assert not _skip_key("model.layers.0.self_attn.q_proj.qweight")
    assert [_layer_dtype_size("a" + s) for s in (".weight", ".qweight", ".qzeros", ".scales", ".g_idx")] == [2, 4, 4, 2, 0]


def test_synthetic_checkpoint_layout(tmp_path):
    dims = synth.LLAMA_TINY_GQA
    t = synth.make_checkpoint(dims, groupsize=128, act_order=True, seed=3)
    h, I, kvd = dims.hidden_size, dims.intermediate_size, dims.num_key_value_heads * dims.head_dim
    assert t["model.layers.1.self_attn.k_proj.qweight"].shape == (h // 8, kvd)
    assert t["model.layers.1.mlp.down_proj.qweight"].shape == (I // 8, h)
    assert t["model.layers.0.mlp.up_proj.qzeros"].shape == (h // 128, I // 8)
    assert t["model.layers.0.mlp.up_proj.scales"].dtype == torch.float16
    g = t["model.layers.0.mlp.down_proj.g_idx"]
    assert g.dtype == torch.int32 and np.array_equal(np.bincount(g.numpy()), np.full(I // 128, 128))
    # seeded: same seed -> same bits
    t2 = synth.make_checkpoint(dims, groupsize=128, act_order=True, seed=3)
    assert all(torch.equal(t[k], t2[k]) for k in t)
    cfg_path, st_path = synth.save_checkpoint(str(tmp_path), dims, groupsize=128, act_order=True, seed=3)
    from safetensors import safe_open
    with safe_open(st_path, framework="pt", device="cpu") as f:
        assert set(f.keys()) == set(t.keys())
        assert torch.equal(f.get_tensor("model.layers.0.self_attn.q_proj.qweight"), t["model.layers.0.self_attn.q_proj.qweight"])
    assert ExLlamaConfig(cfg_path).num_key_value_heads == dims.num_key_value_heads
