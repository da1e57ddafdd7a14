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


# ---- model_init: the reference's command-line surface -> ExLlamaConfig (model_init.py) ---------------------------------
def _parse(argv):
    import argparse
    from exllama_amd import model_init, perplexity
    p = argparse.ArgumentParser()
    model_init.add_args(p)
    perplexity.add_args(p)
    a = p.parse_args(argv)
    model_init.post_parse(a)
    perplexity.post_parse(a)
    return a


def test_model_init_flags_to_config(tmp_path):
    from exllama_amd import model_init
    dims = synth.PRESETS["tiny"]
    cfg_path = tmp_path / "config.json"
    cfg_path.write_text(json.dumps(synth.config_dict(dims)))
    a = _parse(["-t", "tok.model", "-c", str(cfg_path), "-m", "w.safetensors", "-l", "4096", "-gs", "20,7.5", "-a", "2.0",
                "-mmrt", "16", "-fmt", "0", "-nfa", "-mmfr", "-flash", "1024"])
    model_init.get_model_files(a)                       # -t/-c/-m given: nothing to discover
    c = model_init.make_config(a)
    assert (c.max_seq_len, c.matmul_recons_thd, c.fused_mlp_thd, c.fused_attn, c.matmul_fused_remap) == (4096, 16, 0, False, True)
    assert c.auto_map == [20.0, 7.5] and c.model_path == "w.safetensors"
    assert c.use_flash_attn_2 and c.max_input_len == 1024
    hd = dims.hidden_size // dims.num_attention_heads
    assert c.rotary_embedding_base == pytest.approx(10000.0 * 2.0 ** (hd / (hd - 2)))       # NTK alpha (model.py:122-123)
    assert c.rmsnorm_no_half2 and c.silu_no_half2      # ROCm default of the reference: half2 paths off unless -fh2
    assert _parse(["-c", str(cfg_path), "-theta", "500000"]).theta == 500000.0
    model_init.print_options(a)
    # defaults are the reference's
    d = _parse([])
    assert (d.length, d.matmul_recons_thd, d.fused_mlp_thd, d.sdp_thd, d.compress_pos_emb, d.alpha) == (2048, 8, 2, 8, 1.0, 1.0)


def test_model_init_directory_discovery(tmp_path):
    from exllama_amd import model_init
    (tmp_path / "config.json").write_text("{}")
    for n in ("model-00001-of-00002.safetensors", "model-00002-of-00002.safetensors"):
        (tmp_path / n).write_bytes(b"")
    a = _parse(["-d", str(tmp_path)])
    model_init.get_model_files(a)
    assert a.config.endswith("config.json") and a.tokenizer.endswith("tokenizer.model") and len(a.model) == 2
    assert model_init._wildcard_name([os.path.basename(m) for m in a.model]) == "model-0000*-of-00002.safetensors"
    with pytest.raises(SystemExit):
        model_init.get_model_files(_parse(["-t", "x"]))          # neither -d nor all of -t -c -m
    empty = tmp_path / "empty"
    empty.mkdir()
    with pytest.raises(SystemExit):
        model_init.get_model_files(_parse(["-d", str(empty)]))


# ---- perplexity: chunking and the NLL arithmetic (perplexity.py) -------------------------------------------------------
class _FixedLogitsModel:
    """forward() returns the same logits row for every position: ppl is then known in closed form."""
    def __init__(self, row):
        self.row = torch.as_tensor(row, dtype=torch.float32)
        self.calls = []

    def forward(self, ids, cache, last_id_only=True, lora=None):
        self.calls.append((tuple(ids.shape), cache.current_seq_len))
        cache.current_seq_len += ids.shape[1]
        return self.row.expand(ids.shape[0], ids.shape[1], -1).clone()


class _Cache:
    current_seq_len = 0


def test_perplexity_chunking_and_value():
    from exllama_amd.perplexity import Perplexity
    V = 7
    m = _FixedLogitsModel(torch.zeros(V))
    p = Perplexity(model=m, cache=_Cache())
    p.add_tokens(torch.arange(25) % V, chunk_size=10, overlap=2)             # windows start every 8 tokens
    assert [c.shape[1] for c in p.dataset_chunks] == [10, 10, 9, 1]
    assert p.dataset_chunks[1][0, 0].item() == 8 % V
    assert p.test(quiet=True) == pytest.approx(V)                            # uniform prediction: ppl = vocabulary size
    assert [c for c, _ in m.calls] == [(1, 9), (1, 9), (1, 8)]               # 1-token tail chunk skipped; cache restarted per chunk
    assert all(pos == 0 for _, pos in m.calls)
    # token-by-token mode makes one call per input token and gives the same number
    m2 = _FixedLogitsModel(torch.tensor([2.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0]))
    p2 = Perplexity(model=m2, cache=_Cache())
    p2.add_tokens(torch.zeros(6, dtype=torch.long), chunk_size=6)
    whole = p2.test(quiet=True)
    assert whole == pytest.approx(1.0 / torch.softmax(m2.row, 0)[0].item())
    n_whole = len(m2.calls)
    assert p2.test(quiet=True, ppl_token=True) == pytest.approx(whole)
    assert len(m2.calls) - n_whole == 5
    # chunk_limit and truncation
    p3 = Perplexity(model=m, cache=_Cache())
    p3.add_tokens(torch.arange(40) % V, chunk_size=16, chunk_truncate=4, overlap=99)     # overlap clamped to chunk_size - 2
    assert all(c.shape[1] <= 4 for c in p3.dataset_chunks) and len(p3.dataset_chunks) == 20
    assert p3.test(chunk_limit=2, quiet=True) == pytest.approx(V)
    with pytest.raises(SystemExit):
        Perplexity(model=m, cache=_Cache()).test()


def test_perplexity_presets_and_json_loader(tmp_path):
    from exllama_amd.perplexity import Perplexity
    a = _parse(["-ppl", "gptq-for-llama"])
    assert (a.perplexity_dataset, a.perplexity_chunk_num, a.perplexity_chunk_min) == ("datasets/wikitext2.txt", 128, 0)
    assert _parse(["-ppl"]).perplexity_dataset == "datasets/wikitext2_val_sample.jsonl"
    assert _parse([]).perplexity is None

    class Tok:
        def encode(self, text):
            return torch.tensor([[ord(ch) % 11 for ch in text]])
    ds = tmp_path / "d.jsonl"
    ds.write_text("\n".join(json.dumps({"text": t}) for t in ["short", "a much longer record of text", "another long enough record"]))
    p = Perplexity(model=_FixedLogitsModel(torch.zeros(11)), cache=_Cache(), tokenizer=Tok())
    p.load(str(ds), chunk_size=12, chunk_truncate=8, minlength=10)
    assert [c.shape[1] for c in p.dataset_chunks] == [8, 8]
    raw = tmp_path / "d.txt"
    raw.write_text("x" * 30)
    p.load(str(raw), chunk_size=12)
    assert [c.shape[1] for c in p.dataset_chunks[2:]] == [12, 12, 6]
    with pytest.raises(ValueError):
        Perplexity(model=p.model, cache=_Cache()).load(str(raw), 8)


# ---- lora: PEFT adapter -> transposed, pre-scaled fp16 halves keyed by target layer (lora.py) ---------------------------
class _Lin:
    def __init__(self, i, o):
        self.in_features, self.out_features = i, o


class _Blk:
    pass


def _stub_model(layers=2, h=16, inter=24):
    class M:
        pass
    m = M()
    m.config = type("C", (), {"device_map": ExLlamaDeviceMap(layers)})()
    m.config.device_map.layers = ["cpu"] * layers
    m.layers = []
    for _ in range(layers):
        l = _Blk()
        l.self_attn, l.mlp = _Blk(), _Blk()
        for n in ("q_proj", "k_proj", "v_proj", "o_proj"):
            setattr(l.self_attn, n, _Lin(h, h))
        l.mlp.gate_proj, l.mlp.up_proj, l.mlp.down_proj = _Lin(h, inter), _Lin(h, inter), _Lin(inter, h)
        m.layers.append(l)
    return m


def test_lora_loader(tmp_path):
    from exllama_amd.lora import ExLlamaLora
    m = _stub_model()
    g = torch.Generator().manual_seed(0)
    r = 4
    pre = "base_model.model.model.layers."
    sd = {pre + "0.self_attn.q_proj.lora_A.weight": torch.randn(r, 16, generator=g),
          pre + "0.self_attn.q_proj.lora_B.weight": torch.randn(16, r, generator=g).to(torch.bfloat16),
          pre + "1.mlp.down_proj.lora_A.weight": torch.randn(r, 24, generator=g).half(),
          pre + "1.mlp.down_proj.lora_B.weight": torch.randn(16, r, generator=g),
          pre + "1.mlp.down_proj.bias": torch.zeros(16)}
    cfg = tmp_path / "adapter_config.json"
    cfg.write_text(json.dumps({"r": r, "lora_alpha": 8, "fan_in_fan_out": False}))
    lora = ExLlamaLora(m, str(cfg), "adapter.bin", tensors=sd)
    assert (lora.lora_r, lora.lora_alpha, lora.lora_scaling, lora.bias_ignored) == (4, 8.0, 2.0, True)
    assert sorted(lora.tensors) == ["model.layers.0.self_attn.q_proj.lora_A.weight", "model.layers.0.self_attn.q_proj.lora_B.weight",
                                    "model.layers.1.mlp.down_proj.lora_A.weight", "model.layers.1.mlp.down_proj.lora_B.weight"]
    a = lora.tensors["model.layers.0.self_attn.q_proj.lora_A.weight"]
    b = lora.tensors["model.layers.0.self_attn.q_proj.lora_B.weight"]
    assert a.shape == (16, r) and b.shape == (r, 16) and a.dtype == b.dtype == torch.float16 and a.is_contiguous()
    assert torch.equal(a, sd[pre + "0.self_attn.q_proj.lora_A.weight"].T.half())
    assert torch.equal(b, (sd[pre + "0.self_attn.q_proj.lora_B.weight"].T * 2.0).half())          # alpha / r folded into B
    # through a real file too
    from safetensors.torch import save_file
    st = tmp_path / "adapter_model.safetensors"
    save_file({k: v.contiguous() for k, v in sd.items()}, str(st))
    again = ExLlamaLora(m, str(cfg), str(st))
    assert all(torch.equal(again.tensors[k], lora.tensors[k]) for k in lora.tensors)
    # rejections (reference: lora.py:33-34, 51, 60-63, 94-95)
    bad = [({"r": 4, "lora_alpha": 4, "fan_in_fan_out": True}, sd),
           ({"r": 4, "lora_alpha": 4}, {"lm_head.lora_A.weight": torch.zeros(4, 16)}),
           ({"r": 4, "lora_alpha": 4}, {pre + "0.self_attn.q_proj.lora_A.weight": torch.zeros(4, 17)}),
           ({"r": 4, "lora_alpha": 4}, {pre + "0.mlp.q_proj.lora_A.weight": torch.zeros(4, 16)}),
           ({"r": 4, "lora_alpha": 4}, {pre + "0.self_attn.q_proj.bias": torch.ones(16)}),
           ({"r": 4, "lora_alpha": 4}, {pre + "0.self_attn.q_proj.lora_A.weight": torch.zeros(4, 16, dtype=torch.int32)})]
    for c, tensors in bad:
        with pytest.raises(ValueError):
            ExLlamaLora(m, c, "x.bin", tensors=tensors)


# ---- bench.py: the algorithmic byte / FLOP counts behind the roofline numbers equal SURVEY.md 8d --------------------------
def test_bench_algorithmic_counts_match_the_survey_table():
    import bench
    table = {  # model: (groupsize, decode GB at ctx 0, decode GB at ctx 2048, prefill TFLOP at S = 1920, at S = 2048)
        "7b": (128, 3.627, 4.701, 26.80, 28.73), "13b": (128, 6.920, 8.598, 51.74, 55.41),
        "33b": (32, 18.987, 22.26, 129.2, 138.2), "65b": (128, 34.172, 39.54, 258.3, 276.3)}
    for name, (g, b0, b2048, f1920, f2048) in table.items():
        d = synth.PRESETS[name]
        assert bench.decode_bytes_per_token(d, g, 0) / 1e9 == pytest.approx(b0, abs=6e-3)
        assert bench.decode_bytes_per_token(d, g, 2048) / 1e9 == pytest.approx(b2048, rel=1e-3)
        assert bench.prefill_flops(d, 1920) / 1e12 == pytest.approx(f1920, rel=1e-3)
        assert bench.prefill_flops(d, 2048) / 1e12 == pytest.approx(f2048, rel=1e-3)
    # per-matmul bytes: K N / 2 + (K / g) N 2.5 + 2 M (K + N); the 7B gate + up pair of the decode step
    per = 2 * bench.algorithmic_bytes_per_matmul(4096, 11008, 128, M=1)
    assert per == pytest.approx(46.9e6, rel=2e-3)
    r = bench.whole_job_rates(4, 2048, 128, prefill_ms=20.0, worst_ms=200.0, best_ms=100.0)
    assert r == {"prefill": pytest.approx(4 * 2048 / 0.02), "worst": pytest.approx(4 * 128 / 0.2), "best": pytest.approx(4 * 128 / 0.1)}


def test_config_and_device_map_equal_the_references_own_classes(tmp_path):
    """Pinned by the reference: tests/golden/host_ref.json was written by /root/reference/model.py's ExLlamaConfig /
    ExLlamaDeviceMap (oracle/make_host_golden.py, run where the reference lives): every scalar attribute of a config built from
    three config.json files (incl. GQA and a rope_theta), the NTK-scaled rotary base, the parsed auto map, and where the device
    map sends each kind of tensor key."""
    import json
    import os
    with open(os.path.join(os.path.dirname(__file__), "golden", "host_ref.json")) as f:
        gold = json.load(f)
    from oracle.make_host_golden import CONFIGS, KEYS
    for name, cfg in CONFIGS.items():
        p = tmp_path / f"{name}.json"
        p.write_text(json.dumps(cfg))
        c = ExLlamaConfig(str(p))
        want = gold["configs"][name]
        for k, v in want.items():
            if k.startswith("rotary_embedding_base_after") or k.startswith("auto_map_of"):
                continue
            assert hasattr(c, k), (name, k)
            assert getattr(c, k) == v, (name, k, getattr(c, k), v)
        c.alpha_value = 2.5
        c.calculate_rotary_embedding_base()
        assert c.rotary_embedding_base == want["rotary_embedding_base_after_alpha_2.5"]
        c.set_auto_map("10.5,24")
        assert c.auto_map == want["auto_map_of_10.5,24"]
    dm = gold["device_map"]
    m = ExLlamaDeviceMap(8)
    m.layers = list(dm["layers"])
    m.norm = m.lm_head = "cuda:1"
    # the ONE deliberate difference: the reference keeps the embedding table on the CPU (model.py:640), this repository on the first
    # GPU (the lookup is part of the decode graph); with the reference's placement the maps are identical
    assert dm["defaults"]["embed_tokens"] == "cpu" and ExLlamaDeviceMap(3).embed_tokens == "cuda:0"
    m.embed_tokens = "cpu"
    assert {k: m.map(k) for k in KEYS} == dm["map"]
    assert m.get_layers_devs() == dm["layers_devs"] and m.get_all_devs() == dm["all_devs"]
    fresh = ExLlamaDeviceMap(3)
    fresh.embed_tokens = "cpu"
    assert {"embed_tokens": fresh.embed_tokens, "lm_head": fresh.lm_head, "norm": fresh.norm, "layers": fresh.layers} == dm["defaults"]


def test_model_init_equals_the_references_own_model_init(tmp_path):
    """Pinned by the reference: tests/golden/init_ref.json holds the ExLlamaConfig that /root/reference/model_init.py's own
    add_args / post_parse / get_model_files / make_config build for 11 argument vectors (oracle/make_init_golden.py, run where
    the reference lives; on this ROCm torch post_parse switches the half2 paths off unless -fh2, model_init.py:43).  This
    repository's model_init must build the same config, attribute for attribute."""
    import argparse
    import os
    from exllama_amd import model_init
    from oracle.make_init_golden import make_dir
    with open(os.path.join(os.path.dirname(__file__), "golden", "init_ref.json")) as f:
        cases = json.load(f)
    d = str(tmp_path / "m")
    make_dir(d)
    assert len(cases) >= 10
    for case in cases:
        parser = argparse.ArgumentParser()
        model_init.add_args(parser)
        args = parser.parse_args([a.replace("{DIR}", d) for a in case["argv"]])
        model_init.post_parse(args)
        model_init.get_model_files(args)
        c = model_init.make_config(args)
        for k, want in case["config"].items():
            got = getattr(c, k)
            if isinstance(got, str):
                got = got.replace(d, "{DIR}")
            if isinstance(got, list):
                got = [x.replace(d, "{DIR}") if isinstance(x, str) else x for x in got]
            assert got == want, (case["argv"], k, got, want)


def test_lora_loader_equals_the_references_own_loader(tmp_path):
    """Pinned by the reference: tests/golden/lora_ref.json holds what /root/reference/lora.py's own ExLlamaLora.__init__ makes of a
    seeded synthetic adapter (oracle/make_lora_golden.py, run where the reference lives): keys, shapes, dtypes and SHA-256 of every
    tensor after transposition, alpha / r pre-scaling of B and fp32 / bf16 -> fp16 conversion, and the ignored zero bias."""
    import os
    from safetensors.torch import save_file
    from exllama_amd.lora import ExLlamaLora
    from oracle.make_lora_golden import ALPHA, H, INTER, LAYERS, R, adapter_state, describe
    with open(os.path.join(os.path.dirname(__file__), "golden", "lora_ref.json")) as f:
        gold = json.load(f)
    cfg = tmp_path / "adapter_config.json"
    cfg.write_text(json.dumps({"r": R, "lora_alpha": ALPHA, "fan_in_fan_out": False}))
    st = tmp_path / "adapter_model.safetensors"
    save_file({k: v.contiguous() for k, v in adapter_state().items()}, str(st))
    got = describe(ExLlamaLora(_stub_model(layers=LAYERS, h=H, inter=INTER), str(cfg), str(st)))
    assert len(gold["tensors"]) == 10
    assert got == gold


def test_perplexity_harness_equals_the_references_own_perplexity(tmp_path):
    """Pinned by the reference (BASELINE's `-ppl` leg): tests/golden/ppl_ref.json holds the chunking and the perplexity that
    /root/reference/perplexity.py's own Perplexity.load / .test produce for a raw text and a .jsonl dataset under six settings, with a
    deterministic position-dependent stand-in model and a byte tokenizer (oracle/make_ppl_golden.py, run where the reference lives).
    This repository's harness must cut the same chunks and print the same number, whole-chunk and token by token."""
    import os
    from exllama_amd.perplexity import Perplexity
    from oracle.make_ppl_golden import ByteTokenizer, StandInCache, StandInModel, write_datasets
    with open(os.path.join(os.path.dirname(__file__), "golden", "ppl_ref.json")) as f:
        gold = json.load(f)
    raw, js = write_datasets(str(tmp_path))
    assert len(gold) == 6
    for rec in gold:
        path = raw if rec["kind"] == "raw" else js
        for mode, key in ((False, "ppl_chunk"), (True, "ppl_token")):
            p = Perplexity("default", StandInModel(), StandInCache(), ByteTokenizer())
            p.load(path, **rec["args"])
            assert [[int(c.shape[1]), int(c[0, 0]), int(c[0, -1])] for c in p.dataset_chunks] == rec["chunks"], rec["args"]
            got = p.test(ppl_token=mode, quiet=True)
            assert round(got, 4) == pytest.approx(rec[key], abs=1.5e-4), (rec["args"], mode, got, rec[key])   # the reference prints 4 decimals


def test_bench_per_rank_report_single_process():
    """bench.py's multi-GPU modes print the communicator size and every rank's own timings next to the MAX over ranks; without a
    process group the report has one row and rccl_ranks = 0."""
    import bench
    info = bench.per_rank_report([1.5, 2.25], None, "cpu", ("a_ms", "b_ms"))
    assert info["rccl_ranks"] == 0 and info["backend"] is None
    assert info["per_rank"] == [{"rank": 0, "a_ms": 1.5, "b_ms": 2.25}]
    assert bench.reduce_over_ranks([3.0, 4.0], None, "cpu") == [3.0, 4.0]


# ---- bench.py: the sub-runs that put every quoted number into the one JSON line (other configs, drop-in, sharded modes) ------
def test_bench_sub_run_plumbing(tmp_path, monkeypatch):
    import json
    import sys
    import bench
    assert bench._last_json_line('noise\n{"a": 1}\n{"value": 2, "config": {}}\nRCCL banner') == {"value": 2, "config": {}}
    assert bench._last_json_line("nothing here") is None
    line = {"value": 402.5, "unit": "tokens/s", "prefill_tokens_per_s": 40125.1, "config": {"workload": "Llama-13B ...", "decode_mode": "hipGraph replay"},
            "path_roofline": {"decode_worst": {"frac_of_8TBps": 0.43}}, "roofline": {"kernel": "k", "frac": 0.5, "achieved": 4000.0, "classes": {}},
            "device_state": {"x": 1}}
    rec = bench._sub_record(line, 12.34, None, "BASELINE configs[2]")
    assert rec["value"] == 402.5 and rec["workload"] == "Llama-13B ..." and rec["measured"].startswith("this run")
    assert rec["roofline"] == {"kernel": "k", "achieved": 4000.0, "frac": 0.5} and "device_state" not in rec
    assert bench._sub_record(None, 900.0, "timed out after 900 s", "x") == {"what": "x", "error": "timed out after 900 s", "seconds": 900.0}
    # a sub-run that prints no JSON line, exits non-zero or outlives its limit is an error record, never an exception
    d, secs, err = bench._run_sub([sys.executable, "-c", "print('no json')"], 30)
    assert d is None and "rc 0" in err
    d, secs, err = bench._run_sub([sys.executable, "-c", "import sys; print('{\"value\": 1}'); sys.exit(3)"], 30)
    assert d is None and "rc 3" in err
    d, secs, err = bench._run_sub([sys.executable, "-c", "import time; time.sleep(30)"], 1)
    assert d is None and "timed out" in err
    d, secs, err = bench._run_sub([sys.executable, "-c", "print('{\"value\": 7, \"config\": {}}')"], 30)
    assert d == {"value": 7, "config": {}} and err is None
    # the sharded sub-runs: one torch.distributed.run job per mode over the same N GPUs, fresh port, the parent's rank variables dropped
    seen = []

    def fake(cmd, limit_s, env=None):
        seen.append((cmd, env))
        return {"value": 1.0, "config": {"workload": "w", "parallelism": "p"}, "n_gpus": 2, "rccl_ranks": 2}, 1.0, None
    monkeypatch.setattr(bench, "_run_sub", fake)
    monkeypatch.setenv("RANK", "0"); monkeypatch.setenv("WORLD_SIZE", "2"); monkeypatch.setenv("LOCAL_RANK", "0"); monkeypatch.setenv("MASTER_PORT", "1")
    out = bench.sharded_runs(2)
    assert set(out) == {"layer_split_65b", "layer_split_33b_g32_actorder", "layer_split_7b", "tensor_parallel_7b"}
    assert set(bench.sharded_runs(8)) == {"layer_split_65b", "layer_split_7b", "tensor_parallel_7b"}
    ports = set()
    for cmd, env in seen[:4]:
        assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=2" in cmd and "--brief" in cmd
        assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
        ports.add(cmd[cmd.index("--master-port") + 1])
        assert not ({"RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"} & set(env))
        assert ("--layer-split" in cmd) != ("--tensor-parallel" in cmd)
    assert out["layer_split_65b"]["rccl_ranks"] == 2 and out["tensor_parallel_7b"]["decode_mode"] == "p"
    # the PMC traffic file: newest round wins, its own stamp is quoted
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    (tmp_path / "profiles").mkdir()
    assert bench.newest_pmc_profile() == (None, None)
    (tmp_path / "profiles" / "r04_pmc_traffic.json").write_text(json.dumps({"decode_classes": {}, "round": 4}))
    (tmp_path / "profiles" / "r11_pmc_traffic.json").write_text(json.dumps({"decode_classes": {}, "round": 11, "git_head": "abc1234", "collected": "2026-09-23"}))
    d, src = bench.newest_pmc_profile()
    assert d["round"] == 11 and "r11_pmc_traffic.json" in src and "abc1234" in src and "2026-09-23" in src
