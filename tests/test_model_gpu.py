"""GPU model-level parity: exllama_amd.model.ExLlama (HIP kernels through the C ABI) against the CPU oracle model and
the committed golden logits, on the seeded synthetic checkpoints; plus the properties that hold at BASELINE sizes."""
import json
import math
import os

import numpy as np
import pytest
import torch

from exllama_amd import synth
from oracle.model_oracle import OracleLlama
from parity import LORA_TOL, ORACLE_TOL, PATHS_TOL, _direct_steps_close, _model_close, _oracle_steps, _truth_close

pytestmark = pytest.mark.gpu


def _build(name, gs, act, seed=11, max_seq_len=64, num_layers=None, zeros="rand", **cfg_over):
    from exllama_amd.model import ExLlama, ExLlamaCache, ExLlamaConfig
    dims = synth.PRESETS[name]
    tensors = synth.make_checkpoint(dims, groupsize=gs, act_order=act, seed=seed, device="cpu", zeros=zeros, num_layers=num_layers)
    cfg = ExLlamaConfig(synth.config_dict(dims, num_layers))
    cfg.max_seq_len = max_seq_len
    cfg.max_input_len = max(max_seq_len, 16)
    for k, v in cfg_over.items():
        setattr(cfg, k, v)
    model = ExLlama(cfg, tensors={k: v.clone() for k, v in tensors.items()})
    return model, ExLlamaCache(model), tensors, dims


def _ppl(logits, ids):
    """exp(-mean log p(target)) over positions (reference: perplexity.py:121-137)."""
    lp = torch.log_softmax(torch.as_tensor(logits).float(), dim=-1)
    tgt = torch.as_tensor(ids)[:, 1:]
    tok = lp[:, :-1].gather(-1, tgt.unsqueeze(-1)).squeeze(-1)
    return math.exp(-tok.mean().item())


@pytest.mark.parametrize("name,gs,act", [("tiny", 64, False), ("tiny_gqa", 128, True)])
def test_tiny_model_matches_golden_and_oracle(name, gs, act, golden_dir):
    g = np.load(os.path.join(golden_dir, "tiny_model.npz"))
    model, cache, tensors, dims = _build(name, gs, act)
    ids = torch.from_numpy(g[f"{name}_ids"])
    logits = model.forward(ids.to("cuda:0"), cache, last_id_only=False).cpu().numpy()
    ref = g[f"{name}_logits"].astype(np.float32)
    scale = np.abs(ref).max()
    assert np.isfinite(logits).all()
    _model_close(logits, ref, ORACLE_TOL, f"golden prefill {name}")
    # perplexity "equal to 2 dp" (north_star) on the same token stream
    assert abs(_ppl(logits, ids) - _ppl(ref, ids)) < 5e-3 * _ppl(ref, ids)
    # greedy continuation through the fused decode path reproduces the oracle's tokens (integer result)
    tok = int(np.argmax(logits[0, -1]))
    toks = [tok]
    for i in range(4):
        lg = model.forward(torch.tensor([[tok]], device="cuda:0"), cache).cpu().numpy()
        _model_close(lg[0, 0], g[f"{name}_step_logits"][i].astype(np.float32), ORACLE_TOL, f"golden step {name} {i}")
        tok = int(np.argmax(lg[0, 0]))
        toks.append(tok)
    assert toks == g[f"{name}_tokens"].tolist()
    assert cache.current_seq_len == ids.shape[1] + 4
    model.free_unmanaged()


def test_prefill_and_token_by_token_agree():
    """The -v check of the reference (test_benchmark_inference.py:237-246: "should produce roughly equal results"):
    MFMA-GEMM prefill vs GEMV one-token-at-a-time give the same logits within fp16 tolerance and the same ppl to 2 dp."""
    model, cache, tensors, dims = _build("tiny_gqa", 128, True, seed=5)
    from exllama_amd.model import ExLlamaCache
    ids = torch.randint(1, dims.vocab_size, (1, 40), generator=torch.Generator().manual_seed(1)).to("cuda:0")
    a = model.forward(ids, cache, last_id_only=False).cpu()
    cache2 = ExLlamaCache(model)
    b = torch.cat([model.forward(ids[:, i:i + 1], cache2, last_id_only=False).cpu() for i in range(ids.shape[1])], dim=1)
    scale = a.abs().max().item()
    _model_close(b, a, PATHS_TOL, "prefill vs token by token")
    assert abs(_ppl(a, ids.cpu()) - _ppl(b, ids.cpu())) < 5e-3 * _ppl(a, ids.cpu())
    # KV caches written by the two paths agree too
    for l in range(len(cache.key_states)):
        ka, kb = cache.key_states[l][:, :, :40].float(), cache2.key_states[l][:, :, :40].float()
        _model_close(kb, ka, PATHS_TOL, f"prefill vs token by token, K cache {l}")
    model.free_unmanaged()


def test_chunked_prefill_and_threshold_variants():
    """Chunking by max_input_len (model.py:948-984) and flipping the matmul threshold must not change results beyond
    fp16 tolerance; batch rows are independent."""
    model, cache, tensors, dims = _build("tiny", 64, False, seed=9, max_seq_len=64)
    from exllama_amd.model import ExLlamaCache
    ids = torch.randint(1, dims.vocab_size, (1, 37), generator=torch.Generator().manual_seed(2)).to("cuda:0")
    full = model.forward(ids, cache).cpu()
    model.config.max_input_len = 16
    c2 = ExLlamaCache(model)
    chunked = model.forward(ids, c2).cpu()
    assert c2.current_seq_len == 37
    _model_close(chunked, full, PATHS_TOL, "chunked prefill")
    model.config.max_input_len = 64
    # batched: two different prompts in one batch == each alone
    ids2 = torch.randint(1, dims.vocab_size, (2, 9), generator=torch.Generator().manual_seed(3)).to("cuda:0")
    cb = ExLlamaCache(model, batch_size=2)
    both = model.forward(ids2, cb, last_id_only=False).cpu()
    for r in range(2):
        c1 = ExLlamaCache(model)
        one = model.forward(ids2[r:r + 1], c1, last_id_only=False).cpu()
        _model_close(both[r], one[0], PATHS_TOL, f"batch row {r}")
    model.free_unmanaged()


def test_cache_clone_roll_and_copy_states():
    model, cache, tensors, dims = _build("tiny", 64, False, seed=4, max_seq_len=32)
    ids = torch.randint(1, dims.vocab_size, (1, 10)).to("cuda:0")
    model.forward(ids, cache, preprocess_only=True)
    c2 = cache.clone()
    assert torch.equal(c2.key_states[0], cache.key_states[0])
    from exllama_amd.model import ExLlamaCache
    big = ExLlamaCache(model, batch_size=3, max_seq_len=32)
    cache.copy_states(big, 0, 10, 0, 10, 0, 1, 0, 3)
    assert torch.equal(big.value_states[1][2, :, :10], cache.value_states[1][0, :, :10])
    before = cache.key_states[0][:, :, 1].clone()
    cache.roll_left()
    assert torch.equal(cache.key_states[0][:, :, 0], before)
    model.free_unmanaged()


def test_7b_layer_shapes_finite_and_consistent():
    """BASELINE shape (h 4096, I 11008, 32 heads, g128), two layers: 2048-token prefill is finite, its last-token
    logits equal a 2047-token prefill followed by one fused decode step (prefill GEMM path vs decode GEMV path at full
    context), and greedy tokens agree."""
    from exllama_amd.model import ExLlama, ExLlamaCache, ExLlamaConfig
    dims = synth.LLAMA_7B
    tensors = synth.make_checkpoint(dims, groupsize=128, act_order=False, seed=0, device="cuda:0", zeros="sym", num_layers=2)
    cfg = ExLlamaConfig(synth.config_dict(dims, 2))
    cfg.max_seq_len = 2048
    model = ExLlama(cfg, tensors=tensors)
    cache = ExLlamaCache(model)
    ids = torch.randint(0, 31999, (1, 2048), generator=torch.Generator().manual_seed(0)).to("cuda:0")
    a = model.forward(ids, cache).cpu()
    assert torch.isfinite(a).all()
    c2 = ExLlamaCache(model)
    model.forward(ids[:, :2047], c2, preprocess_only=True)
    b = model.forward(ids[:, 2047:], c2).cpu()
    scale = a.abs().max().item()
    _model_close(b, a[:, -1:], PATHS_TOL, "7B shapes: 2048-token prefill vs 2047 + one decode step")
    assert int(a[0, -1].argmax()) == int(b[0, -1].argmax())
    with pytest.raises(RuntimeError, match="exceeds the cache length"):
        model.forward(ids[:, :1], cache)
    model.free_unmanaged()


_DIRECT_STEPS = []          # per case of the test below: decode steps compared with the fp16 oracle DIRECTLY (parity._direct_steps_close)


@pytest.mark.parametrize("name,gs,act,prompt,max_seq", [("tiny_hd128", 128, False, 20, 96), ("tiny_hd128_gqa", 64, True, 20, 96),
                                                        ("tiny_hd128", 128, False, 200, 320), ("tiny_hd128_gqa", 64, True, 700, 1024),
                                                        ("tiny_hd128", 128, False, 2900, 3072)])
def test_native_decode_executor_matches_op_path_and_oracle(name, gs, act, prompt, max_seq):
    """The native decode executor (decode_fused.hip), eager and as a replayed hipGraph, against (a) the
    op-by-op fused path (q4_attn -> attention -> q4_attn_2 -> q4_mlp) and (b) the CPU oracle model.  Prompts of 20 / 200 /
    700 tokens put the decode steps into the 1- / 4- / 16-split buckets (several splits: merged inside the o_proj kernel);
    2900 tokens make a split longer than the 160 keys one pass of the attention kernel holds (its chunk loop)."""
    from exllama_amd.model import ExLlamaCache
    model, cache, tensors, dims = _build(name, gs, act, seed=21, max_seq_len=max_seq)
    ids = torch.randint(1, dims.vocab_size, (1, prompt), generator=torch.Generator().manual_seed(4)).to("cuda:0")
    n_new = 12

    def run(mode, forced=None):
        """Greedy decode; with `forced` the token history of another run is replayed (teacher forcing), so that a near-tie
        in the argmax of two numerically different paths cannot make the histories -- and every later logit -- diverge."""
        c = ExLlamaCache(model)
        model.disable_decode_graph()
        logits = model.forward(ids, c)
        if mode != "ops":
            model.enable_decode_graph(c, use_graph=(mode == "graph"))
        outs, toks = [], []
        for i in range(n_new):
            own = int(logits[0, -1].argmax())
            toks.append(own)
            tok = torch.tensor([[forced[i] if forced is not None else own]], device="cuda:0")
            logits = model.forward(tok, c)
            outs.append(logits[0, 0].float().cpu())
        assert c.current_seq_len == prompt + n_new
        return torch.stack(outs), toks, c

    ops, toks_ops, c_ops = run("ops")
    eager, toks_eager, c_eager = run("eager", forced=toks_ops)
    graph, toks_graph, c_graph = run("graph", forced=toks_ops)
    scale = ops.abs().max().item()
    assert torch.isfinite(eager).all()
    _model_close(eager, ops, PATHS_TOL, f"executor vs op path {name} {prompt}")
    # same greedy choice wherever the op path's top-2 margin is not within the tolerance
    prev = torch.cat([model.forward(ids, ExLlamaCache(model))[0, -1:].float().cpu(), ops[:-1]])
    top2 = prev.topk(2, dim=-1).values
    clear = (top2[:, 0] - top2[:, 1]) > 4e-2 * scale
    assert all(a == b for a, b, ok in zip(toks_eager, toks_ops, clear.tolist()) if ok)
    # replayed graphs and eager launches pick their KV-split count from the same context buckets: the same kernels
    assert (graph - eager).abs().max().item() <= 2e-3 * scale
    graph2, toks_graph2, _ = run("graph", forced=toks_ops)
    assert torch.equal(graph, graph2) and toks_graph == toks_graph2         # bit-reproducible run to run
    for l in range(len(c_ops.key_states)):
        ka, kb = c_ops.key_states[l][:, :, :prompt + n_new].float(), c_graph.key_states[l][:, :, :prompt + n_new].float()
        _model_close(kb, ka, PATHS_TOL, f"executor vs op path, K cache {l}")
    # oracle model on the same tokens
    ref = OracleLlama(synth.config_dict(dims), tensors, max_seq_len=max_seq)
    rl = ref.forward(ids.cpu().numpy())
    # the decode steps are compared in isolation: the oracle continues from the K / V rows the GPU's prompt pass wrote (its own
    # differ from them by fp16 rounding in up to 2900 rows x 3 layers, which the conditioning probe of ONE step cannot see; the
    # prompt pass itself is held to the oracle by the golden, prefill-vs-token and end-to-end tests)
    for l in range(dims.num_hidden_layers):
        ref.kc[l][0, :, :prompt] = c_graph.key_states[l][0, :, :prompt].cpu().numpy()
        ref.vc[l][0, :, :prompt] = c_graph.value_states[l][0, :, :prompt].cpu().numpy()
    ref_steps, runs, truth = _oracle_steps(ref, toks_ops, prompt)
    for i in range(n_new):
        _truth_close(graph[i].numpy(), runs, truth, i, f"executor vs truth {name} {prompt} step {i}")
    # ... and against the fp16 oracle itself wherever the step is well-conditioned (parity._direct_steps_close: why)
    _DIRECT_STEPS.append(_direct_steps_close([graph[i].numpy() for i in range(n_new)], runs, truth, f"executor {name} {prompt}"))
    # rewinding the cache on the host is picked up by the device-side position
    model.enable_decode_graph(c_graph, use_graph=True)
    c_graph.current_seq_len = prompt
    again = model.forward(torch.tensor([[toks_ops[0]]], device="cuda:0"), c_graph)[0, 0].float().cpu()
    assert torch.equal(again, graph[0])
    model.free_unmanaged()


def test_some_decode_step_was_compared_with_the_fp16_oracle_directly():
    """Runs behind the five cases of test_native_decode_executor_matches_op_path_and_oracle (60 decode steps): the float64 truth
    criterion must not be the ONLY thing between a decode step and the oracle."""
    if not _DIRECT_STEPS:
        pytest.skip("test_native_decode_executor_matches_op_path_and_oracle did not run in this session")
    assert sum(_DIRECT_STEPS) >= 1, _DIRECT_STEPS


def test_perplexity_module_chunk_and_token_modes_and_oracle():
    """exllama_amd.perplexity (the reference's -ppl leg, perplexity.py:93-138) on a seeded token stream: whole-chunk
    (MFMA GEMM + flash prefill) and token-by-token (decode kernels) evaluation agree to the reference's own precision
    (2 decimals relative), and both agree with the CPU oracle model."""
    from exllama_amd.perplexity import Perplexity
    model, cache, tensors, dims = _build("tiny_gqa", 64, True, seed=3, max_seq_len=48)
    ids = torch.randint(1, dims.vocab_size, (1, 100), generator=torch.Generator().manual_seed(7))
    p = Perplexity(model=model, cache=cache)
    p.add_tokens(ids.to("cuda:0"), chunk_size=40, overlap=4)
    assert [c.shape[1] for c in p.dataset_chunks] == [40, 40, 28]
    whole = p.test(quiet=True)
    token = p.test(quiet=True, ppl_token=True)
    orc = OracleLlama(synth.config_dict(dims), tensors, max_seq_len=48)
    lp_sum, n = 0.0, 0
    for c in p.dataset_chunks:
        c = c.cpu()
        orc.reset()
        lg = torch.from_numpy(np.asarray(orc.forward(c[:, :-1].numpy(), last_id_only=False), dtype=np.float32))
        lp = torch.log_softmax(lg, dim=-1).gather(-1, c[:, 1:].unsqueeze(-1))
        lp_sum += lp.sum().item()
        n += c.shape[1] - 1
    ref = math.exp(-lp_sum / n)
    assert math.isfinite(whole) and whole > 1.0
    assert abs(whole - token) < 5e-3 * whole, (whole, token)
    assert abs(whole - ref) < 5e-3 * ref, (whole, ref)
    model.free_unmanaged()


def test_perplexity_agrees_to_the_second_decimal_of_the_readme_range():
    """north_star: "perplexity equal to 2 dp".  The reference prints perplexity with 4 decimals (perplexity.py:121-138) and its
    README quotes 5.68 .. 3.53 for 7B .. 65B.  No real checkpoint exists here, so the claim is tested on BASELINE configs[1] layer
    shapes (two layers, vocabulary 32000) with the head sharpened until the model's own text scores IN that range (head x 4.6:
    next-token entropy 1.7 - 1.8 nats, perplexity ~6; calibrated on the CPU oracle), over 1535 tokens SAMPLED from the model's
    next-token distribution (decode path):
      * whole-chunk HIP path (MFMA GEMMs, flash attention) vs CPU oracle, neither of which chose the text: ABSOLUTE
        |delta| < 0.005, i.e. equal to the second decimal (first-order logit noise averages out over the tokens: relative 1.3e-4
        measured at head x 3 in round 3);
      * token-by-token HIP path vs oracle: this path SAMPLED the text, and a path that samples the text scores it with its own
        entropy while every other path pays the KL divergence to it on top (log E_p[exp(d)] - E_p[d] ~ s^2 / 2 for logit noise of
        standard deviation s nats, whatever the text): its perplexity is the lowest of the three by construction.  The term is
        second order in the fp16 noise of the logits (s ~ 0.04 - 0.06 nats at logits of +-30, fp16 spacing 0.016 - 0.03 there), so it is
        bounded, not equal: |delta| < 0.02 absolute, measured value recorded through EXL_TOL_STATS (profiles/)."""
    from exllama_amd.model import ExLlama, ExLlamaCache, ExLlamaConfig
    from exllama_amd.perplexity import Perplexity
    dims, L, S = synth.PRESETS["7b"], 2, 1536
    tensors = synth.make_checkpoint(dims, groupsize=128, act_order=False, seed=23, device="cpu", zeros="rand", num_layers=L)
    tensors["lm_head.weight"] = (tensors["lm_head.weight"].float() * 4.6).half()
    cfg = ExLlamaConfig(synth.config_dict(dims, L))
    cfg.max_seq_len = S + 64
    cfg.max_input_len = 2048
    model = ExLlama(cfg, tensors={k: v.clone() for k, v in tensors.items()})
    gen = torch.Generator().manual_seed(17)
    cache = ExLlamaCache(model)
    seq = torch.randint(1, dims.vocab_size, (4,), generator=gen).tolist()
    lg = model.forward(torch.tensor([seq], device="cuda:0"), cache)
    while len(seq) < S:
        nxt = int(torch.multinomial(torch.softmax(lg[0, -1].float().cpu(), -1), 1, generator=gen))
        seq.append(nxt)
        lg = model.forward(torch.tensor([[nxt]], device="cuda:0"), cache)
    ids = torch.tensor([seq])
    p = Perplexity(model=model, cache=ExLlamaCache(model))
    p.add_tokens(ids.to("cuda:0"), chunk_size=S, overlap=0)
    whole = p.test(quiet=True)
    token = p.test(quiet=True, ppl_token=True)
    orc = OracleLlama(synth.config_dict(dims, L), tensors, max_seq_len=cfg.max_seq_len)
    orc.prepare()
    lp_sum, n = 0.0, 0
    for c in p.dataset_chunks:
        c = c.cpu()
        orc.reset()
        lgo = torch.from_numpy(np.asarray(orc.forward(c[:, :-1].numpy(), last_id_only=False), dtype=np.float32))
        lp = torch.log_softmax(lgo, dim=-1).gather(-1, c[:, 1:].unsqueeze(-1))
        lp_sum += lp.sum().item()
        n += c.shape[1] - 1
    ref = math.exp(-lp_sum / n)
    stats = os.environ.get("EXL_TOL_STATS")
    if stats:
        with open(stats, "a") as f:
            f.write(json.dumps({"tag": "perplexity whole / token / oracle", "values": [whole, token, ref], "tokens": n}) + "\n")
    assert 3.0 < ref < 12.0, ref                                     # the README's range (5.68 .. 3.53) or just above it
    assert abs(whole - ref) < 0.005, (whole, token, ref)             # equal to the second decimal
    assert abs(token - ref) < 0.02 and token < whole + 0.005, (whole, token, ref)
    model.free_unmanaged()


@pytest.mark.parametrize("name,gs,act", [("tiny", 64, False), ("tiny_gqa", 128, True)])
def test_lora_adapter_end_to_end(name, gs, act):
    """exllama_amd.lora.ExLlamaLora (PEFT layout in, transposed + pre-scaled halves out) through every projection of the
    model, prefill (q4_matmul_lora) and single-token decode (q4_attn / q4_mlp with LoRA operands), against the oracle
    model with the same adapter; and the adapter must actually change the logits."""
    from exllama_amd.lora import ExLlamaLora
    from exllama_amd.model import ExLlamaCache
    model, cache, tensors, dims = _build(name, gs, act, seed=9)
    g = torch.Generator().manual_seed(21)
    r, sd = 8, {}
    kvd = dims.hidden_size // dims.num_attention_heads * dims.num_key_value_heads
    shapes = {"self_attn.q_proj": (dims.hidden_size, dims.hidden_size), "self_attn.k_proj": (dims.hidden_size, kvd),
              "self_attn.v_proj": (dims.hidden_size, kvd), "self_attn.o_proj": (dims.hidden_size, dims.hidden_size),
              "mlp.gate_proj": (dims.hidden_size, dims.intermediate_size), "mlp.up_proj": (dims.hidden_size, dims.intermediate_size),
              "mlp.down_proj": (dims.intermediate_size, dims.hidden_size)}
    for i in range(dims.num_hidden_layers):
        for key, (fin, fout) in shapes.items():
            if i == 1 and key == "mlp.up_proj":
                continue                                                     # a projection without an adapter stays on the plain path
            sd[f"base_model.model.model.layers.{i}.{key}.lora_A.weight"] = torch.randn(r, fin, generator=g) * 0.05
            sd[f"base_model.model.model.layers.{i}.{key}.lora_B.weight"] = torch.randn(fout, r, generator=g) * 0.05
    lora = ExLlamaLora(model, {"r": r, "lora_alpha": 16}, "synthetic.bin", tensors=sd)
    assert all(t.device.type == "cuda" and t.dtype == torch.float16 for t in lora.tensors.values())
    ids = torch.randint(1, dims.vocab_size, (1, 12), generator=torch.Generator().manual_seed(2))
    base = model.forward(ids.to("cuda:0"), cache, last_id_only=False).float().cpu()
    cache2 = ExLlamaCache(model)
    got = model.forward(ids.to("cuda:0"), cache2, last_id_only=False, lora=lora).float().cpu()
    orc = OracleLlama(synth.config_dict(dims), tensors, max_seq_len=64)
    # the product's loader put the adapter halves that touch the MLP's intermediate activations into the order of the folded
    # act-order down_proj (exllama_amd.lora / model._fold_act_order_down_proj); the oracle works on the checkpoint as it is
    plain = {k: v.cpu() for k, v in lora.tensors.items()}
    for i, layer in enumerate(model.layers):
        fold = layer.mlp.fold_map
        if fold is None:
            continue
        inv = torch.empty_like(fold)
        inv[fold] = torch.arange(fold.numel())
        for proj in ("gate_proj", "up_proj"):
            k = f"model.layers.{i}.mlp.{proj}.lora_B.weight"
            if k in plain:
                plain[k] = plain[k][:, inv].contiguous()
        k = f"model.layers.{i}.mlp.down_proj.lora_A.weight"
        if k in plain:
            plain[k] = plain[k][inv, :].contiguous()
    orc.set_lora(plain)
    ref = torch.from_numpy(np.asarray(orc.forward(ids.numpy(), last_id_only=False), dtype=np.float32))
    scale = ref.abs().max().item()
    _model_close(got, ref, LORA_TOL, f"LoRA prefill {name}")
    assert (got - base).abs().max().item() > 5e-2 * scale                    # the adapter is not a no-op
    tok = torch.tensor([[int(ref[0, -1].argmax())]])
    step = model.forward(tok.to("cuda:0"), cache2, lora=lora).float().cpu()   # rows == 1: fused decode ops with LoRA operands
    ref_step = torch.from_numpy(np.asarray(orc.forward(tok.numpy()), dtype=np.float32))
    _model_close(step, ref_step, LORA_TOL, f"LoRA step {name}")
    model.free_unmanaged()


def test_device_greedy_generation_matches_the_host_loop():
    """generate_greedy (argmax inside the captured graph, exl_decoder_step_greedy) produces the token sequence of the
    reference's loop -- forward, torch.argmax on the logits, forward ... -- and leaves the same final logits and cache."""
    from exllama_amd.model import ExLlamaCache
    model, cache, tensors, dims = _build("tiny_hd128", 128, False, seed=4, max_seq_len=256)
    ids = torch.randint(1, dims.vocab_size, (1, 150), generator=torch.Generator().manual_seed(3)).to("cuda:0")   # crosses the 160-key bucket limit
    n = 24
    lg = model.forward(ids, cache)
    first = lg[0, -1].argmax().view(1, 1)
    model.enable_decode_graph(cache)
    # host loop: graph replay, torch.argmax, graph replay ... (same kernels, so the same logits bit for bit)
    toks, tok, cur = [], first, lg
    for _ in range(n):
        cur = model.forward(tok, cache)
        tok = cur[0, -1].argmax().view(1, 1)
        toks.append(int(tok))
    assert len(set(toks)) > 1
    # device loop from the same position
    cache.current_seq_len = 150
    got = model.generate_greedy(first, cache, n)
    assert cache.current_seq_len == 150 + n
    assert got.dtype == torch.int64 and got.tolist() == toks
    last = model.last_decoder_logits()
    assert torch.equal(last, cur.to(last.device).view_as(last))
    # the executor can continue token by token afterwards, and be rewound
    nxt = model.forward(got[-1].view(1, 1), cache)
    assert torch.isfinite(nxt).all() and cache.current_seq_len == 150 + n + 1
    cache.current_seq_len = 150
    again = model.generate_greedy(first, cache, n)
    assert again.tolist() == toks
    with pytest.raises(RuntimeError):
        model.generate_greedy(first, cache, 1000)
    model.free_unmanaged()


def test_weight_arena_is_one_allocation_and_changes_nothing():
    """config.weight_arena (default): every device tensor of the model is a 2 MiB-aligned view into one allocation;
    logits are bit-identical to a model loaded tensor by tensor."""
    model, cache, tensors, dims = _build("tiny_gqa", 64, True, seed=8)
    ids = torch.randint(1, dims.vocab_size, (1, 17), generator=torch.Generator().manual_seed(9)).to("cuda:0")
    a = model.forward(ids, cache, last_id_only=False).cpu()
    assert len(model._arenas) == 1
    base, size = model._arenas[0].data_ptr(), model._arenas[0].numel()
    lin = model.layers[1].mlp.down_proj
    for t in (model.embed_weight, model.lm_head_weight, model.norm.weight, lin.qweight, lin.scales, lin.qzeros):
        assert base <= t.data_ptr() < base + size and (t.data_ptr() - base) % (2 << 20) == 0
    model.free_unmanaged()
    model2, cache2, _, _ = _build("tiny_gqa", 64, True, seed=8, weight_arena=False)
    assert not getattr(model2, "_arenas", [])
    b = model2.forward(ids, cache2, last_id_only=False).cpu()
    assert torch.equal(a, b)
    model2.free_unmanaged()


# ---- the native decode executor at the REAL layer shapes (BASELINE configs 1-4) ------------------------------------------------
# What bench.py times are dec_ring_kernel (decode_ring.hip: the hand-counted rolling-ring weight stream) and, for what the ring
# does not cover (group sizes 32 / 64, launches that gather through an act-order map, the deepest two-per-CU down_proj streams),
# dec_stream_kernel instantiations picked by layer shape (decode_fused.hip: launch_dec_gemv_cfg): (U, NP) by row-blocks per
# wave, NV by K, the stand-alone split merge for hidden > 4096, act-order staging through x_map.  The tiny presets above only
# ever launch <4,1,.,.,.,1>; these cases run the executor (eager, graph replay, greedy generation) on truncated models of the
# real dimensions against the CPU oracle model, and assert through exl_decoder_plan WHICH instantiation ran, so that every
# configuration the benchmark launches is compared with the oracle.
#   per kernel class, [1] = launched at all: dec_ring_kernel (U, UL, 2, PNORM, EMODE, NV) -- U loads in flight per lane, UL row-blocks
#   per wave and tile -- or dec_stream_kernel (U, NP, G16, PNORM, EMODE, NV) with the scale applied per row-block (G16 = 1: group
#   size % 128 == 0) / per k-group (0: group sizes 32, 64)
_REAL_SHAPES = {
    #        name   gs   act    L  qkv                    o_proj (>1 split)       gate_up                down                   merge kernel
    # 7B o_proj at one KV split and down_proj: 256 tiles = one block per CU -> 16-wave blocks (2 / 6 row-blocks per wave)
    "7b":  ("7b", 128, False, 2, (3, 4, 2, 1, 0, 1), (3, 4, 2, 3, 1, 1), (3, 8, 2, 1, 2, 1), (4, 6, 2, 0, 1, 2), False),
    # (U, NP) fits the row-blocks per wave exactly where it can: 13B 5 / 10 / 14, 33B 7 / 13 / 18, 65B 8 / 16 / 22, 70B 8 / 16 / 28
    # 13B act-order: q/k/v and gate/up gather one image per matrix through their maps (ring kernel, PNORM 2; gate/up also stores
    # through down_proj's inverse map); o_proj / down_proj get their input already in row order
    "13b": ("13b", 128, True, 1, (4, 5, 2, 2, 0, 2), (4, 5, 2, 0, 1, 2), (3, 10, 2, 2, 2, 2), (3, 14, 2, 0, 1, 6), True),
    # 33B group size 32 act-order: the ring kernel's GM form (a scale / zero pair per lane and piece), gathers as for 13B
    "33b": ("33b", 32, True, 1, (3, 7, 2, 2, 0, 2), (3, 7, 2, 0, 1, 2), (3, 13, 2, 2, 2, 2), (3, 18, 2, 0, 1, 6), True),
    "65b": ("65b", 128, False, 1, (3, 8, 2, 1, 0, 2), (3, 8, 2, 0, 1, 2), (3, 16, 2, 1, 2, 2), (3, 22, 2, 0, 1, 6), True),
    # Llama-2-70B: GQA (8 kv heads) and K = 28672 in down_proj -- 28 row-blocks per wave, 7 -> 8 vectors per thread
    "70b": ("70b", 128, False, 1, (3, 8, 2, 1, 0, 2), (3, 8, 2, 0, 1, 2), (3, 16, 2, 1, 2, 2), (3, 28, 2, 0, 1, 8), True),
}


_PLAIN_O_PROJ = {"7b": (2, 2, 2, 0, 1, 1)}      # o_proj at ONE KV split where it differs from "the merge variant with PNORM 0"


def _plan(model, cls):
    import ctypes as C
    from exllama_amd import cuda_ext
    out = (C.c_int * 10)()
    cuda_ext.check(cuda_ext.exllama_ext._lib.exl_decoder_plan(model._decoder["handle"], cls, out), "decoder_plan")
    return list(out)


@pytest.mark.parametrize("key", list(_REAL_SHAPES))
def test_native_decode_executor_at_real_layer_shapes(key):
    from exllama_amd.model import ExLlama, ExLlamaCache, ExLlamaConfig
    name, gs, act, L, p_qkv, p_o, p_gu, p_down, merge_kernel = _REAL_SHAPES[key]
    dims = synth.PRESETS[name]
    tensors = synth.make_checkpoint(dims, groupsize=gs, act_order=act, seed=3, device="cpu", zeros="rand", num_layers=L)
    cfg = ExLlamaConfig(synth.config_dict(dims, L))
    full_ctx = key == "7b"                                           # the benchmarked configuration also gets a step at context >= 2048
    cfg.max_seq_len = 2176 if full_ctx else 1408                    # > 1280: the decoder sizes for 512 attention blocks, so its maximum split count exceeds 4
    cfg.max_input_len = 2048
    model = ExLlama(cfg, tensors=tensors)                            # copies to the device; the CPU tensors stay as generated
    ref = OracleLlama(synth.config_dict(dims, L), tensors, max_seq_len=cfg.max_seq_len)
    ref.prepare()                                                    # dequantise once (reconstruct bits), fp32 BLAS per step
    rs = np.random.RandomState(5)
    ids = torch.from_numpy(rs.randint(1, dims.vocab_size, size=(1, 2060))).to("cuda:0")
    prompts = [20, 200, 700] + ([2047] if full_ctx else [])          # 1- / 4- / max-split buckets (+ the full context)
    n_new = 3
    seen = set()
    for P in prompts:
        cache = ExLlamaCache(model)
        model.disable_decode_graph()
        model.forward(ids[:, :P], cache, preprocess_only=True)       # prompt through the MFMA GEMM / flash path
        # the oracle continues from the SAME cache contents: the decode step is compared in isolation
        for i in range(L):
            ref.kc[i][0, :, :P] = cache.key_states[i][0, :, :P].cpu().numpy()
            ref.vc[i][0, :, :P] = cache.value_states[i][0, :, :P].cpu().numpy()
        toks = ids[0, P:P + n_new].tolist()
        ref_steps, runs, truth = _oracle_steps(ref, toks, P)
        for mode in ("eager", "graph"):
            c = ExLlamaCache(model, copy_from=cache)
            c.current_seq_len = P
            model.enable_decode_graph(c, use_graph=(mode == "graph"))
            for i, t in enumerate(toks):
                lg = model.forward(torch.tensor([[t]], device="cuda:0"), c)[0, 0].float().cpu().numpy()
                assert np.isfinite(lg).all()
                _truth_close(lg, runs, truth, i, f"real shapes {key} ctx {P} {mode} step {i}")
                if mode == "eager":                                  # which kernels this step launched
                    model._set_eager_splits(model._decoder, P + i)
                    plans = {cls: _plan(model, j) for j, cls in enumerate(model.DECODER_CLASSES)}
                    nsplit = plans["attn"][1]
                    assert tuple(plans["qkv"][1:7]) == p_qkv and tuple(plans["gate_up"][1:7]) == p_gu and tuple(plans["down"][1:7]) == p_down
                    exp_o = p_o if nsplit > 1 else _PLAIN_O_PROJ.get(key, p_o[:3] + (0,) + p_o[4:])   # one split: plain o_proj, nothing to merge
                    assert tuple(plans["o_proj"][1:7]) == exp_o, (plans["o_proj"], exp_o)
                    assert bool(plans["merge"][0]) == (merge_kernel and nsplit > 1)
                    seen.add(nsplit)
            # the K/V rows the executor appended match the oracle's (RoPE + scatter inside the attention kernel)
            for l in range(L):
                ka = c.key_states[l][0, :, P:P + n_new].float().cpu().numpy()
                _model_close(ka, ref.kc[l][0, :, P:P + n_new].astype(np.float32), ORACLE_TOL, f"real shapes {key} ctx {P} K rows layer {l}")
            if mode == "graph":                                      # device-side greedy generation from the same state
                c.current_seq_len = P
                first = torch.tensor([[toks[0]]], device="cuda:0")
                got = model.generate_greedy(first, c, 2)
                c2 = ExLlamaCache(model, copy_from=cache)
                c2.current_seq_len = P
                model.enable_decode_graph(c2, use_graph=True)
                l1 = model.forward(first, c2)
                t1 = int(l1[0, -1].argmax())
                l2 = model.forward(torch.tensor([[t1]], device="cuda:0"), c2)
                assert got.tolist() == [t1, int(l2[0, -1].argmax())]
    assert len(seen) >= 3, seen                                      # 1 split, 4 splits and the decoder's maximum all ran
    model.free_unmanaged()


@pytest.mark.parametrize("name,gs,act,L", [("7b", 128, False, 2), ("13b", 128, True, 1), ("33b", 32, True, 1), ("65b", 128, False, 1)])
def test_real_shape_prefill_end_to_end_vs_oracle(name, gs, act, L):
    """BASELINE configs[1] / [2] / [3] / [4] shapes (7B g128; 13B g128 act-order; 33B g32 act-order; 65B g128), the whole 2048-token prompt through
    the product's prefill path (act-order: the gather folded into the GEMM's activation staging; fused q/k/v + RoPE + cache GEMM
    -> flash attention -> o_proj GEMM -> dual gate/up GEMM + SiLU -> down GEMM), compared END TO END with the CPU oracle model
    that ran the same prompt itself: last-token logits, K / V cache rows at sampled positions of every layer, and the next
    token's logits through the decode executor continuing from that cache -- nothing is seeded from the GPU.
    (test_native_decode_executor_at_real_layer_shapes isolates the decode step by copying the GPU's cache into the oracle.)"""
    from exllama_amd.model import ExLlama, ExLlamaCache, ExLlamaConfig
    dims, S = synth.PRESETS[name], 2048
    tensors = synth.make_checkpoint(dims, groupsize=gs, act_order=act, seed=13, device="cpu", zeros="rand", num_layers=L)
    cfg = ExLlamaConfig(synth.config_dict(dims, L))
    cfg.max_seq_len = S + 128
    cfg.max_input_len = S
    model = ExLlama(cfg, tensors=tensors)
    ref = OracleLlama(synth.config_dict(dims, L), tensors, max_seq_len=cfg.max_seq_len)
    ref.prepare()
    ids = np.random.RandomState(6).randint(1, dims.vocab_size, size=(1, S))
    want = ref.forward(ids)[0, 0]                                      # fp32 BLAS on the reconstructed weights, fp16 at the reference's points
    cache = ExLlamaCache(model)
    got = model.forward(torch.from_numpy(ids).to("cuda:0"), cache)[0, 0].float().cpu().numpy()
    _model_close(got, want, ORACLE_TOL, f"{name} shapes, 2048-token prefill, last-token logits")
    rows = [0, 1, 255, 256, 1023, 1024, 2046, 2047]
    for l in range(L):
        _model_close(cache.key_states[l][0][:, rows].float().cpu().numpy(), ref.kc[l][0][:, rows].astype(np.float32), ORACLE_TOL, f"{name} K rows layer {l}")
        _model_close(cache.value_states[l][0][:, rows].float().cpu().numpy(), ref.vc[l][0][:, rows].astype(np.float32), ORACLE_TOL, f"{name} V rows layer {l}")
    # the next token through the decode executor, against the float64 truth continuing from the ORACLE's own cache
    tok = int(np.argmax(want))
    _, runs, truth = _oracle_steps(ref, [tok], S)
    model.enable_decode_graph(cache, use_graph=True)
    got2 = model.forward(torch.tensor([[tok]], device="cuda:0"), cache)[0, 0].float().cpu().numpy()
    _truth_close(got2, runs, truth, 0, f"{name} shapes, decode step after the 2048-token prefill")
    model.free_unmanaged()


@pytest.mark.parametrize("name,gs,act,L", [("7b", 128, False, 2), ("13b", 128, "gptq", 1), ("33b", 32, "gptq", 1)])
def test_real_shape_short_prompts_vs_oracle(name, gs, act, L):
    """Short prompts (2 .. 256 rows: BASELINE configs[0]'s 128 tokens, chat turns) at real layer shapes: one native call per layer, GEMMs on
    fragment-order activations (exl_q4_layer_prompt, csrc/q4_gemm_frag.hip; reference: model.py:421-552 op by op) -- q / k / v as one
    launch, gate / up + SiLU * mul as one launch writing down_proj's input in fragment order, act-order maps applied by the
    producers.  Whole-sequence logits and cache rows against the CPU oracle model; lengths around the 16- / 64-row tile edges, a
    second chunk on top of a filled cache (past_len > 0), and 257 rows (the first length NOT on this path).
    The checkpoint carries the statistics GPTQ writes (symmetric zero points, nibbles centred on them: synth.make_checkpoint), not the
    random-zero-point stress variant of the sampled-position tests: EVERY logit of EVERY position and EVERY cache row is compared here,
    and with random zero points a few of the 256 x 32000 logits behind a filled cache sit where the attention scores are nearly tied
    -- the long-established op-by-op kernels miss the same bound there by the same amount (5.9e-3 against 4e-3 at 13B shapes, op by op
    and on this path alike: gpurun_out/r06h/tol_short*.jsonl).  Random zero points are covered per GEMM, at 1.5 ulps, by
    test_ops_gpu.py::test_q4_matmul_frag_vs_oracle."""
    from exllama_amd.model import ExLlama, ExLlamaCache, ExLlamaConfig
    dims = synth.PRESETS[name]
    tensors = synth.make_checkpoint(dims, groupsize=gs, act_order=act, seed=29, device="cpu", zeros="sym", nibbles="centered", num_layers=L)
    cfg = ExLlamaConfig(synth.config_dict(dims, L))
    cfg.max_seq_len = 512
    cfg.max_input_len = 512
    model = ExLlama(cfg, tensors=tensors)
    ref = OracleLlama(synth.config_dict(dims, L), tensors, max_seq_len=cfg.max_seq_len)
    ref.prepare()
    rs = np.random.RandomState(3)
    for S, S2 in ((2, 0), (17, 0), (70, 47), (128, 128), (250, 6), (256, 0), (257, 0)):
        ids = rs.randint(1, dims.vocab_size, size=(1, S + S2))
        ref.reset()
        cache = ExLlamaCache(model)
        got = model.forward(torch.from_numpy(ids[:, :S]).to("cuda:0"), cache, last_id_only=False).float().cpu().numpy()
        want = np.asarray(ref.forward(ids[:, :S], last_id_only=False), dtype=np.float32)
        _model_close(got, want, ORACLE_TOL, f"{name} shapes, {S}-token prompt, all logits")
        if S2:
            got = model.forward(torch.from_numpy(ids[:, S:]).to("cuda:0"), cache, last_id_only=False).float().cpu().numpy()
            want = np.asarray(ref.forward(ids[:, S:], last_id_only=False), dtype=np.float32)
            _model_close(got, want, ORACLE_TOL, f"{name} shapes, {S2} more tokens behind {S} cached ones")
        n = S + S2
        for l in range(L):
            _model_close(cache.key_states[l][0][:, :n].float().cpu().numpy(), ref.kc[l][0][:, :n].astype(np.float32), ORACLE_TOL, f"{name} K rows, {n} tokens")
            _model_close(cache.value_states[l][0][:, :n].float().cpu().numpy(), ref.vc[l][0][:, :n].astype(np.float32), ORACLE_TOL, f"{name} V rows, {n} tokens")
        assert cache.current_seq_len == n
    model.free_unmanaged()


@pytest.mark.parametrize("key", ["7b", "13b", "33b", "65b", "70b"])
def test_ring_stream_equals_compiler_stream_bit_for_bit(key):
    """decode_ring.hip (every vector-memory instruction inline asm, every wait counted by hand) against dec_stream_kernel
    (ordinary loads, hipcc's waits) on the same decoder: same arithmetic in the same order, so the logits, the appended K/V
    rows and the greedy tokens must be IDENTICAL bit patterns -- at the real layer shapes (7B; 13B act-order: gathered images and the permuted
    gate/up store; 33B group size 32 act-order: a scale / zero pair per lane and piece; 65B / 70B long units), over the ring's options (start-up barrier, depth 2 - 4), in the one-split and the many-split attention buckets, several tokens in a row (each step consumes
    what the previous one wrote).  A miscounted wait shows up here as a different bit somewhere, not as a tolerance question.
    (16-wave blocks split K differently: those runs are held to fp32-reordering noise instead.)"""
    import ctypes as C
    from exllama_amd import cuda_ext
    from exllama_amd.model import ExLlama, ExLlamaCache, ExLlamaConfig
    name, gs, act, L = _REAL_SHAPES[key][:4]
    dims = synth.PRESETS[name]
    tensors = synth.make_checkpoint(dims, groupsize=gs, act_order=act, seed=11, device="cpu", zeros="rand", num_layers=L)
    cfg = ExLlamaConfig(synth.config_dict(dims, L))
    cfg.max_seq_len = 1408
    cfg.max_input_len = 1024
    model = ExLlama(cfg, tensors=tensors)
    lib = cuda_ext.exllama_ext._lib
    rs = np.random.RandomState(9)
    ids = torch.from_numpy(rs.randint(1, dims.vocab_size, size=(1, 1000))).to("cuda:0")
    for P in (30, 900):
        cache = ExLlamaCache(model)
        model.disable_decode_graph()
        model.forward(ids[:, :P], cache, preprocess_only=True)
        toks = ids[0, P:P + 4].tolist()
        runs = {}
        # (ring classes, start-up barrier, loads in flight per lane, 16-wave blocks)
        modes = ((0, 0, 4, 0), (15, 1, 4, 0), (15, 0, 3, 0), (15, 1, 2, 0), (15, 1, 4, 1), (15, 0, 3, 1))
        for mode in modes:
            c = ExLlamaCache(model, copy_from=cache)
            c.current_seq_len = P
            model.enable_decode_graph(c, use_graph=False)
            for sg in model._decoder["stages"]:
                for opt, val in enumerate(mode):
                    cuda_ext.check(lib.exl_decoder_set_option(sg["handle"], opt, val), "set_option")
            kinds = {cls: _plan(model, j)[3] for j, cls in enumerate(model.DECODER_CLASSES) if cls in ("qkv", "o_proj", "gate_up", "down")}
            if mode[0]:
                assert 2 in kinds.values(), kinds                    # the ring really is what ran
            else:
                assert 2 not in kinds.values(), kinds
            out = [model.forward(torch.tensor([[t]], device="cuda:0"), c)[0, 0].float().cpu().numpy().copy() for t in toks]
            kv = [c.key_states[l][0, :, P:P + 4].cpu().numpy().copy() for l in range(L)]
            runs[mode] = (out, kv)
        ref_out, ref_kv = runs[modes[0]]
        scale = float(np.abs(np.stack(ref_out)).max())
        for mode in modes[1:]:
            out, kv = runs[mode]
            for i in range(len(toks)):
                assert np.isfinite(out[i]).all()
                if mode[3] == 0:                                     # same split of K over the waves: the same additions in the same order
                    assert np.array_equal(out[i].view(np.uint32), ref_out[i].view(np.uint32)), (key, P, mode, i, float(np.abs(out[i] - ref_out[i]).max()))
                else:                                                # 16-wave blocks sum 16 partial dot products instead of 8: fp32 rounding only
                    assert float(np.abs(out[i] - ref_out[i]).max()) <= 2.0 ** -8 * scale, (key, P, mode, i, float(np.abs(out[i] - ref_out[i]).max()), scale)
            if mode[3] == 0:
                for l in range(L):
                    assert np.array_equal(kv[l].view(np.uint16), ref_kv[l].view(np.uint16)), (key, P, mode, l)
    model.free_unmanaged()


def test_decode_executor_as_stages_matches_the_single_executor_bit_for_bit():
    """The layer split of the reference (ExLlamaDeviceMap, model.py:636-668; hop at :1053-1058) on the native executor:
    (a) ONE model whose executor is cut into three stages (config.decoder_stage_split stands in for a device boundary: the
    test boxes have one GPU), residual stream handed over through the stages' buffers;
    (b) TWO models, each holding half of the layers (pipeline.stage_tensors), driven as the two links of a split across
    processes (decode_stage_step: hidden state out of link 0 into link 1).
    Both must reproduce the single-executor logits BIT FOR BIT, eager and replayed, and keep working across KV-split buckets."""
    from exllama_amd.model import ExLlama, ExLlamaCache, ExLlamaConfig
    from exllama_amd.pipeline import stage_tensors
    dims = synth.LLAMA_TINY_HD128
    L = 4
    tensors = synth.make_checkpoint(dims, groupsize=128, act_order=True, seed=17, device="cpu", zeros="rand", num_layers=L)

    def build(t, n_layers, **over):
        cfg = ExLlamaConfig(synth.config_dict(dims, n_layers))
        cfg.max_seq_len = 256
        for k, v in over.items():
            setattr(cfg, k, v)
        m = ExLlama(cfg, tensors={k: v.clone() for k, v in t.items()})
        return m, ExLlamaCache(m)

    ids = torch.randint(1, dims.vocab_size, (1, 150), generator=torch.Generator().manual_seed(6)).to("cuda:0")    # decode crosses the 160-key bucket
    n_new = 14
    whole, c_whole = build(tensors, L)
    staged, c_staged = build(tensors, L, decoder_stage_split=[1, 3])
    lo, c_lo = build(stage_tensors(tensors, 0, 2), 2)
    hi, c_hi = build(stage_tensors(tensors, 2, 4), 2)
    # prompt: whole model vs the two links run back to back (op path: embed -> layers | layers -> head)
    ref_logits = whole.forward(ids, c_whole)
    staged_logits = staged.forward(ids, c_staged)
    h = lo.forward_layers(lo.embed(ids), c_lo)
    c_lo.current_seq_len += ids.shape[1]
    h = hi.forward_layers(h, c_hi)
    c_hi.current_seq_len += ids.shape[1]
    assert torch.equal(hi.head(h), ref_logits) and torch.equal(staged_logits, ref_logits)
    for use_graph in (False, True):
        for m, c in ((whole, c_whole), (staged, c_staged), (lo, c_lo), (hi, c_hi)):
            c.current_seq_len = ids.shape[1]
        whole.enable_decode_graph(c_whole, use_graph=use_graph)
        staged.enable_decode_graph(c_staged, use_graph=use_graph)
        assert len(staged._decoder["stages"]) == 3 and len(whole._decoder["stages"]) == 1
        lo.enable_decode_graph(c_lo, use_graph=use_graph, last_stage=False)
        hi.enable_decode_graph(c_hi, use_graph=use_graph, first_stage=False)
        tok = ref_logits[0, -1].argmax().view(1, 1)
        for i in range(n_new):
            a = whole.forward(tok, c_whole)
            b = staged.forward(tok, c_staged)
            hid = lo.decode_stage_step(c_lo, input_ids=tok)
            assert hid.shape == (1, 1, dims.hidden_size) and hid.dtype == torch.float16
            c2 = hi.decode_stage_step(c_hi, hidden_in=hid.clone())
            assert torch.equal(a, b), (use_graph, i)
            assert torch.equal(a, c2), (use_graph, i)
            tok = a[0, -1].argmax().view(1, 1)
        assert c_lo.current_seq_len == c_hi.current_seq_len == c_whole.current_seq_len == ids.shape[1] + n_new
        for l in range(L):                                            # and the caches the stages wrote are the whole model's
            src = c_lo if l < 2 else c_hi
            assert torch.equal(src.key_states[l % 2][:, :, :c_whole.current_seq_len], c_whole.key_states[l][:, :, :c_whole.current_seq_len])
    with pytest.raises(RuntimeError):
        staged.generate_greedy(tok, c_staged, 1)                      # device-side generation needs one stage
    c_whole.roll_left()                                               # in place: the executor keeps working after a roll
    c_whole.current_seq_len = ids.shape[1]
    assert torch.isfinite(whole.forward(tok, c_whole)).all()
    for m in (whole, staged, lo, hi):
        m.disable_decode_graph()
    whole.free_unmanaged()


def test_layer_split_runner_token_ring_on_one_rank_equals_generate_greedy():
    """pipeline.LayerSplitRunner.generate_greedy with the hand-off captured INSIDE the rank's graphs (StageHop; one rank here: the
    ring is a device-side copy of the argmax into the token buffer, the same torch ops the multi-rank graphs carry next to the
    RCCL calls): the tokens of ExLlama.generate_greedy, whose argmax is the executor's own kernel; eager exchanges give the same."""
    from exllama_amd.model import ExLlamaCache
    from exllama_amd.pipeline import LayerSplitRunner

    class _Solo:
        def get_rank(self): return 0
        def get_world_size(self): return 1
        def broadcast(self, t, src): return None

    model, cache, tensors, dims = _build("tiny_hd128", 128, True, seed=6, max_seq_len=256)
    ids = torch.randint(1, dims.vocab_size, (1, 150), generator=torch.Generator().manual_seed(2)).to("cuda:0")   # crosses the 160-key bucket
    n = 20
    lg = model.forward(ids, cache)
    first = lg[0, -1].argmax().view(1, 1)
    model.enable_decode_graph(cache)
    want = model.generate_greedy(first, cache, n).tolist()
    for capture in (True, False):
        cache.current_seq_len = 150
        runner = LayerSplitRunner(model, cache, _Solo(), dims.hidden_size, "cuda:0")
        runner.enable_decode_executor(use_graph=True, capture_hop=capture, token_ring=True)
        assert bool(model._decoder["hop_captured"]) == capture
        got = runner.generate_greedy(first, n)
        assert got.tolist() == want, (capture, got.tolist(), want)
        assert cache.current_seq_len == 150 + n
    model.free_unmanaged()


@pytest.mark.parametrize("name,gs", [("tiny_hd128", 128), ("tiny_gqa", 64)])
def test_act_order_down_proj_folded_at_load_changes_no_bit(name, gs):
    """exllama_amd.model._fold_act_order_down_proj: an act-order down_proj's row permutation is moved into the COLUMN order of its
    producers (gate_proj, up_proj) when the model is loaded, and down_proj becomes an ordinary matrix: the reference's column_remap
    of the intermediate activations (q4_matmul.cu:320-325; the largest gather of a layer) and the decode executor's permuted store
    disappear.  Same weights, same activations, same summation order: prompt logits (short-prompt kernels and the fused long-prompt
    launches), K / V rows and decode steps (op path and executor) must be IDENTICAL to the model loaded with the fold switched off."""
    from exllama_amd.model import ExLlama, ExLlamaCache, ExLlamaConfig
    dims = synth.PRESETS[name]
    tensors = synth.make_checkpoint(dims, groupsize=gs, act_order="gptq", seed=41, device="cpu", zeros="rand")
    runs = {}
    for fold in (True, False):
        cfg = ExLlamaConfig(synth.config_dict(dims))
        cfg.max_seq_len = 700
        cfg.max_input_len = 640
        cfg.fold_act_order_mlp = fold
        model = ExLlama(cfg, tensors={k: v.clone() for k, v in tensors.items()})
        assert all((l.mlp.fold_map is not None) == fold for l in model.layers)
        assert all((l.mlp.down_proj.g_idx is None) == fold for l in model.layers)
        out = {}
        for S in (40, 600):                                           # <= 256 rows: skinny GEMM; > 512: the fused prompt launches
            ids = torch.randint(1, dims.vocab_size, (1, S), generator=torch.Generator().manual_seed(S)).to("cuda:0")
            cache = ExLlamaCache(model)
            out[f"prefill{S}"] = model.forward(ids, cache, last_id_only=False).cpu()
            out[f"k{S}"] = cache.key_states[-1][:, :, :S].cpu()
            tok = out[f"prefill{S}"][0, -1].argmax().view(1, 1).to("cuda:0")
            out[f"ops{S}"] = model.forward(tok, cache).cpu()              # q4_attn / q4_mlp op path
            cache.current_seq_len = S
            if dims.head_dim == 128:
                model.enable_decode_graph(cache)
                out[f"exec{S}"] = torch.stack([model.forward(tok, cache).cpu() for _ in range(3)])
                model.disable_decode_graph()
        runs[fold] = out
        model.free_unmanaged()
    for k, v in runs[True].items():
        assert torch.isfinite(v.float()).all(), k
        assert torch.equal(v, runs[False][k]), k


def _unfolded_adapter(model, lora):
    """lora.tensors as the oracle wants them: the halves exllama_amd.lora put into the order of a folded act-order down_proj, back in
    the checkpoint's order."""
    plain = {k: v.cpu() for k, v in lora.tensors.items()}
    for i, layer in enumerate(model.layers):
        fold = layer.mlp.fold_map
        if fold is None:
            continue
        inv = torch.empty_like(fold)
        inv[fold] = torch.arange(fold.numel())
        for proj in ("gate_proj", "up_proj"):
            k = f"model.layers.{i}.mlp.{proj}.lora_B.weight"
            if k in plain:
                plain[k] = plain[k][:, inv].contiguous()
        k = f"model.layers.{i}.mlp.down_proj.lora_A.weight"
        if k in plain:
            plain[k] = plain[k][inv, :].contiguous()
    return plain


@pytest.mark.parametrize("act,r", [(False, 16), ("gptq", 12), (False, 64)])
def test_lora_adapter_inside_the_decode_executor(act, r):
    """exl_decoder_set_lora: the adapter products out = W x + (x A) B ride INSIDE the token step (dec_lora_down_kernel / dec_lora_up_kernel
    behind the GEMV launches; gate / up go out un-fused, the adapter kernel applies SiLU * mul), so a model with an adapter keeps the
    graph path -- the reference passes the same operands to q4_attn / q4_attn_2 / q4_mlp per token (model.py:254-289).  Against the op
    path with the same adapter (two HIP paths) and the oracle model with it; greedy generation in the graph equals the host loop;
    one projection without an adapter, a rank that is no multiple of 8, rank 64, a folded act-order down_proj (its adapter halves
    follow the fold); an act-order o_proj is refused (the op path stays)."""
    from exllama_amd.lora import ExLlamaLora
    from exllama_amd.model import ExLlamaCache
    name = "tiny_hd128"
    dims = synth.PRESETS[name]
    tensors = synth.make_checkpoint(dims, groupsize=128, act_order=act, seed=51, device="cpu", zeros="rand")
    if act:                                                          # an act-order o_proj is what the executor does NOT take with an adapter:
        for i in range(dims.num_hidden_layers):                      # this model's o_proj carries no map (q / k / v / gate / up / down do)
            del tensors[f"model.layers.{i}.self_attn.o_proj.g_idx"]
    from exllama_amd.model import ExLlama, ExLlamaConfig
    cfg = ExLlamaConfig(synth.config_dict(dims))
    cfg.max_seq_len = 256
    model = ExLlama(cfg, tensors={k: v.clone() for k, v in tensors.items()})
    g = torch.Generator().manual_seed(r)
    kvd = dims.num_key_value_heads * dims.head_dim
    shapes = {"self_attn.q_proj": (dims.hidden_size, dims.hidden_size), "self_attn.k_proj": (dims.hidden_size, kvd),
              "self_attn.v_proj": (dims.hidden_size, kvd), "self_attn.o_proj": (dims.hidden_size, dims.hidden_size),
              "mlp.gate_proj": (dims.hidden_size, dims.intermediate_size), "mlp.up_proj": (dims.hidden_size, dims.intermediate_size),
              "mlp.down_proj": (dims.intermediate_size, dims.hidden_size)}
    sd = {}
    for i in range(dims.num_hidden_layers):
        for key, (fin, fout) in shapes.items():
            if i == 1 and key in ("mlp.up_proj", "self_attn.k_proj"):
                continue                                             # projections without an adapter inside launches that have some
            sd[f"base_model.model.model.layers.{i}.{key}.lora_A.weight"] = torch.randn(r, fin, generator=g) * 0.05 * (16.0 / r) ** 0.5
            sd[f"base_model.model.model.layers.{i}.{key}.lora_B.weight"] = torch.randn(fout, r, generator=g) * 0.05     # (same strength at every rank)
    lora = ExLlamaLora(model, {"r": r, "lora_alpha": 16}, "synthetic.bin", tensors=sd)
    ids = torch.randint(1, dims.vocab_size, (1, 150), generator=torch.Generator().manual_seed(3)).to("cuda:0")   # decode crosses the 160-key bucket
    n = 14
    cache = ExLlamaCache(model)
    lg = model.forward(ids, cache, lora=lora)
    toks = [int(lg[0, -1].argmax())]
    ops = []
    for i in range(n):                                               # op path with the adapter: the token history of every other run
        lg = model.forward(torch.tensor([[toks[-1]]], device="cuda:0"), cache, lora=lora)
        ops.append(lg[0, 0].float().cpu())
        toks.append(int(lg[0, 0].argmax()))
    c0 = ExLlamaCache(model)
    model.forward(ids, c0, lora=lora)
    base = model.forward(torch.tensor([[toks[0]]], device="cuda:0"), c0)                # the same step WITHOUT the adapter
    outs = {}
    for mode in ("eager", "graph"):
        c = ExLlamaCache(model)
        model.forward(ids, c, lora=lora)
        model.enable_decode_graph(c, use_graph=(mode == "graph"), lora=lora)
        outs[mode] = [model.forward(torch.tensor([[toks[i]]], device="cuda:0"), c, lora=lora)[0, 0].float().cpu() for i in range(n)]
        assert c.current_seq_len == 150 + n
        if mode == "graph":
            c.current_seq_len = 150
            got = model.generate_greedy(torch.tensor([[toks[0]]], device="cuda:0"), c, n).tolist()
            scale = float(torch.stack(ops).abs().max())
            margins = [float(o.topk(2).values[0] - o.topk(2).values[1]) for o in ops]
            for i in range(n):                                       # same greedy choice wherever the op path's top-2 margin is clear
                if margins[i] > 4e-2 * scale:
                    assert got[i] == toks[i + 1], (i, got, toks)
                else:
                    break
    scale = float(torch.stack(ops).abs().max())
    for i in range(n):
        assert torch.isfinite(outs["eager"][i]).all()
        assert float((outs["graph"][i] - outs["eager"][i]).abs().max()) <= 2e-3 * scale
    # every step of the three paths against the float64 truth WITH the adapter (TruthLlama adds (x A) B in float64), the fp16 oracle
    # with the same adapter as the yardstick: the decode steps of a random model with a strong adapter are often ill-conditioned
    # (round 4, first attempt: executor vs op path 6.7e-3 RMS at one step, both inside the oracle's own spread)
    orc = OracleLlama(synth.config_dict(dims), tensors, max_seq_len=256)
    orc.set_lora(_unfolded_adapter(model, lora))
    for l in range(dims.num_hidden_layers):                           # the steps are compared in isolation: same cached rows as the GPU
        orc.kc[l][0, :, :150] = cache.key_states[l][0, :, :150].cpu().numpy()
        orc.vc[l][0, :, :150] = cache.value_states[l][0, :, :150].cpu().numpy()
    _, runs, truth = _oracle_steps(orc, toks[:n], 150)
    for i in range(n):
        _truth_close(ops[i], runs, truth, i, f"op path with adapter, step {i}")
        _truth_close(outs["eager"][i], runs, truth, i, f"executor (eager) with adapter, step {i}")
        _truth_close(outs["graph"][i], runs, truth, i, f"executor (graph) with adapter, step {i}")
    assert float((ops[0] - base[0, 0].float().cpu()).abs().max()) > 2e-2 * scale       # the adapter is not a no-op
    # without the adapter the logits differ; forward(lora=None) on this executor leaves the graph path (the executor is bound to the adapter)
    model.disable_decode_graph()
    model.free_unmanaged()


def test_executor_refuses_an_adapter_on_an_act_order_o_proj():
    from exllama_amd.lora import ExLlamaLora
    model, cache, tensors, dims = _build("tiny_hd128", 128, True, seed=52, max_seq_len=128)
    sd = {}
    for i in range(dims.num_hidden_layers):
        sd[f"base_model.model.model.layers.{i}.self_attn.o_proj.lora_A.weight"] = torch.randn(8, dims.hidden_size) * 0.05
        sd[f"base_model.model.model.layers.{i}.self_attn.o_proj.lora_B.weight"] = torch.randn(dims.hidden_size, 8) * 0.05
    lora = ExLlamaLora(model, {"r": 8, "lora_alpha": 16}, "synthetic.bin", tensors=sd)
    with pytest.raises(RuntimeError, match="act-order"):
        model.enable_decode_graph(cache, lora=lora)
    assert any("act-order o_proj" in w for w in model.executor_obstacles(lora=lora)) and model.executor_obstacles() == []
    rep = model.decode_path_report(cache, lora=lora)                   # ... and the report names the tier that serves instead, and why
    assert rep["tier"] == "ops_fused" and any("act-order o_proj" in w for w in rep["why_not_faster"]), rep
    ids = torch.randint(1, dims.vocab_size, (1, 9)).to("cuda:0")
    model.disable_decode_graph()
    assert torch.isfinite(model.forward(ids, cache, lora=lora)).all()       # the op path takes it
    model.enable_decode_graph(cache)                                   # without the adapter the executor takes the model
    rep = model.decode_path_report(cache, lora=lora)
    assert rep["tier"] == "ops_fused" and any("act-order o_proj" in w for w in rep["why_not_faster"]), rep
    assert model.decode_path_report(cache)["tier"] == "executor_graph"
    model.free_unmanaged()


# ---- which decode tier ran (ExLlama.decode_path_report): one test per tier and per fallback, each with oracle parity ---------
def test_decode_path_report_follows_the_executor_on_and_off():
    """forward() picks its single-token path silently (graph replay / eager executor / fused ops): the report names the tier a
    forward would take now, why a faster one is not in use, and counts the forwards each tier served."""
    from exllama_amd.model import ExLlamaCache
    model, cache, tensors, dims = _build("tiny_hd128", 128, False, seed=9, max_seq_len=96)
    ids = torch.randint(1, dims.vocab_size, (1, 12), generator=torch.Generator().manual_seed(3)).to("cuda:0")
    tok = torch.tensor([[5]], device="cuda:0")
    assert model.executor_obstacles() == []
    rep = model.decode_path_report(cache)
    assert rep["tier"] == "ops_fused" and rep["why_not_faster"] == ["enable_decode_graph(cache) has not been called"], rep
    model.forward(ids, cache)                                          # a prompt is not a decode step: not counted
    assert model.decode_path_report(cache)["forwards_by_tier"] == {}
    model.forward(tok, cache)
    model.enable_decode_graph(cache)
    rep = model.decode_path_report(cache)
    assert rep["tier"] == "executor_graph" and rep["why_not_faster"] == [] and rep["executor_stages"] == 1, rep
    model.forward(tok, cache); model.forward(tok, cache)
    other = ExLlamaCache(model)
    assert model.decode_path_report(other)["why_not_faster"] == ["the executor was enabled for another cache"]
    model.forward(ids, other); model.forward(tok, other)               # ... and really runs op by op
    model.enable_decode_graph(cache, use_graph=False)
    assert model.decode_path_report(cache)["tier"] == "executor_eager"
    model.forward(tok, cache)
    rep = model.decode_path_report(cache)
    assert rep["forwards_by_tier"] == {"ops_fused": 2, "executor_graph": 2, "executor_eager": 1} and rep["last_forward_tier"] == "executor_eager", rep
    model.reset_decode_path_counts()
    assert model.decode_path_report(cache)["forwards_by_tier"] == {}
    assert set(model.DECODE_TIERS) >= {rep["tier"], "ops_general", "executor_pieces_tp"}
    model.free_unmanaged()


@pytest.mark.parametrize("dims_args,gs,bsz,expect", [
    ((256, 512, 1, 2), 128, 1, None),                                  # head_dim 128, everything a multiple of 128: taken
    ((256, 320, 1, 2), 64, 1, "multiple of 128"),                      # intermediate size 320
    ((8320, 128, 1, 65), 128, 1, "hidden > 8192"),                     # 65 heads of 128
    ((256, 33024, 1, 2), 128, 1, "intermediate > 32768"),
    ((256, 512, 1, 4), 64, 1, "head_dim 64"),
    ((256, 512, 1, 2), 128, 2, "batch size 2"),
])
def test_executor_obstacles_agree_with_what_the_executor_refuses(dims_args, gs, bsz, expect):
    """executor_obstacles() restates in Python conditions that live in csrc/decode_fused.hip (exl_decoder_create) and in
    enable_decode_graph: one case per listed condition, held against the REAL refusal -- enable_decode_graph raises exactly when an
    obstacle is reported, and the report names the condition."""
    from exllama_amd.model import ExLlama, ExLlamaCache, ExLlamaConfig
    dims = synth.LlamaDims(*dims_args, vocab_size=256)
    tensors = synth.make_checkpoint(dims, groupsize=gs, act_order=False, seed=5, device="cpu", zeros="rand")
    cfg = ExLlamaConfig(synth.config_dict(dims))
    cfg.max_seq_len = cfg.max_input_len = 32
    model = ExLlama(cfg, tensors=tensors)
    cache = ExLlamaCache(model, batch_size=bsz)
    why = model.executor_obstacles(batch_size=bsz)
    refused = False
    try:
        model.enable_decode_graph(cache, use_graph=False)
    except RuntimeError:
        refused = True
    assert refused == bool(why), (why, refused)
    if expect is None:
        assert why == []
    else:
        assert any(expect in w for w in why), why
    model.free_unmanaged()


def test_head_dim_100_falls_to_the_fused_ops_and_says_so():
    """OpenLLaMA-3B's geometry (hidden 3200, 32 heads of 100, intermediate 8640: the reference benchmarks it, README.md:33-41): the
    executor's kernels are written for head_dim 128, so enable_decode_graph refuses and decode runs the reference's op sequence
    (q4_attn -> attention -> q4_attn_2 -> q4_mlp).  The report says which tier and why; the tier is held to the oracle."""
    from exllama_amd.model import ExLlama, ExLlamaCache, ExLlamaConfig
    dims = synth.LlamaDims(3200, 8640, 2, 32, vocab_size=640)
    assert dims.head_dim == 100
    tensors = synth.make_checkpoint(dims, groupsize=64, act_order=False, seed=13, device="cpu", zeros="rand")
    cfg = ExLlamaConfig(synth.config_dict(dims))
    cfg.max_seq_len = cfg.max_input_len = 64
    model = ExLlama(cfg, tensors={k: v.clone() for k, v in tensors.items()})
    cache = ExLlamaCache(model)
    why = model.executor_obstacles()
    assert any("head_dim 100" in w for w in why) and any("multiple of 128" in w for w in why), why
    with pytest.raises(RuntimeError):
        model.enable_decode_graph(cache)
    rep = model.decode_path_report(cache)
    assert rep["tier"] == "ops_fused" and rep["why_not_faster"] == why and not rep["executor_enabled"], rep
    S, n = 23, 5
    ids = torch.randint(1, dims.vocab_size, (1, S), generator=torch.Generator().manual_seed(8))
    lg = model.forward(ids.to("cuda:0"), cache, last_id_only=False).float().cpu().numpy()
    ref = OracleLlama(synth.config_dict(dims), tensors, max_seq_len=64)
    _model_close(lg, np.asarray(ref.forward(ids.numpy(), last_id_only=False), dtype=np.float32), ORACLE_TOL, "hd 100 prompt vs oracle")
    toks, got = [int(lg[0, -1].argmax())], []
    for i in range(n):
        step = model.forward(torch.tensor([[toks[-1]]], device="cuda:0"), cache).float().cpu().numpy()[0, 0]
        got.append(step)
        toks.append(int(step.argmax()))
    assert model.decode_path_report(cache)["forwards_by_tier"] == {"ops_fused": n}
    for l in range(dims.num_hidden_layers):                           # the steps in isolation, from the rows the GPU's prompt pass wrote
        ref.kc[l][0, :, :S] = cache.key_states[l][0, :, :S].cpu().numpy()
        ref.vc[l][0, :, :S] = cache.value_states[l][0, :, :S].cpu().numpy()
    _, runs, truth = _oracle_steps(ref, toks[:n], S)
    for i in range(n):
        _truth_close(got[i], runs, truth, i, f"hd 100 fused ops vs truth, step {i}")
    model.free_unmanaged()


def test_batched_decode_runs_the_general_ops_and_says_so():
    """Batch 2, one token per row (what the reference's generator does with a batched cache, generator.py:344-381 on
    ExLlamaCache(batch_size=2)): the executor handles batch 1, the fused ops need rows == 1; since round 6 the two rows take the
    one-call-per-layer path of short prompts (tier `layer_call`), and with EXL_GEMM_NO_FRAG the general op sequence.  Reported as
    what ran, and equal to the oracle's batched forward."""
    from exllama_amd.model import ExLlamaCache
    model, cache1, tensors, dims = _build("tiny_hd128", 128, "gptq", seed=15, max_seq_len=96)
    model.enable_decode_graph(cache1)                                 # an executor on ANOTHER cache does not catch the batched one
    cache = ExLlamaCache(model, batch_size=2)
    rep = model.decode_path_report(cache)
    want_tier = "ops_general" if os.environ.get("EXL_GEMM_NO_FRAG") else "layer_call"
    assert rep["tier"] == want_tier and any("batch size 2" in w for w in rep["why_not_faster"]), rep
    with pytest.raises(RuntimeError):
        model.enable_decode_graph(cache)
    model.enable_decode_graph(cache1)
    S, n = 19, 4
    ids = torch.randint(1, dims.vocab_size, (2, S), generator=torch.Generator().manual_seed(6))
    lg = model.forward(ids.to("cuda:0"), cache, last_id_only=False).float().cpu().numpy()
    ref = OracleLlama(synth.config_dict(dims), tensors, max_seq_len=96)
    ref.reset(bsz=2)
    _model_close(lg, np.asarray(ref.forward(ids.numpy(), last_id_only=False), dtype=np.float32), ORACLE_TOL, "batch 2 prompt vs oracle")
    tok = lg[:, -1].argmax(-1).reshape(2, 1)
    for i in range(n):
        step = model.forward(torch.from_numpy(tok).to("cuda:0"), cache).float().cpu().numpy()
        want = np.asarray(ref.forward(tok), dtype=np.float32)
        _model_close(step, want, PATHS_TOL, f"batch 2 decode step {i} vs oracle")
        tok = want[:, -1].argmax(-1).reshape(2, 1)                    # the oracle's choice drives both
    assert model.decode_path_report(cache)["forwards_by_tier"] == {want_tier: n}
    assert cache.current_seq_len == S + n
    model.free_unmanaged()


@pytest.mark.parametrize("name,layers,act", [("7b", 32, False), ("13b", 40, "gptq")], ids=["7b_g128", "13b_g128_actorder"])
def test_perplexity_full_depth(golden_dir, name, layers, act):
    """north_star's accuracy bar at the depth it is stated for: ALL layers of BASELINE configs[1] (7B g128, 32 layers) and configs[2]
    (13B g128 act-order, 40 layers), 1535 scored tokens of the model's own sampled text, HIP whole-chunk path vs the CPU oracle over
    the same layers (reference: perplexity.py:92-138; README.md:139-148 quotes two decimals).  Asserted the way it is stated: the
    two numbers PRINT the same to 2 dp, and |delta| < 0.005.  Should the oracle's value sit within |delta| of a x.xx5 rounding
    boundary, the strings can differ although the values agree to 3 dp: that case is reported with both values (warning + stats
    record) and held to |delta| < 0.005 and |delta| < 5 % of the standard error of the perplexity estimate itself (what 1535 tokens
    can resolve) -- the bound is not widened.  The oracle's prompt pass over a whole model is 5-12 minutes of host time (8 cores: 45),
    so the TEXT and the oracle's per-token log-likelihoods of it are a committed fixture (tests/golden/ppl_full_depth_<model>.npz, made
    by oracle/make_ppl_full_depth_golden.py from a text this checkpoint sampled on the HIP decode path): the test scores that text.
    (Until round 6 the test sampled the text anew and used the fixture only if it came out identical, id for id -- which any change
    of a rounding point in the 4-token prompt pass that seeds the sampling undoes: the round-6 short-prompt path did, and the suite
    ran both oracles for 12 minutes.)  EXL_PPL_ORACLE=1 -- or a missing fixture -- samples a fresh text and runs the oracle here.
    More texts: scripts/ppl_full_depth.py -> profiles/rNN_model_tolerance_stats.jsonl.  EXL_SKIP_SLOW=1 skips the cases."""
    if os.environ.get("EXL_SKIP_SLOW"):
        pytest.skip("EXL_SKIP_SLOW set")
    import warnings
    from parity import perplexity_hip, perplexity_oracle, perplexity_record
    dims = synth.PRESETS[name]
    assert dims.num_hidden_layers == layers
    gname = "ppl_full_depth_%s%s.npz" % (name, "_act" if act else "")
    gpath = os.path.join(golden_dir, gname)
    g = np.load(gpath) if os.path.exists(gpath) and not os.environ.get("EXL_PPL_ORACLE") else None
    if g is not None:
        assert g["meta"].tolist() == [layers, 128, 17, 23, 4600], g["meta"]      # (layers, group size, seed, checkpoint seed, 1000 x head scale)
        hip, ids = perplexity_hip(dims, layers, 128, act, tokens=1536, seed=17, text=g["ids"])
        assert np.array_equal(g["ids"], ids.numpy()) and hip["ckpt_seed"] == 23 and hip["head_scale"] == 4.6
        rec = perplexity_record(hip, g["oracle_nll"])
        rec["oracle_source"] = "tests/golden/%s (text and oracle log-likelihoods)" % gname
    else:
        hip, ids = perplexity_hip(dims, layers, 128, act, tokens=1536, seed=17)
        rec = perplexity_oracle(hip, ids, dims)
        rec["oracle_source"] = "a fresh text, oracle run in this test"
    stats = os.environ.get("EXL_TOL_STATS")
    if stats:
        with open(stats, "a") as f:
            f.write(json.dumps(rec) + "\n")
    whole, token, ref = rec["values"]
    assert rec["layers"] == layers and rec["tokens"] == 1535
    assert 2.5 < ref < 12.0, rec                                      # the README's range (5.68 .. 3.53) or next to it (13B act-order text: 2.75)
    assert abs(whole - ref) < 0.005, rec
    if not rec["equal_to_2dp"]:
        assert rec["oracle_distance_to_rounding_boundary"] <= abs(whole - ref), rec        # only a straddled boundary can do this
        assert rec["delta_whole_over_standard_error"] < 0.05, rec
        warnings.warn(f"perplexity straddles a rounding boundary: HIP {whole:.4f} vs oracle {ref:.4f} (standard error {rec['oracle_standard_error']:.3f})")
    else:
        assert f"{whole:.2f}" == f"{ref:.2f}"
    # token by token (op path; the text was sampled on the executor's graph, a third path): bounded as in the 2-layer test -- the KL term of
    # scoring a text another path sampled is second order in the logit noise (measured at full depth: 8.520 vs 8.509 whole, 13B 3.6598 vs 3.6596)
    assert abs(token - ref) < 0.02, rec
