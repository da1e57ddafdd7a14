"""CPU tests of the oracle itself: golden fixtures (regression pins), the reference's own rep_penalty.cpp,
and algebraic self-consistency of the restated kernels."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

from exllama_amd import synth
from oracle import exl_oracle as O
from oracle.model_oracle import OracleLlama

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def ops(golden_dir):
    return np.load(os.path.join(golden_dir, "ops_small.npz"))


@pytest.mark.parametrize("tag,act", [("a", True), ("b", False), ("c", True)])
def test_q4_golden(ops, tag, act):
    qw, qz, sc, x = (ops[f"q4{tag}_{k}"] for k in ("qweight", "qzeros", "scales", "x"))
    x_map, qws = None, qw
    if act:
        x_map, qws = O.make_sequential(qw, ops[f"q4{tag}_g_idx"], qz.shape[0])
        assert np.array_equal(x_map, ops[f"q4{tag}_x_map"])                      # integer work: bit-exact
        assert np.array_equal(qws, ops[f"q4{tag}_qweight_seq"])
    assert np.array_equal(O.dequant_w16(qws, qz, sc).view(np.uint16), ops[f"q4{tag}_w16"].view(np.uint16))
    assert np.array_equal(O.q4_matmul_recons(x, qws, qz, sc, x_map).view(np.uint16), ops[f"q4{tag}_out_recons"].view(np.uint16))
    assert np.array_equal(O.q4_matmul_gemv_f32(x[:3], qws, qz, sc, x_map).view(np.uint16), ops[f"q4{tag}_out_gemv"].view(np.uint16))


def test_pack_unpack_roundtrip():
    rs = np.random.RandomState(0)
    q = rs.randint(0, 16, size=(64, 24)).astype(np.uint8)
    assert np.array_equal(O.unpack_qweight(O.pack_qweight(q)), q)
    z = rs.randint(0, 16, size=(4, 24))
    assert np.array_equal(O.unpack_qzeros(O.pack_qzeros(z)), z + 1)              # the +1 of GPTQ v1 storage


def test_act_order_is_a_pure_permutation():
    """x' @ W_seq == x @ W for the dequantised weights (make_sequential + column_remap)."""
    gen = torch.Generator().manual_seed(5)
    K, N, gs = 384, 64, 128
    lin = synth.make_q4_linear(K, N, gs, True, gen, "cpu", zeros="rand", std=0.05)
    qw, qz, sc, gi = (lin[k].numpy() for k in ("qweight", "qzeros", "scales", "g_idx"))
    x = torch.randn(4, K, generator=gen).half().numpy()
    x_map, qws = O.make_sequential(qw, gi, qz.shape[0])
    assert sorted(x_map.tolist()) == list(range(K))
    assert np.array_equal(gi[x_map.astype(np.int64)], np.repeat(np.arange(K // gs), gs))     # groups now contiguous
    # direct evaluation with per-row group index on the ORIGINAL row order
    q = O.unpack_qweight(qw).astype(np.int32)
    z = O.unpack_qzeros(qz)
    W = ((q - z[gi]).astype(np.float16).astype(np.float32) * sc.astype(np.float32)[gi]).astype(np.float16)
    ref = x.astype(np.float64) @ W.astype(np.float64)
    got = O.column_remap(x, x_map).astype(np.float64) @ O.dequant_w16(qws, qz, sc).astype(np.float64)
    np.testing.assert_allclose(got, ref, rtol=0, atol=1e-9)


def test_gemv_forms_agree_within_fp16_tolerance(ops):
    """fp32-accumulated target vs dequant-then-GEMM vs the reference's fp16-accumulating arithmetic."""
    for tag in "abc":
        a = ops[f"q4{tag}_out_gemv"][:1].astype(np.float32)
        b = ops[f"q4{tag}_out_recons"][:1].astype(np.float32)
        e = ops[f"q4{tag}_out_f16emu"].astype(np.float32)
        scale = np.abs(b).max()
        assert np.abs(a - b).max() <= 4e-3 * scale
        assert np.abs(e - b).max() <= 2e-2 * scale      # the reference's own fp16 accumulation is this far off


def test_elementwise_golden(ops):
    assert np.array_equal(O.rms_norm(ops["rms_x"], ops["rms_w"], 1e-6).view(np.uint16), ops["rms_out"].view(np.uint16))
    assert np.array_equal(O.rope(ops["rope_x"], ops["rope_sin"], ops["rope_cos"], 5, 4, 32).view(np.uint16), ops["rope_out"].view(np.uint16))
    with np.errstate(over="ignore"):
        assert np.array_equal(O.silu_mul(ops["silu_x"], ops["silu_y"]).view(np.uint16), ops["silu_out"].view(np.uint16))
    got = O.attention(ops["att_q"], ops["att_k"], ops["att_v"], causal_past_len=8)
    assert np.array_equal(got.view(np.uint16), ops["att_out"].view(np.uint16))


def test_rms_norm_matches_plain_formula():
    rs = np.random.RandomState(1)
    x = rs.randn(3, 256).astype(np.float16)
    w = (1 + 0.1 * rs.randn(256)).astype(np.float16)
    ref = x.astype(np.float64) / np.sqrt((x.astype(np.float64) ** 2).mean(-1, keepdims=True) + 1e-6) * w.astype(np.float64)
    np.testing.assert_allclose(O.rms_norm(x, w, 1e-6).astype(np.float64), ref, rtol=3e-3, atol=3e-3)


def test_rope_is_a_rotation():
    sin, cos = O.rope_tables(32, 16)
    x = np.random.RandomState(2).randn(1, 2 * 16).astype(np.float16)          # 1 token, 2 heads
    y = O.rope(x, sin, cos, 7, 2, 16).astype(np.float64).reshape(2, 16)
    xx = x.astype(np.float64).reshape(2, 16)
    np.testing.assert_allclose((y ** 2).sum(-1), (xx ** 2).sum(-1), rtol=5e-3)     # norm preserved
    assert np.array_equal(O.rope(x, sin, cos, 0, 2, 16).view(np.uint16), x.view(np.uint16))   # position 0 = identity


def test_update_cache_scatter():
    k = np.arange(1, 25, dtype=np.float16).reshape(1, 2, 12)          # bsz 1, q_len 2, kv_heads 3 * hd 4
    v = -k
    kc = np.zeros((1, 3, 10, 4), dtype=np.float16)
    vc = np.zeros_like(kc)
    O.update_cache(k, v, kc, vc, past_len=5)
    assert np.array_equal(kc[0, :, 5, :].reshape(-1), k[0, 0])
    assert np.array_equal(vc[0, :, 6, :].reshape(-1), v[0, 1])
    assert not kc[0, :, :5].any() and not kc[0, :, 7:].any()


def test_rep_penalty_against_reference_binary(golden_dir):
    """Golden vectors were produced by the reference's rep_penalty.cpp (oracle/_ref); the numpy restatement and, when
    present, the freshly built reference binary must reproduce them bit for bit."""
    g = np.load(os.path.join(golden_dir, "rep_penalty.npz"))
    ref_path = os.path.join(ROOT, "oracle", "_ref", "librep_penalty_ref.so")
    ref = C.CDLL(ref_path) if os.path.exists(ref_path) else None
    for n in range(int(g["ncases"])):
        vocab, seq_len, sustain, decay = (int(v) for v in g[f"c{n}_params"])
        pmax = float(g[f"c{n}_pmax"])
        seq = g[f"c{n}_seq"]
        assert np.array_equal(O.rep_penalty_mask(vocab, seq, pmax, sustain, decay).view(np.uint32), g[f"c{n}_mask"].view(np.uint32))
        lg = g[f"c{n}_logits"].copy()[None]
        O.apply_rep_penalty(seq[None], pmax, sustain, decay, lg)
        assert np.array_equal(lg[0].view(np.uint32), g[f"c{n}_applied"].view(np.uint32))
        if ref is not None:
            s = np.ascontiguousarray(seq.astype(np.uint64) if seq_len else np.zeros(1, dtype=np.uint64))
            m = np.zeros(vocab, dtype=np.float32)
            ref._Z15rep_penalty_cpuiPKmPffiii(vocab, s.ctypes.data_as(C.c_void_p), m.ctypes.data_as(C.c_void_p), C.c_float(pmax), sustain, decay, seq_len)
            assert np.array_equal(m.view(np.uint32), g[f"c{n}_mask"].view(np.uint32))


def test_tiny_model_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "tiny_model.npz"))
    dims = synth.PRESETS["tiny"]
    tensors = synth.make_checkpoint(dims, groupsize=64, act_order=False, seed=11, device="cpu", zeros="rand")
    m = OracleLlama(synth.config_dict(dims), tensors, max_seq_len=64)
    logits = m.forward(g["tiny_ids"], last_id_only=False)
    assert np.isfinite(logits).all()
    np.testing.assert_array_equal(logits.astype(np.float16).view(np.uint16), g["tiny_logits"].view(np.uint16))
    # prefill-all vs token-by-token must agree within fp16 tolerance (same algebra, different matmul shapes)
    m2 = OracleLlama(synth.config_dict(dims), tensors, max_seq_len=64)
    step = np.concatenate([m2.forward(g["tiny_ids"][:, i:i + 1]) for i in range(g["tiny_ids"].shape[1])], axis=1)
    np.testing.assert_allclose(step, logits, rtol=0, atol=2e-2 * np.abs(logits).max())


def test_oracle_model_lora_path():
    """OracleLlama.set_lora (the checker of tests/test_model_gpu.py::test_lora_adapter_end_to_end): an all-zero B leaves the
    logits unchanged bit for bit, a rank-1 adapter on one projection changes them, and OracleLinear.with_lora follows the
    reference's order (exllama_ext.cpp:245-324): adapter product first (two fp16-rounded GEMMs), quantised product onto it."""
    import torch
    from exllama_amd import synth
    from oracle.model_oracle import OracleLlama
    dims = synth.PRESETS["tiny"]
    tensors = synth.make_checkpoint(dims, groupsize=64, act_order=False, seed=9, device="cpu")
    o = OracleLlama(synth.config_dict(dims), tensors, max_seq_len=32)
    ids = np.array([[3, 5, 7, 9, 11]])
    base = o.forward(ids, last_id_only=False)
    g = torch.Generator().manual_seed(1)
    key = "model.layers.0.self_attn.q_proj"
    a = (torch.randn(dims.hidden_size, 4, generator=g) * 0.05).half()
    zero_b = torch.zeros(4, dims.hidden_size).half()
    o.reset(); o.set_lora({key + ".lora_A.weight": a, key + ".lora_B.weight": zero_b})
    assert np.array_equal(o.forward(ids, last_id_only=False), base)
    b = (torch.randn(4, dims.hidden_size, generator=g) * 0.05).half()
    o.reset(); o.set_lora({key + ".lora_A.weight": a, key + ".lora_B.weight": b})
    assert np.abs(o.forward(ids, last_id_only=False) - base).max() > 1e-3
    o.reset(); o.set_lora(None)
    assert np.array_equal(o.forward(ids, last_id_only=False), base)
    lin = o.layers[0]["q"]
    x = (torch.randn(3, dims.hidden_size, generator=g)).half().numpy()
    t = (x.astype(np.float32) @ a.numpy().astype(np.float32)).astype(np.float16)
    d = (t.astype(np.float32) @ b.numpy().astype(np.float32)).astype(np.float16)
    assert np.array_equal(lin.with_lora(x, a.numpy(), b.numpy()), lin(x, residual=d))


def test_truth_model_and_the_decode_step_criterion():
    """oracle.model_oracle.TruthLlama (float64, no fp16 rounding) is what the fp16 oracle approximates, and tests/parity.py's
    decode-step criterion |x - truth| <= 1.5 |oracle - truth| + 1 ulp accepts a second correct fp16 implementation (the oracle
    on prepared weights: fp32 BLAS instead of the per-group reconstruction order) and refuses small defects: one 16-column
    block of the logits off by 4e-3 x scale, or a K row of the cache rotated with the wrong position."""
    from oracle.model_oracle import TruthLlama
    from parity import _oracle_steps, _truth_close
    dims = synth.PRESETS["tiny_gqa"]
    tensors = synth.make_checkpoint(dims, groupsize=128, act_order=True, seed=5, device="cpu", zeros="rand")
    cfg = synth.config_dict(dims)
    ref = OracleLlama(cfg, tensors, max_seq_len=64)
    ids = np.random.RandomState(0).randint(1, dims.vocab_size, size=(1, 20))
    full = ref.forward(ids, last_id_only=False)
    truth_full = TruthLlama(cfg, tensors, max_seq_len=64).forward(ids, last_id_only=False)
    scale = float(np.abs(truth_full).max())
    assert np.abs(full - truth_full).max() <= 3e-3 * scale            # the fp16 pipeline sits a few logit ulps from the truth
    toks = [5, 9, 200]
    other = OracleLlama(cfg, tensors, max_seq_len=64)                 # the "implementation under test": same cache, fp32 BLAS matmuls
    other.prepare()
    other.kc, other.vc = [a.copy() for a in ref.kc], [a.copy() for a in ref.vc]
    clean, runs, truth = _oracle_steps(ref, toks, 20)
    assert len(runs) == 4 and runs[0] is clean
    other.past = 20
    got = [other.forward(np.array([[t]]))[0, 0] for t in toks]
    for i in range(len(toks)):
        _truth_close(got[i], runs, truth, i, f"second implementation, step {i}")
        bad = got[i].copy()
        bad[32:48] += np.float32(4e-3 * scale)
        with pytest.raises(AssertionError):
            _truth_close(bad, runs, truth, i)
    # a defect inside the model: the key of the first generated token cached one position off in RoPE
    broken = OracleLlama(cfg, tensors, max_seq_len=64)
    broken.kc, broken.vc = [a.copy() for a in other.kc], [a.copy() for a in other.vc]
    for l in range(broken.L):
        broken.kc[l][0, :, 20] = broken.kc[l][0, :, 19]
    broken.past = 21
    with pytest.raises(AssertionError):
        _truth_close(broken.forward(np.array([[toks[1]]]))[0, 0], runs, truth, 1)


def test_centered_nibbles_keep_a_deep_synthetic_model_in_the_fp16_range():
    """synth.make_checkpoint(nibbles=...): uniform nibbles 0..15 against the symmetric zero point 8 bias every weight by -0.5 steps; through
    a deep model that bias becomes a common-mode drift of the residual stream, linear in the depth (measured on the GPU at 13B shapes:
    ~1.8e3 per layer, inf at layer 36-37 of 40) -- the benchmark models of rounds 1-4 beyond 7B computed on inf / NaN from there on.
    "centered" (every 0 nibble -> 8: symmetric about the zero point, as the weights GPTQ writes are) keeps it bounded."""
    import torch
    from exllama_amd import synth
    from oracle.model_oracle import OracleLlama
    dims, L = synth.LLAMA_TINY_HD128, 40
    growth = {}
    for nib in ("uniform", "centered"):
        t = synth.make_checkpoint(dims, groupsize=128, seed=3, num_layers=L, nibbles=nib)
        q = t["model.layers.7.mlp.down_proj.qweight"]
        vals = torch.stack([(q >> (4 * j)) & 0xF for j in range(8)]).flatten().float()
        assert abs(float(vals.mean()) - (8.0 if nib == "centered" else 7.5)) < 0.02
        assert (int((vals == 0).sum()) == 0) == (nib == "centered")
        o = OracleLlama(synth.config_dict(dims, L), t, max_seq_len=16)
        h = o.embed[np.array([[5, 9, 11, 200, 17, 3]])]
        peaks = []
        for i in range(L):
            h = o.layer_forward(i, h)
            peaks.append(float(np.abs(h.astype(np.float32)).max()))
        growth[nib] = peaks
    assert growth["uniform"][-1] > 2000 and growth["uniform"][-1] > 1.8 * growth["uniform"][L // 2 - 1]      # linear in the depth
    assert growth["centered"][-1] < 100
    # the default stays "uniform": the committed golden vectors were generated with it
    a = synth.make_checkpoint(synth.LLAMA_TINY, seed=1, num_layers=1)
    b = synth.make_checkpoint(synth.LLAMA_TINY, seed=1, num_layers=1, nibbles="uniform")
    assert all(torch.equal(a[k], b[k]) for k in a)


def test_perplexity_record_states_the_two_decimal_claim():
    """tests/parity.py:perplexity_record -- the bookkeeping behind "perplexity equal to 2 dp": printed strings, |delta|, the standard
    error of the estimate, and the one case in which equal values print differently (a straddled x.xx5 boundary)."""
    import math
    from parity import perplexity_record
    nll = np.full(1535, math.log(8.1102))
    nll[0:1534:2] += 0.3; nll[1:1534:2] -= 0.3                        # spread with zero mean: the perplexity stays 8.1102
    ref = math.exp(float(nll.mean()))
    rec = perplexity_record({"hip_whole": ref + 0.0033, "hip_token": ref + 0.004, "layers": 32}, nll)
    assert rec["tokens"] == 1535 and rec["two_dp"][0] == rec["two_dp"][2] and rec["equal_to_2dp"] and not rec["boundary_straddled"]
    assert abs(rec["delta_whole"] - 0.0033) < 1e-9 and abs(rec["oracle_standard_error"] - ref * float(np.std(nll, ddof=1)) / math.sqrt(1535)) < 1e-9
    # a value 0.002 away from the oracle's, on the other side of 8.115: different strings, flagged as straddled, not as unequal values
    nll2 = np.full(1535, math.log(8.1140))
    rec2 = perplexity_record({"hip_whole": 8.1160, "hip_token": 8.1160, "layers": 32}, nll2)
    assert rec2["two_dp"][0] == "8.12" and rec2["two_dp"][2] == "8.11" and not rec2["equal_to_2dp"] and rec2["boundary_straddled"]
    assert rec2["oracle_distance_to_rounding_boundary"] <= abs(rec2["delta_whole"]) < 0.005
    # ... and a real disagreement is neither
    rec3 = perplexity_record({"hip_whole": 8.20, "hip_token": 8.20, "layers": 32}, nll2)
    assert not rec3["equal_to_2dp"] and not rec3["boundary_straddled"]
