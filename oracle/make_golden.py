"""Generates the committed golden fixtures under tests/golden/ -- run from the repo root:

    make -C oracle && python -m oracle.make_golden

Sources of truth:
  rep_penalty.npz   outputs of the REFERENCE's own rep_penalty.cpp (oracle/_ref/librep_penalty_ref.so, built by
                    oracle/Makefile from /root/reference/exllama_ext/cpu_func/rep_penalty.cpp, unmodified)
  ops_small.npz     outputs of the numpy restatement oracle/exl_oracle.py on seeded inputs (the reference has no
                    golden vectors and no CPU path for these ops: regression pins of the oracle)
  tiny_model.npz    logits of oracle/model_oracle.py on the seeded synthetic "tiny" checkpoints
"""

import ctypes as C
import os

import numpy as np
import torch

from exllama_amd import synth
from oracle import exl_oracle as O
from oracle.model_oracle import OracleLlama

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def rep_penalty_cases():
    lib = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref", "librep_penalty_ref.so"))
    f_mask = lib._Z15rep_penalty_cpuiPKmPffiii
    f_apply = lib._Z21apply_rep_penalty_cpuiPKmfiiiPf
    rs = np.random.RandomState(7)
    out = {}
    cases = [(64, 50, 1.15, 10, 12), (64, 50, 1.3, -1, 0), (64, 5, 1.2, 100, 100), (32, 40, 1.05, 0, 8), (16, 0, 1.2, 4, 4),
             (128, 90, 1.25, 30, 0)]
    for n, (vocab, seq_len, pmax, sustain, decay) in enumerate(cases):
        seq = rs.randint(0, vocab, size=max(seq_len, 1)).astype(np.uint64)[:seq_len]
        mask = np.zeros(vocab, dtype=np.float32)
        seq_c = np.ascontiguousarray(seq if seq_len else np.zeros(1, dtype=np.uint64))
        f_mask(vocab, seq_c.ctypes.data_as(C.c_void_p), mask.ctypes.data_as(C.c_void_p), C.c_float(pmax), sustain, decay, seq_len)
        logits = rs.randn(vocab).astype(np.float32) * 3
        applied = logits.copy()
        f_apply(vocab, seq_c.ctypes.data_as(C.c_void_p), C.c_float(pmax), sustain, decay, seq_len, applied.ctypes.data_as(C.c_void_p))
        out[f"c{n}_params"] = np.array([vocab, seq_len, sustain, decay], dtype=np.int64)
        out[f"c{n}_pmax"] = np.float32(pmax)
        out[f"c{n}_seq"] = seq.astype(np.int64)
        out[f"c{n}_mask"] = mask
        out[f"c{n}_logits"] = logits
        out[f"c{n}_applied"] = applied
    out["ncases"] = np.int64(len(cases))
    np.savez_compressed(os.path.join(OUT, "rep_penalty.npz"), **out)


def ops_cases():
    gen = torch.Generator().manual_seed(1234)
    out = {}
    # q4 linear, act-order, random zeros, two group sizes
    for tag, (K, N, gs, act) in {"a": (256, 128, 64, True), "b": (512, 96, 128, False), "c": (256, 64, 32, True)}.items():
        lin = synth.make_q4_linear(K, N, gs, act, gen, "cpu", zeros="rand", std=0.05)
        qw = lin["qweight"].numpy().view(np.uint32)
        qz = lin["qzeros"].numpy().view(np.uint32)
        sc = lin["scales"].numpy()
        x = torch.randn(5, K, generator=gen).half().numpy()
        res = (torch.randn(5, N, generator=gen) * 0.5).half().numpy()
        out[f"q4{tag}_qweight"], out[f"q4{tag}_qzeros"], out[f"q4{tag}_scales"], out[f"q4{tag}_x"], out[f"q4{tag}_res"] = qw, qz, sc, x, res
        x_map = None
        qws = qw
        if act:
            g_idx = lin["g_idx"].numpy()
            out[f"q4{tag}_g_idx"] = g_idx
            x_map, qws = O.make_sequential(qw, g_idx, qz.shape[0])
            out[f"q4{tag}_x_map"] = x_map
            out[f"q4{tag}_qweight_seq"] = qws
        out[f"q4{tag}_w16"] = O.dequant_w16(qws, qz, sc)
        out[f"q4{tag}_out_recons"] = O.q4_matmul_recons(x, qws, qz, sc, x_map)
        out[f"q4{tag}_out_gemv"] = O.q4_matmul_gemv_f32(x[:3], qws, qz, sc, x_map)
        out[f"q4{tag}_out_gemv_acc"] = O.q4_matmul_gemv_f32(x[:3], qws, qz, sc, x_map, out=res[:3])
        out[f"q4{tag}_out_f16emu"] = O.q4_matmul_gemv_f16emu(x[:1], qws, qz, sc, x_map)
    # rms norm / rope / silu / attention
    x = (torch.randn(6, 320, generator=gen) * 2).half().numpy()
    w = (1 + 0.1 * torch.randn(320, generator=gen)).half().numpy()
    out["rms_x"], out["rms_w"], out["rms_out"] = x, w, O.rms_norm(x, w, 1e-6)
    sin, cos = O.rope_tables(64, 32)
    xr = torch.randn(2, 3 * 4 * 32, generator=gen).half().numpy()      # bsz 2, q_len 3, heads 4, hd 32
    out["rope_x"], out["rope_out"] = xr, O.rope(xr, sin, cos, 5, 4, 32)
    out["rope_sin"], out["rope_cos"] = sin, cos
    a = (torch.randn(4, 64, generator=gen) * 3).half().numpy()
    b = torch.randn(4, 64, generator=gen).half().numpy()
    out["silu_x"], out["silu_y"], out["silu_out"] = a, b, O.silu_mul(a, b)
    q = torch.randn(1, 4, 3, 32, generator=gen).half().numpy()
    k = torch.randn(1, 2, 11, 32, generator=gen).half().numpy()
    v = torch.randn(1, 2, 11, 32, generator=gen).half().numpy()
    out["att_q"], out["att_k"], out["att_v"] = q, k, v
    out["att_out"] = O.attention(q, k, v, causal_past_len=8)
    np.savez_compressed(os.path.join(OUT, "ops_small.npz"), **out)


def tiny_model_cases():
    out = {}
    for name, act, gs in (("tiny", False, 64), ("tiny_gqa", True, 128)):
        dims = synth.PRESETS[name]
        tensors = synth.make_checkpoint(dims, groupsize=gs, act_order=act, seed=11, device="cpu", zeros="rand")
        m = OracleLlama(synth.config_dict(dims), tensors, max_seq_len=64)
        ids = np.random.RandomState(3).randint(1, dims.vocab_size, size=(1, 12))
        logits_all = m.forward(ids, last_id_only=False)
        step = []
        tok = int(np.argmax(logits_all[0, -1]))
        toks = [tok]
        for _ in range(4):
            lg = m.forward(np.array([[tok]]))
            step.append(lg[0, 0])
            tok = int(np.argmax(lg[0, 0]))
            toks.append(tok)
        out[f"{name}_ids"] = ids
        out[f"{name}_logits"] = logits_all.astype(np.float16)
        out[f"{name}_step_logits"] = np.stack(step).astype(np.float16)
        out[f"{name}_tokens"] = np.array(toks, dtype=np.int64)
    np.savez_compressed(os.path.join(OUT, "tiny_model.npz"), **out)


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    rep_penalty_cases()
    ops_cases()
    tiny_model_cases()
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))
