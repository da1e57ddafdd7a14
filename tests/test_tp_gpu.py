"""Tensor parallelism (exllama_amd/tp.py): W rank models built from the shards of ONE checkpoint must
reproduce the unsharded model -- prompt pass (op-by-op path with the collectives) and single-token decode (native executor in
pieces, exl_decoder_step_part, residual stream all-reduced after every half layer).  The ranks are threads on this one GPU
(tests/tp_emul.py); the collectives over RCCL are the same two calls on torch.distributed."""

import numpy as np
import pytest
import torch

from exllama_amd import synth, tp
from exllama_amd.model import ExLlama, ExLlamaCache, ExLlamaConfig

from oracle.model_oracle import OracleLlama
from parity import TP_TOL, _model_close
from tp_emul import LocalGroup

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _build(cfg_dict, tensors, tpobj=None, max_seq=256):
    cfg = ExLlamaConfig(dict(cfg_dict))
    cfg.max_seq_len = max_seq
    cfg.max_input_len = max_seq
    cfg.tp = tpobj
    model = ExLlama(cfg, tensors={k: v.clone() for k, v in tensors.items()})
    return model, ExLlamaCache(model)


CASES = [
    # dims preset, layers, groupsize, act_order, world
    ("tiny_hd128", 3, 128, False, 2),          # 4 heads, intermediate 1408 = 11 blocks of 128: an uneven 6 + 5 split
    ("tiny_hd128", 2, 128, False, 4),          # one head per rank
    ("tiny_hd128_gqa", 2, 64, False, 2),       # GQA: one kv head per rank; groupsize 64
    ("tiny_hd128", 2, 128, True, 2),           # act-order: o_proj / down_proj in gather mode (op path only)
    ("7b", 2, 128, False, 2),                  # real shapes: o_proj K = 2048, gate/up N = down K = 5504
]


@pytest.mark.parametrize("preset,layers,gs,act,world", CASES)
def test_tensor_parallel_ranks_reproduce_the_unsharded_model(preset, layers, gs, act, world):
    dims = synth.PRESETS[preset]
    cfg_dict = synth.config_dict(dims, num_layers=layers)
    tensors = synth.make_checkpoint(dims, groupsize=gs, act_order=act, seed=11, num_layers=layers)
    gen = torch.Generator().manual_seed(5)
    prompt = torch.randint(3, dims.vocab_size, (1, 37), generator=gen)
    steps = 6
    # the unsharded model first: its greedy tokens drive every rank (teacher forcing: no divergence through a near-tie)
    full, fcache = _build(cfg_dict, tensors)
    ref = [full.forward(prompt.to(DEV), fcache, last_id_only=False).float().cpu()]
    full.enable_decode_graph(fcache, use_graph=False)
    tokens = [ref[0][0, -1].argmax().view(1, 1)]
    for _ in range(steps):
        lg = full.forward(tokens[-1].to(DEV), fcache).float().cpu()
        ref.append(lg)
        tokens.append(lg[0, -1].argmax().view(1, 1))

    group = LocalGroup(world)
    keep = []

    def rank_program(r):
        def run():
            local, plan = tp.shard_tensors(tensors, cfg_dict, r, world)
            model, cache = _build(tp.shard_config_dict(cfg_dict, plan), local, tp.TensorParallel(plan, group.comm(r)))
            keep.append((model, cache))                                  # (free_unmanaged releases EVERY handle of the process: once, at the end)
            outs = [model.forward(prompt.to(DEV), cache, last_id_only=False).float().cpu()]
            if not act:
                model.enable_decode_graph(cache, use_graph=False)        # executor in pieces; act-order shards stay on the op path
            for i in range(steps):
                outs.append(model.forward(tokens[i].to(DEV), cache).float().cpu())
            return outs
        return run

    rank_outs = group.run([rank_program(r) for r in range(world)])
    full.free_unmanaged()

    for r in range(world):
        assert len(rank_outs[r]) == len(ref)
        for a, b in zip(rank_outs[r], ref):
            # same weights, same fp32 accumulation inside each matmul; the partial sums are rounded to fp16 before they are added
            # (one extra rounding per rank and half layer)
            scale = float(b.abs().max())
            err = float((a - b).abs().max())
            assert err <= 6e-3 * scale, (r, err, scale)
    # ... and the CPU oracle on the same tokens, not only the product's own unsharded model: the sharded ranks compute what the
    # reference algorithm computes (the extra fp16 rounding of the per-rank partial sums included in the tolerance)
    orc = OracleLlama(cfg_dict, tensors, max_seq_len=256)
    orc.prepare()                                                        # dequantise once
    want = [np.asarray(orc.forward(prompt.numpy(), last_id_only=False), dtype=np.float32)]
    for i in range(steps):
        want.append(np.asarray(orc.forward(tokens[i].numpy()), dtype=np.float32))
    for i, (a, b) in enumerate(zip(rank_outs[0], want)):
        _model_close(a.numpy(), b, TP_TOL, f"tensor-parallel rank 0 vs oracle, output {i}")   # the criteria of every other model test, at 5e-3 (parity.py: why)
    for r in range(1, world):                                            # the replicas of the residual stream agree exactly
        for a, b in zip(rank_outs[r], rank_outs[0]):
            assert torch.equal(a, b)


def test_generation_on_tensor_parallel_ranks():
    """generate_greedy / generate_sample on the ranks of a tensor-parallel model: every rank picks the token itself from the
    all-gathered logits (argmax, or the sampler kernel with the same draw per position), so the ranks must produce IDENTICAL token
    streams without exchanging a token -- and the stream of the unsharded model, except where that model's own top-2 margin is
    inside the tolerance the sharded logits are held to (then the first such step is where the comparison ends)."""
    from exllama_amd import _lib
    dims = synth.PRESETS["tiny_hd128"]
    layers, world, n = 3, 2, 10
    cfg_dict = synth.config_dict(dims, num_layers=layers)
    tensors = synth.make_checkpoint(dims, groupsize=128, act_order=False, seed=19, num_layers=layers)
    prompt = torch.randint(3, dims.vocab_size, (1, 29), generator=torch.Generator().manual_seed(8))
    settings = _lib.ExlSampler(temperature=0.9, top_k=40, top_p=0.8, rep_penalty_max=1.15, rep_sustain=16, rep_decay=8)
    uniforms = torch.rand(257, generator=torch.Generator().manual_seed(3)).to(DEV)

    def program(tpobj_for, local_tensors, cfgd):
        def run():
            model, cache = _build(cfgd, local_tensors, tpobj_for)
            keep.append((model, cache))
            model.forward(prompt[:, :-1].to(DEV), cache, preprocess_only=True)
            model.enable_decode_graph(cache, use_graph=tpobj_for is None)
            # greedy, with the logits of every step kept for the margin rule
            c = cache.current_seq_len
            greedy = model.generate_greedy(prompt[:, -1:].to(DEV), cache, n).cpu().tolist()
            cache.current_seq_len = c                                       # rewind: the same positions again, sampled
            sampled = model.generate_sample(prompt.to(DEV), cache, n, settings=settings, uniforms=uniforms).cpu().tolist()
            cache.current_seq_len = c
            steps, tok = [], prompt[:, -1:].to(DEV)
            for t in greedy:                                                # teacher-forced logits of the greedy stream
                lg = model.forward(tok, cache).float().cpu()[0, -1]
                steps.append(lg)
                tok = torch.tensor([[t]], device=DEV)
            return greedy, sampled, steps
        return run

    keep = []
    full_greedy, full_sampled, full_steps = program(None, tensors, cfg_dict)()
    group = LocalGroup(world)
    progs = []
    for r in range(world):
        local, plan = tp.shard_tensors(tensors, cfg_dict, r, world)
        progs.append(program(tp.TensorParallel(plan, group.comm(r)), local, tp.shard_config_dict(cfg_dict, plan)))
    outs = group.run(progs)
    keep[0][0].free_unmanaged()
    assert outs[0][0] == outs[1][0] and outs[0][1] == outs[1][1], "the ranks of one model disagree on a token"
    assert len(set(outs[0][0])) > 1
    for got, want, what in ((outs[0][0], full_greedy, "greedy"),):
        for i, (a, b) in enumerate(zip(got, want)):
            if a != b:                                                       # allowed only at a near-tie of the unsharded model
                lg = full_steps[i]
                assert float(lg[b] - lg[a]) <= 1.2e-2 * float(lg.abs().max()), (what, i, a, b)
                break
