"""The multi-GPU code paths of the product run by REAL processes against REAL kernels on the one GPU a test box has.

Every rank is its own process (spawned here), joined into a `torch.distributed` group over 127.0.0.1 with backend "gloo" -- RCCL
refuses two ranks on one device, so the device tensors travel through exllama_amd.pipeline.HostStagedGroup (device -> host ->
gloo -> host -> device).  What runs in each rank is the product's own multi-process code and nothing else:

  * layer split (reference: model.py:636-668 device map, :1053-1058 the hop between devices): pipeline.stage_tensors ->
    ExLlama on the rank's layers only -> pipeline.LayerSplitRunner.forward (prompt; one hidden-state hand-off per boundary),
    enable_decode_executor + forward (single tokens through the native executor stage of each rank, StageHop.before / after
    on rank > 0 and rank < last), enable_decode_executor(token_ring=True) + generate_greedy (the token travels from the last
    rank to rank 0, the first-token send and the final spare receive included);
  * tensor parallel: tp.shard_tensors -> ExLlama with tp.TensorParallel over the process group -> prompt
    pass (op path with all-reduce / all-gather), executor in pieces with the residual stream all-reduced after every half
    layer, generate_greedy on every rank.

Results are compared (a) BIT FOR BIT with the same checkpoint run by one process (layer split: nothing is re-rounded at a hop;
the ranks of a TP model must agree with each other exactly) and (b) with the CPU oracle through tests/parity.py's criteria.
bench.py's --layer-split / --tensor-parallel modes use the same classes with backend "nccl" (= RCCL over xGMI)."""
import os
import socket
import traceback

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from exllama_amd import synth

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _join_group(rank, world, port):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    return dist


def _cfg(cfg_dict, max_seq, **over):
    from exllama_amd.model import ExLlamaConfig
    cfg = ExLlamaConfig(dict(cfg_dict))
    cfg.max_seq_len = max_seq
    cfg.max_input_len = max_seq
    for k, v in over.items():
        setattr(cfg, k, v)
    return cfg


# ---- layer split ---------------------------------------------------------------------------------------------------------------
def _layer_split_rank(rank, world, port, spec, out):
    try:
        dist = _join_group(rank, world, port)
        from exllama_amd.model import ExLlama, ExLlamaCache
        from exllama_amd.pipeline import HostStagedGroup, LayerSplitRunner, split_layers, stage_tensors
        dims = synth.PRESETS[spec["preset"]]
        L, S, n = spec["layers"], spec["prompt"], spec["steps"]
        tensors = synth.make_checkpoint(dims, groupsize=spec["gs"], act_order=spec["act"], seed=spec["seed"], num_layers=L, zeros="rand")
        first, last = split_layers(L, world)[rank]
        local = stage_tensors(tensors, first, last)
        assert sum(k.startswith("model.layers.") and k.endswith("q_proj.qweight") for k in local) == last - first
        del tensors
        model = ExLlama(_cfg(synth.config_dict(dims, last - first), spec["max_seq"]), tensors=local)
        cache = ExLlamaCache(model)
        group = HostStagedGroup(dist)
        runner = LayerSplitRunner(model, cache, group, dims.hidden_size, DEV)
        ids = torch.tensor(spec["ids"], dtype=torch.int64).view(1, -1).to(DEV)
        res = {"rank": rank, "layers": (first, last)}
        # prompt pass: embedding on rank 0, one hand-off per boundary, head on the last rank
        logits = runner.forward(ids, last_id_only=False)
        assert (logits is not None) == (rank == world - 1)
        tok0 = runner.next_token(logits)
        res["tok0"] = int(tok0)
        if logits is not None:
            res["prompt_logits"] = logits.float().cpu().numpy()
        res["k_rows"] = [cache.key_states[i][:, :, :S].cpu().numpy() for i in range(last - first)]
        for use_graph in (True, False):
            tag = "graph" if use_graph else "eager"
            # (a) single tokens through every rank's executor stage, token brought by the host (no ring)
            cache.current_seq_len = S
            runner.enable_decode_executor(use_graph=use_graph, capture_hop=False, token_ring=False)
            assert model._decoder["has_embed"] == (rank == 0) and model._decoder["has_head"] == (rank == world - 1)
            assert not model._decoder["hop_captured"]
            tok, steps, toks = tok0.clone(), [], []
            for _ in range(n):
                lg = runner.forward(tok)
                assert (lg is not None) == (rank == world - 1)
                if lg is not None:
                    steps.append(lg.float().cpu().numpy()[0, 0])
                tok = runner.next_token(lg)
                toks.append(int(tok))
            assert cache.current_seq_len == S + n
            res[f"steps_{tag}"], res[f"toks_{tag}"] = steps, toks
            # (b) the token ring: no host in the loop, generate_greedy on every rank
            cache.current_seq_len = S
            runner.enable_decode_executor(use_graph=use_graph, capture_hop=False, token_ring=True)
            with pytest.raises(RuntimeError):
                runner.forward(tok0)                                    # single-token forward() needs the ring off
            res[f"ring_{tag}"] = runner.generate_greedy(tok0, n).cpu().tolist()
            assert cache.current_seq_len == S + n
        res["k_rows_after"] = [cache.key_states[i][:, :, :S + n].cpu().numpy() for i in range(last - first)]
        model.disable_decode_graph()
        dist.barrier()
        out.put((rank, None, res))
        dist.destroy_process_group()
    except BaseException:                                               # noqa: BLE001 -- reported to the parent
        out.put((rank, traceback.format_exc(), None))


def _run_ranks(target, world, spec, timeout=600):
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=target, args=(r, world, port, spec, out)) for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    try:
        for _ in range(world):
            rank, err, res = out.get(timeout=timeout)
            if err is not None:
                raise AssertionError(f"rank {rank} failed:\n{err}")
            got[rank] = res
    finally:
        for p in procs:
            p.join(timeout=60)
            if p.is_alive():
                p.kill()                                                # the exact process this test started
    return [got[r] for r in range(world)]


LAYER_SPLIT_CASES = [
    # preset, layers, groupsize, act_order, world, prompt tokens
    ("7b", 4, 128, False, 2, 45),                 # BASELINE configs[1] layer shapes, two ranks x 2 layers
    ("tiny_hd128", 3, 128, "gptq", 3, 150),       # three ranks x 1 layer: a MIDDLE rank (receive and send); crosses the 160-key bucket
    ("tiny_hd128_gqa", 2, 64, False, 2, 33),      # GQA, groupsize 64
]


@pytest.mark.parametrize("preset,layers,gs,act,world,S", LAYER_SPLIT_CASES)
def test_layer_split_across_processes_equals_one_process(preset, layers, gs, act, world, S):
    from exllama_amd.model import ExLlama, ExLlamaCache
    from oracle.model_oracle import OracleLlama
    from parity import ORACLE_TOL, _model_close
    dims = synth.PRESETS[preset]
    n = 12
    ids = torch.randint(3, dims.vocab_size, (1, S), generator=torch.Generator().manual_seed(S)).tolist()
    spec = dict(preset=preset, layers=layers, gs=gs, act=act, seed=77, prompt=S, steps=n, max_seq=S + n + 4, ids=ids)
    # one process, the whole checkpoint: the reference of the bit-for-bit comparison
    tensors = synth.make_checkpoint(dims, groupsize=gs, act_order=act, seed=77, num_layers=layers, zeros="rand")
    model = ExLlama(_cfg(synth.config_dict(dims, layers), S + n + 4), tensors={k: v.clone() for k, v in tensors.items()})
    cache = ExLlamaCache(model)
    want_prompt = model.forward(torch.tensor(ids).to(DEV), cache, last_id_only=False).float().cpu().numpy()
    tok0 = int(want_prompt[0, -1].argmax())
    model.enable_decode_graph(cache)
    want_steps, want_toks, tok = [], [], tok0
    for _ in range(n):
        lg = model.forward(torch.tensor([[tok]], device=DEV), cache).float().cpu().numpy()[0, 0]
        want_steps.append(lg)
        tok = int(lg.argmax())
        want_toks.append(tok)
    cache.current_seq_len = S
    want_ring = model.generate_greedy(torch.tensor([[tok0]], device=DEV), cache, n).cpu().tolist()
    assert want_ring == want_toks
    want_k = [cache.key_states[i][:, :, :S + n].cpu().numpy() for i in range(layers)]
    model.free_unmanaged()
    del model, cache
    torch.cuda.empty_cache()

    ranks = _run_ranks(_layer_split_rank, world, spec)
    last = ranks[-1]
    assert [r["layers"] for r in ranks] == [tuple(x) for x in __import__("exllama_amd.pipeline", fromlist=["x"]).split_layers(layers, world)]
    assert all(r["tok0"] == tok0 for r in ranks)
    assert np.array_equal(last["prompt_logits"], want_prompt), "prompt logits of the split differ from the one-process model"
    for tag in ("graph", "eager"):
        for r in ranks:
            assert r[f"toks_{tag}"] == want_toks, (tag, r["rank"])
            assert r[f"ring_{tag}"] == want_toks, (tag, r["rank"], "token ring")
        for i, (a, b) in enumerate(zip(last[f"steps_{tag}"], want_steps)):
            assert np.array_equal(a, b), (tag, i)
    for r in ranks:                                                      # the K rows every rank wrote are the whole model's
        first, _ = r["layers"]
        for j, k in enumerate(r["k_rows_after"]):
            assert np.array_equal(k, want_k[first + j]), (r["rank"], j)
    # ... and the CPU oracle, not only the product's own one-process model
    orc = OracleLlama(synth.config_dict(dims, layers), tensors, max_seq_len=S + n + 4)
    orc.prepare()
    ref = np.asarray(orc.forward(np.asarray(ids), last_id_only=False), dtype=np.float32)
    _model_close(last["prompt_logits"], ref, ORACLE_TOL, f"layer split x{world} {preset}: prompt logits vs oracle")


# ---- tensor parallel -----------------------------------------------------------------------------------------------------------
def _tensor_parallel_rank(rank, world, port, spec, out):
    try:
        dist = _join_group(rank, world, port)
        from exllama_amd import tp
        from exllama_amd.model import ExLlama, ExLlamaCache
        from exllama_amd.pipeline import HostStagedGroup
        dims = synth.PRESETS[spec["preset"]]
        L, S = spec["layers"], spec["prompt"]
        cfg_dict = synth.config_dict(dims, L)
        tensors = synth.make_checkpoint(dims, groupsize=spec["gs"], act_order=spec["act"], seed=spec["seed"], num_layers=L)
        local, plan = tp.shard_tensors(tensors, cfg_dict, rank, world)
        del tensors
        tpobj = tp.TensorParallel(plan, HostStagedGroup(dist))
        model = ExLlama(_cfg(tp.shard_config_dict(cfg_dict, plan), spec["max_seq"], tp=tpobj), tensors=local)
        cache = ExLlamaCache(model)
        ids = torch.tensor(spec["ids"], dtype=torch.int64).view(1, -1).to(DEV)
        res = {"rank": rank}
        res["prompt_logits"] = model.forward(ids, cache, last_id_only=False).float().cpu().numpy()
        gather = any(l.self_attn.o_gather or l.mlp.down_gather for l in model.layers)
        res["gather_mode"] = gather
        if not gather:
            model.enable_decode_graph(cache, use_graph=False)            # executor in pieces (exl_decoder_step_part) + all-reduces
        steps = []
        for t in spec["tokens"]:                                         # teacher-forced: the one-process model's greedy stream
            steps.append(model.forward(torch.tensor([[t]], device=DEV), cache).float().cpu().numpy()[0, 0])
        res["steps"] = steps
        if not gather:
            cache.current_seq_len = S
            res["greedy"] = model.generate_greedy(torch.tensor([[spec["tokens"][0]]], device=DEV), cache, len(spec["tokens"]) - 1).cpu().tolist()
        dist.barrier()
        out.put((rank, None, res))
        dist.destroy_process_group()
    except BaseException:                                               # noqa: BLE001
        out.put((rank, traceback.format_exc(), None))


TP_CASES = [
    ("tiny_hd128", 3, 128, False, 2, 37),          # 4 heads, intermediate 1408 = 11 blocks of 128: an uneven 6 + 5 split
    ("tiny_hd128", 2, 128, True, 2, 37),           # act-order: o_proj in gather mode (two all-gathers), op path
    ("7b", 2, 128, False, 2, 37),                  # real shapes: o_proj K = 2048, gate/up N = down K = 5504 per rank
    ("tiny_hd128", 2, 128, False, 4, 21),          # four processes, one head each
]


@pytest.mark.parametrize("preset,layers,gs,act,world,S", TP_CASES)
def test_tensor_parallel_across_processes_over_torch_distributed(preset, layers, gs, act, world, S):
    from exllama_amd.model import ExLlama, ExLlamaCache
    from oracle.model_oracle import OracleLlama
    from parity import TP_TOL, _model_close
    dims = synth.PRESETS[preset]
    n = 6
    ids = torch.randint(3, dims.vocab_size, (1, S), generator=torch.Generator().manual_seed(5)).tolist()
    tensors = synth.make_checkpoint(dims, groupsize=gs, act_order=act, seed=11, num_layers=layers)
    cfg_dict = synth.config_dict(dims, layers)
    full = ExLlama(_cfg(cfg_dict, 256), tensors={k: v.clone() for k, v in tensors.items()})
    fcache = ExLlamaCache(full)
    ref = [full.forward(torch.tensor(ids).to(DEV), fcache, last_id_only=False).float().cpu().numpy()]
    full.enable_decode_graph(fcache, use_graph=False)
    tokens = [int(ref[0][0, -1].argmax())]
    for _ in range(n):
        lg = full.forward(torch.tensor([[tokens[-1]]], device=DEV), fcache).float().cpu().numpy()
        ref.append(lg[0, 0])
        tokens.append(int(lg[0, 0].argmax()))
    full.free_unmanaged()
    del full, fcache
    torch.cuda.empty_cache()

    spec = dict(preset=preset, layers=layers, gs=gs, act=act, seed=11, prompt=S, max_seq=256, ids=ids, tokens=tokens[:n])
    ranks = _run_ranks(_tensor_parallel_rank, world, spec)
    assert all(r["gather_mode"] == bool(act) for r in ranks)
    for r in ranks:
        outs = [r["prompt_logits"]] + r["steps"]
        assert len(outs) == n + 1
        for a, b in zip(outs, ref):
            scale = float(np.abs(b).max())
            # the partial sums are rounded to fp16 by the kernels that produce them, travel and are ADDED as fp32 and the total is rounded
            # once (tp.TensorParallel.all_reduce): the same bound at every world size (round 5 added pairwise in fp16 and needed
            # (world - 1) x 6e-3: 1.47e-2 x scale at four ranks, gpurun_out/r05a)
            assert float(np.abs(a - b).max()) <= 6e-3 * scale, r["rank"]
    for r in ranks[1:]:                                                   # the replicas of the residual stream agree exactly
        assert np.array_equal(r["prompt_logits"], ranks[0]["prompt_logits"])
        for a, b in zip(r["steps"], ranks[0]["steps"]):
            assert np.array_equal(a, b)
        if not act:
            assert r["greedy"] == ranks[0]["greedy"]
    if not act:                                                           # the greedy stream of the unsharded model, up to its first near-tie
        for i, (a, b) in enumerate(zip(ranks[0]["greedy"], tokens[1:n])):
            if a != b:
                lg = ref[i + 1]
                assert float(lg[b] - lg[a]) <= 1.2e-2 * float(np.abs(lg).max()), (i, a, b)
                break
    orc = OracleLlama(cfg_dict, tensors, max_seq_len=256)
    orc.prepare()
    want = [np.asarray(orc.forward(np.asarray(ids), last_id_only=False), dtype=np.float32)]
    for t in tokens[:n]:
        want.append(np.asarray(orc.forward(np.array([[t]])), dtype=np.float32)[0, 0])
    for i, (a, b) in enumerate(zip([ranks[0]["prompt_logits"]] + ranks[0]["steps"], want)):
        _model_close(a, b, TP_TOL, f"tensor parallel x{world} over torch.distributed, {preset}: output {i} vs oracle")   # unscaled at every world size
