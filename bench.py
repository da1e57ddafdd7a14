#!/usr/bin/env python3
"""Benchmark of the 4-bit GPTQ Llama hot path on MI355X -- the reference's `-p` protocol
(/root/reference/test_benchmark_inference.py:155-197) with device synchronisation added.

One STEP = one pass of the protocol on one synthetic prompt:
    prefill of `--prompt` random token ids (last_id_only), then `--gen` greedy decode steps at full context
    ("worst case" of the reference's table), then `--gen` greedy decode steps from a 4-token context ("best").
`value` is the single-token decode rate at full context (tokens/s, whole job = sum over ranks); prefill and
best-case rates ride along in the same JSON line.  Inputs (weights, prompt) are resident in HBM before the
timed region.  Multi-GPU: one process per GPU, replicas of the model on different prompts, no data-path
collective (weak scaling) -- the reference itself only shards layers sequentially.

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
"""

import argparse
import json
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL across processes needs it on this driver
os.environ.setdefault("EXL_REQUIRE_FAST_BINDING", "1")         # a benchmark never runs on the ctypes path by accident (cuda_ext.py)

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured streaming copy)
MFMA_PEAK_TFLOPS = 2500.0        # fp16/bf16 dense


def parse_args():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=3)
    p.add_argument("--warmup", type=int, default=1)
    p.add_argument("--model", default="7b", choices=["7b", "13b", "33b", "65b", "70b", "tiny", "tiny_gqa"])
    p.add_argument("--groupsize", type=int, default=128)
    p.add_argument("--act-order", action="store_true")
    p.add_argument("--act-order-maps", default="gptq", choices=["gptq", "independent"],
                   help="with --act-order: 'gptq' = q/k/v and gate/up share their row permutation, as GPTQ (desc_act) writes them "
                        "(quantised against the same input); 'independent' = one random permutation per matrix (the general case)")
    p.add_argument("--prompt", type=int, default=2048)
    p.add_argument("--gen", type=int, default=128)
    p.add_argument("--layers", type=int, default=None, help="debug: truncate the model (INVALID as a benchmark)")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--cpu-baseline-full", action="store_true",
                   help="time the CPU oracle on ALL layers instead of 2 extrapolated (needs ~30 GB of host memory and ~2 minutes for 7B)")
    p.add_argument("--no-graph", action="store_true", help="decode eagerly instead of replaying the captured hipGraph")
    p.add_argument("--host-argmax", action="store_true", help="greedy argmax by torch between graph replays (the reference's loop) instead of inside the graph")
    p.add_argument("--no-roofline-probe", action="store_true")
    p.add_argument("--nibbles", default="centered", choices=["centered", "uniform"],
                   help="synthetic weight nibbles: 'centered' (default: symmetric about the zero point, finite activations at any depth) or "
                        "'uniform' (rounds 1-4: every weight biased by -0.5 steps; models deeper than ~36 layers compute on inf / NaN) -- A/B aid")
    p.add_argument("--brief", action="store_true",
                   help="the timed protocol, path_roofline and the kernel probes only: no other prompt lengths, host-loop tiers, CPU baseline, "
                        "drop-in run or other configs (what the sub-runs of the default invocation use)")
    p.add_argument("--no-other-configs", action="store_true",
                   help="default 7B run on one GPU: do not run BASELINE configs[2..4] (13B act-order, 33B g32 act-order, 65B) and the drop-in "
                        "path as sub-runs after the headline")
    p.add_argument("--sharded-at-one-gpu", action="store_true",
                   help="test aid: run the sharded sub-runs (one-rank RCCL groups) behind a --gpus 1 headline too, so that the launching "
                        "machinery of the N > 1 line is exercised on a one-GPU box")
    p.add_argument("--no-sharded", action="store_true",
                   help="--gpus N > 1: do not run the sharded sub-runs (--layer-split / --tensor-parallel over the same N GPUs) after the "
                        "replica headline")
    p.add_argument("--layer-split", action="store_true",
                   help="ONE model split by layers across the ranks (hidden states handed off by RCCL send/recv, exllama_amd/pipeline.py) "
                        "instead of one replica per rank; capacity mode: the stages run one after the other at batch 1")
    p.add_argument("--tensor-parallel", action="store_true",
                   help="ONE model split by heads / intermediate columns across the ranks (exllama_amd/tp.py): every rank streams 1/N of "
                        "the weights per token, two all-reduces of the residual stream per layer over RCCL")
    args = p.parse_args()
    if args.act_order:                                    # what synth.make_checkpoint takes: False, True (independent maps) or "gptq"
        args.act_order = "gptq" if args.act_order_maps == "gptq" else True
    return args


def algorithmic_bytes_per_matmul(K, N, g, M=1):
    """K*N/2 packed nibbles + (K/g)*N*2.5 (fp16 scale + 4-bit zero) + 2*M*(K+N) activations."""
    return K * N / 2 + (K / g) * N * 2.5 + 2 * M * (K + N)


def decode_bytes_per_token(dims, g, ctx):
    h, I, L, V = dims.hidden_size, dims.intermediate_size, dims.num_hidden_layers, dims.vocab_size
    kvd = dims.num_key_value_heads * dims.head_dim
    per_layer = (2 * h * h + 2 * h * kvd + 3 * h * I) * (0.5 + 2.5 / g) + 4 * h
    weights = L * per_layer + V * h * 2 + 2 * h
    kv = 2 * L * ctx * kvd * 2
    return weights + kv


def prefill_flops(dims, S, causal=False):
    """SURVEY.md 8d counts the attention of a prompt as 4 * S^2 * h per layer (every key for every query); the kernels skip the
    masked half: causal=True counts what is actually multiplied (S (S + 1) / 2 pairs).  Both ride in path_roofline.prefill."""
    h, I, L, V = dims.hidden_size, dims.intermediate_size, dims.num_hidden_layers, dims.vocab_size
    kvd = dims.num_key_value_heads * dims.head_dim
    linear = 2 * S * L * (2 * h * h + 2 * h * kvd + 3 * h * I)
    attn = 4 * (S * (S + 1) // 2 if causal else S * S) * h * L
    return linear + attn + 2 * h * V


def rocprof_average_us(kernel_substring):
    """Average launch duration of a kernel from the newest committed profiles/rNN_bench_kernel_stats.csv (rocprofv3 --kernel-trace --stats of
    this same command, collected by scripts/gpu_rNN_profiles.sh): the figure the event timing beside it must agree with.  (None, None) when
    there is no such file / row."""
    import csv
    import glob
    import re
    best = None
    for path in glob.glob(os.path.join(ROOT, "profiles", "r*_bench_kernel_stats.csv")):
        m = re.match(r"r(\d+)_bench_kernel_stats\.csv$", os.path.basename(path))
        if m and (best is None or int(m.group(1)) > best[0]):
            best = (int(m.group(1)), path)
    if best is None:
        return None, None
    try:
        with open(best[1]) as f:
            for row in csv.DictReader(f):
                if kernel_substring in row.get("Name", ""):
                    return round(float(row["AverageNs"]) / 1e3, 3), "profiles/" + os.path.basename(best[1])
    except Exception:                                                 # noqa: BLE001
        pass
    return None, None


def layer_split_main(args, dims, L, S, G, rank, world, dev, dist):
    """--layer-split: rank r builds ONLY its contiguous run of layers (synthetic weights, own seed), the prompt and every
    decoded token travel through the ranks once per forward pass.  Same protocol and timing rules as the replica mode;
    the job's rate is the single pipeline's rate."""
    from exllama_amd import synth
    from exllama_amd.model import ExLlama, ExLlamaCache, ExLlamaConfig
    from exllama_amd.pipeline import LayerSplitRunner, split_layers

    class _Solo:                                                  # world == 1: same code path without a process group
        def get_rank(self): return 0
        def get_world_size(self): return 1
        def broadcast(self, t, src): return None
    d = dist if dist is not None else _Solo()
    first, last = split_layers(L, world)[rank]
    n_local = last - first
    tensors = synth.make_checkpoint(dims, groupsize=args.groupsize, act_order=args.act_order, seed=100 + rank, device=dev,
                                    zeros="sym", num_layers=n_local, nibbles=args.nibbles)
    cfg = ExLlamaConfig(synth.config_dict(dims, n_local))
    cfg.max_seq_len = S + G
    cfg.max_input_len = S
    cfg.device_map.layers = [dev] * n_local
    cfg.device_map.embed_tokens = cfg.device_map.norm = cfg.device_map.lm_head = dev
    model = ExLlama(cfg, tensors=tensors)
    del tensors
    cache = ExLlamaCache(model)
    runner = LayerSplitRunner(model, cache, d, dims.hidden_size, dev)
    if not args.no_graph:
        # each rank: native executor stage; the hand-off (hidden state in, hidden state / greedy token out) is captured into the rank's
        # hipGraph when RCCL allows it, so a token is one replay per rank with no host work (pipeline.StageHop)
        runner.enable_decode_executor(token_ring=True)
    gen = torch.Generator(device=dev)
    gen.manual_seed(1234)                                         # every rank needs the same prompt SHAPE; values matter on rank 0
    ids = torch.randint(0, min(31999, dims.vocab_size - 1), (1, S), device=dev, generator=gen)
    ev = lambda: torch.cuda.Event(enable_timing=True)

    def step(record):
        e = [ev() for _ in range(3)]
        cache.current_seq_len = 0
        e[0].record()
        logits = runner.forward(ids)
        e[1].record()
        tok = runner.next_token(logits)
        if not args.no_graph:
            runner.generate_greedy(tok, G)                        # G tokens: the token travels last rank -> rank 0 on the device
        else:
            for _ in range(G):
                lg = runner.forward(tok)
                tok = runner.next_token(lg)
        e[2].record()
        if record is not None:
            record.append(e)
        return logits

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step(None)
    barrier()
    events = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        last_logits = step(events)
    barrier()
    elapsed = time.perf_counter() - t0
    bad = 0.0 if last_logits is None or bool(torch.isfinite(last_logits).all()) else 1.0     # (the prompt's logits, on the last rank)
    bad = reduce_over_ranks([bad], dist, dev)[0]
    mean = lambda v: sum(v) / len(v)
    pre = mean([e[0].elapsed_time(e[1]) for e in events])
    dec = mean([e[1].elapsed_time(e[2]) for e in events])
    ranks = per_rank_report([elapsed * 1e3 / args.steps, pre, dec / G], dist, dev, ("ms_per_step", "prefill_ms", "decode_ms_per_token"))
    elapsed, pre, dec = reduce_over_ranks([elapsed, pre, dec], dist, dev)
    line = None
    if rank == 0:
        line = json.dumps(dict(**{
            **ranks,
            "metric": "single-token decode tokens/s at full context (prefill tokens/s alongside), Llama GPTQ 4-bit",
            "value": round(G / (dec / 1e3), 2), "unit": "tokens/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "logits_finite": bad == 0.0,
            "dtype": "int4 weights (GPTQ) x fp16 activations, fp32 accumulate",
            "data": "synthetic (seeded random GPTQ weights of the named architecture, random token ids)",
            "config": {"workload": f"Llama-{args.model.upper()} 4-bit GPTQ g{args.groupsize}, {S}-token prefill + {G}-token greedy decode, "
                                   f"ONE model split by layers over {world} rank(s)", "layers": L, "prompt_tokens": S, "gen_tokens": G,
                       "parallelism": f"layer split x{world} (sequential stages, P2P hidden-state hand-off, "
                                      + ("op-by-op decode path)" if args.no_graph else "one native executor stage + hipGraph per rank, hand-off "
                                         + ("captured in the graph)" if model._decoder.get("hop_captured") else "issued eagerly around the replay)"))},
            "prefill_tokens_per_s": round(S / (pre / 1e3), 1), "decode_worst_tokens_per_s": round(G / (dec / 1e3), 2),
            "prefill_ms": round(pre, 3), "decode_worst_ms_per_token": round(dec / G, 4)}))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if line is not None:                                           # the contract's ONE JSON line is the last thing on stdout
        _print_last(line)


def tensor_parallel_main(args, dims, L, S, G, rank, world, dev, dist):
    """--tensor-parallel: every rank builds ITS shard of ONE model (exllama_amd/tp.py: its heads, its intermediate columns;
    the same seeded checkpoint on every rank, cut locally), two collectives per layer on the residual stream.  Same protocol
    and timing rules as the replica mode; the job's rate is the one model's rate ("strong" scaling)."""
    from exllama_amd import synth, tp
    from exllama_amd.model import ExLlama, ExLlamaCache, ExLlamaConfig

    cfg_dict = synth.config_dict(dims, L)
    full = synth.make_checkpoint(dims, groupsize=args.groupsize, act_order=args.act_order, seed=100, device=dev, zeros="sym", num_layers=L, nibbles=args.nibbles)
    local, plan = tp.shard_tensors(full, cfg_dict, rank, world)
    del full
    cfg = ExLlamaConfig(tp.shard_config_dict(cfg_dict, plan))
    cfg.max_seq_len = S + G
    cfg.max_input_len = S
    cfg.device_map.layers = [dev] * L
    cfg.device_map.embed_tokens = cfg.device_map.norm = cfg.device_map.lm_head = dev
    cfg.tp = tp.TensorParallel(plan, dist)
    model = ExLlama(cfg, tensors=local)
    del local
    cache = ExLlamaCache(model)
    gather_mode = any(l.self_attn.o_gather or l.mlp.down_gather for l in model.layers)
    if not gather_mode:
        model.enable_decode_graph(cache, use_graph=not args.no_graph)
    gen = torch.Generator(device=dev)
    gen.manual_seed(1234)                                         # the same prompt on every rank (replicated residual stream)
    ids = torch.randint(0, min(31999, dims.vocab_size - 1), (1, S), device=dev, generator=gen)
    ev = lambda: torch.cuda.Event(enable_timing=True)

    def step(record):
        e = [ev() for _ in range(3)]
        cache.current_seq_len = 0
        e[0].record()
        logits = first_logits = model.forward(ids, cache)
        e[1].record()
        tok = logits[0, -1].argmax().view(1, 1)                   # identical logits on every rank: no token broadcast needed
        if model._decoder is not None:
            model.generate_greedy(tok, cache, G)                   # every rank picks its own (identical) tokens on the device
        else:
            for _ in range(G):
                logits = model.forward(tok, cache)
                tok = logits[0, -1].argmax().view(1, 1)
        e[2].record()
        if record is not None:
            record.append(e)
        return first_logits

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step(None)
    barrier()
    events = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        last_logits = step(events)
    barrier()
    elapsed = time.perf_counter() - t0
    logits_finite = bool(torch.isfinite(last_logits).all())
    mean = lambda v: sum(v) / len(v)
    pre = mean([e[0].elapsed_time(e[1]) for e in events])
    dec = mean([e[1].elapsed_time(e[2]) for e in events])
    ranks = per_rank_report([elapsed * 1e3 / args.steps, pre, dec / G], dist, dev, ("ms_per_step", "prefill_ms", "decode_ms_per_token"))
    elapsed, pre, dec = reduce_over_ranks([elapsed, pre, dec], dist, dev)
    line = None
    if rank == 0:
        st = model._decoder
        mode = ("op-by-op path (act-order o_proj / down_proj shards: gather mode)" if gather_mode else
                "native executor in half-layer pieces, " + ("captured in one hipGraph per token" if (st and st["graph"] is not None) else "eager launches"))
        line = json.dumps(dict(**{
            **ranks,
            "metric": "single-token decode tokens/s at full context (prefill tokens/s alongside), Llama GPTQ 4-bit",
            "value": round(G / (dec / 1e3), 2), "unit": "tokens/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "logits_finite": logits_finite,
            "dtype": "int4 weights (GPTQ) x fp16 activations, fp32 accumulate",
            "data": "synthetic (seeded random GPTQ weights of the named architecture, random token ids)",
            "config": {"workload": f"Llama-{args.model.upper()} 4-bit GPTQ g{args.groupsize}, {S}-token prefill + {G}-token greedy decode, "
                                   f"ONE model, tensor parallel over {world} rank(s)", "layers": L, "prompt_tokens": S, "gen_tokens": G,
                       "parallelism": f"tensor parallel x{world} (heads / intermediate columns per rank, 2 all-reduces of the residual stream "
                                      f"per layer over RCCL; decode: {mode})"},
            "prefill_tokens_per_s": round(S / (pre / 1e3), 1), "decode_worst_tokens_per_s": round(G / (dec / 1e3), 2),
            "prefill_ms": round(pre, 3), "decode_worst_ms_per_token": round(dec / G, 4)}))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if line is not None:                                           # the contract's ONE JSON line is the last thing on stdout
        _print_last(line)


def _print_last(line):
    """The contract's ONE JSON line must be the LAST thing on stdout.  RCCL writes its version banner through C stdio, which is fully
    buffered when stdout is a file or pipe and would otherwise be flushed AFTER this line, at exit (seen on the GPU box: five banner
    lines behind the JSON): flush the C buffers first, then write the line unbuffered."""
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:                                                 # noqa: BLE001
        pass
    sys.stdout.flush()
    print(line, flush=True)


def device_state():
    """Best effort, after the timed region: what the management interface says about this box (the same binaries measured 600-650
    tokens/s on 7B and a wider spread on the big models across the boxes of one pool: the power cap and the clocks a box
    grants are the first things to compare).  Never fails the run."""
    import shutil
    import subprocess
    exe = shutil.which("rocm-smi") or "/opt/rocm/bin/rocm-smi"
    try:
        out = subprocess.run([exe, "-d", "0", "--showmaxpower", "--showpower", "--showclocks", "--showperflevel", "--showmemuse", "--json"],
                             capture_output=True, text=True, timeout=20).stdout
        card = next(iter(json.loads(out).values()))
        keep = {}
        for k, v in card.items():
            kl = k.lower()
            if any(t in kl for t in ("max graphics package power", "package power", "sclk", "mclk", "fclk", "performance level", "memory")):
                keep[k] = v
        keep["note"] = "sampled once after the timed steps (idle clocks; the cap and the levels are the comparable part)"
        return keep
    except Exception as e:                                            # noqa: BLE001
        return {"unavailable": str(e)[:120]}


def per_rank_report(local_values, dist, device, names):
    """What every rank measured, gathered on all ranks (rank 0 prints it): the multi-GPU modes report the communicator they ran on
    (`rccl_ranks`) and each rank's own timings next to the MAX the contract asks for, so one command on an N-GPU node yields the
    whole picture (a slow rank, a slow link)."""
    t = torch.tensor(list(local_values), dtype=torch.float64, device=device)
    if dist is None:
        rows = [t.tolist()]
        info = {"rccl_ranks": 0, "backend": None}
    else:
        out = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
        dist.all_gather(out, t)
        rows = [o.tolist() for o in out]
        info = {"rccl_ranks": dist.get_world_size(), "backend": dist.get_backend()}
    info["per_rank"] = [{"rank": r, **{n: round(v, 3) for n, v in zip(names, row)}} for r, row in enumerate(rows)]
    info["device"] = torch.cuda.get_device_name(device) if torch.cuda.is_available() else str(device)
    return info


def reduce_over_ranks(local_values, dist, device):
    """The contract's timing rule: every rank contributes its own timings, the job's timing is the MAX over ranks.
    `dist` is torch.distributed (initialised) or None for a single process."""
    t = torch.tensor(list(local_values), dtype=torch.float64, device=device)
    if dist is not None:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return t.tolist()


def whole_job_rates(world, S, G, prefill_ms, worst_ms, best_ms):
    """Replicas are independent (one model copy and one prompt per GPU, no data-path collective): the job's throughput
    is the sum over ranks of tokens / the slowest rank's time."""
    return {"prefill": world * S / (prefill_ms / 1e3), "worst": world * G / (worst_ms / 1e3), "best": world * G / (best_ms / 1e3)}


def main():
    args = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: become the launcher -- the command the contract names, one rank per GPU -- so
        # that an N > 1 invocation produces its line however it is started
        return self_launch(args.gpus)
    assert world == args.gpus, f"WORLD_SIZE ({world}) != --gpus ({args.gpus}); launch with torch.distributed.run"
    if os.environ.get("EXL_BENCH_DRY_RUN"):
        return dry_run_main(args, rank, world)
    assert torch.cuda.is_available(), "bench.py needs a HIP device"
    torch.cuda.set_device(local_rank)
    dev = f"cuda:{local_rank}"
    dist = None
    if world > 1 or os.environ.get("EXL_BENCH_FORCE_DIST"):        # the env switch exercises the RCCL code path with one rank
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(dev))

    from exllama_amd import synth
    from exllama_amd.cuda_ext import exllama_ext as ext
    from exllama_amd.model import ExLlama, ExLlamaCache, ExLlamaConfig

    dims = synth.PRESETS[args.model]
    L = dims.num_hidden_layers if args.layers is None else args.layers
    S, G = args.prompt, args.gen
    if args.layer_split:
        return layer_split_main(args, dims, L, S, G, rank, world, dev, dist)
    if args.tensor_parallel:
        return tensor_parallel_main(args, dims, L, S, G, rank, world, dev, dist)
    # nibbles="centered": weights symmetric about the zero point, as GPTQ writes them -- uniform nibbles carry a -0.5-step bias per weight
    # that pushes a deep model's residual stream out of the fp16 range around layer 37 (synth.make_checkpoint)
    tensors = synth.make_checkpoint(dims, groupsize=args.groupsize, act_order=args.act_order, seed=0, device=dev,
                                    zeros="sym", num_layers=L, nibbles=args.nibbles)
    cfg = ExLlamaConfig(synth.config_dict(dims, L))
    cfg.max_seq_len = S + G
    cfg.max_input_len = S
    cfg.device_map.layers = [dev] * L
    cfg.device_map.embed_tokens = cfg.device_map.norm = cfg.device_map.lm_head = dev
    model = ExLlama(cfg, tensors=tensors)
    del tensors
    cache = ExLlamaCache(model)
    use_graph = (not args.no_graph) and hasattr(model, "enable_decode_graph")
    if use_graph:
        model.enable_decode_graph(cache)

    gen = torch.Generator(device=dev)
    gen.manual_seed(1234 + rank)                                  # each replica gets its own prompt
    ids = torch.randint(0, min(31999, dims.vocab_size - 1), (1, S), device=dev, generator=gen)

    ev = lambda: torch.cuda.Event(enable_timing=True)
    phase_ms = {"prefill": [], "worst": [], "best": []}

    device_greedy = use_graph and not args.host_argmax and hasattr(model, "generate_greedy")

    def decode(n, logits):
        if device_greedy:                                         # n graph replays: decode kernels + argmax feeding the next step
            model.generate_greedy(logits[0, -1].argmax().view(1, 1), cache, n)
            return model.last_decoder_logits()
        for _ in range(n):                                        # the reference's loop (test_benchmark_inference.py:188-191)
            tok = logits[0, -1].argmax().view(1, 1)               # stays on the device: no host sync per token
            logits = model.forward(tok, cache)
        return logits

    def step(record):
        e = [ev() for _ in range(5)]
        cache.current_seq_len = 0
        e[0].record()
        logits = model.forward(ids, cache)                        # prefill, last_id_only
        e[1].record()
        logits = decode(G, logits)                                # worst case: context S .. S+G
        e[2].record()
        cache.current_seq_len = 4                                 # reference: test_benchmark_inference.py:196-197
        e[3].record()
        logits = decode(G, logits)                                # best case: context 4 .. 4+G
        e[4].record()
        if record is not None:
            record.append(e)
        return logits

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step(None)
    barrier()
    events = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        last_logits = step(events)
    barrier()
    elapsed = time.perf_counter() - t0
    # the timed steps ran on real numbers: a synthetic model whose residual stream has left the fp16 range executes the same
    # instructions on inf / NaN (and draws less power doing so) -- such a line is marked invalid (checked after the clock stopped)
    logits_finite = bool(torch.isfinite(last_logits).all()) and bool(torch.isfinite(model.forward(ids[:, :8], ExLlamaCache(model), last_id_only=False)).all())
    for e in events:
        phase_ms["prefill"].append(e[0].elapsed_time(e[1]))
        phase_ms["worst"].append(e[1].elapsed_time(e[2]))
        phase_ms["best"].append(e[3].elapsed_time(e[4]))
    mean = lambda v: sum(v) / len(v)
    elapsed, prefill_ms, worst_ms, best_ms = reduce_over_ranks(
        [elapsed, mean(phase_ms["prefill"]), mean(phase_ms["worst"]), mean(phase_ms["best"])], dist, dev)   # MAX over ranks
    rates = whole_job_rates(world, S, G, prefill_ms, worst_ms, best_ms)
    decode_tps, best_tps, prefill_tps = rates["worst"], rates["best"], rates["prefill"]

    result = {
        "metric": "single-token decode tokens/s at full context (prefill tokens/s alongside), Llama GPTQ 4-bit",
        "value": round(decode_tps, 2),
        "unit": "tokens/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 3),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "int4 weights (GPTQ) x fp16 activations, fp32 accumulate",
        "data": "synthetic (seeded random GPTQ weights of the named architecture -- random nibbles, " + (
            "every 0 nibble replaced by 8 so that the weights are symmetric about the zero point and the activations stay finite at any depth"
            if args.nibbles == "centered" else "uniform 0..15 as in rounds 1-4: biased by -0.5 steps, deep models overflow fp16") + "; random token ids)",
        "config": {
            "workload": f"Llama-{args.model.upper()} 4-bit GPTQ g{args.groupsize}{(' act-order (shared q/k/v and gate/up maps)' if args.act_order == 'gptq' else ' act-order (one map per matrix)') if args.act_order else ''}, "
                        f"{S}-token prefill + {G}-token greedy decode at context {S}..{S + G} (BASELINE configs[1]); "
                        f"plus {G} tokens from context 4",
            "layers": L, "prompt_tokens": S, "gen_tokens": G, "parallelism": f"replicas x{world}" if world > 1 else "single GPU",
            "decode_mode": ("hipGraph replay, greedy argmax inside the graph" if device_greedy else
                            "hipGraph replay, torch.argmax between replays" if use_graph else "eager launches"),
            "decode_path_report": model.decode_path_report(cache),     # which tier a token step takes on this model / cache, and why
        },
        "prefill_tokens_per_s": round(prefill_tps, 1),
        "decode_worst_tokens_per_s": round(decode_tps, 2),
        "decode_best_tokens_per_s": round(best_tps, 2),
        "prefill_ms": round(prefill_ms, 3), "decode_worst_ms_per_token": round(worst_ms / G, 4),
        "decode_best_ms_per_token": round(best_ms / G, 4),
    }
    result["logits_finite"] = logits_finite
    if not logits_finite:
        result["config"]["INVALID"] = "the model's activations left the fp16 range (inf / NaN logits): timings are not those of a real model"
    if args.layers is not None:
        result["config"]["INVALID"] = "truncated model (--layers): not a benchmark result"
    if rank == 0:
        result["device_state"] = device_state()

    # ---- further protocol points, measured after the timed region (not part of `value`) -----------------------------
    # the reference's own `-p` lengths (test_benchmark_inference.py:157-180: S = max_seq_len - 128 = 1920 with -l 2048) and
    # the 128-token prompt of BASELINE configs[0]; each: 2 warm-up passes, then the mean of 3
    if rank == 0 and world == 1 and not args.brief:
        def timed_prefill(n_tok, reps=3):
            for _ in range(2):
                cache.current_seq_len = 0
                model.forward(ids[:, :n_tok], cache)
            torch.cuda.synchronize()
            tot = 0.0
            for _ in range(reps):
                cache.current_seq_len = 0
                a, b = ev(), ev()
                a.record()
                lg = model.forward(ids[:, :n_tok], cache)
                b.record()
                torch.cuda.synchronize()
                tot += a.elapsed_time(b)
            return tot / reps, lg
        extra = {}
        if S > 1920:
            ms_1920, lg = timed_prefill(1920)
            a, b = ev(), ev()
            a.record()
            decode(G, lg)                                         # the reference's worst case: 128 tokens from context 1920
            b.record()
            torch.cuda.synchronize()
            extra["reference_protocol_S1920"] = {"prefill_tokens_per_s": round(1920 / (ms_1920 / 1e3), 1), "prefill_ms": round(ms_1920, 3),
                                                 "decode_tokens_per_s": round(G / (a.elapsed_time(b) / 1e3), 2)}
        if S >= 128:
            ms_128, _ = timed_prefill(128)
            extra["prompt_128_tokens"] = {"prefill_tokens_per_s": round(128 / (ms_128 / 1e3), 1), "prefill_ms": round(ms_128, 3),
                                          "note": "BASELINE configs[0] prompt length (2 .. 256 rows: one native call per layer, GEMMs on fragment-order "
                                                  "activations, csrc/q4_gemm_frag.hip; round 5 ran 12 eager launches per layer here: 5.27 ms)"}
            ms_16, _ = timed_prefill(16)
            extra["prompt_16_tokens"] = {"prefill_tokens_per_s": round(16 / (ms_16 / 1e3), 1), "prefill_ms": round(ms_16, 3),
                                         "note": "a chat turn: one row tile per block (the weights stream once, at about the decode step's rate)"}
        result["other_lengths"] = extra

        # ---- the reference's own generation loop on THIS repository's model.py (test_benchmark_inference.py:182-197: forward, then
        # torch.argmax on the logits, per token -- no generate_greedy): (a) forward() replaying the executor's hipGraph, the token
        # picked by a torch kernel between replays; (b) the executor off: forward() through the op-by-op fused ops (q4_attn ->
        # attention -> q4_attn_2 -> q4_mlp per layer, compiled binding), what the reference's model.py runs on the shim too, minus
        # its ATen attention.  With `value` (argmax inside the graph) and dropin_reference_model_py these are the tiers of the path.
        def host_loop(n, ctx):
            cache.current_seq_len = 0
            lg = model.forward(ids[:, :ctx], cache)
            for _ in range(8):                                    # warm
                lg = model.forward(lg[0, -1].argmax().view(1, 1), cache)
            cache.current_seq_len = ctx
            torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(n):
                lg = model.forward(lg[0, -1].argmax().view(1, 1), cache)
            torch.cuda.synchronize()
            return n / (time.perf_counter() - t)
        tiers = {}
        if use_graph:
            tiers["forward_graph_replay_plus_torch_argmax"] = {"worst": round(host_loop(G, S), 2), "best": round(host_loop(G, 4), 2)}
            model.disable_decode_graph()
        tiers["forward_op_by_op_plus_torch_argmax"] = {"worst": round(host_loop(G, S), 2), "best": round(host_loop(G, 4), 2)}
        if use_graph:
            model.enable_decode_graph(cache)
        tiers["unit"] = "tokens/s; wall clock around %d tokens incl. host time, context %d.. (worst) / 4.. (best)" % (G, S)
        result["host_argmax_loop"] = tiers

        # ---- batch > 1 (the reference's batched / CFG / beam generators call forward with bsz > 1, model.py:528): outside the executor
        # (batch 1 only) -- the general op path, q4_matmul at 2 .. 8 rows + attention per layer from Python.  Measured, so that the tier
        # has a number: sequences x tokens per second at context S (the K / V rows behind the position are whatever the cache holds:
        # a rate measurement, no text), 16 steps after 4 warm ones.
        if not args.brief:
            from exllama_amd.model import ExLlamaCache as _Cache
            batched = {}
            try:
                for bsz in (2, 4, 8):
                    bc = _Cache(model, batch_size=bsz)
                    tok = torch.randint(1, 31999, (bsz, 1), device=dev)
                    bc.current_seq_len = S
                    for _ in range(4):
                        lg = model.forward(tok, bc)
                        bc.current_seq_len = S
                    torch.cuda.synchronize()
                    t = time.perf_counter()
                    for i in range(16):
                        lg = model.forward(tok, bc)
                    torch.cuda.synchronize()
                    dt = time.perf_counter() - t
                    batched["bsz_%d" % bsz] = {"steps_per_s": round(16 / dt, 2), "tokens_per_s_all_sequences": round(16 * bsz / dt, 1),
                                              "logits_finite": bool(torch.isfinite(lg).all())}
                    del bc, lg
                    torch.cuda.empty_cache()
                batched["tier"] = "%s (decode_path_report): no executor for batch > 1; context %d .. %d" % (getattr(model, "_last_path", "?"), S, S + 16)
            except Exception as e:                                    # noqa: BLE001  (never let this leg take the headline with it)
                batched["error"] = "%s: %s" % (type(e).__name__, e)
            result["batched_decode"] = batched

    # ---- whole-path roofline fractions (algorithmic bytes / flops) ------------------------------
    full = synth.LlamaDims(dims.hidden_size, dims.intermediate_size, L, dims.num_attention_heads, dims.num_key_value_heads,
                           dims.vocab_size)
    b_worst = decode_bytes_per_token(full, args.groupsize, S + G / 2)
    b_best = decode_bytes_per_token(full, args.groupsize, 4 + G / 2)
    result["path_roofline"] = {
        "decode_worst": {"bytes_per_token": int(b_worst), "achieved_GBps": round(b_worst / (worst_ms / G / 1e3) / 1e9, 1),
                         "frac_of_8TBps": round(b_worst / (worst_ms / G / 1e3) / 1e9 / HBM_PEAK_GBS, 4)},
        "decode_best": {"bytes_per_token": int(b_best), "achieved_GBps": round(b_best / (best_ms / G / 1e3) / 1e9, 1),
                        "frac_of_8TBps": round(b_best / (best_ms / G / 1e3) / 1e9 / HBM_PEAK_GBS, 4)},
        "prefill": {"flops": prefill_flops(full, S), "achieved_TFLOPs": round(prefill_flops(full, S) / (prefill_ms / 1e3) / 1e12, 1),
                    "frac_of_2.5PF": round(prefill_flops(full, S) / (prefill_ms / 1e3) / 1e12 / MFMA_PEAK_TFLOPS, 4),
                    "flops_causal": prefill_flops(full, S, causal=True),
                    "frac_of_2.5PF_causal": round(prefill_flops(full, S, causal=True) / (prefill_ms / 1e3) / 1e12 / MFMA_PEAK_TFLOPS, 4),
                    "note": "flops: SURVEY.md 8d's count (attention 4 S^2 h per layer); flops_causal: the masked half not counted -- what the kernels multiply"},
    }

    # ---- dominant-kernel roofline: the q4 decode GEMV, timed per launch with HIP events on the launch stream --
    if rank == 0 and not args.no_roofline_probe:
        if use_graph:
            result["roofline"] = decoder_roofline_probe(model, cache, full, args.groupsize, S)
        else:
            result["roofline"] = gemv_roofline_probe(model, args.groupsize, dev)
    if rank == 0 and not args.no_roofline_probe:
        try:
            result["prefill_roofline"] = prefill_gemm_probe(model, full, S, dev)
        except Exception as exc:                                  # measurement aid only: never fail the benchmark line
            result["prefill_roofline"] = {"error": str(exc)}
    # ---- CPU baseline: the oracle ("port") on a bounded sample of the same workload ----------------------------
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not args.brief:
        result["cpu_baseline"] = cpu_baseline(dims, args.groupsize, ctx=S,
                                              sample_layers=dims.num_hidden_layers if args.cpu_baseline_full else 2)
    # ---- everything below runs AFTER the timed region as sub-runs of this same file / of scripts/, each in its own process
    # (own HIP context, memory released when it ends), on the GPU(s) this job was given, and lands in the ONE JSON line: every
    # number in the line is measured in THIS invocation.  A sub-run that fails or times out leaves an {"error": ...} record.
    extras_ok = rank == 0 and not args.brief and args.model == "7b" and args.groupsize == 128 and not args.act_order and args.layers is None
    if extras_ok and world == 1:
        if not args.sharded_at_one_gpu:
            del model, cache
            torch.cuda.empty_cache()
        if not args.no_other_configs:
            # BASELINE configs[2..4] on this one GPU (every model fits 288 GB): same protocol, fewer steps
            result["other_configs"] = {
                "13b_g128_actorder": sub_bench(["--model", "13b", "--act-order", "--steps", "2", "--warmup", "1"], "BASELINE configs[2]"),
                "33b_g32_actorder": sub_bench(["--model", "33b", "--groupsize", "32", "--act-order", "--steps", "2", "--warmup", "1"],
                                              "BASELINE configs[3] on ONE device (its 2-GPU layer split: --gpus 2)"),
                "65b_g128": sub_bench(["--model", "65b", "--steps", "1", "--warmup", "1"],
                                      "BASELINE configs[4] on ONE device (its 8-GPU layer split: --gpus 8)"),
            }
            # the drop-in path: the reference's UNMODIFIED model.py on the cuda_ext shim, its own -p loop (scripts/bench_dropin.py;
            # the reference's three .py files travel in the git-ignored oracle/_ref/refpy.tgz -- absent: the key is omitted)
            dropin = dropin_run()
            if dropin is not None:
                result["dropin_reference_model_py"] = dropin

    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if extras_ok and (world > 1 or args.sharded_at_one_gpu) and not args.no_sharded:
        # the replicas are done and every other rank is exiting (its GPU is free): rank 0 launches ONE model sharded over the same N GPUs
        # -- the reference's layer split (model.py:636-668; P2P hidden-state hand-off over RCCL) and the tensor-parallel mode -- each as
        # its own torch.distributed.run job with a time limit, so a hang or crash there cannot take the replica headline with it
        del model, cache
        torch.cuda.empty_cache()
        time.sleep(3.0)
        result["sharded"] = sharded_runs(world)
    if rank == 0:                                                  # the ONE JSON line last (after RCCL's output, if any)
        _print_last(json.dumps(result))


def self_launch(n):
    """Re-executes this invocation as `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P
    bench.py <the same arguments>` (a free port) and passes its output and exit code through: rank 0's ONE JSON line stays last on stdout."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    sys.stdout.flush()
    sys.exit(subprocess.call(cmd, cwd=ROOT, env=env))


def dry_run_main(args, rank, world):
    """EXL_BENCH_DRY_RUN=1 (tests/test_multiproc.py, no GPU): the PROCESS choreography of an N-rank invocation and nothing else -- one
    process per rank joined over gloo, fabricated per-rank timings through the contract's reduction (barrier, MAX over ranks, whole-job
    rates), rank 0 alone launching the sharded sub-runs as torch.distributed.run jobs of this same file (dry as well: the variable is
    inherited) once the group is gone, their records nested into the ONE JSON line, that line last on stdout.  No kernel, no model, no
    number that means anything: `config.DRY_RUN` says so."""
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
    S, G = args.prompt, args.gen
    local = [0.300 + 0.010 * rank, 25.0 + rank, 180.0 + 2 * rank, 150.0 + rank]          # elapsed s / step, prefill / worst / best ms
    if dist is not None:
        dist.barrier()
    if args.layer_split or args.tensor_parallel:
        rep = per_rank_report([local[0] * 1e3, local[1], local[2] / G], dist, "cpu", ("ms_per_step", "prefill_ms", "decode_ms_per_token"))
        elapsed, pre, dec, _ = reduce_over_ranks(local, dist, "cpu")
        line = None
        if rank == 0:
            mode = "layer split" if args.layer_split else "tensor parallel"
            line = json.dumps(dict(**{**rep, "metric": "dry run", "value": round(G / (dec / 1e3), 2), "unit": "tokens/s", "n_gpus": world,
                                      "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed * 1e3, 3), "scaling": "strong",
                                      "logits_finite": True, "prefill_tokens_per_s": round(S / (pre / 1e3), 1),
                                      "config": {"workload": f"DRY RUN {args.model} {mode} x{world}", "parallelism": f"{mode} x{world}", "DRY_RUN": True}}))
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        if line is not None:
            _print_last(line)
        return
    elapsed, prefill_ms, worst_ms, best_ms = reduce_over_ranks(local, dist, "cpu")
    rates = whole_job_rates(world, S, G, prefill_ms, worst_ms, best_ms)
    result = {"metric": "dry run", "value": round(rates["worst"], 2), "unit": "tokens/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
              "ms_per_step": round(elapsed * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
              "prefill_tokens_per_s": round(rates["prefill"], 1), "decode_best_tokens_per_s": round(rates["best"], 2),
              "config": {"workload": "DRY RUN", "parallelism": f"replicas x{world}" if world > 1 else "single GPU", "DRY_RUN": True}}
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0 and (world > 1 or args.sharded_at_one_gpu) and not args.no_sharded and not args.brief:
        result["sharded"] = sharded_runs(world, limit_s=120, budget_s=300)
    if rank == 0:
        _print_last(json.dumps(result))


def _last_json_line(text):
    for line in reversed(text.strip().splitlines()):
        line = line.strip()
        if line.startswith("{") and line.endswith("}"):
            try:
                return json.loads(line)
            except ValueError:
                continue
    return None


def _run_sub(cmd, limit_s, env=None):
    """One sub-run with a wall-clock limit; returns (parsed JSON line or None, seconds, error text or None).  The sub-run is its own
    process group: at the limit the WHOLE group (a torch.distributed.run launcher and its workers) is killed, nothing is left holding a GPU."""
    import signal
    import subprocess
    t0 = time.perf_counter()
    try:
        p = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=ROOT, env=env, start_new_session=True)
    except Exception as e:                                            # noqa: BLE001
        return None, time.perf_counter() - t0, str(e)[:300]
    try:
        out, err = p.communicate(timeout=limit_s)
    except subprocess.TimeoutExpired:
        try:
            os.killpg(p.pid, signal.SIGKILL)                          # the group this call created (start_new_session), nothing else
        except OSError:
            pass
        try:
            p.communicate(timeout=30)
        except Exception:                                             # noqa: BLE001
            pass
        return None, time.perf_counter() - t0, f"timed out after {limit_s} s"
    d = _last_json_line(out)
    if p.returncode != 0 or d is None:
        return None, time.perf_counter() - t0, f"rc {p.returncode}: {(err or out)[-400:]}"
    return d, time.perf_counter() - t0, None


_SUB_KEEP = ("value", "unit", "logits_finite", "ms_per_step", "steps", "warmup", "prefill_tokens_per_s", "decode_worst_tokens_per_s", "decode_best_tokens_per_s",
             "prefill_ms", "decode_worst_ms_per_token", "decode_best_ms_per_token", "path_roofline", "scaling", "n_gpus", "rccl_ranks", "backend",
             "per_rank")


def _sub_record(d, secs, err, what):
    if err is not None:
        return {"what": what, "error": err, "seconds": round(secs, 1)}
    rec = {"what": what, "workload": d["config"]["workload"], "decode_mode": d["config"].get("decode_mode") or d["config"].get("parallelism"),
           "measured": "this run (sub-run of the same bench.py after the timed region)", "seconds": round(secs, 1)}
    rec.update({k: d[k] for k in _SUB_KEEP if k in d})
    for k in ("roofline", "prefill_roofline"):
        if isinstance(d.get(k), dict) and "frac" in d[k]:
            rec[k] = {kk: d[k][kk] for kk in ("kernel", "achieved", "peak", "unit", "frac", "avg_launch_us") if kk in d[k]}
    return rec


# The one-GPU sub-runs behind the headline (13B / 33B / 65B, the drop-in run) share ONE wall-clock budget: the headline is printed
# after them, and a slow or hung box must not push it past the driver's own limit (measured: ~65 s for all four).
_SUB_BUDGET_S = 780.0
_sub_t0 = [None]


def _sub_left(limit_s):
    if _sub_t0[0] is None:
        _sub_t0[0] = time.perf_counter()
    return min(limit_s, _SUB_BUDGET_S - (time.perf_counter() - _sub_t0[0]))


def sub_bench(flags, what, limit_s=420):
    left = _sub_left(limit_s)
    if left < 30:
        return {"what": what, "error": f"skipped: the sub-runs' shared time budget ({_SUB_BUDGET_S:.0f} s) was used up"}
    d, secs, err = _run_sub([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--brief"] + flags, left)
    return _sub_record(d, secs, err, what)


def dropin_run(limit_s=420):
    if not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "refpy.tgz")):
        return None
    import tempfile
    left = _sub_left(limit_s)
    if left < 30:
        return {"error": f"skipped: the sub-runs' shared time budget ({_SUB_BUDGET_S:.0f} s) was used up"}
    out = os.path.join(tempfile.mkdtemp(prefix="exl_dropin_"), "dropin.json")
    _, secs, err = _run_sub([sys.executable, os.path.join(ROOT, "scripts", "bench_dropin.py"), "--out", out], left)
    try:
        with open(out) as f:
            d = json.load(f)
    except Exception:                                                 # noqa: BLE001
        return {"error": err or "no output", "seconds": round(secs, 1)}
    d["measured"] = "this run (scripts/bench_dropin.py as a sub-run after the timed region)"
    d["seconds"] = round(secs, 1)
    return d


def sharded_runs(world, limit_s=240, budget_s=600):
    """ONE model over the N GPUs of this job, launched by rank 0 after the replica run: `python -m torch.distributed.run ... bench.py
    --layer-split / --tensor-parallel` on a fresh rendezvous port.  Every job has its own time limit and all of them share a budget:
    these modes have never run on more than one GPU (no such box was available to any round), and a hang in one of them must not
    cost the replica headline its place in the driver's time limit."""
    import socket
    out = {}
    t_start = time.perf_counter()

    def launch(flags, what):
        if time.perf_counter() - t_start > budget_s:
            return {"what": what, "error": f"skipped: the sharded sub-runs' time budget ({budget_s} s) was used up"}
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "GROUP_RANK", "ROLE_RANK",
                                                                  "LOCAL_WORLD_SIZE", "ROLE_WORLD_SIZE", "TORCHELASTIC_RUN_ID")}
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--brief"] + flags
        d, secs, err = _run_sub(cmd, limit_s, env=env)
        return _sub_record(d, secs, err, what)
    out["layer_split_7b"] = launch(["--layer-split", "--steps", "2", "--warmup", "1"], f"Llama-7B g128 split by layers over {world} GPUs")
    out["layer_split_65b"] = launch(["--layer-split", "--model", "65b", "--steps", "1", "--warmup", "1"],
                                    f"BASELINE configs[4]: Llama-65B g128 split by layers over {world} GPUs (reference: model.py:636-668)")
    if world == 2:
        out["layer_split_33b_g32_actorder"] = launch(["--layer-split", "--model", "33b", "--groupsize", "32", "--act-order", "--steps", "2", "--warmup", "1"],
                                                     "BASELINE configs[3]: Llama-33B g32 act-order split by layers over 2 GPUs")
    out["tensor_parallel_7b"] = launch(["--tensor-parallel", "--steps", "2", "--warmup", "1"],
                                       f"Llama-7B g128 tensor parallel over {world} GPUs (not in the reference; SURVEY.md 8 row N4)")
    return out


def newest_pmc_profile():
    """profiles/rNN_pmc_traffic.json of the highest round present (collected by scripts/gpu_rNN_profiles.sh in separate rocprofv3 --pmc
    passes: counters cannot be read inside this process).  Returns (dict, provenance string) or (None, None)."""
    import glob
    import re
    best = None
    for path in glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")):
        m = re.match(r"r(\d+)_pmc_traffic\.json$", os.path.basename(path))
        if m and (best is None or int(m.group(1)) > best[0]):
            best = (int(m.group(1)), path)
    if best is None:
        return None, None
    try:
        with open(best[1]) as f:
            d = json.load(f)
    except Exception:                                                 # noqa: BLE001
        return None, None
    stamp = ", ".join(f"{k} {d[k]}" for k in ("git_head", "collected") if k in d) or "no git / date stamp in the file"
    return d, (f"offline PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate runs; FETCH_SIZE doubled per MI355X_MICROARCH.md): "
               f"profiles/{os.path.basename(best[1])} [{stamp}]; counters cannot be collected inside this process")


def decoder_roofline_probe(model, cache, dims, g, ctx, steps=6):
    """Per-kernel-class timing of the decode path itself: the native executor launched eagerly with a hipEvent between
    consecutive kernels on the launch stream (exl_decoder_step_timed), at context `ctx`, averaged over `steps` tokens.
    The dominant kernel is the fused gate/up projection (dec_gemv_kernel<..., PNORM=1, EMODE=2>): one launch streams the
    packed gate and up matrices of one layer."""
    h, I, L = dims.hidden_size, dims.intermediate_size, dims.num_hidden_layers
    kvd = dims.num_key_value_heads * dims.head_dim
    wbytes = lambda K, N: K * N / 2 + (K / g) * N * 2.5
    cache.current_seq_len = ctx
    tok = torch.zeros((1, 1), dtype=torch.int64, device=model.embed_weight.device)
    ms = model.decoder_profile(tok, cache, steps=steps)
    # algorithmic bytes per launch of each class (weights + activations in/out; attention: K and V rows of the context)
    per_launch = {
        "qkv": wbytes(h, h) + 2 * wbytes(h, kvd) + 2 * h + 2 * (h + 2 * kvd),
        "attn": 2 * (ctx + 1) * kvd * 2 + 2 * (h + 2 * kvd),
        "merge": 0,
        "o_proj": wbytes(h, h) + 2 * h + 4 * h,
        "gate_up": 2 * wbytes(h, I) + 2 * h + 2 * I,
        "down": wbytes(I, h) + 2 * I + 4 * h,
        "head": dims.vocab_size * h * 2 + 2 * h + 4 * dims.vocab_size,
    }
    classes = {}
    for k, v in ms.items():
        n = 1 if k == "head" else L
        us = v * 1e3 / n
        classes[k] = {"us_per_launch": round(us, 3), "GBps": round(per_launch[k] / us / 1e3, 1) if per_launch[k] else None}
    dom = "gate_up"
    achieved = per_launch[dom] / (ms[dom] / L / 1e3) / 1e9
    # HBM traffic of the same kernel from the PMC counters (newest round's file; only valid for the shapes it was measured on)
    traffic = None
    pmc, pmc_src = newest_pmc_profile()
    try:
        if pmc is not None and (h, I, g) == (4096, 11008, 128):
            traffic = pmc["decode_classes"][dom]["hbm_bytes_per_launch"]
    except Exception:
        traffic = None
    return {"bound": "hbm", "kernel": "dec_ring_kernel<PNORM=1, EMODE=2> (fused RMSNorm + gate/up projections + SiLU*mul, one launch per layer)",
            "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
            "traffic": traffic,
            "traffic_source": pmc_src if traffic else None,
            "launches": L * steps, "avg_launch_us": round(ms[dom] * 1e3 / L, 3),
            "rocprof_avg_us": rocprof_average_us("dec_ring_kernel<3, 8, 1, 2,")[0] if (h, I, g) == (4096, 11008, 128) else None,
            "rocprof_source": rocprof_average_us("dec_ring_kernel<3, 8, 1, 2,")[1] if (h, I, g) == (4096, 11008, 128) else None,
            "algorithmic_bytes_per_launch": int(per_launch[dom]), "classes": classes,
            "token_ms_sum_of_classes": round(sum(ms.values()), 4),
            "note": "per class: %d passes over all %d layers' launches of that kernel, back to back between two HIP events on the "
                    "launch stream (kernel + launch boundary), context %d; peak = 8.0 TB/s spec (6.29 TB/s measured copy => "
                    "frac_of_measured = %.4f)" % (steps, L, ctx, achieved / 6290.0)}


def prefill_gemm_probe(model, dims, S, dev, reps=2):
    """Dominant PREFILL kernel: the fused gate+up projection of the prompt pass (q4_gemm_t16d_kernel: x @ Wgate and x @ Wup
    from one activation tile, SiLU*mul in the epilogue).  Every layer's launch (different weights each), bracketed as a group
    by two HIP events on the launch stream; flops per launch = 2 matmuls x 2 S h I."""
    from exllama_amd.cuda_ext import exllama_ext as ext
    h, I = dims.hidden_size, dims.intermediate_size
    x = (torch.randn(S, h, device=dev) * 0.5).half()
    out = torch.empty(S, I, dtype=torch.float16, device=dev)
    mlps = [layer.mlp for layer in model.layers]
    launched = ext.q4_matmul_dual(x, mlps[0].gate_proj.q4, mlps[0].up_proj.q4, out, None, True)
    if not launched:
        return {"error": "fused gate/up kernel not eligible for this shape"}
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        for m in mlps:
            ext.q4_matmul_dual(x, m.gate_proj.q4, m.up_proj.q4, out, None, True)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / (reps * len(mlps))
    flops = 2 * 2.0 * S * h * I
    tf = flops / us / 1e6
    traffic = None
    pmc, pmc_src = newest_pmc_profile()                           # L2 <-> fabric bytes of the same kernel at the 7B shape
    try:
        if pmc is not None and (h, I, S) == (4096, 11008, 2048):
            traffic = [v["hbm_bytes_per_launch"] for k, v in pmc["prefill"].items() if "q4_gemm_t16d2" in k][0]
    except Exception:
        traffic = None
    return {"bound": "mfma", "kernel": "q4_gemm_t16d2_kernel (fused int4 dequant + gate/up MFMA GEMMs + SiLU*mul, one launch per layer)",
            "achieved": round(tf, 1), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(tf / MFMA_PEAK_TFLOPS, 4),
            "launches": reps * len(mlps), "avg_launch_us": round(us, 2), "flops_per_launch": int(flops), "rows": S,
            "traffic": traffic, "algorithmic_bytes_per_launch": int(h * I + 2.5 * 2 * (h // 128) * I + 2 * S * (h + I)),
            "traffic_source": pmc_src if traffic else None}


def gemv_roofline_probe(model, groupsize, dev, tokens=3):
    """One token's worth of q4 GEMV launches (every layer's 7 matrices, so the weights stream from HBM exactly as in
    decode: 3.4 GB >> 256 MB Infinity Cache), each launch bracketed by its own pair of HIP events on the stream the
    kernels run on (torch's current stream).  achieved = algorithmic bytes per launch / mean launch duration."""
    from exllama_amd.cuda_ext import exllama_ext as ext
    mats = []
    for layer in model.layers:
        a, m = layer.self_attn, layer.mlp
        mats += [a.q_proj, a.k_proj, a.v_proj, a.o_proj, m.gate_proj, m.up_proj, m.down_proj]
    xs = {}
    outs = {}
    for lin in mats:
        xs.setdefault(lin.height, torch.randn(1, lin.height, device=dev).half())
        outs.setdefault(lin.width, torch.empty(1, lin.width, dtype=torch.float16, device=dev))
    for lin in mats:                                              # warm-up pass
        ext.q4_matmul_gemv(xs[lin.height], lin.q4, outs[lin.width])
    torch.cuda.synchronize()
    pairs = []
    # keep the GPU busy while the host enqueues, so that event-to-event time is kernel time, not host launch latency
    torch.cuda._sleep(int(2.0e9 * 0.08 * tokens))
    for _ in range(tokens):
        for lin in mats:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            ext.q4_matmul_gemv(xs[lin.height], lin.q4, outs[lin.width])
            e1.record()
            pairs.append((lin, e0, e1))
    torch.cuda.synchronize()
    total_ms = sum(e0.elapsed_time(e1) for _, e0, e1 in pairs)
    total_bytes = sum(algorithmic_bytes_per_matmul(lin.height, lin.width, groupsize) for lin, _, _ in pairs)
    n = len(pairs)
    avg_us = total_ms * 1e3 / n
    achieved = total_bytes / (total_ms / 1e3) / 1e9
    return {"bound": "hbm", "kernel": "q4_gemv_kernel (+ q4_gemv_reduce_kernel when split-K)", "achieved": round(achieved, 1),
            "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None,
            "launches": n, "avg_launch_us": round(avg_us, 3), "algorithmic_bytes_per_launch": int(total_bytes / n),
            "note": "per-launch HIP events over one token's 7 x layers GEMV launches, x%d tokens; peak = 8.0 TB/s spec "
                    "(6.29 TB/s measured copy => frac_of_measured = %.4f)" % (tokens, achieved / 6290.0)}


def cpu_baseline(dims, groupsize, sample_layers=2, prompt=128, gen=4, ctx=2048):
    """The CPU oracle (a port: the reference has no CPU path) on a bounded sample: `sample_layers` layers of the same
    architecture -- the op sequence of the reference's forward pass (oracle/model_oracle.py composes the oracle's
    cuda_ext-shaped ops in model.py's order) -- a 128-token prompt (BASELINE configs[0]) and a few decode tokens AT THE
    BENCHMARK'S CONTEXT (`ctx` keys already in the cache), weights dequantised once up front.  Extrapolated linearly to
    the full depth: per-layer time x L + measured head time."""
    import numpy as np
    from exllama_amd import synth
    from oracle.model_oracle import OracleLlama
    try:
        from threadpoolctl import threadpool_info
        threads = max([p.get("num_threads", 1) for p in threadpool_info()] + [1])
    except Exception:
        threads = os.cpu_count() or 1
    tensors = synth.make_checkpoint(dims, groupsize=groupsize, act_order=False, seed=0, device="cpu", zeros="sym",
                                    num_layers=sample_layers, nibbles="centered")
    m = OracleLlama(synth.config_dict(dims, sample_layers), tensors, max_seq_len=max(prompt, ctx) + gen)
    m.prepare()
    ids = np.random.RandomState(0).randint(0, min(31999, dims.vocab_size - 1), size=(1, prompt))
    t0 = time.perf_counter()
    hidden = m.embed[ids]
    for i in range(sample_layers):
        hidden = m.layer_forward(i, hidden)
    t_layers_prefill = time.perf_counter() - t0
    rs = np.random.RandomState(1)
    for i in range(sample_layers):                                # a full-length context: the decode step reads `ctx` keys per layer
        m.kc[i][:, :, :ctx] = (rs.standard_normal(m.kc[i][:, :, :ctx].shape) * 0.5).astype(np.float16)
        m.vc[i][:, :, :ctx] = (rs.standard_normal(m.vc[i][:, :, :ctx].shape) * 0.5).astype(np.float16)
    m.past = ctx
    t0 = time.perf_counter()
    for _ in range(gen):
        hd = m.embed[ids[:, :1]]
        for i in range(sample_layers):
            hd = m.layer_forward(i, hd)
        m.past += 1
    t_layers_decode = (time.perf_counter() - t0) / gen
    t0 = time.perf_counter()
    m.past = 0
    _ = (hidden[:, -1].astype(np.float32) @ m.lm_head.astype(np.float32).T)
    t_head = time.perf_counter() - t0
    Lfull = dims.num_hidden_layers
    prefill_s = t_layers_prefill / sample_layers * Lfull + t_head
    decode_s = t_layers_decode / sample_layers * Lfull + t_head
    return {"value": round(1.0 / decode_s, 3), "unit": "tokens/s", "cores": int(threads), "kind": "port",
            "prefill_tokens_per_s": round(prompt / prefill_s, 2),
            "decode_context": ctx,
            "sample": (f"{sample_layers} of {Lfull} layers of the same shapes, {prompt}-token prompt, then {gen} decode tokens at context {ctx}, "
                       f"numpy/OpenBLAS fp32 GEMMs on weights dequantised once (untimed), "
                       + (f"extrapolated x{Lfull / sample_layers:.0f} layers" if sample_layers < Lfull else "full depth, nothing extrapolated")
                       + f" + lm_head; CPU: {os.cpu_count()} logical cores visible; measured once at FULL depth (bench.py --cpu-baseline-full, "
                       "profiles/r03_cpu_baseline_full_depth.json): 0.225 tokens/s -- a 2-layer sample keeps its weights cache-resident and reads 10-18 % high")}


if __name__ == "__main__":
    main()
