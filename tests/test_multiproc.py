"""N > 1 path of bench.py on CPU: two processes over gloo (127.0.0.1) run the contract's reduction -- every rank's
timings, MAX over ranks, whole-job rate = ranks x tokens / slowest time.  The GPU work itself is per-replica and has no
collective on the data path, so this is all the cross-rank logic there is."""
import os
import socket

import pytest

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    # rank r pretends its phases took (10 + r, 20 + 2r, 30 + 3r, 40 + 4r) ms
    local = [10.0 + rank, 20.0 + 2 * rank, 30.0 + 3 * rank, 40.0 + 4 * rank]
    red = bench.reduce_over_ranks(local, dist, "cpu")
    rep = bench.per_rank_report(local[:2], dist, "cpu", ("prefill_ms", "decode_ms"))      # what --tensor-parallel / --layer-split print
    dist.barrier()
    if rank == 0:
        out.put((red, rep))
    dist.destroy_process_group()


def test_rank_reduction_is_max_and_rates_are_whole_job():
    world = 2
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, PORT, out)) for r in range(world)]
    for p in procs:
        p.start()
    red, rep = out.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert red == [11.0, 22.0, 33.0, 44.0]                      # MAX over ranks of every phase
    assert rep["rccl_ranks"] == 2 and rep["backend"] == "gloo"  # (the communicator the run used; "nccl" = RCCL on the GPU box)
    assert rep["per_rank"] == [{"rank": 0, "prefill_ms": 10.0, "decode_ms": 20.0}, {"rank": 1, "prefill_ms": 11.0, "decode_ms": 22.0}]
    import bench
    rates = bench.whole_job_rates(world, 2048, 128, red[1], red[2], red[3])
    assert abs(rates["prefill"] - world * 2048 / 0.022) < 1e-6
    assert abs(rates["worst"] - world * 128 / 0.033) < 1e-6
    assert abs(rates["best"] - world * 128 / 0.044) < 1e-6


def test_single_process_reduction_is_identity():
    import bench
    assert bench.reduce_over_ranks([1.0, 2.0], None, "cpu") == [1.0, 2.0]


PORT = _free_port()


# ---- layer split across processes (exllama_amd/pipeline.py): plumbing test with a torch-only stage on CPU ------------
class _ToyStage:
    """embed / forward_layers / head with plain torch ops, deterministic weights (seeded per GLOBAL layer index)."""

    def __init__(self, first, last, hidden=32, vocab=50):
        g = torch.Generator().manual_seed(123)
        self.emb = torch.randn(vocab, hidden, generator=g).half()
        self.out = torch.randn(vocab, hidden, generator=g).half()
        self.ws = []
        for i in range(8):
            w = (torch.randn(hidden, hidden, generator=g) * 0.1).half()
            if first <= i < last:
                self.ws.append(w)

    def embed(self, ids):
        return self.emb[ids]

    def forward_layers(self, hidden, cache):
        for w in self.ws:
            hidden = (hidden.float() + torch.tanh(hidden.float() @ w.float())).half()
        return hidden

    def head(self, hidden, last_id_only=True):
        if last_id_only:
            hidden = hidden[:, -1:, :]
        return hidden.float() @ self.out.float().t()


def _pipe_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from exllama_amd.pipeline import LayerSplitRunner, split_layers
    first, last = split_layers(8, world)[rank]
    runner = LayerSplitRunner(_ToyStage(first, last), None, dist, 32, "cpu")
    ids = torch.tensor([[3, 7, 11, 13]])
    logits = runner.forward(ids)
    toks = []
    for _ in range(3):
        tok = runner.next_token(logits)
        toks.append(int(tok))
        logits = runner.forward(tok)
    if rank == world - 1:
        out.put((toks, logits.numpy()))              # by value: a torch tensor travels as a file descriptor the exiting worker may close first
    dist.barrier()
    dist.destroy_process_group()


def test_layer_split_matches_single_process():
    from exllama_amd.pipeline import split_layers, stage_tensors
    assert split_layers(32, 8) == [(4 * i, 4 * i + 4) for i in range(8)]
    assert split_layers(7, 3) == [(0, 3), (3, 5), (5, 7)]
    t = {"model.embed_tokens.weight": 1, "model.layers.0.a": 2, "model.layers.5.b.qweight": 3, "model.layers.6.c": 4, "lm_head.weight": 5}
    assert stage_tensors(t, 5, 7) == {"model.embed_tokens.weight": 1, "model.layers.0.b.qweight": 3, "model.layers.1.c": 4, "lm_head.weight": 5}
    # reference run: all 8 layers in one process
    full = _ToyStage(0, 8)
    ids = torch.tensor([[3, 7, 11, 13]])
    logits = full.head(full.forward_layers(full.embed(ids), None))
    toks = []
    for _ in range(3):
        tok = logits[0, -1].argmax().view(1, 1)
        toks.append(int(tok))
        logits = full.head(full.forward_layers(full.embed(tok), None))
    world = 2
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_pipe_worker, args=(r, world, port, out)) for r in range(world)]
    for p in procs:
        p.start()
    got_toks, got_logits = out.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got_toks == toks
    assert torch.equal(torch.from_numpy(got_logits), logits)


# ---- the same split with CHECKPOINT-shaped stages: every rank builds its link from pipeline.stage_tensors() of a real GPTQ
# checkpoint layout.  exllama_amd.model.ExLlama has no CPU path by design (the product fails loudly without a HIP device),
# so on CPU the stage is the oracle model behind the same three-method interface (embed / forward_layers / head): what is
# covered here is the cross-rank logic with real tensors -- layer re-indexing, per-rank caches, one fp16 hidden-state
# hand-off per boundary, token broadcast -- against the unsplit model, bit for bit.
class _OracleStage:
    def __init__(self, cfg, tensors, n_layers):
        import numpy as np
        from oracle.model_oracle import OracleLlama
        self.np = np
        self.m = OracleLlama(cfg, tensors, max_seq_len=32, num_layers=n_layers)

    def embed(self, ids):
        return torch.from_numpy(self.m.embed[ids.numpy()])

    def forward_layers(self, hidden, cache):
        h = hidden.numpy()
        for i in range(self.m.L):
            h = self.m.layer_forward(i, h)
        self.m.past += hidden.shape[1]
        return torch.from_numpy(self.np.ascontiguousarray(h))

    def head(self, hidden, last_id_only=True):
        from oracle import exl_oracle as O
        h = hidden.numpy()
        if last_id_only:
            h = h[:, -1:, :]
        b, q, d = h.shape
        hn = O.rms_norm(h.reshape(-1, d), self.m.norm_w, self.m.eps)
        lg = (hn.astype(self.np.float32) @ self.m.lm_head.astype(self.np.float32).T).astype(self.np.float16).astype(self.np.float32)
        return torch.from_numpy(lg.reshape(b, q, -1))


def _ckpt():
    from exllama_amd import synth
    dims = synth.LLAMA_TINY
    L = 4
    return dims, L, synth.make_checkpoint(dims, groupsize=64, act_order=True, seed=31, device="cpu", zeros="rand", num_layers=L)


def _oracle_pipe_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from exllama_amd import synth
    from exllama_amd.pipeline import LayerSplitRunner, split_layers, stage_tensors
    dims, L, tensors = _ckpt()
    first, last = split_layers(L, world)[rank]
    stage = _OracleStage(synth.config_dict(dims, last - first), stage_tensors(tensors, first, last), last - first)
    runner = LayerSplitRunner(stage, None, dist, dims.hidden_size, "cpu")
    ids = torch.tensor([[3, 7, 11, 13, 17]])
    logits = runner.forward(ids)
    toks = []
    for _ in range(3):
        tok = runner.next_token(logits)
        toks.append(int(tok))
        logits = runner.forward(tok)
    if rank == world - 1:
        out.put((toks, logits.numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_layer_split_of_a_real_checkpoint_layout_matches_the_unsplit_model():
    import numpy as np
    from exllama_amd import synth
    from oracle.model_oracle import OracleLlama
    dims, L, tensors = _ckpt()
    full = OracleLlama(synth.config_dict(dims, L), tensors, max_seq_len=32)
    logits = full.forward(np.array([[3, 7, 11, 13, 17]]))
    toks = []
    for _ in range(3):
        t = int(np.argmax(logits[0, -1]))
        toks.append(t)
        logits = full.forward(np.array([[t]]))
    world = 2
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_oracle_pipe_worker, args=(r, world, port, out)) for r in range(world)]
    for p in procs:
        p.start()
    got_toks, got_logits = out.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got_toks == toks
    assert np.array_equal(got_logits, logits)


# ---- the executor-side hand-off (pipeline.StageHop) and the token ring of LayerSplitRunner.generate_greedy: every rank's single-token
# step receives its input and sends its output ON THE STAGE'S OWN BUFFERS (on the GPU: inside the rank's captured hipGraph; RCCL
# point-to-point is capturable), the greedy token travels from the last rank back to rank 0 without the host.  On CPU the stage is
# the oracle model behind the executor's stage interface (enable_decode_graph / decode_hop_buffers / decode_stage_step) and gloo
# carries the exchanges eagerly -- the protocol (who receives what, when; the pre-sent first token; the drained last one; the final
# broadcast) is what is covered, against the unsplit model's greedy tokens.
class _OracleExecutorStage(_OracleStage):
    def enable_decode_graph(self, cache, use_graph=True, first_stage=True, last_stage=True, hop=None, hop_capture=True):
        h, V = self.m.h, self.m.lm_head.shape[0]
        self.first, self.last, self.hop = first_stage, last_stage, hop
        self.tok = torch.zeros((1, 1), dtype=torch.int64)
        self.hid_in = torch.zeros((1, 1, h), dtype=torch.float16)
        self.hid_out = torch.zeros((1, 1, h), dtype=torch.float16)
        self.logits = torch.zeros((1, 1, V), dtype=torch.float32)
        self.steps = 0
        if hop is not None:
            hop.bind(self)

    def decode_hop_buffers(self):
        return self.tok, self.hid_in, self.hid_out, self.logits

    def decode_stage_step(self, cache, input_ids=None, hidden_in=None):
        if input_ids is not None:
            self.tok.copy_(input_ids.view(1, 1))
        if hidden_in is not None:
            self.hid_in.copy_(hidden_in.view(1, 1, -1))
        if self.hop is not None:
            self.hop.before()
        hidden = self.embed(self.tok) if self.first else self.hid_in
        hidden = self.forward_layers(hidden, None)
        if self.last:
            self.logits.copy_(self.head(hidden))
        else:
            self.hid_out.copy_(hidden)
        if self.hop is not None:
            self.hop.after()
        self.steps += 1
        return self.logits.clone() if self.last else self.hid_out


def _ring_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from exllama_amd import synth
    from exllama_amd.pipeline import LayerSplitRunner, split_layers, stage_tensors
    dims, L, tensors = _ckpt()
    first, last = split_layers(L, world)[rank]
    stage = _OracleExecutorStage(synth.config_dict(dims, last - first), stage_tensors(tensors, first, last), last - first)
    runner = LayerSplitRunner(stage, None, dist, dims.hidden_size, "cpu")
    ids = torch.tensor([[3, 7, 11, 13, 17]])
    logits = runner.forward(ids)                                  # the prompt: the chain of forward()
    tok = runner.next_token(logits)                               # known to every rank
    runner.enable_decode_executor(use_graph=False, token_ring=True)
    toks = runner.generate_greedy(tok, 4)
    again = None
    try:
        runner.forward(tok)                                       # a host-fed single token needs the executor without the ring
    except RuntimeError as e:
        again = str(e)
    # the plain chain through the executor stages (no ring): the hidden state arrives through the hop, the token from the host
    runner.enable_decode_executor(use_graph=False, token_ring=False)
    lg = runner.forward(toks[-1].view(1, 1))
    out.put((rank, int(tok), toks.tolist(), again, stage.steps, None if lg is None else lg.numpy().copy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_layer_split_token_ring_generates_the_unsplit_models_tokens(world):
    import numpy as np
    from exllama_amd import synth
    from oracle.model_oracle import OracleLlama
    dims, L, tensors = _ckpt()
    full = OracleLlama(synth.config_dict(dims, L), tensors, max_seq_len=32)
    logits = full.forward(np.array([[3, 7, 11, 13, 17]]))
    want = []
    for _ in range(6):
        t = int(np.argmax(logits[0, -1]))
        want.append(t)
        logits = full.forward(np.array([[t]]))
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ring_worker, args=(r, world, port, out)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted([out.get(timeout=240) for _ in range(world)], key=lambda g: g[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, first, toks, refusal, steps, lg in got:
        assert first == want[0] and toks == want[1:5], (rank, first, toks, want)       # every rank holds the same, right tokens
        assert refusal is not None and "token_ring" in refusal
        assert steps == 1                                          # (the second executor was stepped once)
        if rank == world - 1:                                      # the chain step after the ring continues the same sequence
            assert int(lg[0, -1].argmax()) == want[5]


# ---- pipeline.HostStagedGroup: the torch.distributed-shaped adapter the one-GPU multi-process tests (tests/test_multiproc_gpu.py)
# and RCCL-less boxes use; here with host tensors on three ranks: every call it forwards, and the toy pipeline through it.
def _staged_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from exllama_amd.pipeline import HostStagedGroup, LayerSplitRunner, split_layers
    g = HostStagedGroup(dist)
    assert g.get_rank() == rank and g.get_world_size() == world and g.get_backend().startswith("gloo")
    t = torch.full((5,), float(rank + 1), dtype=torch.float16)
    g.all_reduce(t)
    parts = [torch.zeros(3, dtype=torch.int64) for _ in range(world)]
    g.all_gather(parts, torch.arange(3) + 10 * rank)
    b = torch.tensor([[rank]], dtype=torch.int64)
    g.broadcast(b, src=world - 1)
    ring = torch.zeros(4, dtype=torch.float16)
    if rank == 0:
        g.send(torch.arange(4, dtype=torch.float16), dst=1)
        g.recv(ring, src=world - 1)
    else:
        g.recv(ring, src=rank - 1)
        g.send(ring + 1, dst=(rank + 1) % world)
    m = torch.tensor([float(rank)], dtype=torch.float64)
    g.all_reduce(m, op=g.ReduceOp.MAX)
    first, last = split_layers(8, world)[rank]
    runner = LayerSplitRunner(_ToyStage(first, last), None, g, 32, "cpu")
    logits = runner.forward(torch.tensor([[3, 7, 11, 13]]))
    tok = runner.next_token(logits)
    g.barrier()
    out.put((rank, t.tolist(), [p.tolist() for p in parts], int(b), ring.tolist(), float(m), int(tok)))
    dist.destroy_process_group()


def test_host_staged_group_forwards_every_call():
    world = 3
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_staged_worker, args=(r, world, port, out)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted(out.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    full = _ToyStage(0, 8)
    want_tok = int(full.head(full.forward_layers(full.embed(torch.tensor([[3, 7, 11, 13]])), None))[0, -1].argmax())
    for rank, red, parts, b, ring, m, tok in got:
        assert red == [6.0] * 5
        assert parts == [[0, 1, 2], [10, 11, 12], [20, 21, 22]]
        assert b == world - 1 and m == float(world - 1) and tok == want_tok
        assert ring == [float(v + (world - 1 if rank == 0 else rank - 1)) for v in range(4)]


# ---- bench.py --gpus 2 as the driver launches it (torch.distributed.run, one process per rank), DRY: the process choreography of the
# N > 1 line under gloo on CPU -- both ranks through the barrier / MAX reduction, rank 0 alone launching the sharded sub-runs as
# torch.distributed.run jobs of the same file once the group is gone, their records nested in the ONE JSON line, that line LAST on stdout.
def test_bench_two_ranks_dry_run_prints_one_line_with_the_sharded_records():
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, EXL_BENCH_DRY_RUN="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=root, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.strip().splitlines() if l.strip()]
    json_lines = [l for l in lines if l.startswith("{")]
    assert len(json_lines) == 1 and lines[-1] == json_lines[0], lines[-3:]        # ONE line, and it is the last thing on stdout
    d = json.loads(json_lines[0])
    assert d["n_gpus"] == 2 and d["config"]["DRY_RUN"] and d["scaling"] == "weak"
    # MAX over ranks of the fabricated timings (rank 1 is the slower one), whole-job rate = 2 replicas
    assert abs(d["ms_per_step"] - 310.0) < 1e-6 and abs(d["value"] - 2 * 128 / 0.182) < 0.01 and abs(d["prefill_tokens_per_s"] - 2 * 2048 / 0.026) < 0.1
    sh = d["sharded"]
    assert set(sh) == {"layer_split_7b", "layer_split_65b", "layer_split_33b_g32_actorder", "tensor_parallel_7b"}
    for k, v in sh.items():
        assert "error" not in v, (k, v)
        assert v["n_gpus"] == 2 and v["rccl_ranks"] == 2 and v["backend"] == "gloo" and v["scaling"] == "strong" and v["logits_finite"]
        assert [p["rank"] for p in v["per_rank"]] == [0, 1]
        assert ("layer split x2" in v["decode_mode"]) == k.startswith("layer_split")


def test_bench_gpus_2_without_a_launcher_launches_itself():
    """`python bench.py --gpus 2` with no WORLD_SIZE in the environment (how the driver starts the N = 1 run): bench.py re-executes itself
    under torch.distributed.run instead of dying on the WORLD_SIZE assert -- same ONE line, n_gpus 2 (dry run; --no-sharded keeps it short)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, EXL_BENCH_DRY_RUN="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--no-sharded"],
                       capture_output=True, text=True, timeout=300, cwd=root, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.strip().splitlines() if l.strip()]
    json_lines = [l for l in lines if l.startswith("{")]
    assert len(json_lines) == 1 and lines[-1] == json_lines[0], lines[-3:]
    d = json.loads(json_lines[0])
    assert d["n_gpus"] == 2 and d["config"]["DRY_RUN"] and d["scaling"] == "weak" and "sharded" not in d
    assert abs(d["ms_per_step"] - 310.0) < 1e-6
