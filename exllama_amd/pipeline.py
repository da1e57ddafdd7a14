"""Layer split across PROCESSES (one process per GPU): the reference's device_map / auto-split (model.py:636-668,
770-801) restated for torch.distributed.  Rank r holds a contiguous run of decoder layers; per forward pass there is
exactly ONE exchange per rank boundary -- the hidden states [bsz, q_len, hidden] fp16 -- as a point-to-point
send/recv between neighbours (backend "nccl" = RCCL over xGMI on MI355X; "gloo" in the CPU tests), no collective on the
data path.  At batch 1 the stages run one after the other (capacity, not speed-up): throughput scaling is
done with replicas (bench.py --gpus N), this module is for models split by choice.

A stage is any object with the three methods exllama_amd.model.ExLlama provides: embed(ids), forward_layers(hidden,
cache), head(hidden, last_id_only).  Every rank constructs its ExLlama from the tensors of ITS layers only
(`stage_tensors`), so the weights of the other ranks never touch this GPU.
"""

import re

import torch


def split_layers(num_layers, world):
    """Contiguous, near-equal layer ranges [(first, last_exclusive), ...], earlier ranks get the remainder
    (the greedy fill of the reference's auto-split with equal budgets)."""
    assert world >= 1 and num_layers >= world, "need at least one layer per rank"
    base, rem = divmod(num_layers, world)
    out, start = [], 0
    for r in range(world):
        n = base + (1 if r < rem else 0)
        out.append((start, start + n))
        start += n
    return out


_LAYER_KEY = re.compile(r"^model\.layers\.(\d+)\.(.*)$")


def stage_tensors(tensors, first, last):
    """The subset of a checkpoint dict one rank needs, with its layers re-indexed from 0: embedding, final norm and lm_head
    are kept everywhere (small; only rank 0 / the last rank use them)."""
    out = {}
    for k, v in tensors.items():
        m = _LAYER_KEY.match(k)
        if m is None:
            out[k] = v
        else:
            i = int(m.group(1))
            if first <= i < last:
                out[f"model.layers.{i - first}.{m.group(2)}"] = v
    return out


class HostStagedGroup:
    """A `torch.distributed`-shaped object for process groups whose backend cannot move DEVICE tensors point to point (gloo:
    only broadcast / all_reduce take them): every call stages through pinned-size host copies -- device -> host (which drains
    the stream the kernels ran on), the backend's own call on the host tensor, host -> device.  StageHop, LayerSplitRunner and
    exllama_amd.tp.TensorParallel take it wherever they take `torch.distributed`; results are the backend's (gloo adds fp16
    all-reduce operands pairwise in fp16, like RCCL's ring).  With backend "nccl" (= RCCL) use `torch.distributed` itself: the
    transfers then are stream-ordered device kernels and can be captured into the rank's hipGraph -- these calls cannot
    (capture_hop=False / eager token steps).  This is what lets N processes share ONE GPU in tests/test_multiproc_gpu.py
    (RCCL refuses two ranks on one device) and what a box without a working RCCL falls back to."""

    def __init__(self, dist, group=None):
        self.dist, self.group = dist, group
        self.ReduceOp = dist.ReduceOp

    def get_rank(self): return self.dist.get_rank(self.group)
    def get_world_size(self): return self.dist.get_world_size(self.group)
    def get_backend(self): return self.dist.get_backend(self.group) + " (host-staged)"
    def barrier(self): return self.dist.barrier(group=self.group)

    def send(self, t, dst):
        self.dist.send(t.detach().to("cpu"), dst=dst, group=self.group)

    def recv(self, t, src):
        h = torch.empty(t.shape, dtype=t.dtype, device="cpu")
        self.dist.recv(h, src=src, group=self.group)
        t.copy_(h)

    def broadcast(self, t, src):
        h = t.detach().to("cpu")
        self.dist.broadcast(h, src=src, group=self.group)
        t.copy_(h)

    def all_reduce(self, t, op=None, group=None):
        h = t.detach().to("cpu")
        self.dist.all_reduce(h, op=self.dist.ReduceOp.SUM if op is None else op, group=self.group)
        t.copy_(h)

    def all_gather(self, parts, t, group=None):
        hs = [torch.empty(p.shape, dtype=p.dtype, device="cpu") for p in parts]
        self.dist.all_gather(hs, t.detach().to("cpu").contiguous(), group=self.group)
        for p, h in zip(parts, hs):
            p.copy_(h)


class StageHop:
    """This rank's point-to-point exchanges of ONE token step of a layer split, issued on the executor's own device buffers
    (ExLlama.decode_hop_buffers): before() ahead of the first stage's kernels, after() behind the last stage's.

        hidden-state chain:      rank r > 0 receives the fp16 hidden state [1, 1, hidden] from r - 1, rank r < last sends its own on
        token ring (token_ring): in addition the last rank picks the greedy token from its logits (argmax on the device) and sends
                                 it to rank 0, which receives it where its embedding lookup reads the token: no host sees a token

    exllama_amd.model.ExLlama.enable_decode_graph(hop=...) captures these calls into the rank's hipGraph when the process group
    allows it (backend "nccl" = RCCL: stream-ordered point-to-point kernels), so a replay is receive -> kernels -> send; with a
    backend that cannot be captured (gloo) or a failed capture the same calls run eagerly around the replay."""

    def __init__(self, dist, rank, world, token_ring=False):
        self.dist, self.rank, self.world, self.token_ring = dist, rank, world, token_ring
        self.next_tok = None                        # last rank: the token picked by the latest step (device tensor [1, 1] int64)

    def bind(self, stage):
        self.tok, self.hid_in, self.hid_out, self.logits = stage.decode_hop_buffers()
        if self.rank == self.world - 1 and self.next_tok is None:
            self.next_tok = torch.zeros((1, 1), dtype=torch.int64, device=self.logits.device)

    def before(self, st=None):
        if self.rank > 0:
            self.dist.recv(self.hid_in, src=self.rank - 1)
        elif self.token_ring and self.world > 1:
            self.dist.recv(self.tok, src=self.world - 1)

    def after(self, st=None):
        if self.rank < self.world - 1:
            self.dist.send(self.hid_out, dst=self.rank + 1)
            return
        torch.argmax(self.logits.view(1, -1), dim=-1, keepdim=True, out=self.next_tok)
        if self.token_ring:
            if self.world > 1:
                self.dist.send(self.next_tok, dst=0)
            else:
                self.tok.copy_(self.next_tok)                          # one rank: the ring is a copy


class LayerSplitRunner:
    """Drives one rank's stage.  forward() returns fp32 logits on the LAST rank and None elsewhere."""

    def __init__(self, stage, cache, dist, hidden_size, device, dtype=torch.float16):
        self.stage, self.cache, self.dist = stage, cache, dist
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.hidden_size, self.device, self.dtype = hidden_size, device, dtype

    def enable_decode_executor(self, use_graph=True, capture_hop=True, token_ring=False):
        """Single-token steps of this rank run through the native decode executor (exllama_amd.model.ExLlama.enable_decode_graph)
        as one link of the split: rank 0 owns the embedding, the last rank the head, the fp16 hidden state [1, 1, hidden]
        travels between them exactly as in forward() -- received straight into / sent straight from the executor's buffers
        (StageHop), and, with capture_hop, INSIDE the rank's captured graph: no host round trip per token and boundary.
        token_ring = True prepares generate_greedy() (the token travels from the last rank to rank 0 inside the graphs too);
        single-token forward() calls then need token_ring = False (they bring their token from the host)."""
        self.hop = StageHop(self.dist, self.rank, self.world, token_ring)
        if self.world > 1 and use_graph and capture_hop:
            self._warm_up_links(token_ring)
        self.stage.enable_decode_graph(self.cache, use_graph=use_graph, first_stage=self.rank == 0, last_stage=self.rank == self.world - 1,
                                       hop=self.hop, hop_capture=capture_hop)
        self._executor = True

    def _warm_up_links(self, token_ring):
        """One eager exchange over every point-to-point pair the captured token step will use (hidden state r -> r + 1, token last -> 0)
        BEFORE anything is captured: RCCL sets a pair's connection up at its first use (allocations, IPC handles), which has no place
        inside a stream capture.  Same order as a token step, so it cannot deadlock; the payloads are scratch."""
        last = self.world - 1
        hid = torch.zeros((1, 1, self.hidden_size), dtype=self.dtype, device=self.device)
        tok = torch.zeros((1, 1), dtype=torch.int64, device=self.device)
        if self.rank > 0:
            self.dist.recv(hid, src=self.rank - 1)
        if self.rank < last:
            self.dist.send(hid, dst=self.rank + 1)
        if token_ring:
            if self.rank == last:
                self.dist.send(tok, dst=0)
            elif self.rank == 0:
                self.dist.recv(tok, src=last)
        if torch.cuda.is_available() and str(self.device).startswith("cuda"):
            torch.cuda.synchronize(self.device)

    def _decode_token(self, input_ids):
        if self.hop.token_ring:
            raise RuntimeError("this executor was enabled with token_ring=True: use generate_greedy(), or enable it without the ring")
        out = self.stage.decode_stage_step(self.cache, input_ids=input_ids if self.rank == 0 else None)
        return out if self.rank == self.world - 1 else None

    def generate_greedy(self, first_token, num_tokens):
        """num_tokens greedy tokens after `first_token` (the token at the cache's position -- the same value on every rank, e.g. from
        next_token()) with NO host in the loop: every rank runs its step num_tokens times -- rank 0: receive the token from the
        last rank, embedding + its layers, send the hidden state; middle ranks: receive, layers, send; last rank: receive, layers +
        head, argmax, send the token to rank 0 -- each step one graph replay when the hand-off was captured.  The reference's
        loop does forward + torch.argmax per token across all its devices (test_benchmark_inference.py:188-191).  Returns the
        tokens (LongTensor [num_tokens]) on every rank."""
        if not getattr(self, "_executor", False) or not self.hop.token_ring:
            raise RuntimeError("generate_greedy needs enable_decode_executor(token_ring=True)")
        last = self.world - 1
        first = first_token.view(1, 1).to(device=self.device, dtype=torch.int64)
        if self.world > 1:
            if self.rank == last:
                self.dist.send(first, dst=0)                            # rank 0's step ALWAYS receives its token: this one is the first
        else:
            self.hop.tok.copy_(first)
        picked = []
        for _ in range(num_tokens):
            self.stage.decode_stage_step(self.cache)                     # inputs and outputs travel through the hop
            if self.rank == last:
                picked.append(self.hop.next_tok.clone())
        if self.world > 1 and self.rank == 0:
            spare = torch.zeros((1, 1), dtype=torch.int64, device=self.device)
            self.dist.recv(spare, src=last)                              # the token of the final step is on the wire: take it off
        toks = torch.cat(picked).view(-1) if picked else torch.zeros((num_tokens,), dtype=torch.int64, device=self.device)
        if self.world > 1:
            self.dist.broadcast(toks, src=last)
        return toks

    def forward(self, input_ids, last_id_only=True):
        """input_ids [bsz, q_len] must be the same on every rank (its VALUES are only read on rank 0)."""
        bsz, q_len = input_ids.shape
        if getattr(self, "_executor", False) and bsz == 1 and q_len == 1:
            return self._decode_token(input_ids)
        if self.rank == 0:
            hidden = self.stage.embed(input_ids)
        else:
            hidden = torch.empty((bsz, q_len, self.hidden_size), dtype=self.dtype, device=self.device)
            self.dist.recv(hidden, src=self.rank - 1)                  # the one hand-off per boundary
        hidden = self.stage.forward_layers(hidden, self.cache)
        if self.cache is not None:
            self.cache.current_seq_len += q_len
        if self.rank < self.world - 1:
            self.dist.send(hidden.contiguous(), dst=self.rank + 1)
            return None
        return self.stage.head(hidden, last_id_only)

    def next_token(self, logits):
        """Greedy token chosen on the last rank, made known to every rank (rank 0 embeds it next): one 8-byte broadcast."""
        tok = torch.zeros((1, 1), dtype=torch.int64, device=self.device)
        if self.rank == self.world - 1:
            tok = logits[0, -1].argmax().view(1, 1).to(torch.int64)
        self.dist.broadcast(tok, src=self.world - 1)
        return tok
