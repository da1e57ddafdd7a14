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


class LayerSplitRunner:
    """Drives one rank's stage.  forward() returns fp32 logits on the LAST rank and None elsewhere."""

    def __init__(self, stage, cache, dist, hidden_size, device, dtype=torch.float16):
        self.stage, self.cache, self.dist = stage, cache, dist
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.hidden_size, self.device, self.dtype = hidden_size, device, dtype

    def enable_decode_executor(self, use_graph=True):
        """Single-token steps of this rank run through the native decode executor (exllama_amd.model.ExLlama.enable_decode_graph)
        as one link of the split: rank 0 owns the embedding, the last rank the head, the fp16 hidden state [1, 1, hidden]
        travels between them exactly as in forward()."""
        self.stage.enable_decode_graph(self.cache, use_graph=use_graph, first_stage=self.rank == 0, last_stage=self.rank == self.world - 1)
        self._executor = True

    def _decode_token(self, input_ids):
        hidden_in = None
        if self.rank > 0:
            hidden_in = torch.empty((1, 1, self.hidden_size), dtype=self.dtype, device=self.device)
            self.dist.recv(hidden_in, src=self.rank - 1)
        out = self.stage.decode_stage_step(self.cache, input_ids=input_ids if self.rank == 0 else None, hidden_in=hidden_in)
        if self.rank < self.world - 1:
            self.dist.send(out.contiguous(), dst=self.rank + 1)       # (the send completes before the next step overwrites the buffer)
            return None
        return out

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
