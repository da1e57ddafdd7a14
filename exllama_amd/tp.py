"""Tensor parallelism for the 4-bit Llama path: ONE model's matrices split across the ranks of a process group (one process
per GPU, `torch.distributed` backend "nccl" = RCCL over xGMI on MI355X, "gloo" in CPU tests).  Not in the reference (its
multi-GPU mode is the layer split of model.py:636-668, which adds capacity but no speed at batch 1; doc/TODO.md:19 lists
multi-GPU matmul as open): SURVEY.md 8 row N4, second half.

Partitioning (Megatron-style, restated for GPTQ tensors):
  * q / k / v / gate / up projections: split by OUTPUT columns (heads; intermediate columns in whole quantisation groups).
    qweight[K/8, N], qzeros[G, N/8] and scales[G, N] are cut along N; g_idx is untouched.
  * o / down projections: split by INPUT rows -- the rows that multiply the columns this rank produced -- so a layer needs no
    exchange between its two matmuls; every rank ends with a partial sum of the full output and ONE all-reduce per half layer
    (2 per layer) restores the replicated residual stream.  qweight is cut along K/8, qzeros / scales along their group axis.
  * act-order (g_idx) matrices cannot be cut by rows as they are: the rows of a quantisation group are scattered over the input
    features, a row range holds ragged pieces of many groups.  down_proj's row permutation is therefore FOLDED into the column
    order of its producers first (exllama_amd.model._fold_act_order_down_proj, round 4: gate / up columns and down rows in
    down_proj's sequential order, bit-identical results) -- after which down_proj is an ordinary matrix, cut by rows, in the
    executor like any other.  o_proj's input is the attention output, whose column order belongs to the heads: an act-order o_proj
    is cut by output columns instead; the rank then needs the FULL input (all-gather of the attention output) and the outputs are
    all-gathered ("gather mode": two all-gathers instead of one all-reduce, op-by-op path only).
  * lm_head is cut by vocabulary rows (the logits are all-gathered); embedding (one row read per token) and norms are replicated.
The KV cache of a rank holds its own kv heads only.

At batch 1 the win is HBM streaming time: every rank reads 1/W of the weights per token.  The price is two collectives per
layer on a 16 KiB vector: latency-bound on xGMI (point-to-point links: a ring all-reduce of this size costs ~2 (W - 1) hops).
"""

import copy
import math
import os

import torch


def _even_bounds(units, world):
    """`units` whole units over `world` ranks, earlier ranks take the remainder: [(lo, hi), ...] in units."""
    base, rem = divmod(units, world)
    out, start = [], 0
    for r in range(world):
        n = base + (1 if r < rem else 0)
        out.append((start, start + n))
        start += n
    return out


class TPPlan:
    """Which heads / kv heads / intermediate columns one rank owns."""

    def __init__(self, config_dict, rank, world, groupsize):
        heads = config_dict["num_attention_heads"]
        kv = config_dict.get("num_key_value_heads", heads)
        inter = config_dict["intermediate_size"]
        hidden = config_dict["hidden_size"]
        self.rank, self.world = rank, world
        self.head_dim = config_dict.get("head_dim", hidden // heads)
        if kv % world or heads % world:
            raise ValueError(f"tensor parallel: {heads} heads / {kv} kv heads do not divide by {world} ranks")
        self.heads = (rank * heads // world, (rank + 1) * heads // world)
        self.kv_heads = (rank * kv // world, (rank + 1) * kv // world)
        # intermediate columns: whole quantisation groups of down_proj and whole 128-row blocks of the T16 layout
        gs = groupsize if groupsize and 0 < groupsize < inter else 128
        unit = math.lcm(128, gs)
        if inter % unit:
            raise ValueError(f"tensor parallel: intermediate size {inter} is not a multiple of {unit}")
        if inter // unit < world:
            raise ValueError(f"tensor parallel: {inter // unit} blocks of {unit} intermediate columns cannot feed {world} ranks")
        self.inter_all = [(lo * unit, hi * unit) for lo, hi in _even_bounds(inter // unit, world)]
        self.inter = self.inter_all[rank]
        self.hidden = hidden
        self.hidden_cols_all = [(lo * 128, hi * 128) for lo, hi in _even_bounds(hidden // 128, world)]   # gather mode: output columns of o / down
        self.hidden_cols = self.hidden_cols_all[rank]
        self.hidden_sizes = [hi - lo for lo, hi in self.hidden_cols_all]
        self.inter_sizes = [hi - lo for lo, hi in self.inter_all]
        self.inter_full = inter
        vocab = config_dict["vocab_size"]
        vunit = 32 if vocab % 32 == 0 and vocab // 32 >= world else 1       # lm_head rows per rank (the head kernel works on 32-row blocks)
        self.vocab_all = [(lo * vunit, hi * vunit) for lo, hi in _even_bounds(vocab // vunit, world)]
        self.vocab = self.vocab_all[rank]
        self.vocab_sizes = [hi - lo for lo, hi in self.vocab_all]


def _is_act_order(g_idx, groupsize):
    if g_idx is None:
        return False
    g = g_idx.cpu().to(torch.int64)
    if bool((g == 0).all()):
        return False
    return not bool((g == torch.arange(g.numel()) // groupsize).all())


def _cols(t, key, lo, hi, out):
    """Column range [lo, hi) of a GPTQ linear."""
    out[key + ".qweight"] = t[key + ".qweight"][:, lo:hi].clone()          # (a full-width slice would be the caller's tensor itself)
    out[key + ".qzeros"] = t[key + ".qzeros"][:, lo // 8:hi // 8].contiguous()
    out[key + ".scales"] = t[key + ".scales"][:, lo:hi].contiguous()
    if key + ".g_idx" in t:
        out[key + ".g_idx"] = t[key + ".g_idx"]
    if key + ".bias" in t:
        out[key + ".bias"] = t[key + ".bias"][lo:hi].contiguous()


def _rows(t, key, lo, hi, out, rank):
    """Row range [lo, hi) of a GPTQ linear without act-order (whole groups)."""
    qw = t[key + ".qweight"]
    groups = t[key + ".qzeros"].shape[0]
    gs = qw.shape[0] * 8 // groups
    if groups > 1 and (lo % gs or hi % gs):
        raise ValueError(f"{key}: rows {lo}:{hi} do not fall on quantisation-group boundaries (groupsize {gs})")
    # clone(): a row slice of a contiguous tensor is a VIEW, and make_q4 re-tiles qweight in place -- the shard must not share
    # storage with the caller's checkpoint (column slices copy anyway)
    out[key + ".qweight"] = qw[lo // 8:hi // 8].clone()
    g_idx = t.get(key + ".g_idx")
    # an all-zero g_idx is the "no act-order" placeholder some checkpoints carry (the reference and Ex4bitLinear treat it as
    # absent): it is passed through unchanged; only a real sequential index is rebased to the shard's first group
    placeholder = g_idx is not None and not bool((g_idx != 0).any())
    if groups > 1:
        out[key + ".qzeros"] = t[key + ".qzeros"][lo // gs:hi // gs].clone()
        out[key + ".scales"] = t[key + ".scales"][lo // gs:hi // gs].clone()
        if g_idx is not None:
            out[key + ".g_idx"] = g_idx[lo:hi].clone() if placeholder else (g_idx[lo:hi] - lo // gs).contiguous()
    else:                                                           # one group for the whole K: every rank keeps it
        out[key + ".qzeros"] = t[key + ".qzeros"]
        out[key + ".scales"] = t[key + ".scales"]
        if g_idx is not None:
            out[key + ".g_idx"] = g_idx[lo:hi].clone()
    if key + ".bias" in t and rank == 0:                             # a bias is added once
        out[key + ".bias"] = t[key + ".bias"]


def shard_tensors(tensors, config_dict, rank, world):
    """The checkpoint one rank loads (keys unchanged), plus its TPPlan.  `tensors`: the full checkpoint dict."""
    k0 = "model.layers.0.mlp.down_proj"
    groups = tensors[k0 + ".qzeros"].shape[0]
    groupsize = tensors[k0 + ".qweight"].shape[0] * 8 // groups if groups > 1 else None
    plan = TPPlan(config_dict, rank, world, groupsize)
    hd = plan.head_dim
    out = {}
    done = set()
    plan.fold_maps = {}                                             # layer -> down_proj's row map folded into gate / up (new row -> old row)
    from .model import _fold_act_order_down_proj
    tensors = dict(tensors)                                         # (folded MLP tensors replace the caller's in this view only)
    for i in range(config_dict["num_hidden_layers"]):
        p = f"model.layers.{i}."
        gd = tensors.get(p + "mlp.down_proj.g_idx")
        if gd is not None and _is_act_order(gd, tensors[p + "mlp.down_proj.qweight"].shape[0] * 8 // tensors[p + "mlp.down_proj.qzeros"].shape[0]):
            for leaf in ("gate_proj", "up_proj", "down_proj"):      # the fold works in place: on copies
                for suffix in (".qweight", ".qzeros", ".scales"):
                    tensors[p + "mlp." + leaf + suffix] = tensors[p + "mlp." + leaf + suffix].clone()
            fm = _fold_act_order_down_proj(tensors, p + "mlp")
            if fm is not None:
                plan.fold_maps[i] = fm
        _cols(tensors, p + "self_attn.q_proj", plan.heads[0] * hd, plan.heads[1] * hd, out)
        _cols(tensors, p + "self_attn.k_proj", plan.kv_heads[0] * hd, plan.kv_heads[1] * hd, out)
        _cols(tensors, p + "self_attn.v_proj", plan.kv_heads[0] * hd, plan.kv_heads[1] * hd, out)
        _cols(tensors, p + "mlp.gate_proj", plan.inter[0], plan.inter[1], out)
        _cols(tensors, p + "mlp.up_proj", plan.inter[0], plan.inter[1], out)
        for key, (lo, hi) in ((p + "self_attn.o_proj", (plan.heads[0] * hd, plan.heads[1] * hd)), (p + "mlp.down_proj", plan.inter)):
            g = tensors.get(key + ".g_idx")
            gcount = tensors[key + ".qzeros"].shape[0]
            gs = tensors[key + ".qweight"].shape[0] * 8 // gcount
            if _is_act_order(g, gs):
                _cols(tensors, key, plan.hidden_cols[0], plan.hidden_cols[1], out)          # gather mode
            else:
                _rows(tensors, key, lo, hi, out, rank)
        for leaf in ("self_attn.q_proj", "self_attn.k_proj", "self_attn.v_proj", "self_attn.o_proj", "mlp.gate_proj", "mlp.up_proj", "mlp.down_proj"):
            for suffix in (".qweight", ".qzeros", ".scales", ".g_idx", ".bias"):
                done.add(p + leaf + suffix)
    for k, v in tensors.items():
        if k not in done:
            out[k] = v                                               # embedding, norms: replicated
    # lm_head: this rank's vocabulary rows (0.26 of 3.6 GB per token at 7B, more than the layers' share at 8 ranks); the logits
    # are all-gathered (ExLlama.head / the executor's last piece)
    out["lm_head.weight"] = tensors["lm_head.weight"][plan.vocab[0]:plan.vocab[1]].clone()
    return out, plan


def shard_config_dict(config_dict, plan):
    """config.json of the local (narrower) model a rank builds: its own heads and intermediate columns, the full hidden size."""
    c = copy.deepcopy(config_dict)
    c["head_dim"] = plan.head_dim
    c["num_attention_heads"] = plan.heads[1] - plan.heads[0]
    c["num_key_value_heads"] = plan.kv_heads[1] - plan.kv_heads[0]
    c["intermediate_size"] = plan.inter[1] - plan.inter[0]
    return c


class TensorParallel:
    """What a model needs at run time: rank, world and the two collectives, over `torch.distributed` (or any object with
    all_reduce(tensor) / all_gather(list, tensor): the in-process emulation of tests/tp_emul.py)."""

    def __init__(self, plan, dist, group=None):
        self.plan, self.rank, self.world = plan, plan.rank, plan.world
        self.dist, self.group = dist, group
        # EXL_TP_ALWAYS_COLLECTIVE=1: issue the collectives even with one rank (exercises RCCL + graph capture on a one-GPU box)
        self.always = os.environ.get("EXL_TP_ALWAYS_COLLECTIVE") == "1" and dist is not None
        self._f32 = {}

    def _wide(self, t):
        """Persistent fp32 staging buffer for the partial sums of `t` (one per shape and device: no allocation inside a captured graph)."""
        key = (tuple(t.shape), str(t.device))
        buf = self._f32.get(key)
        if buf is None:
            buf = self._f32[key] = torch.empty(t.shape, dtype=torch.float32, device=t.device)
        return buf

    def all_reduce(self, t):
        """Sum over ranks, in place.  fp16 partial sums travel and are added as fp32 and the total is rounded to fp16 ONCE: the
        payload is 8-32 KiB per half layer (latency-bound on xGMI, the width costs nothing measurable), and a backend that adds
        fp16 pairwise rounds once per rank -- an error that grows with the node size (measured 1.5e-2 x scale at four ranks against
        4.3e-3 at two before this; tests/test_multiproc_gpu.py now holds every world size to the same bound)."""
        if self.world > 1 or self.always:
            wide = self._wide(t) if t.dtype == torch.float16 else t
            if wide is not t:
                wide.copy_(t)
            if self.group is None:
                self.dist.all_reduce(wide)
            else:
                self.dist.all_reduce(wide, group=self.group)
            if wide is not t:
                t.copy_(wide)
        return t

    def all_gather_into(self, out, t, sizes):
        """out (1-D, sum(sizes) elements) = concatenation of the ranks' 1-D tensors t (sizes[r] elements each); capturable."""
        if self.world == 1 and not self.always:
            out.copy_(t)
            return out
        if len(set(sizes)) == 1 and hasattr(self.dist, "all_gather_into_tensor"):
            if self.group is None:
                self.dist.all_gather_into_tensor(out, t.contiguous())
            else:
                self.dist.all_gather_into_tensor(out, t.contiguous(), group=self.group)
            return out
        out.copy_(self.all_gather_last(t, sizes))
        return out

    def all_gather_last(self, t, sizes=None):
        """Concatenate the ranks' tensors along the last dimension (equal sizes unless `sizes` lists them)."""
        if self.world == 1:
            return t
        t = t.contiguous()
        if sizes is None or len(set(sizes)) == 1:
            parts = [torch.empty_like(t) for _ in range(self.world)]
            if self.group is None:
                self.dist.all_gather(parts, t)
            else:
                self.dist.all_gather(parts, t, group=self.group)
            return torch.cat(parts, dim=-1)
        # ragged: pad to the widest, gather, trim
        width = max(sizes)
        padded = torch.zeros(t.shape[:-1] + (width,), dtype=t.dtype, device=t.device)
        padded[..., :t.shape[-1]] = t
        parts = [torch.empty_like(padded) for _ in range(self.world)]
        if self.group is None:
            self.dist.all_gather(parts, padded)
        else:
            self.dist.all_gather(parts, padded, group=self.group)
        return torch.cat([p[..., :n] for p, n in zip(parts, sizes)], dim=-1)
