"""ExLlama model API (ExLlamaConfig / ExLlama / ExLlamaCache / ExLlamaDeviceMap) on the MI355X kernels.

Public names, constructor arguments, attributes and `forward()` semantics follow /root/reference/model.py
(citations inline) so code written against the reference -- generator.py, perplexity.py,
test_benchmark_inference.py -- drives this class unchanged.  The implementation is not the reference's:

  * every matmul / norm / RoPE / cache / attention op is a hand-written HIP kernel reached through the C ABI
    (exllama_amd/cuda_ext.py); nothing on the token path falls back to ATen except the embedding gather and the
    fp16 lm_head GEMM ("may stay ATen initially");
  * residual adds are fused into the o_proj / down_proj matmul epilogues for every row count (the reference
    does this only in its rows == 1 fused ops, q4_attn.cu:227 / q4_mlp.cu:194);
  * attention never materialises the score matrix and never copies K/V for GQA;
  * the embedding table lives in HBM next to layer 0 (288 GB per GPU; the reference keeps it on the host and
    pays an H2D copy per forward, model.py:642,1042-1043);
  * single-token decode can be captured into one hipGraph per model (`ExLlama.enable_decode_graph()`), with the
    position read from device memory so the same graph replays at every context length.
"""

import json
import math

import os

import torch

from . import cuda_ext
from .cuda_ext import exllama_ext as ext


class ExLlamaDeviceMap:
    """Layer -> device placement (reference: model.py:636-668)."""

    def __init__(self, num_layers):
        self.num_layers = num_layers
        self.embed_tokens = "cuda:0"      # reference default: "cpu" (model.py:642); see module docstring
        self.lm_head = "cuda:0"
        self.norm = "cuda:0"
        self.layers = ["cuda:0"] * self.num_layers

    def get_layers_devs(self):
        return sorted(set(self.layers))

    def get_all_devs(self):
        return sorted(set(self.layers + [self.lm_head, self.norm, self.embed_tokens]))

    def map(self, key):
        if key.startswith("lm_head."):
            return self.lm_head
        if key.startswith("model.embed_tokens."):
            return self.embed_tokens
        if key.startswith("model.norm."):
            return self.norm
        if key.startswith("model.layers."):
            return self.layers[int(key.split(".")[2])]
        raise ValueError("Unknown key: " + key)


class ExLlamaConfig:
    """Model + tuning configuration (reference: model.py:39-127). `source` is a config.json path or a dict."""

    def __init__(self, source):
        if isinstance(source, dict):
            cfg = source
        else:
            with open(source) as f:
                cfg = json.load(f)

        self.bos_token_id = cfg.get("bos_token_id", 1)
        self.eos_token_id = cfg.get("eos_token_id", 2)
        self.pad_token_id = cfg.get("pad_token_id", 0)
        self.hidden_size = cfg["hidden_size"]
        self.initializer_range = cfg["initializer_range"]
        self.intermediate_size = cfg["intermediate_size"]
        self.num_attention_heads = cfg["num_attention_heads"]
        self.num_hidden_layers = cfg["num_hidden_layers"]
        self.rms_norm_eps = cfg["rms_norm_eps"]
        self.vocab_size = cfg["vocab_size"]
        self.num_key_value_heads = cfg.get("num_key_value_heads", self.num_attention_heads)
        self.num_key_value_groups = self.num_attention_heads // self.num_key_value_heads
        self.rotary_embedding_base = cfg.get("rope_theta", 10000.0)
        self.head_dim = cfg.get("head_dim", self.hidden_size // self.num_attention_heads)   # explicit for a tensor-parallel shard (tp.py)
        self.tp = None              # exllama_amd.tp.TensorParallel when this model is one rank's shard

        self.groupsize = None       # autodetected
        self.act_order = False      # autodetected
        self.empty_g_idx = False    # autodetected

        self.model_path = None      # str | list[str]
        self.device_map = ExLlamaDeviceMap(self.num_hidden_layers)

        self.max_seq_len = 2048
        self.max_input_len = 2048
        self.max_attention_size = 2048 ** 2     # kept for API parity; flash attention has no such limit
        self.compress_pos_emb = 1.0
        self.alpha_value = 1.0
        self.gpu_peer_fix = False
        self.auto_map = None
        self.weight_arena = True                # one allocation per device for all weight tensors (not in the reference)

        # tuning (reference: model.py:92-103)
        self.use_flash_attn_2 = False           # ignored: attention is always the in-tree HIP kernel
        self.matmul_recons_thd = 8
        self.fused_mlp_thd = 2
        self.sdp_thd = 8
        self.fused_attn = True
        self.matmul_fused_remap = False
        self.rmsnorm_no_half2 = False
        self.rope_no_half2 = False
        self.matmul_no_half2 = False
        self.silu_no_half2 = False
        self.concurrent_streams = False

    def set_tuning_params(self):
        ext.set_tuning_params(self.matmul_recons_thd, self.fused_mlp_thd, self.sdp_thd, self.matmul_fused_remap,
                              self.rmsnorm_no_half2, self.rope_no_half2, self.matmul_no_half2, self.silu_no_half2,
                              self.concurrent_streams)

    def set_auto_map(self, map_string):
        self.auto_map = None if map_string is None else [float(a) for a in map_string.split(",")]

    def calculate_rotary_embedding_base(self):
        self.rotary_embedding_base = self.rotary_embedding_base * self.alpha_value ** (self.head_dim / (self.head_dim - 2))


class Ex4bitLinear:
    """4-bit GPTQ linear layer holding a native Q4 handle (reference: model.py:132-221)."""

    def __init__(self, config, in_features, out_features, has_bias, tensors, key):
        self.config = config
        self.key = key
        self.in_features = in_features
        self.out_features = out_features
        self.qweight = tensors[key + ".qweight"]
        self.qzeros = tensors[key + ".qzeros"]
        self.scales = tensors[key + ".scales"]
        self.g_idx = tensors[key + ".g_idx"].cpu() if key + ".g_idx" in tensors else None
        self.bias = tensors[key + ".bias"] if has_bias else None
        if self.g_idx is not None and bool((self.g_idx == 0).all()):
            self.config.empty_g_idx = True
            self.g_idx = None
        self.device = self.qweight.device
        self.device_index = self.device.index
        self.height = self.qweight.shape[0] * 8
        self.width = self.qweight.shape[1]
        if self.height != in_features or self.width != out_features:
            raise ValueError(f"{key}: qweight is {self.height}x{self.width}, expected {in_features}x{out_features}")
        self.q4 = cuda_ext.ext_make_q4(self.qweight, self.qzeros, self.scales, self.g_idx, self.device_index)
        self.groupsize = None
        if self.qzeros.shape[0] > 1:
            self.groupsize = self.height // self.qzeros.shape[0]
            if self.config.groupsize is None:
                self.config.groupsize = self.groupsize
        if self.g_idx is not None:
            if self.groupsize is None:
                raise ValueError("Found group index but no groupsize. What do?")
            self.config.act_order = True

    def lora_applies(self, lora):
        return lora is not None and (self.key + ".lora_A.weight") in lora.tensors

    def get_lora_tensors_or_meta(self, lora):
        if not self.lora_applies(lora):
            return cuda_ext.none_tensor, cuda_ext.none_tensor
        return lora.tensors[self.key + ".lora_A.weight"], lora.tensors[self.key + ".lora_B.weight"]

    def forward(self, x, lora=None, out=None, accumulate=False):
        """x @ W (+ LoRA).  `out`/`accumulate` expose the residual-fusing epilogue."""
        if self.lora_applies(lora):
            a, b = self.get_lora_tensors_or_meta(lora)
            res = cuda_ext.ext_q4_matmul(x, self.q4, self.width, a, b)
            if out is not None:
                res = out.add_(res.view_as(out)) if accumulate else out.copy_(res.view_as(out))
        elif out is None:
            res = cuda_ext.ext_q4_matmul(x, self.q4, self.width)
        else:
            x2 = x.view(-1, x.shape[-1])
            rows = x2.shape[0]
            fn = ext._lib.exl_q4_matmul
            with cuda_ext._Guard(x.device):
                cuda_ext.check(fn(self.q4, x2.data_ptr(), rows, out.data_ptr(), int(accumulate), cuda_ext._stream(x)), "q4_matmul")
            res = out
        if self.bias is not None:
            res.add_(self.bias)
        return res


class ExLlamaRMSNorm:
    def __init__(self, config, tensors, key):
        self.config = config
        self.variance_epsilon = config.rms_norm_eps
        self.weight = tensors[key]

    def forward(self, hidden_states, buffer=None):
        return cuda_ext.ext_rms_norm(hidden_states, self.weight, self.variance_epsilon)


def _fold_act_order_down_proj(tensors, key):
    """Act-order down_proj, resolved at LOAD time instead of per forward pass.  The reference gathers down_proj's input through
    its row map before every matmul (column_remap, q4_matmul.cu:320-325) after repacking the rows into group order
    (make_sequential, q4_matrix.cu:104-168).  down_proj's input has ONE producer -- silu(gate) * up, elementwise in the columns of
    gate_proj / up_proj -- so the permutation can live in the producers instead: the COLUMNS of gate_proj and up_proj (qweight,
    scales, qzeros nibbles) are put into down_proj's sequential row order, and down_proj's own rows are repacked into that order
    here (the same integer repack make_q4 would do).  down_proj then is an ordinary matrix without a group index: no gather in
    the prompt pass (113 MB per 13B layer), no permuted store in the decode executor.  Every output element is computed from the
    same weights and activations in the same summation order, so the logits do not change by a bit (tested).  In place on the
    device tensors (the weight arena keeps its layout); returns the map (new row -> old row, CPU int64) or None."""
    gk = key + ".down_proj.g_idx"
    if gk not in tensors:
        return None
    g_idx = tensors[gk].detach().cpu().to(torch.int64)
    if bool((g_idx == 0).all()):
        return None
    names = ("qweight", "qzeros", "scales")
    gate, up, down = ({n: tensors[f"{key}.{p}.{n}"] for n in names} for p in ("gate_proj", "up_proj", "down_proj"))
    dev = down["qweight"].device
    K8, N = down["qweight"].shape
    K = K8 * 8
    if any(t.device != dev for d in (gate, up) for t in d.values()) or gate["qweight"].shape[1] != K or up["qweight"].shape[1] != K:
        return None
    x_map = torch.sort(g_idx, stable=True).indices                      # make_q4's counting sort: new row -> old row
    xm = x_map.to(dev)
    with torch.no_grad():
        # down_proj: nibble k of packed row r is weight row 8 r + k (matrix.cuh:73-77); new row i takes old row x_map[i]
        qw = down["qweight"]
        nib = torch.empty((K, N), dtype=torch.uint8, device=dev)
        for k in range(8):
            nib[k::8] = ((qw >> (4 * k)) & 0xF).to(torch.uint8)
        nib = nib[xm]
        acc = torch.zeros_like(qw)
        for k in range(8):
            acc |= nib[k::8].to(torch.int32) << (4 * k)
        qw.copy_(acc)
        del nib, acc
        # gate_proj / up_proj: column c of the new matrices is old column x_map[c]
        sh = torch.arange(0, 32, 4, device=dev, dtype=torch.int32)
        for m in (gate, up):
            m["qweight"].copy_(m["qweight"][:, xm])
            m["scales"].copy_(m["scales"][:, xm])
            qz = m["qzeros"]                                             # [G, N / 8]: nibble j of word w is column 8 w + j
            z = ((qz.unsqueeze(-1) >> sh) & 0xF).reshape(qz.shape[0], -1)[:, xm].reshape(qz.shape[0], -1, 8)
            packed = torch.zeros_like(qz)
            for j in range(8):
                packed |= z[:, :, j].to(torch.int32) << (4 * j)
            qz.copy_(packed)
    for proj in ("gate_proj", "up_proj"):                                # an output bias of a producer follows its columns
        bk = f"{key}.{proj}.bias"
        if bk in tensors and tensors[bk] is not None:
            tensors[bk] = tensors[bk][x_map.to(tensors[bk].device)].contiguous()
    del tensors[gk]
    return x_map


class ExLlamaMLP:
    def __init__(self, config, tensors, key):
        self.config = config
        # act-order down_proj folded into the column order of gate_proj / up_proj (not for tensor-parallel shards, whose cuts assume
        # the checkpoint's order; config.fold_act_order_mlp = False keeps the reference's run-time gather)
        self.fold_map = (_fold_act_order_down_proj(tensors, key)
                         if config.tp is None and getattr(config, "fold_act_order_mlp", True) and tensors[key + ".down_proj.qweight"].is_cuda else None)
        h, i = config.hidden_size, config.intermediate_size
        self.gate_proj = Ex4bitLinear(config, h, i, False, tensors, key + ".gate_proj")
        self.up_proj = Ex4bitLinear(config, h, i, False, tensors, key + ".up_proj")
        d_in, d_out = i, h
        self.down_gather = False
        if config.tp is not None and tensors[key + ".down_proj.qweight"].shape[1] != h:
            self.down_gather = True                                  # act-order down_proj of a tensor-parallel shard (tp.py)
            d_in, d_out = tensors[key + ".down_proj.qweight"].shape[0] * 8, tensors[key + ".down_proj.qweight"].shape[1]
        self.down_proj = Ex4bitLinear(config, d_in, d_out, False, tensors, key + ".down_proj")

    def fused(self, x, buffer, post_attention_layernorm, lora):
        """rows <= fused_mlp_thd: one native call, residual added in place (reference: model.py:238-263)."""
        bsz, q_len, _ = x.shape
        ga, gb = self.gate_proj.get_lora_tensors_or_meta(lora)
        ua, ub = self.up_proj.get_lora_tensors_or_meta(lora)
        da, db = self.down_proj.get_lora_tensors_or_meta(lora)
        ranks = [t.shape[1] for t in (ga, ua, da) if not t.is_meta]
        lora_temp = (torch.empty((1, bsz * q_len * max(ranks)), dtype=torch.float16, device=x.device)
                     if ranks else cuda_ext.none_tensor)
        ext.q4_mlp(x.view(-1, x.shape[-1]), post_attention_layernorm.weight, self.config.rms_norm_eps,
                   self.gate_proj.q4, self.up_proj.q4, self.down_proj.q4, ga, gb, ua, ub, da, db, lora_temp)

    def _activation(self, x, lora):
        """silu(gate(x)) * up(x): one fused kernel for long prompts (exl_q4_matmul_dual), else the reference's three calls."""
        if not (self.gate_proj.lora_applies(lora) or self.up_proj.lora_applies(lora)):
            g = torch.empty(x.shape[:-1] + (self.gate_proj.out_features,), dtype=torch.float16, device=x.device)
            if ext.q4_matmul_dual(x.view(-1, x.shape[-1]), self.gate_proj.q4, self.up_proj.q4, g.view(-1, g.shape[-1]), None, True):
                return g
        g = self.gate_proj.forward(x, lora)
        u = self.up_proj.forward(x, lora)
        ext.silu_mul(g, u)
        return g

    def forward_residual(self, normed, hidden, lora, norm=None):
        """hidden += down(silu(gate(normed)) * up(normed)); residual fused in the down_proj epilogue.  `normed` None + `norm`: the
        post-attention RMSNorm has not run yet -- a long prompt without adapters goes through exl_q4_mlp_prompt (norm -> fused
        gate / up / SiLU kernel -> down_proj), otherwise the norm runs here."""
        tp = self.config.tp
        if normed is None:
            if tp is None and not any(p.lora_applies(lora) or p.bias is not None for p in (self.gate_proj, self.up_proj, self.down_proj)):
                h2 = hidden.view(-1, hidden.shape[-1])
                if h2.shape[0] > 512:
                    act = torch.empty((h2.shape[0], self.gate_proj.out_features), dtype=torch.float16, device=hidden.device)
                    if ext.q4_mlp_prompt(h2, norm.weight, norm.variance_epsilon, self.gate_proj.q4, self.up_proj.q4, self.down_proj.q4, act):
                        return
            normed = norm.forward(hidden)
        if tp is None:
            self.down_proj.forward(self._activation(normed, lora), lora, out=hidden, accumulate=True)
        elif self.down_gather:
            act = tp.all_gather_last(self._activation(normed, lora), tp.plan.inter_sizes)
            hidden.add_(tp.all_gather_last(self.down_proj.forward(act, lora), tp.plan.hidden_sizes))
        else:
            hidden.add_(tp.all_reduce(self.down_proj.forward(self._activation(normed, lora), lora)))

    def forward(self, x, buffer=None, lora=None):
        """Non-residual form of the reference (model.py:266-273)."""
        return self.down_proj.forward(self._activation(x, lora), lora)


class ExLlamaAttention:
    def __init__(self, config, tensors, key, sin, cos, index):
        self.config = config
        self.sin, self.cos, self.index = sin, cos, index
        h, hd = config.hidden_size, config.head_dim
        self.q_proj = Ex4bitLinear(config, h, config.num_attention_heads * hd, False, tensors, key + ".q_proj")
        self.k_proj = Ex4bitLinear(config, h, config.num_key_value_heads * hd, False, tensors, key + ".k_proj")
        self.v_proj = Ex4bitLinear(config, h, config.num_key_value_heads * hd, False, tensors, key + ".v_proj")
        o_in, o_out = config.num_attention_heads * hd, h
        self.o_gather = False
        if config.tp is not None and tensors[key + ".o_proj.qweight"].shape[1] != h:
            # act-order o_proj of a tensor-parallel shard: cut by output columns, needs the full attention output (tp.py)
            self.o_gather = True
            o_in, o_out = tensors[key + ".o_proj.qweight"].shape[0] * 8, tensors[key + ".o_proj.qweight"].shape[1]
        self.o_proj = Ex4bitLinear(config, o_in, o_out, False, tensors, key + ".o_proj")

    def fused(self, hidden_states, cache, buffer, input_layernorm, lora):
        """rows == 1: q4_attn -> HIP attention -> q4_attn_2, all in place on hidden_states (reference: model.py:322-418)."""
        cfg = self.config
        bsz, q_len, _ = hidden_states.shape
        past_len = cache.current_seq_len
        qa, qb = self.q_proj.get_lora_tensors_or_meta(lora)
        ka, kb = self.k_proj.get_lora_tensors_or_meta(lora)
        va, vb = self.v_proj.get_lora_tensors_or_meta(lora)
        oa, ob = self.o_proj.get_lora_tensors_or_meta(lora)
        ranks = [t.shape[1] for t in (qa, ka, va, oa) if not t.is_meta]
        dev = hidden_states.device
        lora_temp = (torch.empty((1, bsz * q_len * max(ranks)), dtype=torch.float16, device=dev) if ranks else cuda_ext.none_tensor)
        q = torch.empty((bsz, q_len, cfg.num_attention_heads * cfg.head_dim), dtype=torch.float16, device=dev)
        k = torch.empty((bsz, q_len, cfg.num_key_value_heads * cfg.head_dim), dtype=torch.float16, device=dev)
        v = torch.empty_like(k)
        kc, vc = cache.key_states[self.index], cache.value_states[self.index]
        ext.q4_attn(hidden_states, input_layernorm.weight, cfg.rms_norm_eps, q, k, v, self.q_proj.q4, self.k_proj.q4,
                    self.v_proj.q4, self.sin, self.cos, q_len, past_len, cfg.num_attention_heads, cfg.num_key_value_heads,
                    cfg.head_dim, kc, vc, cache.max_seq_len, qa, qb, ka, kb, va, vb, lora_temp)
        attn = torch.empty_like(q)
        ext.attention(q, kc, vc, attn, past_len, cfg.num_attention_heads)
        ext.q4_attn_2(hidden_states, attn, self.o_proj.q4, oa, ob, lora_temp)

    def forward_residual(self, normed, hidden, cache, buffer, lora, norm=None):
        """General path: hidden += o_proj(attention(rope(q), cache <- rope(k), v)) (reference: model.py:421-502).
        `normed` None + `norm`: the input RMSNorm has not run yet -- the fused prompt launch takes it as its prologue
        (exl_q4_attn_prompt), otherwise it runs here."""
        cfg = self.config
        bsz, q_len, _ = hidden.shape
        past_len = cache.current_seq_len
        kc, vc = cache.key_states[self.index], cache.value_states[self.index]
        q = None
        if not any(p.lora_applies(lora) or p.bias is not None for p in (self.q_proj, self.k_proj, self.v_proj)):
            # long prompts: the three projections, both RoPEs and the cache write as one kernel (exl_q4_qkv_rope_cache)
            q = torch.empty((bsz, q_len, cfg.num_attention_heads * cfg.head_dim), dtype=torch.float16, device=hidden.device)
            src = hidden if normed is None else normed
            if not ext.q4_qkv_rope_cache(src.view(-1, src.shape[-1]), self.q_proj.q4, self.k_proj.q4, self.v_proj.q4,
                                         q.view(-1, q.shape[-1]), self.sin, self.cos, kc, vc, q_len, past_len,
                                         cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim, cache.max_seq_len,
                                         norm_weight=norm.weight if normed is None else None,
                                         eps=norm.variance_epsilon if normed is None else 0.0):
                q = None
        if q is None and normed is None:
            normed = norm.forward(hidden, buffer)
        if q is None:
            q = self.q_proj.forward(normed, lora)
            k = self.k_proj.forward(normed, lora)
            v = self.v_proj.forward(normed, lora)
            ext.rope_(q, self.sin, self.cos, past_len, cfg.num_attention_heads, cfg.head_dim)
            ext.rope_(k, self.sin, self.cos, past_len, cfg.num_key_value_heads, cfg.head_dim)
            ext.update_cache(k, v, kc, vc, past_len)
        attn = torch.empty_like(q)
        mask = buffer.attn_mask if (buffer is not None and buffer.needs_mask) else None
        ext.attention(q, kc, vc, attn, past_len, cfg.num_attention_heads, mask=mask)
        if cfg.tp is None:
            self.o_proj.forward(attn, lora, out=hidden, accumulate=True)
        elif self.o_gather:                                          # every rank: full input -> its output columns -> gathered
            hidden.add_(cfg.tp.all_gather_last(self.o_proj.forward(cfg.tp.all_gather_last(attn), lora), cfg.tp.plan.hidden_sizes))
        else:                                                        # partial sums over this rank's heads -> all-reduce
            hidden.add_(cfg.tp.all_reduce(self.o_proj.forward(attn, lora)))


def _rows(x):
    n = 1
    for d in x.shape[:-1]:
        n *= d
    return n


class ExLlamaDecoderLayer:
    def __init__(self, config, tensors, key, index, sin, cos):
        self.config = config
        self.index = index
        self.self_attn = ExLlamaAttention(config, tensors, key + ".self_attn", sin, cos, index)
        self.mlp = ExLlamaMLP(config, tensors, key + ".mlp")
        self.input_layernorm = ExLlamaRMSNorm(config, tensors, key + ".input_layernorm.weight")
        self.post_attention_layernorm = ExLlamaRMSNorm(config, tensors, key + ".post_attention_layernorm.weight")

    def forward(self, hidden_states, cache, buffer, lora):
        """In place on hidden_states; same branch conditions as the reference (model.py:524-552)."""
        cfg = self.config
        rows = _rows(hidden_states)
        if cfg.tp is not None:                                       # a shard: the paths with the collectives (never the fused ops,
            normed = self.input_layernorm.forward(hidden_states, buffer)      # which add their PARTIAL result into the residual stream)
            self.self_attn.forward_residual(normed, hidden_states, cache, buffer, lora)
            normed = self.post_attention_layernorm.forward(hidden_states, buffer)
            self.mlp.forward_residual(normed, hidden_states, lora)
            return hidden_states
        self.took_layer_call = 2 <= rows <= 256 and self._short_prompt(hidden_states, cache, buffer, lora)
        if self.took_layer_call:
            return hidden_states
        if cfg.fused_attn and rows == 1:
            self.self_attn.fused(hidden_states, cache, buffer, self.input_layernorm, lora)
        else:                                                        # (the norms run inside: a long prompt takes them as launch prologues)
            self.self_attn.forward_residual(None, hidden_states, cache, buffer, lora, norm=self.input_layernorm)
        if cfg.fused_mlp_thd > 0 and rows <= cfg.fused_mlp_thd:
            self.mlp.fused(hidden_states, buffer, self.post_attention_layernorm, lora)
        else:
            self.mlp.forward_residual(None, hidden_states, lora, norm=self.post_attention_layernorm)
        return hidden_states


    def _short_prompt(self, hidden_states, cache, buffer, lora):
        """A short prompt (2 .. 256 rows: BASELINE configs[0], a chat turn): the whole layer as ONE native call -- ten launches on
        fragment-order activations (csrc/q4_gemm_frag.hip) instead of the fourteen op-by-op launches the reference's loop drives
        from Python (model.py:421-552).  False: not taken (an adapter, a bias, an attention mask, a shape the kernels do not cover)."""
        a, m = self.self_attn, self.mlp
        projs = (a.q_proj, a.k_proj, a.v_proj, a.o_proj, m.gate_proj, m.up_proj, m.down_proj)
        if any(p.lora_applies(lora) or p.bias is not None for p in projs) or a.o_gather or m.down_gather:
            return False
        if (buffer is not None and buffer.needs_mask) or not hidden_states.is_contiguous():
            return False
        cfg = self.config
        bsz, q_len, hid = hidden_states.shape
        # RMSNorm's sums of squares travel from a layer's down_proj to the next layer's first norm in buffer.rowsq (include/exl_amd.h:
        # exl_q4_layer_prompt); they are valid for THIS x only if the previous layer of the same pass left them for it
        rowsq, slots, tag = None, 0, (hidden_states.data_ptr(), bsz * q_len, self.index - 1, str(hidden_states.device))
        if buffer is not None:
            need = bsz * q_len * (hid // 16 + 4)
            rowsq = getattr(buffer, "rowsq", None)
            if rowsq is None or rowsq.numel() < need or rowsq.device != hidden_states.device:
                rowsq = buffer.rowsq = torch.empty(need, dtype=torch.float32, device=hidden_states.device)
                buffer.rowsq_tag = None
            if getattr(buffer, "rowsq_tag", None) == tag:
                slots = buffer.rowsq_slots
        done, out_slots = ext.q4_layer_prompt(hidden_states.view(-1, hid), bsz, q_len, cache.current_seq_len, self.input_layernorm.weight,
                                              self.post_attention_layernorm.weight, self.input_layernorm.variance_epsilon, *(p.q4 for p in projs),
                                              a.sin, a.cos, cache.key_states[self.index], cache.value_states[self.index], cfg.num_attention_heads,
                                              cfg.num_key_value_heads, cfg.head_dim, cache.max_seq_len, rowsq=rowsq, rowsq_in_slots=slots)
        if buffer is not None:
            buffer.rowsq_slots = out_slots if done else 0
            buffer.rowsq_tag = (tag[0], tag[1], self.index, tag[3]) if done and out_slots > 0 else None
        return done


class ExLlamaCache:
    """Preallocated K/V cache [bsz, kv_heads, max_seq_len, head_dim] per layer (reference: model.py:557-631)."""

    def __init__(self, model, batch_size=1, max_seq_len=-1, copy_from=None):
        self.model = model
        self.config = model.config
        self.max_seq_len = max_seq_len if max_seq_len != -1 else self.config.max_seq_len
        self.batch_size = batch_size
        self.key_states, self.value_states = [], []
        self.current_seq_len = 0
        cfg = self.config
        for i in range(cfg.num_hidden_layers):
            if copy_from is None:
                shape = (batch_size, cfg.num_key_value_heads, self.max_seq_len, cfg.head_dim)
                dev = cfg.device_map.layers[i]
                self.key_states.append(torch.zeros(shape, dtype=torch.float16, device=dev))
                self.value_states.append(torch.zeros(shape, dtype=torch.float16, device=dev))
            else:
                self.key_states.append(copy_from.key_states[i].clone())
                self.value_states.append(copy_from.value_states[i].clone())

    def zero(self):
        for k, v in zip(self.key_states, self.value_states):
            k.zero_()
            v.zero_()

    def clone(self):
        return ExLlamaCache(self.model, batch_size=self.batch_size, max_seq_len=self.max_seq_len, copy_from=self)

    def roll_left(self):
        """Drop the oldest position (reference: model.py:601-607).  Rolled IN PLACE: the native decode executor and its
        captured hipGraphs hold the raw addresses of these tensors (enable_decode_graph), so the tensors must not be
        rebound to new storage."""
        for i in range(len(self.key_states)):
            self.key_states[i].copy_(torch.roll(self.key_states[i], shifts=-1, dims=2))
            self.value_states[i].copy_(torch.roll(self.value_states[i], shifts=-1, dims=2))
        self.current_seq_len -= 1

    def copy_states(self, target, from_column, from_columns, to_column, to_columns, from_row, from_rows, to_row, to_rows):
        assert from_rows == 1
        assert from_columns == to_columns
        assert to_column + to_columns <= target.max_seq_len
        assert from_column + from_columns <= self.max_seq_len
        for i in range(len(self.key_states)):
            for src_t, dst_t in ((self.key_states[i], target.key_states[i]), (self.value_states[i], target.value_states[i])):
                src = src_t.narrow(0, from_row, from_rows).narrow(2, from_column, from_columns)
                dst = dst_t.narrow(0, to_row, to_rows).narrow(2, to_column, to_columns)
                dst.copy_(src.expand_as(dst) if to_rows > 1 else src)


class ExLlamaBuffer:
    """Per-forward attention mask holder (reference: model.py:671-690)."""

    def __init__(self, config):
        self.config = config
        self.attn_mask = None
        self.needs_mask = False         # True only when an input_mask (padding) was supplied: causal masking is built in

    def to(self, device):
        new = ExLlamaBuffer(self.config)
        new.needs_mask = self.needs_mask
        new.attn_mask = None if self.attn_mask is None else _move_tensor(self.attn_mask, device, "attn_mask", self.config)
        return new


def _skip_key(key):
    return key.endswith("_proj.bias") or key.endswith(".rotary_emb.inv_freq")


def _move_tensor(tensor, new_device, name, config):
    if str(tensor.device) == str(new_device):
        return tensor
    if config.gpu_peer_fix and str(tensor.device).startswith("cuda:") and str(new_device).startswith("cuda:"):
        tensor = tensor.to("cpu")
    return tensor.to(new_device)


def _layer_dtype_size(key):
    for suffix, size in ((".weight", 2), (".qweight", 4), (".qzeros", 4), (".scales", 2), (".g_idx", 0)):
        if key.endswith(suffix):
            return size
    raise ValueError("Unrecognized layer: " + key)


class ExLlama:
    """The model. `ExLlama(config)` loads config.model_path (safetensors, one or many files);
    `ExLlama(config, tensors=dict)` adopts already-materialised tensors (synthetic checkpoints)."""

    def __init__(self, config, tensors=None):
        self.config = config
        cfg = config
        cfg.set_tuning_params()
        if not torch.cuda.is_available():
            raise RuntimeError("exllama_amd.ExLlama needs a HIP device: there is no CPU execution path")

        if tensors is None:
            tensors = self._load_safetensors()
        else:
            tensors = self._place_tensors(tensors)
        self.max_dq_buffer_size = max([t.numel() * 8 for k, t in tensors.items() if k.endswith(".qweight")] + [1])

        self.lm_head_weight = tensors["lm_head.weight"]
        self.embed_weight = tensors["model.embed_tokens.weight"]
        with torch.no_grad():
            self.embed_weight[cfg.pad_token_id] = 0            # reference: model.py:853-854
        self.norm = ExLlamaRMSNorm(cfg, tensors, "model.norm.weight")

        # RoPE tables per device: fp32 math, fp16 storage (reference: model.py:862-877)
        self.sincos = {}
        for device in cfg.device_map.get_layers_devs():
            inv_freq = 1.0 / (cfg.rotary_embedding_base ** (torch.arange(0, cfg.head_dim, 2, device=device).float() / cfg.head_dim))
            t = torch.arange(cfg.max_seq_len, device=device, dtype=torch.float32)
            if cfg.compress_pos_emb != 1.0:
                t /= cfg.compress_pos_emb
            freqs = torch.einsum("i,j->ij", t, inv_freq)
            emb = torch.cat((freqs, freqs), dim=-1)
            self.sincos[device] = (emb.sin()[None, None, :, :].half().contiguous(), emb.cos()[None, None, :, :].half().contiguous())

        self.layers = []
        for i in range(cfg.num_hidden_layers):
            sin, cos = self.sincos[cfg.device_map.layers[i]]
            self.layers.append(ExLlamaDecoderLayer(cfg, tensors, f"model.layers.{i}", i, sin, cos))

        # Scratch buffers registered with the native library (reference: model.py:897-917).  temp_dq is not needed by
        # the fused-dequant GEMM; a token-sized stub keeps the reference's prepare_buffers signature.
        self.buffers = []
        for dev in cfg.device_map.get_layers_devs():
            b = {
                "temp_state": torch.zeros((cfg.max_input_len, cfg.intermediate_size if cfg.tp is None else cfg.tp.plan.inter_full),
                                          dtype=torch.float16, device=dev),
                "temp_mlp": torch.zeros((max(cfg.fused_mlp_thd, 1) * 2, cfg.intermediate_size), dtype=torch.float16, device=dev),
                "temp_zeros_float": torch.zeros((1, 65536), dtype=torch.float32, device=dev),
                "temp_dq": torch.zeros((1, 64), dtype=torch.float16, device=dev),
            }
            self.buffers.append(b)
            ext.prepare_buffers(torch.device(dev), b["temp_state"], b["temp_mlp"], b["temp_zeros_float"], b["temp_dq"])
        self._decoder = None
        torch.cuda.empty_cache()

    # ---- loading -------------------------------------------------------------------------------------------
    @staticmethod
    def _cast(key, tensor, device):
        if key.endswith(".scales") or key.endswith("layernorm.weight") or key == "model.norm.weight" \
                or key.endswith(".embed_tokens.weight"):
            return tensor.half()
        if key == "lm_head.weight":
            return tensor.float() if device == "cpu" else tensor.half()
        return tensor

    def _place_tensors(self, tensors):
        """Casts and moves the checkpoint tensors to their devices.  With config.weight_arena (default) all tensors of a
        device live in ONE allocation, 2 MiB-aligned each: a 33B model is ~2,000 tensors, and one contiguous range lets the
        driver back it with its largest page fragments (TLB reach of the weight stream) instead of whatever each small
        allocation happened to get."""
        out, plan = {}, {}
        for key, t in tensors.items():
            if _skip_key(key):
                continue
            device = self.config.device_map.map(key)
            if key.endswith(".g_idx"):
                out[key] = t                                    # consumed on the host by make_q4
                continue
            out[key] = self._cast(key, t, device)
            if str(device).startswith("cuda") and getattr(self.config, "weight_arena", True) and not os.environ.get("EXL_NO_WEIGHT_ARENA"):
                plan.setdefault(str(device), []).append(key)
            else:
                out[key] = out[key].to(device).contiguous()
        align = 2 << 20
        for device, keys in plan.items():
            sizes = [out[k].numel() * out[k].element_size() for k in keys]
            total = sum((n + align - 1) // align * align for n in sizes)
            arena = torch.empty(total, dtype=torch.uint8, device=device)
            self._arenas = getattr(self, "_arenas", []) + [arena]
            off = 0
            for k, n in zip(keys, sizes):
                src = out[k]
                view = arena[off:off + n].view(src.dtype).view(src.shape)
                view.copy_(src, non_blocking=True)
                out[k] = view
                off += (n + align - 1) // align * align
        return out

    def _load_safetensors(self):
        from safetensors import safe_open
        cfg = self.config
        paths = [cfg.model_path] if isinstance(cfg.model_path, str) else list(cfg.model_path)
        load_keys, sizes = {}, {"decoder": 0, "norm": 0, "head": 0}
        for path in paths:
            with safe_open(path, framework="pt", device="cpu") as f:
                for key in f.keys():
                    if _skip_key(key):
                        continue
                    load_keys[key] = path
                    bucket = ("decoder" if key.startswith("model.layers.0.") else "norm" if key.startswith("model.norm.")
                              else "head" if key.startswith("lm_head.") else None)
                    if bucket:
                        sizes[bucket] += math.prod(f.get_slice(key).get_shape()) * _layer_dtype_size(key)
        if cfg.auto_map is not None:
            self._auto_split(sizes)
        tensors = {}
        handles = {}
        for key, path in load_keys.items():
            if path not in handles:
                handles[path] = safe_open(path, framework="pt", device="cpu")
            t = handles[path].get_tensor(key)
            device = cfg.device_map.map(key)
            if key.endswith(".g_idx"):
                tensors[key] = t
            else:
                tensors[key] = self._cast(key, t, device).to(device).contiguous()
        return tensors

    def _auto_split(self, sizes):
        """Greedy fill of per-device GB budgets, layers then norm then head (reference: model.py:770-801)."""
        cfg = self.config
        budgets = [g * 1024 ** 3 for g in cfg.auto_map]
        dev, used = 0, 0
        n = cfg.num_hidden_layers
        cfg.device_map.embed_tokens = "cuda:0"
        for item in range(n + 2):
            size = sizes["decoder"] if item < n else sizes["norm"] if item == n else sizes["head"]
            while used + size > budgets[dev]:
                dev += 1
                used = 0
                if dev >= len(budgets):
                    raise ValueError("Model too large for device allocation scheme.")
            target = f"cuda:{dev}"
            if item < n:
                cfg.device_map.layers[item] = target
            elif item == n:
                cfg.device_map.norm = target
            else:
                cfg.device_map.lm_head = target
            used += size

    # ---- forward -------------------------------------------------------------------------------------------
    def forward(self, input_ids, cache, last_id_only=True, preprocess_only=False, lora=None, output_device=None,
                input_mask=None):
        """Same contract as the reference (model.py:924-986): chunks long inputs by max_input_len, returns
        fp32 logits [bsz, 1 or q_len, vocab] on `output_device` (default: input_ids' device), or None."""
        q_len = input_ids.shape[-1]
        bsz = input_ids.shape[0]
        if lora is not None and self.config.tp is not None:
            # an adapter's lora_A spans the full in_features; the row-split o_proj / down_proj of a shard see a slice of them
            raise RuntimeError("LoRA adapters are not supported on a tensor-parallel shard")
        assert input_mask is None or (input_mask.shape[-1] >= input_ids.shape[-1] and input_mask.shape[-2] == input_ids.shape[-2])
        chunk_cap = max(1, self.config.max_input_len // bsz)
        result = None
        begin = 0
        while begin < q_len:
            end = min(begin + chunk_cap, q_len)
            pre = preprocess_only or (end < q_len and last_id_only)
            r = self._forward(input_ids[:, begin:end], cache, last_id_only, pre, lora, output_device, input_mask)
            if not pre:
                result = r if result is None else torch.cat((result, r), dim=1)
            begin = end
        return result

    @torch.no_grad()
    def _forward(self, input_ids, cache, last_id_only=True, preprocess_only=False, lora=None, output_device=None,
                 input_mask=None):
        cfg = self.config
        bsz, seq_len = input_ids.shape
        past_len = cache.current_seq_len
        if past_len + seq_len > cache.max_seq_len:
            raise RuntimeError(f"sequence ({past_len} + {seq_len}) exceeds the cache length {cache.max_seq_len}")
        if output_device is None:
            output_device = input_ids.device
        st = self._decoder
        if (st is not None and cache is st["cache"] and bsz == 1 and seq_len == 1 and lora is st.get("lora")
                and input_mask is None and not preprocess_only and st["has_embed"] and st["has_head"]):
            self._count_path(self._executor_tier(st))
            return self._decode_step(input_ids, cache, str(output_device))
        devs = cfg.device_map.get_layers_devs()

        buffer = ExLlamaBuffer(cfg)
        if input_mask is not None:
            # additive fp16 mask, causal + padding (reference: model.py:1014-1033); only built when padding exists --
            # plain causal masking is part of the attention kernels
            mask = torch.zeros(bsz, 1, seq_len, past_len + seq_len, dtype=torch.float16, device=devs[0])
            if seq_len > 1:
                tri = torch.triu(torch.full((seq_len - 1, seq_len - 1), -65504.0))
                mask[:, :, :seq_len - 1, past_len + 1:past_len + seq_len] = tri
            im = _move_tensor(input_mask[:, :past_len + seq_len], devs[0], "input_mask", cfg)
            im = torch.where(im, 0, -65504.0).half().unsqueeze(1).unsqueeze(2)
            buffer.attn_mask = torch.minimum(mask, im).contiguous()
            buffer.needs_mask = True
        buffers = {devs[0]: buffer}
        for d in devs[1:]:
            buffers[d] = buffer.to(d)

        hidden = self.embed(input_ids)
        hidden = self.forward_layers(hidden, cache, buffers, lora)
        if seq_len == 1:                                              # (counted by what the layers DID: the layer call reports per layer)
            took = bsz > 1 and all(getattr(l, "took_layer_call", False) for l in self.layers)
            if bsz > 1:
                self._layer_call_seen = took
            self._count_path("layer_call" if took else ("ops_fused" if cfg.tp is None and cfg.fused_attn and bsz == 1 else "ops_general"))
        cache.current_seq_len += seq_len
        if preprocess_only:
            return None
        return _move_tensor(self.head(hidden, last_id_only), output_device, "logits", cfg)

    # The three stages of a forward pass, also used one stage per process by exllama_amd/pipeline.py (layer split
    # across processes: rank 0 embeds, every rank runs its own layers, the last rank applies norm + lm_head).
    def embed(self, input_ids):
        """Embedding lookup.  The HIP gather clamps ids to the table (a device-side id cannot raise); ids that arrive from the HOST
        are checked here first, so a tokenizer / vocabulary mismatch fails as loudly as torch's embedding (and the reference,
        model.py:1002) does.  Ids that are already device tensors (the decode loop's own tokens) are not synchronised on."""
        cfg = self.config
        if not input_ids.is_cuda and input_ids.numel():
            lo, hi = int(input_ids.min()), int(input_ids.max())
            if lo < 0 or hi >= self.embed_weight.shape[0]:
                raise IndexError(f"token id out of range: [{lo}, {hi}] not within the embedding table's {self.embed_weight.shape[0]} rows")
        ids = _move_tensor(input_ids, cfg.device_map.embed_tokens, "input_ids", cfg)
        if ids.is_cuda and self.embed_weight.is_cuda and self.embed_weight.dtype == torch.float16 and self.embed_weight.shape[1] % 8 == 0:
            out = torch.empty(tuple(ids.shape) + (self.embed_weight.shape[1],), dtype=torch.float16, device=ids.device)
            ext.embedding(ids.contiguous().to(torch.int64), self.embed_weight, out)      # HIP gather (reference: torch embedding, model.py:1002)
            return out
        return torch.nn.functional.embedding(ids, self.embed_weight).contiguous()

    def forward_layers(self, hidden, cache, buffers=None, lora=None):
        """Runs every layer this model holds on `hidden` [bsz, q_len, hidden]; does NOT advance cache.current_seq_len."""
        cfg = self.config
        if buffers is None:
            devs = cfg.device_map.get_layers_devs()
            first = ExLlamaBuffer(cfg)
            buffers = {devs[0]: first}
            for d in devs[1:]:
                buffers[d] = first.to(d)
        for i, layer in enumerate(self.layers):
            device = cfg.device_map.layers[i]
            hidden = _move_tensor(hidden, device, "hidden_states", cfg)
            hidden = layer.forward(hidden, cache, buffers[device], lora)
        return hidden

    def head(self, hidden, last_id_only=True):
        cfg = self.config
        hidden = _move_tensor(hidden, cfg.device_map.norm, "hidden_states", cfg)
        if last_id_only:
            hidden = hidden[:, -1:, :].contiguous()
        hidden = self.norm.forward(hidden)
        if cfg.device_map.lm_head == "cpu":
            hidden = hidden.float()
        hidden = _move_tensor(hidden, cfg.device_map.lm_head, "hidden_states", cfg)
        logits = None
        rows = hidden.shape[0] * hidden.shape[1]
        if hidden.is_cuda and hidden.dtype == torch.float16 and self.lm_head_weight.is_cuda:
            # HIP kernels over the fp16 head (reference: nn.Linear + .float(), model.py:1077-1078): a GEMV for the last-token logits of
            # a prompt / a short prompt, an LDS-DMA MFMA GEMM for whole-sequence logits (perplexity, validation)
            out = torch.empty((rows, self.lm_head_weight.shape[0]), dtype=torch.float32, device=hidden.device)
            if ext.head_matmul(hidden.reshape(rows, -1).contiguous(), self.lm_head_weight, out):
                logits = out.view(hidden.shape[0], hidden.shape[1], -1)
        if logits is None:                                          # a head on the CPU, or a shape the kernels do not cover (hidden % 64, vocab % 4)
            logits = torch.matmul(hidden, self.lm_head_weight.t()).float()
        if cfg.tp is not None and self.lm_head_weight.shape[0] != cfg.vocab_size:      # this rank's vocabulary rows (tp.py): gather the rest
            logits = cfg.tp.all_gather_last(logits, cfg.tp.plan.vocab_sizes)
        return logits

    # ---- which decode tier runs (and why not a faster one) --------------------------------------------------
    DECODE_TIERS = {
        "executor_graph": "native decode executor, one hipGraph replay per token and device (5 launches per layer + head; greedy argmax / "
                          "sampler inside the graph through generate_greedy / generate_sample)",
        "executor_eager": "native decode executor, eager launches (5 per layer + head)",
        "executor_pieces_tp": "native decode executor in half-layer pieces, the residual stream all-reduced between them (tensor parallel)",
        "ops_fused": "op by op: q4_attn -> attention -> q4_attn_2 -> q4_mlp per layer (the reference's fused decode ops, model.py:524-552)",
        "layer_call": "batch 2 .. 256, one token each: ONE native call per layer (exl_q4_layer_prompt: the short-prompt launches on "
                      "fragment-order activations, csrc/q4_gemm_frag.hip) -- no adapter, bias, mask or tensor-parallel shard",
        "ops_general": "op by op, general path: norm, q/k/v projections, RoPE, cache update, attention, o_proj, norm, gate/up, SiLU, down "
                       "(batched generation, fused_attn off, or a tensor-parallel shard outside the executor; <= fused_mlp_thd rows still "
                       "take q4_mlp for the MLP half, as in the reference)",
    }

    def _count_path(self, tier, n=1):
        c = self.__dict__.setdefault("_path_counts", {})
        c[tier] = c.get(tier, 0) + n
        self._last_path = tier

    def _executor_tier(self, st):
        if self.config.tp is not None:
            return "executor_pieces_tp"
        return "executor_graph" if st["graph"] is not None else "executor_eager"

    def _op_tier(self, bsz, lora=None):
        cfg = self.config
        if cfg.tp is None and cfg.fused_attn and bsz == 1:
            return "ops_fused"
        if cfg.tp is None and 2 <= bsz <= 256 and lora is None and not os.environ.get("EXL_GEMM_NO_FRAG") and \
                getattr(self, "_layer_call_seen", True):              # (what ExLlamaDecoderLayer.forward tries first; False once a layer refused)
            return "layer_call"
        return "ops_general"

    def executor_obstacles(self, batch_size=1, lora=None):
        """Why enable_decode_graph() would refuse this model (empty list: it takes it).  The conditions are the executor's own
        (csrc/decode_fused.hip: exl_decoder_create / exl_decoder_set_layer / exl_decoder_set_lora) and enable_decode_graph's."""
        cfg, why = self.config, []
        if batch_size != 1:
            why.append(f"batch size {batch_size}: the executor handles batch 1 (batched generation runs op by op)")
        if cfg.head_dim != 128:
            why.append(f"head_dim {cfg.head_dim}: the executor's attention / RoPE kernels are written for 128")
        if cfg.hidden_size % 128 or cfg.intermediate_size % 128:
            why.append("hidden / intermediate size is not a multiple of 128")
        if cfg.hidden_size > 8192 or cfg.intermediate_size > 32768:
            why.append("hidden > 8192 or intermediate > 32768")
        if any(not str(d).startswith("cuda") for d in cfg.device_map.layers):
            why.append("a layer is not on a HIP device")
        if cfg.tp is not None and any(l.self_attn.o_gather or l.mlp.down_gather for l in self.layers):
            why.append("tensor parallel: act-order o_proj / down_proj shards (gather mode) run on the op-by-op path")
        if lora is not None:
            if cfg.tp is not None:
                why.append("LoRA on a tensor-parallel shard")
            for i, l in enumerate(self.layers):
                if (l.self_attn.o_proj.lora_applies(lora) or l.mlp.down_proj.lora_applies(lora)) and \
                        (l.self_attn.o_proj.g_idx is not None or l.mlp.down_proj.g_idx is not None):
                    why.append(f"layer {i}: adapter on an act-order o_proj / unfolded act-order down_proj (their inputs are stored permuted)")
                    break
        return why

    def decode_path_report(self, cache=None, batch_size=None, lora=None):
        """Which tier a [batch_size, 1] forward on `cache` takes RIGHT NOW, what the tiers are, why a faster one is not in use,
        and how many single-token forwards each tier has served since the model was built (reset_decode_path_counts()).  The
        reference has one decode path per branch of model.py:524-552; this model has those plus the executor, and picks
        silently -- this is where it says which."""
        st = self._decoder
        bsz = batch_size if batch_size is not None else (cache.batch_size if cache is not None else 1)
        on_executor = (st is not None and (cache is None or cache is st["cache"]) and bsz == 1 and lora is st.get("lora")
                       and st["has_embed"] and st["has_head"])
        tier = self._executor_tier(st) if on_executor else self._op_tier(bsz, lora)
        why = []
        if not on_executor:
            obstacles = self.executor_obstacles(bsz, lora)
            if obstacles:
                why = obstacles
            elif st is None:
                why = ["enable_decode_graph(cache) has not been called"]
            elif cache is not None and cache is not st["cache"]:
                why = ["the executor was enabled for another cache"]
            elif lora is not st.get("lora"):
                why = ["the executor was enabled " + ("without this adapter" if lora is not None else "with an adapter")
                       + ": enable_decode_graph(cache, lora=...)"]
            elif not (st["has_embed"] and st["has_head"]):
                why = ["this model is one link of a layer split: decode_stage_step() / pipeline.LayerSplitRunner"]
        elif tier == "executor_eager":
            why = ["enable_decode_graph(use_graph=False), or the token step could not be captured (process-group calls)"]
        rep = {"tier": tier, "what": self.DECODE_TIERS[tier], "why_not_faster": why,
               "executor_enabled": st is not None, "executor_stages": len(st["stages"]) if st is not None else 0,
               "hop_captured": bool(st.get("hop_captured")) if st is not None else False,
               "forwards_by_tier": dict(self.__dict__.get("_path_counts", {})), "last_forward_tier": self.__dict__.get("_last_path")}
        return rep

    def reset_decode_path_counts(self):
        self._path_counts, self._last_path = {}, None

    # ---- native decode executor + hipGraph ---------------------------------------------------------------
    def _decode_stages(self):
        """Contiguous runs of layers on one device = the stages of the native executor (the reference's device hops,
        model.py:1053-1058).  config.decoder_stage_split (a list of layer indices, testing aid) forces extra stage
        boundaries inside a device."""
        cfg = self.config
        cuts = set(getattr(cfg, "decoder_stage_split", None) or [])
        stages = []
        for i, dev in enumerate(cfg.device_map.layers):
            if not str(dev).startswith("cuda"):
                raise RuntimeError("the native decode executor needs every layer on a HIP device")
            if stages and stages[-1]["dev"] == dev and i not in cuts:
                stages[-1]["layers"].append(i)
            else:
                stages.append({"dev": dev, "layers": [i]})
        return stages

    def enable_decode_graph(self, cache, use_graph=True, first_stage=True, last_stage=True, hop=None, hop_capture=True, lora=None):
        """Route bsz = 1, q_len = 1 forwards on `cache` through the native decode executor (5 kernels per layer,
        exllama_amd/csrc/decode_fused.hip) and, with use_graph, replay them as ONE captured hipGraph per token and device.
        The position lives in device memory, so the same graph serves every context length.

        A model whose layers sit on several devices (config.device_map / set_auto_map, the reference's layer split) runs as
        one executor STAGE per device: the residual stream [hidden] fp16 is copied from stage to stage, one hop per device
        boundary, exactly the reference's `_move_tensor` between layers (model.py:1053-1058).  first_stage / last_stage =
        False make this model ONE LINK of a split across processes (exllama_amd/pipeline.py): without the embedding the
        first stage starts from the hidden state handed to decode_stage_step(), without the head the last stage returns it.
        `hop` (pipeline.LayerSplitRunner): an object whose before(st) / after(st) issue this rank's point-to-point exchanges -- the
        incoming hidden state (or token) ahead of the first stage's kernels, the outgoing one behind the last stage's; they are
        CAPTURED into the first / last stage's graph when the backend allows it (RCCL point-to-point is capturable), so that a
        replay is receive -> kernels -> send with no host work per token; otherwise (hop_capture False, or a failed capture) they run
        eagerly around the replays.  hop.bind(self) is called once the executor's buffers exist.
        `lora` (an ExLlamaLora): its adapters ride INSIDE the token step (exl_decoder_set_lora: two small launches behind each GEMV
        launch with adapters), so forward(..., lora=lora), generate_greedy and generate_sample keep the graph path with that
        adapter -- the reference's q4_attn / q4_mlp LoRA operands (model.py:254-289).  Raises RuntimeError for what the executor does
        not take (an act-order o_proj / an unfolded act-order down_proj, tensor-parallel shards): forward(lora=...) without
        enable_decode_graph(lora=...) stays on the op-by-op path as before."""
        import ctypes as C
        cfg = self.config
        if cache.batch_size != 1:
            raise RuntimeError("the native decode executor handles batch size 1")
        if cache.max_seq_len > cfg.max_seq_len:
            raise RuntimeError("cache is longer than the RoPE tables (config.max_seq_len)")
        stages = self._decode_stages()
        if cfg.tp is not None and any(l.self_attn.o_gather or l.mlp.down_gather for l in self.layers):
            raise RuntimeError("tensor parallel: act-order o_proj / down_proj shards (gather mode) run on the op-by-op path only")
        if first_stage and str(cfg.device_map.embed_tokens) != stages[0]["dev"]:
            raise RuntimeError("the native decode executor needs the embedding table on the first layer's device")
        if last_stage and not (str(cfg.device_map.norm) == str(cfg.device_map.lm_head) == stages[-1]["dev"]):
            raise RuntimeError("the native decode executor needs the final norm and lm_head on the last layer's device")
        self.disable_decode_graph()
        lib = ext._lib
        for k, sg in enumerate(stages):
            dev = torch.device(sg["dev"])
            sin, cos = self.sincos[sg["dev"]]
            emb = self.embed_weight.data_ptr() if (first_stage and k == 0) else None
            head = last_stage and k == len(stages) - 1
            handle = C.c_void_p()
            sg["hid"] = torch.zeros((1, 1, cfg.hidden_size), dtype=torch.float16, device=dev)
            sg["pos"] = torch.zeros((1,), dtype=torch.int32, device=dev)
            with cuda_ext._Guard(dev):
                cuda_ext.check(lib.exl_decoder_create(dev.index, len(sg["layers"]), cfg.hidden_size, cfg.intermediate_size,
                                                      cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim,
                                                      self.lm_head_weight.shape[0], cache.max_seq_len, float(cfg.rms_norm_eps), emb,
                                                      self.norm.weight.data_ptr() if head else None,
                                                      self.lm_head_weight.data_ptr() if head else None, sin.data_ptr(), cos.data_ptr(),
                                                      C.byref(handle)), "decoder_create")
                cuda_ext.check(lib.exl_decoder_set_hidden(handle, sg["hid"].data_ptr()), "decoder_set_hidden")
                if cfg.tp is not None:
                    cuda_ext.check(lib.exl_decoder_set_tp(handle, int(cfg.tp.rank == 0)), "decoder_set_tp")
                for j, i in enumerate(sg["layers"]):
                    layer = self.layers[i]
                    a, m = layer.self_attn, layer.mlp
                    cuda_ext.check(lib.exl_decoder_set_layer(handle, j, a.q_proj.q4, a.k_proj.q4, a.v_proj.q4, a.o_proj.q4,
                                                             m.gate_proj.q4, m.up_proj.q4, m.down_proj.q4,
                                                             layer.input_layernorm.weight.data_ptr(),
                                                             layer.post_attention_layernorm.weight.data_ptr(),
                                                             cache.key_states[i].data_ptr(), cache.value_states[i].data_ptr()),
                                   "decoder_set_layer")
                    if lora is not None:
                        projs = (a.q_proj, a.k_proj, a.v_proj, a.o_proj, m.gate_proj, m.up_proj, m.down_proj)
                        pa, pb, pr = (C.c_void_p * 7)(), (C.c_void_p * 7)(), (C.c_int * 7)()
                        for n, pj in enumerate(projs):
                            if pj.lora_applies(lora):
                                ta, tb = pj.get_lora_tensors_or_meta(lora)
                                if ta.device != dev or tb.device != dev or not (ta.is_contiguous() and tb.is_contiguous()):
                                    raise RuntimeError(f"{pj.key}: LoRA tensors must be contiguous on {dev}")
                                pa[n], pb[n], pr[n] = ta.data_ptr(), tb.data_ptr(), ta.shape[1]
                        cuda_ext.check(lib.exl_decoder_set_lora(handle, j, pa, pb, pr), "decoder_set_lora")
            sg["handle"], sg["tdev"], sg["graphs"] = handle, dev, []
        d0, dl = stages[0]["tdev"], stages[-1]["tdev"]
        st = {
            "stages": stages, "handle": stages[0]["handle"], "cache": cache, "dev": dl, "graph": None, "graphs": [],
            "has_embed": bool(first_stage), "has_head": bool(last_stage),
            "tok": torch.zeros((1, 1), dtype=torch.int64, device=d0),
            "logits": torch.zeros((1, 1, cfg.vocab_size), dtype=torch.float32, device=dl),
            # tensor parallel with a split lm_head: the head kernel writes this rank's rows here, all-gathered into "logits"
            "logits_local": (torch.zeros((1, 1, self.lm_head_weight.shape[0]), dtype=torch.float32, device=dl)
                             if self.lm_head_weight.shape[0] != cfg.vocab_size else None),
            "pos": stages[0]["pos"], "dev_pos": -1,
            "kv_ptrs": [(k.data_ptr(), v.data_ptr()) for k, v in zip(cache.key_states, cache.value_states)],
            "hop": hop, "hop_captured": False, "lora": lora,
        }
        self._decoder = st
        if hop is not None:
            hop.bind(self)
        if use_graph:
            # one captured graph per context bucket and stage: short contexts use fewer KV splits (1 split = nothing to merge)
            st["bucket_splits"] = []
            start = cache.current_seq_len
            self._set_positions(st, start)
            self._decoder_launch(st, advance=0)                 # eager dry run with the full split count (the K/V written
            for sg in stages:                                    # at the current slot are overwritten by the real token later)
                torch.cuda.synchronize(sg["tdev"])
            def capture_all(hop_in):
                """One graph per context bucket and stage; returns False when a tensor-parallel capture had to be given up."""
                for ns, bucket_limit in self.DECODE_BUCKETS:
                    lim = C.c_int()
                    if any(lib.exl_decoder_set_kv_splits(sg["handle"], ns, C.byref(lim)) != 0 for sg in stages):
                        continue                                    # more splits than this decoder has: covered by the last bucket
                    limit = lim.value if bucket_limit is None else min(lim.value, bucket_limit)
                    if st["graphs"] and limit <= st["graphs"][-1][0]:
                        continue
                    per_stage = []
                    for k, sg in enumerate(stages):
                        g = torch.cuda.CUDAGraph()
                        with torch.cuda.device(sg["tdev"]), torch.cuda.graph(g):    # capture only records: nothing runs at this position
                            if hop_in and k == 0:
                                hop.before(st)
                            self._stage_launch(st, k, advance=1)
                            if hop_in and k == len(stages) - 1:
                                hop.after(st)
                        per_stage.append(g)
                    st["graphs"].append((limit, per_stage))
                    st["bucket_splits"].append(ns)
                st["hop_captured"] = bool(hop_in)

            import warnings
            hop_in = hop is not None and hop_capture
            try:
                capture_all(hop_in)
            except Exception as e:                                # noqa: BLE001
                st["graphs"], st["bucket_splits"], st["hop_captured"] = [], [], False
                if hop_in:
                    # the process group's point-to-point calls could not be captured: graphs of the kernels alone, the exchanges
                    # issued eagerly around each replay (same results, two more host calls per token)
                    warnings.warn(f"layer split: hipGraph capture of the hand-off failed ({e}); exchanging eagerly around the replays")
                    capture_all(False)
                elif cfg.tp is not None:
                    # a tensor-parallel step contains the process group's collectives: if this backend cannot be captured,
                    # the pieces run as eager launches (same results, more host time per token)
                    warnings.warn(f"tensor parallel: hipGraph capture of the token step failed ({e}); decoding with eager launches")
                else:
                    raise
            st["graph"] = st["graphs"][-1][1] if st["graphs"] else None
            self._set_positions(st, start)

    DECODE_BUCKETS = ((1, 160), (4, 640), (0, None))                 # (KV splits, last context served); 0 = the decoder's maximum

    @staticmethod
    def _set_positions(st, position):
        for sg in st["stages"]:
            sg["pos"].fill_(position)
        st["dev_pos"] = position

    def _set_eager_splits(self, st, position):
        """Eager (non-graph) launches pick the KV split count per step by the bucket table the graphs are captured with."""
        import ctypes as C
        for ns, limit in self.DECODE_BUCKETS:
            lim = C.c_int()
            if any(ext._lib.exl_decoder_set_kv_splits(sg["handle"], ns, C.byref(lim)) != 0 for sg in st["stages"]):
                continue                                            # more splits than this decoder has
            if position <= (lim.value if limit is None else min(limit, lim.value)):
                return
        raise RuntimeError(f"position {position} beyond the decoder's context limit")

    def _stage_launch(self, st, k, advance):
        sg = st["stages"][k]
        last = k == len(st["stages"]) - 1
        tp = self.config.tp
        if tp is not None:
            # one rank's shard of a tensor-parallel model: the step in pieces, the residual stream all-reduced after each half
            # layer (partial o_proj / down_proj sums; rank 0 carries the incoming residual: exl_decoder_set_tp)
            tok = st["tok"].data_ptr() if (k == 0 and st["has_embed"]) else None
            lbuf = st["logits_local"] if st["logits_local"] is not None else st["logits"]
            logits = lbuf.data_ptr() if (last and st["has_head"]) else None
            with cuda_ext._Guard(sg["tdev"]):
                stream = torch.cuda.current_stream(sg["tdev"]).cuda_stream
                for j in range(len(sg["layers"])):
                    for part in (0, 1):
                        cuda_ext.check(ext._lib.exl_decoder_step_part(sg["handle"], j, part, tok, sg["pos"].data_ptr(), logits, int(advance), stream),
                                       "decoder_step_part")
                        tp.all_reduce(sg["hid"])
                cuda_ext.check(ext._lib.exl_decoder_step_part(sg["handle"], 0, 2, tok, sg["pos"].data_ptr(), logits, int(advance), stream),
                               "decoder_step_part")
                if logits is not None and st["logits_local"] is not None:
                    tp.all_gather_into(st["logits"].view(-1), st["logits_local"].view(-1), tp.plan.vocab_sizes)
            return
        with cuda_ext._Guard(sg["tdev"]):
            cuda_ext.check(ext._lib.exl_decoder_step(sg["handle"], st["tok"].data_ptr() if (k == 0 and st["has_embed"]) else None,
                                                     sg["pos"].data_ptr(), st["logits"].data_ptr() if (last and st["has_head"]) else None,
                                                     int(advance), torch.cuda.current_stream(sg["tdev"]).cuda_stream), "decoder_step")

    @staticmethod
    def _hop(st, k):
        """The one exchange per stage boundary: the residual stream [hidden] fp16, device to device (stream-ordered on both)."""
        st["stages"][k]["hid"].copy_(st["stages"][k - 1]["hid"], non_blocking=True)

    def _decoder_launch(self, st, advance, graphs=None):
        hop = st.get("hop")
        eager_hop = hop is not None and advance and not (graphs is not None and st.get("hop_captured"))
        for k in range(len(st["stages"])):
            if k:
                self._hop(st, k)
            elif eager_hop:
                hop.before(st)                                        # this rank's incoming hand-off (pipeline.LayerSplitRunner)
            if graphs is not None:
                graphs[k].replay()
            else:
                self._stage_launch(st, k, advance)
        if eager_hop:
            hop.after(st)

    def _check_cache_storage(self, st, cache):
        """The executor and its graphs hold raw K/V addresses: a cache whose tensors were rebound (not rolled / copied in
        place) must not be used with them."""
        if st["kv_ptrs"] != [(k.data_ptr(), v.data_ptr()) for k, v in zip(cache.key_states, cache.value_states)]:
            raise RuntimeError("the cache's K/V tensors were replaced after enable_decode_graph(); call it again")

    def _run_token(self, st, cache):
        if cache.current_seq_len + 1 > cache.max_seq_len:
            raise RuntimeError(f"sequence ({cache.current_seq_len} + 1) exceeds the cache length {cache.max_seq_len}")
        self._check_cache_storage(st, cache)
        if st["dev_pos"] != cache.current_seq_len:              # host rewound / advanced the cache outside the executor
            self._set_positions(st, cache.current_seq_len)
        if st["graph"] is not None:
            for limit, per_stage in st["graphs"]:                # first bucket whose context limit covers this position
                if cache.current_seq_len <= limit:
                    self._decoder_launch(st, 1, graphs=per_stage)
                    break
            else:
                raise RuntimeError(f"position {cache.current_seq_len} beyond the decoder's context limit")
        else:
            self._set_eager_splits(st, cache.current_seq_len)       # same split policy as the captured buckets
            self._decoder_launch(st, advance=1)
        cache.current_seq_len += 1
        st["dev_pos"] = cache.current_seq_len

    def _decode_step(self, input_ids, cache, output_device):
        st = self._decoder
        if not (st["has_embed"] and st["has_head"]):
            raise RuntimeError("this model is one stage of a layer split: use decode_stage_step()")
        st["tok"].copy_(input_ids.view(1, 1), non_blocking=True)
        self._run_token(st, cache)
        return _move_tensor(st["logits"].clone(), output_device, "logits", self.config)

    def decode_stage_step(self, cache, input_ids=None, hidden_in=None):
        """One token through THIS process's part of a layer split (enable_decode_graph(..., first_stage / last_stage)):
        the first link takes `input_ids` [1, 1], later links the hidden state [1, 1, hidden] fp16 of the previous one;
        returns fp32 logits [1, 1, vocab] from the last link, the outgoing hidden state (the executor's own buffer: send or
        copy it before the next step) from the others.  With a `hop` (enable_decode_graph) the inputs arrive through it: pass
        neither input_ids nor hidden_in (they are received straight into the executor's buffers, decode_hop_buffers())."""
        st = self._decoder
        if st is None or st["cache"] is not cache:
            raise RuntimeError("decode_stage_step needs enable_decode_graph(cache) first")
        if st["has_embed"] and input_ids is not None:
            st["tok"].copy_(input_ids.view(1, 1), non_blocking=True)
        elif not st["has_embed"] and hidden_in is not None:
            st["stages"][0]["hid"].copy_(hidden_in.view(1, 1, -1), non_blocking=True)
        elif st.get("hop") is None:
            raise RuntimeError("decode_stage_step: no input (input_ids for the first link, hidden_in for the others)")
        self._run_token(st, cache)
        return st["logits"].clone() if st["has_head"] else st["stages"][-1]["hid"]

    def decode_hop_buffers(self):
        """The executor's own device buffers a layer-split hand-off reads / writes: (token [1, 1] int64 -- first link only --, incoming
        hidden state [1, 1, hidden] fp16, outgoing hidden state, logits [1, 1, vocab] fp32 -- last link only)."""
        st = self._decoder
        if st is None:
            raise RuntimeError("decode_hop_buffers needs enable_decode_graph(cache) first")
        return st["tok"], st["stages"][0]["hid"], st["stages"][-1]["hid"], st["logits"]

    def generate_greedy(self, first_token, cache, num_tokens):
        """num_tokens greedy steps entirely on the device: each replay of a captured hipGraph runs the decode kernels AND the
        argmax that feeds the next step (exl_decoder_step_greedy), so there is no host work between tokens -- the reference's
        loop does `torch.argmax(logits)` + forward per token (test_benchmark_inference.py:188-191).  `first_token` is the
        token at position cache.current_seq_len; returns the num_tokens tokens that follow it (LongTensor on the device) and
        leaves the last step's logits in the executor's buffer (self.last_decoder_logits()).  Needs
        enable_decode_graph(cache, use_graph=True) on a model that sits on one device."""
        st = self._decoder
        if st is None or st["cache"] is not cache or (st["graph"] is None and self.config.tp is None):
            raise RuntimeError("generate_greedy needs enable_decode_graph(cache) with graph replay")
        if len(st["stages"]) != 1 or not (st["has_embed"] and st["has_head"]):
            raise RuntimeError("generate_greedy needs the whole model in one executor stage (one device)")
        self._check_cache_storage(st, cache)
        start = cache.current_seq_len
        if start + num_tokens > cache.max_seq_len:
            raise RuntimeError(f"sequence ({start} + {num_tokens}) exceeds the cache length {cache.max_seq_len}")
        if "history" not in st:                                      # shared with generate_sample, whichever runs first
            st["history"] = torch.zeros((cache.max_seq_len + 1,), dtype=torch.int64, device=st["dev"])
        if self.config.tp is not None:
            st["tok"].copy_(first_token.view(1, 1), non_blocking=True)
            return self._generate_tp(st, cache, num_tokens)
        if "ggraphs" not in st:
            st["ggraphs"] = []
            torch.cuda.synchronize(st["dev"])
            keep_tok, keep_pos = st["tok"].clone(), st["pos"].clone()
            for ns, (limit, _) in zip(self._bucket_splits(st), st["graphs"]):
                cuda_ext.check(ext._lib.exl_decoder_set_kv_splits(st["handle"], ns, None), "decoder_set_kv_splits")
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):                        # capture only records
                    with cuda_ext._Guard(st["dev"]):
                        cuda_ext.check(ext._lib.exl_decoder_step_greedy(st["handle"], st["tok"].data_ptr(), st["pos"].data_ptr(),
                                                                        st["logits"].data_ptr(), st["history"].data_ptr(),
                                                                        torch.cuda.current_stream(st["dev"]).cuda_stream),
                                       "decoder_step_greedy")
                st["ggraphs"].append((limit, g))
            st["tok"].copy_(keep_tok); st["pos"].copy_(keep_pos)
        st["tok"].copy_(first_token.view(1, 1), non_blocking=True)
        if st["dev_pos"] != start:
            self._set_positions(st, start)
        self._count_path("executor_graph", num_tokens)                # (decode_path_report: tokens generated inside the graph count too)
        for i in range(num_tokens):
            p = start + i
            for limit, g in st["ggraphs"]:
                if p <= limit:
                    g.replay()
                    break
            else:
                raise RuntimeError(f"position {p} beyond the decoder's context limit")
        cache.current_seq_len = start + num_tokens
        st["dev_pos"] = cache.current_seq_len
        return st["history"][start + 1:start + num_tokens + 1].clone()

    def generate_sample(self, sequence, cache, num_tokens, settings=None, uniforms=None):
        """num_tokens SAMPLED steps entirely on the device: every replay of a captured hipGraph runs the decode kernels and the
        sampler kernel (exllama_amd/csrc/sampler.hip: repetition penalty, temperature, top-k, top-p / min-p, typical, draw --
        the reference's ExLlamaGenerator.sample / gen_single_token, generator.py:91-170, :344-381, which runs on the host
        between two forward passes).  `sequence` [1, n] is the whole token sequence so far (prompt + generated): its last
        token sits at position cache.current_seq_len and is the step's input, the earlier ones feed the repetition penalty.
        `settings`: exllama_amd._lib.ExlSampler (defaults = the reference's Settings); `uniforms`: optional fp32 tensor
        [>= max_seq_len + 1] of draws indexed by position (else Philox(settings.seed, position)).  Returns the new tokens."""
        from ._lib import ExlSampler
        import ctypes as C
        st = self._decoder
        if st is None or st["cache"] is not cache or (st["graph"] is None and self.config.tp is None):
            raise RuntimeError("generate_sample needs enable_decode_graph(cache) with graph replay")
        if len(st["stages"]) != 1 or not (st["has_embed"] and st["has_head"]):
            raise RuntimeError("generate_sample needs the whole model in one executor stage (one device)")
        self._check_cache_storage(st, cache)
        settings = settings or ExlSampler()
        start = cache.current_seq_len
        seq = sequence.view(-1).to(st["dev"], dtype=torch.int64)
        if seq.numel() != start + 1:
            raise RuntimeError(f"sequence holds {seq.numel()} tokens, the cache position says {start} + 1")
        if start + num_tokens > cache.max_seq_len:
            raise RuntimeError(f"sequence ({start} + {num_tokens}) exceeds the cache length {cache.max_seq_len}")
        if "history" not in st:
            st["history"] = torch.zeros((cache.max_seq_len + 1,), dtype=torch.int64, device=st["dev"])
        st["history"][:start + 1].copy_(seq)
        if self.config.tp is not None:
            st["tok"].copy_(seq[-1:].view(1, 1), non_blocking=True)
            return self._generate_tp(st, cache, num_tokens, settings=settings, uniforms=uniforms)
        key = (bytes(settings), None if uniforms is None else uniforms.data_ptr())
        if st.get("sgraphs_key") != key:                             # the settings are kernel arguments: one set of graphs per setting
            st["sgraphs"], st["sgraphs_key"], st["sampler"], st["uniforms"] = [], key, settings, uniforms
            torch.cuda.synchronize(st["dev"])
            keep_tok, keep_pos, keep_hist = st["tok"].clone(), st["pos"].clone(), st["history"].clone()
            for ns, (limit, _) in zip(self._bucket_splits(st), st["graphs"]):
                cuda_ext.check(ext._lib.exl_decoder_set_kv_splits(st["handle"], ns, None), "decoder_set_kv_splits")
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):                        # capture only records
                    with cuda_ext._Guard(st["dev"]):
                        cuda_ext.check(ext._lib.exl_decoder_step_sample(st["handle"], st["tok"].data_ptr(), st["pos"].data_ptr(),
                                                                        st["logits"].data_ptr(), st["history"].data_ptr(),
                                                                        C.byref(settings), None if uniforms is None else uniforms.data_ptr(),
                                                                        torch.cuda.current_stream(st["dev"]).cuda_stream),
                                       "decoder_step_sample")
                st["sgraphs"].append((limit, g))
            st["tok"].copy_(keep_tok); st["pos"].copy_(keep_pos); st["history"].copy_(keep_hist)
        st["tok"].copy_(seq[-1:].view(1, 1), non_blocking=True)
        if st["dev_pos"] != start:
            self._set_positions(st, start)
        self._count_path("executor_graph", num_tokens)
        for i in range(num_tokens):
            p = start + i
            for limit, g in st["sgraphs"]:
                if p <= limit:
                    g.replay()
                    break
            else:
                raise RuntimeError(f"position {p} beyond the decoder's context limit")
        cache.current_seq_len = start + num_tokens
        st["dev_pos"] = cache.current_seq_len
        return st["history"][start + 1:start + num_tokens + 1].clone()

    def _generate_tp(self, st, cache, num_tokens, settings=None, uniforms=None):
        """Token loop of ONE RANK of a tensor-parallel model (reference for the single-process form: generate_greedy / generate_sample).
        Every rank ends a step with the same all-gathered fp32 logits (bit-identical: tests/test_tp_gpu.py), so every rank picks the
        token itself -- torch.argmax, or the sampler kernel (csrc/sampler.hip) with the same settings and the same draw per position
        (Philox(seed, position) or the caller's `uniforms`) -- and no token travels between ranks.  The loop is sequenced by the host
        (the collectives sit between the kernels of a step) but never synchronises with the device: the token stays there."""
        import ctypes as C
        start = cache.current_seq_len
        lib = ext._lib
        if settings is not None and "probs" not in st:
            st["probs"] = torch.empty((self.config.vocab_size,), dtype=torch.float32, device=st["dev"])
        hist = st["history"]
        self._count_path("executor_pieces_tp", num_tokens)
        for i in range(num_tokens):
            self._run_token(st, cache)                               # advances cache.current_seq_len and the device-side position
            p = start + i + 1                                         # position of the token chosen now
            if settings is None:
                tok = st["logits"].view(-1).argmax()
                hist[p] = tok
                st["tok"].copy_(tok.view(1, 1))
            else:
                with cuda_ext._Guard(st["dev"]):
                    cuda_ext.check(lib.exl_sample(st["dev"].index, st["logits"].data_ptr(), st["probs"].data_ptr(), self.config.vocab_size,
                                                  hist.data_ptr(), st["tok"].data_ptr(), st["stages"][0]["pos"].data_ptr(),
                                                  None if uniforms is None else uniforms.data_ptr(), None, C.byref(settings),
                                                  torch.cuda.current_stream(st["dev"]).cuda_stream), "sample")
        return hist[start + 1:start + num_tokens + 1].clone()

    def last_decoder_logits(self):
        """fp32 logits [1, 1, vocab] of the most recent executor step (a copy)."""
        return self._decoder["logits"].clone()

    @staticmethod
    def _bucket_splits(st):
        return st["bucket_splits"]

    DECODER_CLASSES = ("qkv", "attn", "merge", "o_proj", "gate_up", "down", "head")

    def decoder_profile(self, input_ids, cache, steps=4):
        """Measurement aid for bench.py: per kernel class of the native executor, `steps` passes over all layers' launches
        of that class back to back between two hipEvents (include/exl_amd.h: exl_decoder_step_timed), at the cache's current
        position (not advanced; the K/V slot there is overwritten).  Returns {class: ms per token} for DECODER_CLASSES,
        summed over the executor's stages."""
        import ctypes as C
        st = self._decoder
        if st is None or st["cache"] is not cache:
            raise RuntimeError("decoder_profile needs enable_decode_graph(cache) first")
        if st["has_embed"]:
            st["tok"].copy_(input_ids.view(1, 1))
        self._set_positions(st, cache.current_seq_len)
        st["dev_pos"] = -1
        out = {k: 0.0 for k in self.DECODER_CLASSES}
        for k, sg in enumerate(st["stages"]):
            buf = (C.c_float * len(self.DECODER_CLASSES))()
            last = k == len(st["stages"]) - 1
            with cuda_ext._Guard(sg["tdev"]):
                stream = torch.cuda.current_stream(sg["tdev"]).cuda_stream
                cuda_ext.check(ext._lib.exl_decoder_step_timed(sg["handle"], st["tok"].data_ptr() if (k == 0 and st["has_embed"]) else None,
                                                               sg["pos"].data_ptr(), st["logits"].data_ptr() if (last and st["has_head"]) else None,
                                                               int(steps), stream, buf), "decoder_step_timed")
            for j, name in enumerate(self.DECODER_CLASSES):
                out[name] += float(buf[j])
        return out

    def disable_decode_graph(self):
        st = getattr(self, "_decoder", None)
        if st is not None:
            st["graph"] = None
            st["graphs"] = []
            st.pop("ggraphs", None)
            st.pop("sgraphs", None)
            for sg in st["stages"]:
                ext._lib.exl_decoder_free(sg["handle"])
        self._decoder = None

    def free_unmanaged(self):
        """Release native handles/buffers (reference: model.py:1090-1092)."""
        self.disable_decode_graph()
        ext.cleanup()
