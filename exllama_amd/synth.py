"""Seeded synthetic GPTQ Llama checkpoints (there are no real weights in this environment).

Produces tensors in exactly the key / shape / dtype layout the reference's loader expects
(/root/reference/model.py:141-170, :731-766, :805-841):

    model.embed_tokens.weight  fp16 [V, h]
    model.norm.weight          fp16 [h]
    lm_head.weight             fp16 [V, h]
    model.layers.{i}.input_layernorm.weight / post_attention_layernorm.weight   fp16 [h]
    model.layers.{i}.self_attn.{q,k,v,o}_proj.{qweight,qzeros,scales[,g_idx]}
    model.layers.{i}.mlp.{gate,up,down}_proj.{qweight,qzeros,scales[,g_idx]}
        qweight int32 [K/8, N], qzeros int32 [K/g, N/8], scales fp16 [K/g, N], g_idx int32 [K]

Statistics are chosen so that activations stay finite in fp16 through all layers
(SURVEY.md section 8d): nibbles uniform, stored zero nibble 7 (effective zero 8) or uniform
for the stress variant, scales ~ U(0.5,1.5) * std_target / 4.61.
"""

import json
import math
import os
from dataclasses import dataclass

import torch


@dataclass
class LlamaDims:
    hidden_size: int
    intermediate_size: int
    num_hidden_layers: int
    num_attention_heads: int
    num_key_value_heads: int = None
    vocab_size: int = 32000
    rms_norm_eps: float = 1e-6

    def __post_init__(self):
        if self.num_key_value_heads is None:
            self.num_key_value_heads = self.num_attention_heads

    @property
    def head_dim(self):
        return self.hidden_size // self.num_attention_heads


# Sizes from SURVEY.md section 8 header.
LLAMA_7B = LlamaDims(4096, 11008, 32, 32)
LLAMA_13B = LlamaDims(5120, 13824, 40, 40)
LLAMA_33B = LlamaDims(6656, 17920, 60, 52)
LLAMA_65B = LlamaDims(8192, 22016, 80, 64)
LLAMA2_70B = LlamaDims(8192, 28672, 80, 64, num_key_value_heads=8)             # GQA, the widest down_proj (K = 28672) of the family
# Small shapes for tests (all kernel constraints hold: N % 32 == 0, K % groupsize == 0, kv_heads % 4 == 0).
LLAMA_TINY = LlamaDims(256, 704, 2, 4, vocab_size=512)
LLAMA_TINY_GQA = LlamaDims(512, 1408, 2, 8, num_key_value_heads=4, vocab_size=512)

LLAMA_TINY_HD128 = LlamaDims(512, 1408, 3, 4, vocab_size=512)                       # head_dim 128 like the real models
LLAMA_TINY_HD128_GQA = LlamaDims(1024, 2816, 2, 8, num_key_value_heads=2, vocab_size=640)

PRESETS = {"7b": LLAMA_7B, "13b": LLAMA_13B, "33b": LLAMA_33B, "65b": LLAMA_65B, "70b": LLAMA2_70B,
           "tiny": LLAMA_TINY, "tiny_gqa": LLAMA_TINY_GQA, "tiny_hd128": LLAMA_TINY_HD128,
           "tiny_hd128_gqa": LLAMA_TINY_HD128_GQA}


def make_q4_linear(K, N, groupsize, act_order, gen, device, zeros="sym", std=None, g_idx=None, nibbles="uniform"):
    """One GPTQ linear. Returns dict(qweight, qzeros, scales[, g_idx]).  `g_idx`: reuse this group index instead of drawing one.
    nibbles: "uniform" = every 4-bit value 0..15 equally likely (random bits); "centered" = the same bits with every 0 nibble
    replaced by 8, i.e. a distribution symmetric about the symmetric zero point 8 (see make_checkpoint)."""
    G = K // groupsize
    qweight = torch.randint(-2**31, 2**31 - 1, (K // 8, N), dtype=torch.int64, generator=gen, device=device).to(torch.int32)
    if nibbles == "centered":
        # bit 0 of every nibble of `none` is set where the nibble is 0 (the arithmetic right shifts only disturb bits that the mask drops)
        none = ~(qweight | (qweight >> 1) | (qweight >> 2) | (qweight >> 3)) & 0x11111111
        qweight = qweight | (none << 3)
    elif nibbles != "uniform":
        raise ValueError(f"nibbles: {nibbles!r}")
    if zeros == "sym":
        qzeros = torch.full((G, N // 8), 0x77777777, dtype=torch.int32, device=device)
    else:
        qzeros = torch.randint(-2**31, 2**31 - 1, (G, N // 8), dtype=torch.int64, generator=gen, device=device).to(torch.int32)
    if std is None:
        std = 0.02 * math.sqrt(4096.0 / K)
    scales = ((torch.rand((G, N), generator=gen, device=device) + 0.5) * (std / 4.61)).to(torch.float16)
    out = {"qweight": qweight, "qzeros": qzeros, "scales": scales}
    if act_order and g_idx is not None:
        out["g_idx"] = g_idx.clone()
    elif act_order:
        perm = torch.randperm(K, generator=gen, device=device)
        g_idx = torch.empty(K, dtype=torch.int32, device=device)
        g_idx[perm] = (torch.arange(K, device=device) // groupsize).to(torch.int32)
        out["g_idx"] = g_idx
    return out


def make_checkpoint(dims, groupsize=128, act_order=False, seed=0, device="cpu", zeros="sym", num_layers=None, nibbles="uniform"):
    """Full tensor dict for a Llama of shape `dims` (optionally truncated to `num_layers`).

    nibbles: with zeros="sym" (effective zero point 8) uniform nibbles 0..15 have mean 7.5: every weight carries a bias of -0.5
    scale steps.  One layer does not notice; a DEEP model does -- RMSNorm does not centre, SiLU(g) * u has a positive mean, and the
    common-mode shift the down projection adds to ALL channels grows linearly with depth: measured ~1.8e3 per layer in max |hidden|
    at 13B shapes, the fp16 range is left at layer 36-37 (gpurun_out/r05c), i.e. the 13B / 33B / 65B benchmark models of rounds 1-4
    computed on inf / NaN from there on (same instructions, but not the data -- or the power draw -- of a real model).
    "centered" (what bench.py and the full-depth perplexity runs use) replaces every 0 nibble by 8: values 1..15, symmetric about the
    zero point like the weights GPTQ writes for a real model, and the residual stream stays O(10) through 80 layers.  The default stays
    "uniform" because the committed golden vectors were generated with it (few layers: no difference that matters there).

    act_order: False; True = every matrix draws its own row permutation (the general case the reference's per-matrix x_map allows);
    "gptq" = what GPTQ with desc_act actually writes: the permutation is argsort(diag(H)) of the layer INPUT's Hessian, and the
    matrices quantised against the same input -- q / k / v, and gate / up -- see the same H (GPTQ-for-LLaMa / AutoGPTQ quantise
    them as one group of the sequential pass), so their g_idx tensors are identical; o_proj and down_proj have their own."""
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    L = dims.num_hidden_layers if num_layers is None else num_layers
    h, I, V = dims.hidden_size, dims.intermediate_size, dims.vocab_size
    kvd = dims.num_key_value_heads * dims.head_dim
    t = {}
    t["model.embed_tokens.weight"] = (torch.randn((V, h), generator=gen, device=device) * 0.02).half()
    t["lm_head.weight"] = (torch.randn((V, h), generator=gen, device=device) * 0.02).half()
    t["model.norm.weight"] = (1.0 + 0.05 * torch.randn(h, generator=gen, device=device)).half()
    for i in range(L):
        p = f"model.layers.{i}"
        t[p + ".input_layernorm.weight"] = (1.0 + 0.05 * torch.randn(h, generator=gen, device=device)).half()
        t[p + ".post_attention_layernorm.weight"] = (1.0 + 0.05 * torch.randn(h, generator=gen, device=device)).half()
        for name, (K, N) in (("self_attn.q_proj", (h, h)), ("self_attn.k_proj", (h, kvd)),
                             ("self_attn.v_proj", (h, kvd)), ("self_attn.o_proj", (h, h)),
                             ("mlp.gate_proj", (h, I)), ("mlp.up_proj", (h, I)), ("mlp.down_proj", (I, h))):
            share = None
            if act_order == "gptq" and name in ("self_attn.k_proj", "self_attn.v_proj", "mlp.up_proj"):
                share = t[f"{p}.{'self_attn.q_proj' if name.startswith('self_attn') else 'mlp.gate_proj'}.g_idx"]
            lin = make_q4_linear(K, N, groupsize, bool(act_order), gen, device, zeros=zeros, g_idx=share, nibbles=nibbles)
            for k, v in lin.items():
                t[f"{p}.{name}.{k}"] = v
    return t


def config_dict(dims, num_layers=None):
    return {
        "bos_token_id": 1, "eos_token_id": 2, "pad_token_id": 0,
        "hidden_size": dims.hidden_size,
        "initializer_range": 0.02,
        "intermediate_size": dims.intermediate_size,
        "num_attention_heads": dims.num_attention_heads,
        "num_key_value_heads": dims.num_key_value_heads,
        "num_hidden_layers": dims.num_hidden_layers if num_layers is None else num_layers,
        "rms_norm_eps": dims.rms_norm_eps,
        "vocab_size": dims.vocab_size,
    }


def save_checkpoint(directory, dims, groupsize=128, act_order=False, seed=0, zeros="sym", num_layers=None, nibbles="uniform"):
    """Write config.json + model.safetensors into `directory`; returns (config_path, model_path)."""
    from safetensors.torch import save_file
    os.makedirs(directory, exist_ok=True)
    tensors = make_checkpoint(dims, groupsize, act_order, seed, "cpu", zeros, num_layers, nibbles)
    cfg = os.path.join(directory, "config.json")
    with open(cfg, "w") as f:
        json.dump(config_dict(dims, num_layers), f)
    st = os.path.join(directory, "model.safetensors")
    save_file({k: v.contiguous() for k, v in tensors.items()}, st)
    return cfg, st
