"""CPU restatement of the reference's Llama forward pass -- TEST INFRASTRUCTURE ONLY (see exl_oracle.py header).

Composes the op-level oracle (`exl_oracle.py`) in the order /root/reference/model.py:989-1082 runs the ops:
embedding -> per layer [RMSNorm -> q/k/v -> RoPE -> KV scatter -> attention -> o_proj (+residual) ->
RMSNorm -> gate/up -> SiLU*mul -> down (+residual)] -> final RMSNorm -> lm_head -> fp32 logits.

Used (a) by tests/ as the model-level parity checker (logits, greedy tokens, perplexity) and (b) by bench.py's
`cpu_baseline` leg, where the weights are dequantised ONCE up front (`prepare()`), so the timed region measures the
math of the path, not numpy bit-twiddling.  Parity of this composition is unpinned by the reference for the
floating-point ops (no CPU path and no golden vectors exist there); integer work is bit-exact.
"""

import numpy as np

from . import exl_oracle as O

f16, f32 = np.float16, np.float32


def _np(t):
    """torch tensor / array -> numpy (keeps int32/uint32 bit patterns)."""
    if hasattr(t, "detach"):
        t = t.detach().cpu().numpy()
    return np.asarray(t)


def dequant_rows_f32(qweight, qzeros, scales, g_idx=None):
    """W16[k, n] = h( h(q[k, n] - z[g(k), n]) * s[g(k), n] ) held as fp32, for the ORIGINAL row order of the checkpoint.

    The same bits as exl_oracle.dequant_w16 (exllama_ext/cuda_func/q4_matrix.cu:170-210: exact `__int2half_rn` of the integer
    difference, then one fp16 multiply -- the product of two fp16 values is exact in fp32, so one rounding), computed one
    nibble position at a time to keep the temporaries small at 65B shapes.  g(k) = k // groupsize, or g_idx[k] for an
    act-order checkpoint: x @ W in the original row order equals x[:, x_map] @ W_sequential (q4_matrix.cu:104-168,
    column_remap.cu:7-36) term by term, so the prepared oracle needs neither the repack nor the gather.
    """
    w = np.asarray(qweight).view(np.uint32)
    R, N = w.shape
    K = R * 8
    z = O.unpack_qzeros(qzeros).astype(np.int8)             # [G, N], 1..16
    s = np.asarray(scales).astype(f32)
    gs = K // z.shape[0]
    gi = None if g_idx is None else np.asarray(g_idx).astype(np.int64)
    out = np.empty((K, N), dtype=f32)
    for j in range(8):
        qj = ((w >> np.uint32(4 * j)) & np.uint32(0xF)).astype(np.int8)
        rows = np.arange(j, K, 8)
        grp = rows // gs if gi is None else gi[rows]
        d = (qj - z[grp]).astype(f32)                         # exact small integers
        out[j::8] = (d * s[grp]).astype(f16)
    return out


class OracleLinear:
    def __init__(self, tensors, key):
        self.qweight_raw = _np(tensors[key + ".qweight"]).view(np.uint32)
        self.qzeros = _np(tensors[key + ".qzeros"]).view(np.uint32)
        self.scales = _np(tensors[key + ".scales"]).astype(f16)
        self.g_idx = None
        gk = key + ".g_idx"
        if gk in tensors:
            g_idx = _np(tensors[gk])
            if not (g_idx == 0).all():                      # model.py:147-149: an all-zero g_idx means "no act-order"
                self.g_idx = g_idx
        self._seq = None                                    # (x_map, sequential qweight), built on first use
        self.w32 = None

    def _sequential(self):
        if self._seq is None:
            if self.g_idx is None:
                self._seq = (None, self.qweight_raw)
            else:
                self._seq = O.make_sequential(self.qweight_raw.copy(), self.g_idx, self.qzeros.shape[0])
        return self._seq

    @property
    def x_map(self):
        return self._sequential()[0]

    @property
    def qweight(self):
        return self._sequential()[1]

    def prepare(self):
        """Dequantise once: W16 exactly as the reference reconstructs it, held as fp32 for the BLAS call."""
        self.w32 = dequant_rows_f32(self.qweight_raw, self.qzeros, self.scales, self.g_idx)

    def __call__(self, x, residual=None):
        x = np.asarray(x).astype(f16)
        if self.w32 is not None:
            acc = x.astype(f32) @ self.w32                  # original row order: no gather needed (see dequant_rows_f32)
            if residual is not None:
                acc = acc + np.asarray(residual).astype(f32)
            return acc.astype(f16)
        return O.q4_matmul_recons(x, self.qweight, self.qzeros, self.scales, self.x_map, out=residual)

    def with_lora(self, x, a, b, residual=None):
        """x @ W + (x @ A) @ B in the reference's order (exllama_ext.cpp:245-324): the adapter product is formed first
        (two fp16-rounded GEMMs), the quantised product accumulates onto it; a residual stream is added last."""
        x = np.asarray(x).astype(f16)
        t = (x.astype(f32) @ np.asarray(a).astype(f32)).astype(f16)
        d = (t.astype(f32) @ np.asarray(b).astype(f32)).astype(f16)
        y = self(x, residual=d)
        if residual is not None:
            y = (np.asarray(residual).astype(f32) + y.astype(f32)).astype(f16)
        return y


class OracleLlama:
    def __init__(self, cfg, tensors, max_seq_len=2048, num_layers=None):
        """cfg: dict with the config.json keys; tensors: checkpoint dict (torch or numpy)."""
        self.h = cfg["hidden_size"]
        self.heads = cfg["num_attention_heads"]
        self.kv_heads = cfg.get("num_key_value_heads", self.heads)
        self.hd = self.h // self.heads
        self.eps = cfg["rms_norm_eps"]
        self.L = cfg["num_hidden_layers"] if num_layers is None else num_layers
        self.max_seq_len = max_seq_len
        self.embed = _np(tensors["model.embed_tokens.weight"]).astype(f16).copy()
        self.embed[cfg.get("pad_token_id", 0)] = 0
        self.lm_head = _np(tensors["lm_head.weight"]).astype(f16)
        self.norm_w = _np(tensors["model.norm.weight"]).astype(f16)
        self.sin, self.cos = O.rope_tables(max_seq_len, self.hd, cfg.get("rope_theta", 10000.0))
        self.layers = []
        for i in range(self.L):
            p = f"model.layers.{i}"
            self.layers.append({
                "in_norm": _np(tensors[p + ".input_layernorm.weight"]).astype(f16),
                "post_norm": _np(tensors[p + ".post_attention_layernorm.weight"]).astype(f16),
                "q": OracleLinear(tensors, p + ".self_attn.q_proj"),
                "k": OracleLinear(tensors, p + ".self_attn.k_proj"),
                "v": OracleLinear(tensors, p + ".self_attn.v_proj"),
                "o": OracleLinear(tensors, p + ".self_attn.o_proj"),
                "gate": OracleLinear(tensors, p + ".mlp.gate_proj"),
                "up": OracleLinear(tensors, p + ".mlp.up_proj"),
                "down": OracleLinear(tensors, p + ".mlp.down_proj"),
            })
        self.lora = {}                     # "model.layers.i.<block>.<proj>" -> (A [in, r], B [r, out]) as exllama_amd.lora stores them
        self.reset()

    def set_lora(self, tensors):
        """tensors: the `.tensors` dict of an ExLlamaLora (or None to clear)."""
        self.lora = {}
        for k, v in (tensors or {}).items():
            if k.endswith(".lora_A.weight"):
                base = k[:-len(".lora_A.weight")]
                self.lora[base] = (_np(v), _np(tensors[base + ".lora_B.weight"]))

    def _lin(self, i, which, key, x, residual=None):
        lin = self.layers[i][which]
        ab = self.lora.get(f"model.layers.{i}.{key}")
        return lin.with_lora(x, ab[0], ab[1], residual) if ab else lin(x, residual=residual)

    def prepare(self):
        for l in self.layers:
            for k in ("q", "k", "v", "o", "gate", "up", "down"):
                l[k].prepare()

    def reset(self, bsz=1):
        self.past = 0
        shape = (bsz, self.kv_heads, self.max_seq_len, self.hd)
        self.kc = [np.zeros(shape, dtype=f16) for _ in range(self.L)]
        self.vc = [np.zeros(shape, dtype=f16) for _ in range(self.L)]

    def layer_forward(self, i, hidden):
        """hidden [bsz, q_len, h] fp16 -> new hidden (residuals added inside the o/down matmul rounding, like the
        fused ops of the reference and like exllama_amd.model)."""
        l = self.layers[i]
        bsz, q_len, h = hidden.shape
        x2 = hidden.reshape(-1, h)
        xn = O.rms_norm(x2, l["in_norm"], self.eps)
        q = self._lin(i, "q", "self_attn.q_proj", xn).reshape(bsz, -1)
        k = self._lin(i, "k", "self_attn.k_proj", xn).reshape(bsz, -1)
        v = self._lin(i, "v", "self_attn.v_proj", xn).reshape(bsz, q_len, -1)
        q = O.rope(q, self.sin, self.cos, self.past, self.heads, self.hd).reshape(bsz, q_len, self.heads, self.hd)
        k = O.rope(k, self.sin, self.cos, self.past, self.kv_heads, self.hd).reshape(bsz, q_len, -1)
        O.update_cache(k, v, self.kc[i], self.vc[i], self.past)
        kv_len = self.past + q_len
        a = O.attention(q.transpose(0, 2, 1, 3), self.kc[i][:bsz, :, :kv_len], self.vc[i][:bsz, :, :kv_len],
                        causal_past_len=self.past)
        a = a.transpose(0, 2, 1, 3).reshape(-1, h)
        x2 = self._lin(i, "o", "self_attn.o_proj", a, residual=x2)
        xn = O.rms_norm(x2, l["post_norm"], self.eps)
        act = O.silu_mul(self._lin(i, "gate", "mlp.gate_proj", xn), self._lin(i, "up", "mlp.up_proj", xn))
        x2 = self._lin(i, "down", "mlp.down_proj", act, residual=x2)
        return x2.reshape(bsz, q_len, h)

    def forward(self, input_ids, last_id_only=True):
        """input_ids [bsz, q_len] ints -> fp32 logits [bsz, 1 or q_len, V]; advances the cache position."""
        ids = np.asarray(input_ids)
        hidden = self.embed[ids]
        for i in range(self.L):
            hidden = self.layer_forward(i, hidden)
        self.past += ids.shape[1]
        if last_id_only:
            hidden = hidden[:, -1:, :]
        bsz, q, h = hidden.shape
        hn = O.rms_norm(hidden.reshape(-1, h), self.norm_w, self.eps)
        logits = (hn.astype(f32) @ self.lm_head.astype(f32).T).astype(f16).astype(f32)
        return logits.reshape(bsz, q, -1)


class TruthLlama(OracleLlama):
    """The same forward pass with NO fp16 rounding anywhere: every op of `layer_forward` / `forward` in float64, in the same order,
    on the same model constants (the fp16 embedding / norm / head weights, the fp16 RoPE tables of model.py:864-877 and the
    dequantised weights h(h(q - z) * s) of q4_matrix.cu:170-210 -- all of them exact in float64).  It states what the fp16 pipeline
    approximates, so a test can require  |HIP - truth| <= c * |fp16 oracle - truth|  instead of widening a bound where the fp16
    oracle itself is ill-conditioned (tests/test_model_gpu.py: _truth_close).  K / V rows are cached unrounded (float64 cache);
    rows copied in from an fp16 cache are taken as given.  LoRA operands (set_lora): out = x W + (x A) B, all in float64."""

    @classmethod
    def from_oracle(cls, ref, past=None):
        """A truth model on the SAME weight objects as the fp16 oracle `ref` (nothing is dequantised twice), continuing from the
        first `past` (default: ref.past) cached positions of `ref` -- taken as given, widened to float64."""
        t = cls.__new__(cls)
        t.__dict__.update({k: v for k, v in ref.__dict__.items() if k not in ("kc", "vc", "past")})
        t.reset(ref.kc[0].shape[0])
        t.past = ref.past if past is None else past
        for i in range(t.L):
            t.kc[i][:, :, :t.past] = ref.kc[i][:, :, :t.past]
            t.vc[i][:, :, :t.past] = ref.vc[i][:, :, :t.past]
        return t

    def reset(self, bsz=1):
        self.past = 0
        shape = (bsz, self.kv_heads, self.max_seq_len, self.hd)
        self.kc = [np.zeros(shape, dtype=np.float64) for _ in range(self.L)]
        self.vc = [np.zeros(shape, dtype=np.float64) for _ in range(self.L)]

    @staticmethod
    def _norm(x, w, eps):
        return x / np.sqrt((x * x).mean(axis=-1, keepdims=True) + eps) * np.asarray(w).astype(np.float64)

    def _rope(self, x, heads):
        """x [bsz, rows * hd] float64, rotate-half with the fp16 tables (rope.cu:27-87 without its three roundings)."""
        bsz = x.shape[0]
        xr = x.reshape(bsz, -1, self.hd)
        hd2 = self.hd // 2
        pos = self.past + np.arange(xr.shape[1]) // heads
        s = np.asarray(self.sin).reshape(-1, self.hd)[pos].astype(np.float64)[None]
        c = np.asarray(self.cos).reshape(-1, self.hd)[pos].astype(np.float64)[None]
        l, r = xr[:, :, :hd2], xr[:, :, hd2:]
        out = np.concatenate([l * c[:, :, :hd2] - r * s[:, :, :hd2], r * c[:, :, hd2:] + l * s[:, :, hd2:]], axis=-1)
        return out.reshape(x.shape)

    _KEYS = {"q": "self_attn.q_proj", "k": "self_attn.k_proj", "v": "self_attn.v_proj", "o": "self_attn.o_proj",
             "gate": "mlp.gate_proj", "up": "mlp.up_proj", "down": "mlp.down_proj"}

    def _mm(self, i, which, x):
        ab = self.lora.get(f"model.layers.{i}.{self._KEYS[which]}")
        base = self._mm_base(i, which, x)
        if ab is None:
            return base
        return base + (x @ np.asarray(ab[0]).astype(np.float64)) @ np.asarray(ab[1]).astype(np.float64)

    def _mm_base(self, i, which, x):
        lin = self.layers[i][which]
        if lin.w32 is None:
            lin.prepare()
        K = lin.w32.shape[0]
        acc = np.zeros((x.shape[0], lin.w32.shape[1]), dtype=np.float64)
        for k0 in range(0, K, 2048):                        # the fp16-exact weights are held as fp32: widen a slab at a time
            acc += x[:, k0:k0 + 2048] @ lin.w32[k0:k0 + 2048].astype(np.float64)
        return acc

    def layer_forward(self, i, hidden):
        l = self.layers[i]
        bsz, q_len, h = hidden.shape
        x2 = hidden.reshape(-1, h)
        xn = self._norm(x2, l["in_norm"], self.eps)
        q = self._rope(self._mm(i, "q", xn).reshape(bsz, -1), self.heads).reshape(bsz, q_len, self.heads, self.hd)
        k = self._rope(self._mm(i, "k", xn).reshape(bsz, -1), self.kv_heads).reshape(bsz, q_len, -1)
        v = self._mm(i, "v", xn).reshape(bsz, q_len, -1)
        O.update_cache(k, v, self.kc[i], self.vc[i], self.past)
        kv_len = self.past + q_len
        qh = q.transpose(0, 2, 1, 3)
        kk, vv = self.kc[i][:bsz, :, :kv_len], self.vc[i][:bsz, :, :kv_len]
        rep = self.heads // self.kv_heads
        if rep > 1:
            kk, vv = np.repeat(kk, rep, axis=1), np.repeat(vv, rep, axis=1)
        s = np.einsum("bhqd,bhkd->bhqk", qh, kk) / np.sqrt(self.hd)
        qi, kj = np.arange(q_len)[:, None], np.arange(kv_len)[None, :]
        s = np.where(kj > self.past + qi, -np.inf, s)
        p = np.exp(s - s.max(axis=-1, keepdims=True))
        a = np.einsum("bhqk,bhkd->bhqd", p / p.sum(axis=-1, keepdims=True), vv)
        a = a.transpose(0, 2, 1, 3).reshape(-1, h)
        x2 = x2 + self._mm(i, "o", a)
        xn = self._norm(x2, l["post_norm"], self.eps)
        g, u = self._mm(i, "gate", xn), self._mm(i, "up", xn)
        x2 = x2 + self._mm(i, "down", g / (1.0 + np.exp(-g)) * u)
        return x2.reshape(bsz, q_len, h)

    def forward(self, input_ids, last_id_only=True):
        ids = np.asarray(input_ids)
        hidden = self.embed[ids].astype(np.float64)
        for i in range(self.L):
            hidden = self.layer_forward(i, hidden)
        self.past += ids.shape[1]
        if last_id_only:
            hidden = hidden[:, -1:, :]
        bsz, q, h = hidden.shape
        hn = self._norm(hidden.reshape(-1, h), self.norm_w, self.eps)
        return (hn @ self.lm_head.astype(np.float64).T).reshape(bsz, q, -1)
