"""Host-side mirror of the reference's operator surface for the GPTQ hot path.

Same names, argument meaning and error behaviour as /root/reference/cuda_ext.py:87-166 and the pybind
module it JIT-builds (/root/reference/exllama_ext/exllama_ext.cpp:743-762) -- but backed by the
hand-written HIP library `libexl_amd.so` through its C ABI (include/exl_amd.h).  The reference's
`model.py` / `generator.py` run against this module unchanged (`import cuda_ext` resolves to the shim at
the repository root).

    cuda_ext.exllama_ext.<fn>(...)      the 16 functions of the pybind module
    cuda_ext.ext_make_q4 / ext_q4_matmul / ext_half_matmul / ext_rope_ / ext_rms_norm(_) /
    cuda_ext.ext_rep_penalty_mask_cpu / ext_apply_rep_penalty_mask_cpu
    cuda_ext.none_tensor                 the meta-device "None" sentinel (cuda_ext.py:82)

There is no fallback: tensors must live on a HIP device ("cuda:N" in PyTorch-ROCm) and the native
library must be built, otherwise these functions raise RuntimeError.
"""

import ctypes as C

import torch

from . import _lib
from ._lib import check

# Dummy tensor to pass instead of None (reference: cuda_ext.py:80-82)
none_tensor = torch.empty((1, 1), device="meta")


def _is_none(t):
    return t is none_tensor or t is None or t.device.type == "meta"      # (identity first: the reference passes THE sentinel, 14 times per layer)


def _ptr(t):
    return None if _is_none(t) else t.data_ptr()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_raw_device = getattr(torch._C, "_cuda_getDevice", None)


def _stream(t):
    """torch's current HIP stream on t's device as a raw handle.  The private accessor saves building a torch.cuda.Stream object per
    call (measured on the drop-in path, scripts/bench_dropin.py --profile: 4.5 us x 3 ops per layer)."""
    if _raw_stream is not None:
        return _raw_stream(t.device.index)
    return torch.cuda.current_stream(t.device).cuda_stream


def _all_cuda_contiguous(*ts):
    for t in ts:
        if not (t.is_cuda and t.is_contiguous()):
            return False
    return True


def _req(cond, msg):
    if not cond:
        raise RuntimeError(msg)


def _req_dtype(t, dtype, name):
    _req(t.dtype == dtype, f"{name} is incorrect datatype, must be {dtype}")


def _req_cuda(t, name):
    _req(t.is_cuda, f"{name} must be on a HIP device (exllama_amd has no CPU path), got {t.device}")
    _req(t.is_contiguous(), f"{name} must be contiguous")


class _Guard:
    """Make `device` current for the duration of a native call (reference: OptionalCUDAGuard)."""

    def __init__(self, device):
        self.idx = device.index if isinstance(device, torch.device) else int(device)
        self.prev = None

    def __enter__(self):
        cur = _raw_device() if _raw_device is not None else torch.cuda.current_device()
        if cur != self.idx:
            self.prev = cur
            torch.cuda.set_device(self.idx)

    def __exit__(self, *a):
        if self.prev is not None:
            torch.cuda.set_device(self.prev)


class _ExllamaExt:
    """Drop-in for the pybind module `exllama_ext` (exllama_ext.cpp:743-762)."""

    def __init__(self):
        self._lib = _lib.load()
        self.tuning = _lib.ExlTuning(8, 2, 8, 0, 0, 0, 0, 0, 0)
        self._dims = {}                 # handle -> (height, width): avoids an FFI round trip per matmul

    # -- exllama_ext.cpp:89-112
    def set_tuning_params(self, matmul_recons_thd, fused_mlp_thd, sdp_thd, matmul_fused_remap, rmsnorm_no_half2,
                          rope_no_half2, matmul_no_half2, silu_no_half2, concurrent_streams):
        self.tuning = _lib.ExlTuning(int(matmul_recons_thd), int(fused_mlp_thd), int(sdp_thd), int(matmul_fused_remap),
                                     int(rmsnorm_no_half2), int(rope_no_half2), int(matmul_no_half2),
                                     int(silu_no_half2), int(concurrent_streams))
        check(self._lib.exl_set_tuning(C.byref(self.tuning)), "set_tuning_params")

    # -- exllama_ext.cpp:126-152
    def prepare_buffers(self, device, temp_state, temp_mlp, temp_zeros_float, temp_dq):
        device = torch.device(device)
        _req(device.index is not None and device.index >= 0, "no device index")
        for t, n in ((temp_state, "temp_state"), (temp_mlp, "temp_mlp"), (temp_zeros_float, "temp_zeros_float"), (temp_dq, "temp_dq")):
            _req_cuda(t, n)
        with _Guard(device):
            check(self._lib.exl_prepare_buffers(device.index, temp_state.data_ptr(), temp_state.numel(),
                                                temp_mlp.data_ptr(), temp_mlp.numel(), temp_zeros_float.data_ptr(),
                                                temp_zeros_float.size(-1), temp_dq.data_ptr(), temp_dq.numel()),
                  "prepare_buffers")

    # -- exllama_ext.cpp:117-121
    def cleanup(self):
        self._dims.clear()
        check(self._lib.exl_cleanup(), "cleanup")

    # -- exllama_ext.cpp:157-194
    def make_q4(self, qweight, qzeros, scales, g_idx, device):
        _req_dtype(qweight, torch.int32, "qweight")
        _req_dtype(qzeros, torch.int32, "qzeros")
        _req_dtype(scales, torch.float16, "scales")
        if not _is_none(g_idx):
            _req_dtype(g_idx, torch.int32, "g_idx")
        _req(qweight.size(1) == qzeros.size(1) * 8, "qweight and qzeros have incompatible shapes")
        _req(scales.size(1) == qweight.size(1), "scales and qweight have incompatible shapes")
        _req(qzeros.size(0) == scales.size(0), "qzeros and scales have incompatible shapes")
        for t, n in ((qweight, "qweight"), (qzeros, "qzeros"), (scales, "scales")):
            _req_cuda(t, n)
        _req(device is not None and device >= 0, "no device index")
        g_host = None
        if not _is_none(g_idx):
            g_host = g_idx.detach().to("cpu").contiguous()          # the reference passes the CPU tensor (model.py:144)
            _req(g_host.numel() == qweight.size(0) * 8, "g_idx and qweight have incompatible shapes")
        handle = C.c_void_p()
        with _Guard(device):
            check(self._lib.exl_make_q4(int(device), qweight.size(0) * 8, qweight.size(1), qzeros.size(0),
                                        qweight.data_ptr(), qzeros.data_ptr(), scales.data_ptr(),
                                        None if g_host is None else g_host.data_ptr(), _stream(qweight),
                                        C.byref(handle)), "make_q4")
        self._dims[handle.value] = (qweight.size(0) * 8, qweight.size(1))
        return handle.value

    def q4_info(self, w):
        vals = [C.c_int() for _ in range(5)]
        xm = C.c_void_p()
        check(self._lib.exl_q4_info(w, *[C.byref(v) for v in vals], C.byref(xm)), "q4_info")
        keys = ("device", "height", "width", "groups", "groupsize")
        d = {k: v.value for k, v in zip(keys, vals)}
        d["x_map"] = xm.value
        lay = C.c_int()
        check(self._lib.exl_q4_layout(w, C.byref(lay)), "q4_layout")
        d["layout"] = lay.value          # 1: qweight was re-tiled in place into the T16 streaming layout
        return d

    def _check_mm(self, x, w, out):
        _req_dtype(x, torch.float16, "x")
        _req_dtype(out, torch.float16, "out")
        _req_cuda(x, "x")
        _req_cuda(out, "out")
        _req(x.size(0) == out.size(0), "x and out have incompatible shapes")
        dims = self._dims.get(w)
        _req(dims is not None, "invalid q4 handle")
        _req(dims[0] == x.size(-1), "x and w have incompatible shapes")
        _req(out.size(-1) == dims[1], "out and w have incompatible shapes")

    # -- exllama_ext.cpp:199-240
    def q4_matmul(self, x, w, out):
        self._check_mm(x, w, out)
        with _Guard(x.device):
            check(self._lib.exl_q4_matmul(w, x.data_ptr(), x.size(0), out.data_ptr(), 0, _stream(x)), "q4_matmul")

    # individual kernels (used by tests and by -v style A/B comparisons)
    def q4_matmul_gemv(self, x, w, out, no_zero=False):
        self._check_mm(x, w, out)
        with _Guard(x.device):
            check(self._lib.exl_q4_matmul_gemv(w, x.data_ptr(), x.size(0), out.data_ptr(), int(no_zero), _stream(x)), "q4_matmul_gemv")

    def q4_matmul_gemm(self, x, w, out, no_zero=False):
        self._check_mm(x, w, out)
        with _Guard(x.device):
            check(self._lib.exl_q4_matmul_gemm(w, x.data_ptr(), x.size(0), out.data_ptr(), int(no_zero), _stream(x)), "q4_matmul_gemm")

    def q4_matmul_dual(self, x, w1, w2, out1, out2=None, silu=True):
        """out1 = silu(x @ W1) * (x @ W2) (silu) or out1, out2 = x @ W1, x @ W2 in one kernel (include/exl_amd.h:
        exl_q4_matmul_dual).  Returns False when the pair / row count is not eligible: nothing was launched."""
        self._check_mm(x, w1, out1)
        if out2 is not None:
            self._check_mm(x, w2, out2)
        done = C.c_int()
        with _Guard(x.device):
            check(self._lib.exl_q4_matmul_dual(w1, w2, x.data_ptr(), x.size(0), out1.data_ptr(), _ptr(out2), int(silu), _stream(x),
                                               C.byref(done)), "q4_matmul_dual")
        return bool(done.value)

    def q4_qkv_rope_cache(self, x, wq, wk, wv, q_out, sin, cos, key_cache, value_cache, q_len, past_len, num_heads, num_kv_heads,
                          head_dim, max_seq_len, norm_weight=None, eps=0.0):
        """q_out = rope(x @ Wq), key_cache <- rope(x @ Wk), value_cache <- x @ Wv at past_len in one kernel (include/exl_amd.h:
        exl_q4_qkv_rope_cache; with `norm_weight` exl_q4_attn_prompt: x is the residual stream and RMSNorm runs as the launch's
        prologue).  x: [bsz * q_len, hidden].  Returns False when the shapes are not eligible: nothing was launched."""
        if x.dtype != torch.float16 or q_out.dtype != torch.float16 or not x.is_contiguous() or not q_out.is_contiguous():
            raise RuntimeError("q4_qkv_rope_cache: x and q_out must be contiguous fp16")
        rows = x.size(0)
        if rows % q_len != 0 or q_out.size(0) != rows or q_out.size(1) != num_heads * head_dim:
            raise RuntimeError("q4_qkv_rope_cache: x, q_out and q_len have incompatible shapes")
        done = C.c_int()
        with _Guard(x.device):
            check(self._lib.exl_q4_attn_prompt(wq, wk, wv, x.data_ptr(), _ptr(norm_weight), float(eps), rows // q_len, q_len, q_out.data_ptr(),
                                               sin.data_ptr(), cos.data_ptr(), key_cache.data_ptr(), value_cache.data_ptr(), num_heads,
                                               num_kv_heads, head_dim, past_len, max_seq_len, _stream(x), C.byref(done)), "q4_qkv_rope_cache")
        return bool(done.value)

    def q4_mlp_prompt(self, x, norm_weight, eps, gate, up, down, act):
        """x += down(silu(gate(n)) * up(n)), n = rms_norm(x), for a prompt of more than 512 rows (include/exl_amd.h: exl_q4_mlp_prompt).
        x: [rows, hidden] in place; act: scratch [rows, intermediate].  Returns False when not eligible: nothing was launched."""
        for t, n in ((x, "x"), (norm_weight, "rms_norm_weight"), (act, "act")):
            _req_dtype(t, torch.float16, n)
            _req_cuda(t, n)
        _req(x.is_contiguous() and act.is_contiguous() and act.size(0) == x.size(0), "x and act have incompatible shapes")
        done = C.c_int()
        with _Guard(x.device):
            check(self._lib.exl_q4_mlp_prompt(x.data_ptr(), norm_weight.data_ptr(), float(eps), gate, up, down, x.size(0), act.data_ptr(),
                                              _stream(x), C.byref(done)), "q4_mlp_prompt")
        return bool(done.value)

    def q4_layer_prompt(self, x, bsz, q_len, past_len, in_norm_weight, post_norm_weight, eps, wq, wk, wv, wo, wgate, wup, wdown, sin, cos,
                        key_cache, value_cache, num_heads, num_kv_heads, head_dim, max_seq_len, rowsq=None, rowsq_in_slots=0):
        """One decoder layer of a SHORT prompt (2 .. 256 rows), in place on the residual stream x [bsz * q_len, hidden]: every launch of
        the layer enqueued by one call (include/exl_amd.h: exl_q4_layer_prompt).  Returns (taken, rowsq_out_slots); taken False: the
        layer is not covered, nothing that touches x or the cache was launched.  rowsq (fp32 scratch, rows * (hidden / 16 + 4) floats):
        carries the RMSNorm partial sums from this layer's down_proj to the next layer's call (rowsq_in_slots = what the previous call
        returned, 0 when x was written by anything else since)."""
        for t, n in ((x, "x"), (in_norm_weight, "input_layernorm"), (post_norm_weight, "post_attention_layernorm"), (key_cache, "key_cache"),
                     (value_cache, "value_cache")):
            _req_dtype(t, torch.float16, n)
            _req_cuda(t, n)
        _req(x.is_contiguous() and x.dim() == 2 and x.size(0) == bsz * q_len, "x must be a contiguous [bsz * q_len, hidden] tensor")
        done, slots = C.c_int(), C.c_int()
        if rowsq is not None:
            _req_dtype(rowsq, torch.float32, "rowsq")
            _req_cuda(rowsq, "rowsq")
            _req(rowsq.is_contiguous() and rowsq.device == x.device, "rowsq must be a contiguous fp32 tensor on x's device")
        with _Guard(x.device):
            check(self._lib.exl_q4_layer_prompt(x.data_ptr(), bsz, q_len, past_len, in_norm_weight.data_ptr(), post_norm_weight.data_ptr(),
                                                float(eps), wq, wk, wv, wo, wgate, wup, wdown, sin.data_ptr(), cos.data_ptr(),
                                                key_cache.data_ptr(), value_cache.data_ptr(), num_heads, num_kv_heads, head_dim, max_seq_len,
                                                _stream(x), rowsq.data_ptr() if rowsq is not None else None,
                                                rowsq.numel() if rowsq is not None else 0, int(rowsq_in_slots) if rowsq is not None else 0,
                                                C.byref(slots), C.byref(done)), "q4_layer_prompt")
        return bool(done.value), int(slots.value)

    def q4_matmul_frag(self, x, ws, outs=None, norm_weight=None, eps=0.0, no_zero=False, dual=False, kernel=0, rowsq_in=None, rowsq_out=None):
        """The short-prompt product by itself (include/exl_amd.h: exl_q4_matmul_frag): x [rows, K] against 1 .. 3 matrices `ws` (handles)
        in one launch -> `outs` (row-major tensors, written / accumulated in place), or dual=True: returns silu(x @ W0) * (x @ W1) in
        FRAGMENT ORDER as a uint8 tensor of exl_frag_bytes(rows, width) bytes (unfrag() turns it back).  Returns None when the launch is not
        covered.  rowsq_in = (fp32 tensor [rows, slots]) partial sums of squares of x for the norm; rowsq_out = fp32 tensor of at least
        rows * (width / 16 + 4) elements that receives those of the (single) output: self.last_rowsq_slots says how many per row."""
        _req_dtype(x, torch.float16, "x")
        _req_cuda(x, "x")
        _req(x.is_contiguous() and x.dim() == 2, "x must be a contiguous [rows, K] tensor")
        rows = x.size(0)
        harr = (C.c_void_p * len(ws))(*ws)
        done, oslots = C.c_int(), C.c_int()
        for t in (rowsq_in, rowsq_out):
            if t is not None:
                _req_dtype(t, torch.float32, "rowsq")
                _req_cuda(t, "rowsq")
                _req(t.is_contiguous(), "rowsq tensors must be contiguous")
        oarr = None
        frag = None
        if dual:
            info = self.q4_info(ws[0]) if hasattr(self, "q4_info") else None
            width = info["width"] if info else None
            _req(width is not None, "q4_matmul_frag: cannot read the matrix width")
            frag = torch.zeros(int(self._lib.exl_frag_bytes(rows, width)), dtype=torch.uint8, device=x.device)
        else:
            for o in outs:
                _req_dtype(o, torch.float16, "out")
                _req_cuda(o, "out")
                _req(o.is_contiguous() and o.size(0) == rows, "outputs must be contiguous [rows, width] tensors")
            oarr = (C.c_void_p * len(outs))(*[o.data_ptr() for o in outs])
        with _Guard(x.device):
            check(self._lib.exl_q4_matmul_frag(harr, len(ws), x.data_ptr(), rows, norm_weight.data_ptr() if norm_weight is not None else None,
                                               float(eps), oarr, int(bool(no_zero)), int(bool(dual)), frag.data_ptr() if dual else None,
                                               int(kernel), _stream(x), rowsq_in.data_ptr() if rowsq_in is not None else None,
                                               rowsq_in.size(1) if rowsq_in is not None else 0,
                                               rowsq_out.data_ptr() if rowsq_out is not None else None, C.byref(oslots), C.byref(done)),
                  "q4_matmul_frag")
        self.last_rowsq_slots = int(oslots.value)
        if not done.value:
            return None
        return frag if dual else outs

    @staticmethod
    def unfrag(frag, rows, K):
        """Fragment order -> row-major [rows, K] fp16 (index arithmetic of include/exl_amd.h: exl_q4_matmul_frag; tests)."""
        pad = frag.numel() // (2 * K)
        t = frag.view(torch.float16).view(pad // 16, K // 128, 4, 4, 16, 8)        # [mt][rb][j][kg][r][8]
        return t.permute(0, 4, 1, 3, 2, 5).reshape(pad, K)[:rows].contiguous()     # row 16 mt + r, k 128 rb + 32 kg + 8 j + e

    def q4_reconstruct(self, w, out):
        _req_dtype(out, torch.float16, "out")
        _req_cuda(out, "out")
        info = self.q4_info(w)
        _req(out.numel() >= info["height"] * info["width"], "out is too small")
        with _Guard(out.device):
            check(self._lib.exl_q4_reconstruct(w, out.data_ptr(), _stream(out)), "q4_reconstruct")

    # -- exllama_ext.cpp:245-324
    def q4_matmul_lora(self, x, w, out, lora_A, lora_B, lora_temp):
        self._check_mm(x, w, out)
        _req(x.size(0) == lora_temp.size(0), "x and lora_temp have incompatible shapes")
        _req(x.size(1) == lora_A.size(0), "x and lora_A have incompatible shapes")
        _req(lora_A.size(1) == lora_B.size(0), "lora_A and lora_B have incompatible shapes")
        _req(lora_B.size(1) == out.size(1), "lora_B and out have incompatible shapes")
        for t, n in ((lora_A, "lora_A"), (lora_B, "lora_B"), (lora_temp, "lora_temp")):
            _req_dtype(t, torch.float16, n)
            _req_cuda(t, n)
        with _Guard(x.device):
            check(self._lib.exl_q4_matmul_lora(w, x.data_ptr(), x.size(0), out.data_ptr(), lora_A.data_ptr(),
                                               lora_B.data_ptr(), lora_A.size(1), lora_temp.data_ptr(), _stream(x)),
                  "q4_matmul_lora")

    # -- exllama_ext.cpp:328-358
    def column_remap(self, x, x_new, x_map):
        _req_dtype(x, torch.float16, "x")
        _req_dtype(x_new, torch.float16, "x_new")
        _req_dtype(x_map, torch.int32, "x_map")
        _req(x_map.size(0) == x.size(1), "x_map and x have incompatible shapes")
        _req(x_new.numel() >= x.size(0) * x.size(1), "x_new is too small")
        for t, n in ((x, "x"), (x_new, "x_new"), (x_map, "x_map")):
            _req_cuda(t, n)
        with _Guard(x.device):
            check(self._lib.exl_column_remap(x.data_ptr(), x_new.data_ptr(), x.size(0), x.size(1), x_map.data_ptr(),
                                             _stream(x)), "column_remap")

    # -- exllama_ext.cpp:362-422.  half_matmul expects a pre-zeroed `out` and accumulates (the reference's split-K
    #    atomics do); half_matmul_cublas overwrites.
    def _half_mm(self, x, w, out, no_zero, name):
        for t, n in ((x, "x"), (w, "w"), (out, "out")):
            _req_dtype(t, torch.float16, n)
            _req_cuda(t, n)
        _req(x.size(1) == w.size(0), "x and w have incompatible shapes")
        with _Guard(x.device):
            check(self._lib.exl_half_matmul(x.data_ptr(), w.data_ptr(), out.data_ptr(), x.size(0), x.size(1), w.size(1),
                                            int(no_zero), _stream(x)), name)

    def half_matmul(self, x, w, out):
        self._half_mm(x, w, out, True, "half_matmul")

    def half_matmul_cublas(self, x, w, out):
        self._half_mm(x, w, out, False, "half_matmul_cublas")

    # -- exllama_ext.cpp:606-643
    def rms_norm(self, x, w, out, epsilon):
        for t, n in ((x, "x"), (w, "w"), (out, "out")):
            _req_dtype(t, torch.float16, n)
            _req_cuda(t, n)
        _req(x.size(1) == w.size(0), "x and w have incompatible shapes")
        _req(x.size(0) == out.size(0) and x.size(1) == out.size(1), "x and out have incompatible shapes")
        with _Guard(x.device):
            check(self._lib.exl_rms_norm(x.data_ptr(), w.data_ptr(), out.data_ptr(), float(epsilon), x.size(0), x.size(1),
                                         _stream(x)), "rms_norm")

    # -- not exllama_ext functions (the reference calls torch here: model.py:1002, :1077): HIP forms of the embedding lookup and of the
    #    prompt pass' lm_head for a few rows, so that the token path carries no ATen / BLAS kernel
    def embedding(self, ids, table, out):
        _req_dtype(ids, torch.int64, "ids"); _req_dtype(table, torch.float16, "table"); _req_dtype(out, torch.float16, "out")
        for t, n in ((ids, "ids"), (table, "table"), (out, "out")):
            _req_cuda(t, n)
        _req(out.numel() == ids.numel() * table.size(1), "out does not hold ids.numel() rows of the table")
        with _Guard(table.device):
            check(self._lib.exl_embedding(ids.data_ptr(), table.data_ptr(), out.data_ptr(), ids.numel(), table.size(1), table.size(0),
                                          _stream(table)), "embedding")

    def head_matmul(self, x, w, out):
        """out[r, v] = float(half(x[r] . w[v])); False (nothing launched) when x has more rows than the kernel stages."""
        _req_dtype(x, torch.float16, "x"); _req_dtype(w, torch.float16, "w"); _req_dtype(out, torch.float32, "out")
        for t, n in ((x, "x"), (w, "w"), (out, "out")):
            _req_cuda(t, n)
        _req(x.size(1) == w.size(1) and out.size(0) == x.size(0) and out.size(1) == w.size(0), "x, w and out have incompatible shapes")
        with _Guard(x.device):
            rc = self._lib.exl_head_matmul(x.data_ptr(), w.data_ptr(), out.data_ptr(), x.size(0), x.size(1), w.size(0), _stream(x))
        if rc == 1:
            return False
        check(rc, "head_matmul")
        return True

    # -- exllama_ext.cpp:647-680
    def rope_(self, x, sin, cos, past_len, num_heads, head_dim, past_len_dev=None):
        for t, n in ((x, "x"), (sin, "sin"), (cos, "cos")):
            _req_dtype(t, torch.float16, n)
            _req_cuda(t, n)
        _req(head_dim == cos.size(-1), "cos table does not match head_dim")
        _req(head_dim == sin.size(-1), "sin table does not match head_dim")
        bsz = x.size(0)
        rows_per_batch = x.numel() // head_dim // bsz
        with _Guard(x.device):
            check(self._lib.exl_rope(x.data_ptr(), sin.data_ptr(), cos.data_ptr(), bsz, rows_per_batch, head_dim,
                                     num_heads, int(past_len), _ptr(past_len_dev), _stream(x)), "rope_")

    def silu_mul(self, x, y):
        for t, n in ((x, "x"), (y, "y")):
            _req_dtype(t, torch.float16, n)
            _req_cuda(t, n)
        _req(x.shape == y.shape, "x and y have incompatible shapes")
        with _Guard(x.device):
            check(self._lib.exl_silu_mul(x.data_ptr(), y.data_ptr(), x.numel() // x.size(-1), x.size(-1), _stream(x)), "silu_mul")

    def update_cache(self, key_states, value_states, key_cache, value_cache, past_len, past_len_dev=None):
        bsz, q_len, _ = key_states.shape
        kvh, max_seq, hd = key_cache.size(1), key_cache.size(2), key_cache.size(3)
        for t, n in ((key_states, "key_states"), (value_states, "value_states"), (key_cache, "key_cache"), (value_cache, "value_cache")):
            _req_dtype(t, torch.float16, n)
            _req_cuda(t, n)
        with _Guard(key_states.device):
            check(self._lib.exl_update_cache(key_states.data_ptr(), value_states.data_ptr(), key_cache.data_ptr(),
                                             value_cache.data_ptr(), bsz, q_len, kvh, hd, max_seq, int(past_len),
                                             _ptr(past_len_dev), _stream(key_states)), "update_cache")

    def attention(self, q, key_cache, value_cache, out, past_len, num_heads, mask=None, past_len_dev=None):
        """q/out: [bsz, q_len, heads*hd]; caches [>=bsz, kv_heads, max_seq, hd]."""
        bsz, q_len, _ = q.shape
        kvh, max_seq, hd = key_cache.size(1), key_cache.size(2), key_cache.size(3)
        for t, n in ((q, "q"), (key_cache, "key_cache"), (value_cache, "value_cache"), (out, "out")):
            _req_dtype(t, torch.float16, n)
            _req_cuda(t, n)
        _req(q.size(2) == num_heads * hd, "q does not match num_heads * head_dim")
        if bsz > 1:
            _req(key_cache.size(0) == bsz, "cache batch size must equal bsz when bsz > 1")
        if mask is not None:
            _req_dtype(mask, torch.float16, "mask")
            _req_cuda(mask, "mask")
            _req(mask.numel() == bsz * q_len * (past_len + q_len), "mask has the wrong shape")
        with _Guard(q.device):
            check(self._lib.exl_attention(q.data_ptr(), key_cache.data_ptr(), value_cache.data_ptr(), out.data_ptr(),
                                          _ptr(mask), bsz, q_len, num_heads, kvh, hd, max_seq, int(past_len),
                                          _ptr(past_len_dev), _stream(q)), "attention")

    # -- exllama_ext.cpp:424-528
    def q4_attn(self, x, rms_norm_weight, epsilon, query_states, key_states, value_states, q_proj, k_proj, v_proj,
                sin, cos, q_len, past_len, num_heads, num_kv_heads, head_dim, key_cache, value_cache, max_seq_len,
                q_a, q_b, k_a, k_b, v_a, v_b, lora_temp, past_len_dev=None):
        _req_dtype(query_states, torch.float16, "query_states")
        _req_dtype(key_states, torch.float16, "key_states")
        if not _all_cuda_contiguous(x, rms_norm_weight, query_states, key_states, value_states, key_cache, value_cache, sin, cos):
            for t, n in ((x, "x"), (rms_norm_weight, "rms_norm_weight"), (query_states, "query_states"),
                         (key_states, "key_states"), (value_states, "value_states"), (key_cache, "key_cache"),
                         (value_cache, "value_cache"), (sin, "sin"), (cos, "cos")):
                _req_cuda(t, n)                                  # names the offender
        bsz = query_states.size(0)
        dim = query_states.size(2)
        device_index = x.device.index
        _req(device_index is not None and device_index >= 0, "no device index")
        q_rank = 0 if _is_none(q_a) else q_a.size(1)
        k_rank = 0 if _is_none(k_a) else k_a.size(1)
        v_rank = 0 if _is_none(v_a) else v_a.size(1)
        with _Guard(x.device):
            check(self._lib.exl_q4_attn(device_index, x.data_ptr(), rms_norm_weight.data_ptr(), float(epsilon),
                                        query_states.data_ptr(), key_states.data_ptr(), value_states.data_ptr(),
                                        q_proj, k_proj, v_proj, sin.data_ptr(), cos.data_ptr(), bsz, q_len, dim, head_dim,
                                        num_heads, num_kv_heads, int(past_len), _ptr(past_len_dev), key_cache.data_ptr(),
                                        value_cache.data_ptr(), max_seq_len, _ptr(q_a), _ptr(q_b), q_rank, _ptr(k_a),
                                        _ptr(k_b), k_rank, _ptr(v_a), _ptr(v_b), v_rank, _ptr(lora_temp), _stream(x)),
                  "q4_attn")

    # -- exllama_ext.cpp:530-563
    def q4_attn_2(self, x, attn_output, o_proj, o_a, o_b, lora_temp):
        _req_dtype(x, torch.float16, "x")
        _req_dtype(attn_output, torch.float16, "attn_output")
        _req_cuda(x, "x")
        _req_cuda(attn_output, "attn_output")
        # the reference passes x [bsz, q_len, dim] and uses x.size(0) as the row count (rows == 1 there)
        height = attn_output.numel() // attn_output.size(-1)
        o_rank = 0 if _is_none(o_a) else o_a.size(1)
        with _Guard(x.device):
            check(self._lib.exl_q4_attn_2(x.data_ptr(), attn_output.data_ptr(), o_proj, height, _ptr(o_a), _ptr(o_b),
                                          o_rank, _ptr(lora_temp), _stream(x)), "q4_attn_2")

    # -- exllama_ext.cpp:567-602
    def q4_mlp(self, x, rms_norm_weight, epsilon, gate, up, down, gate_a, gate_b, up_a, up_b, down_a, down_b, lora_temp):
        _req_dtype(x, torch.float16, "x")
        _req_dtype(rms_norm_weight, torch.float16, "rms_norm_weight")
        _req_cuda(x, "x")
        _req_cuda(rms_norm_weight, "rms_norm_weight")
        height, dim = x.size(0), x.size(1)
        device_index = x.device.index
        _req(device_index is not None and device_index >= 0, "no device index")
        gr = 0 if _is_none(gate_a) else gate_a.size(1)
        ur = 0 if _is_none(up_a) else up_a.size(1)
        dr = 0 if _is_none(down_a) else down_a.size(1)
        with _Guard(x.device):
            check(self._lib.exl_q4_mlp(device_index, x.data_ptr(), rms_norm_weight.data_ptr(), float(epsilon), gate, up, down,
                                       height, dim, _ptr(gate_a), _ptr(gate_b), gr, _ptr(up_a), _ptr(up_b), ur,
                                       _ptr(down_a), _ptr(down_b), dr, _ptr(lora_temp), _stream(x)), "q4_mlp")

    # -- exllama_ext.cpp:684-741 (host tensors)
    def rep_penalty(self, sequence, rep_mask, penalty_max, sustain, decay):
        _req_dtype(sequence, torch.int64, "sequence")
        _req_dtype(rep_mask, torch.float32, "rep_mask")
        _req(not sequence.is_cuda and not rep_mask.is_cuda, "rep_penalty works on CPU tensors")
        sequence = sequence.contiguous()
        check(self._lib.exl_rep_penalty(rep_mask.size(0), sequence.data_ptr(), rep_mask.data_ptr(), float(penalty_max),
                                        int(sustain), int(decay), sequence.size(-1)), "rep_penalty")

    def apply_rep_penalty(self, sequence, penalty_max, sustain, decay, logits):
        _req_dtype(sequence, torch.int64, "sequence")
        _req_dtype(logits, torch.float32, "logits")
        _req(not sequence.is_cuda and not logits.is_cuda, "apply_rep_penalty works on CPU tensors")
        _req(sequence.size(0) == logits.size(0), "sequence and logits have incompatible shapes")
        _req(logits.is_contiguous(), "logits must be contiguous")
        sequence = sequence.contiguous()
        check(self._lib.exl_apply_rep_penalty(logits.size(-1), sequence.data_ptr(), float(penalty_max), int(sustain),
                                              int(decay), sequence.size(-1), sequence.size(0), logits.data_ptr()),
              "apply_rep_penalty")


exllama_ext = _ExllamaExt()

# The per-token entry points go through the compiled binding (csrc/binding/exl_fast.cpp -> exllama_amd/_exl_fast.so: the same
# argument lists and checks in C++, < 1 us of host time per call against 12-18 us through ctypes).  The methods above stay as the
# A/B reference (EXL_NO_FAST_BINDING=1), as the readable statement of each check -- and as what runs when the binding does not
# load: it links torch_python / c10, so a torch upgrade can break it while libexl_amd.so (plain C ABI, the kernels) still loads.
# That is a host-overhead regression, not a wrong result, so it warns once and continues; EXL_REQUIRE_FAST_BINDING=1 makes it fatal
# (benchmarks, CI).  A missing libexl_amd.so is always fatal (_lib.py): there is no path without the HIP kernels.
import os as _os
import warnings as _warnings

FAST_BINDING = None
FAST_BINDING_ERROR = None
def _env_on(name):
    """An environment switch is ON when it is set to anything but "" / "0" (EXL_X=0 must not enable it)."""
    return _os.environ.get(name, "") not in ("", "0")


if not _env_on("EXL_NO_FAST_BINDING"):
    try:
        from . import _exl_fast as FAST_BINDING
    except (ImportError, OSError) as e:                               # (OSError: a dependency of the module that does not load)
        FAST_BINDING_ERROR = str(e)
        _msg = ("exllama_amd: the compiled binding exllama_amd/_exl_fast.so is missing or does not load (%s); build it with "
                "`make -C exllama_amd/csrc`" % e)
        if _env_on("EXL_REQUIRE_FAST_BINDING"):
            raise RuntimeError(_msg) from e
        _warnings.warn(_msg + " -- continuing on the ctypes path (same kernels, 12-18 us more host time per op call)", RuntimeWarning)
    if FAST_BINDING is not None:
        for _name in ("q4_matmul", "rms_norm", "rope_", "q4_attn", "q4_attn_2", "q4_mlp", "attention"):
            setattr(exllama_ext, _name, getattr(FAST_BINDING, _name))

# re-exports at module level, as the reference does (cuda_ext.py:66-77)
make_q4 = exllama_ext.make_q4
q4_matmul = exllama_ext.q4_matmul
q4_matmul_lora = exllama_ext.q4_matmul_lora
half_matmul = exllama_ext.half_matmul
half_matmul_cublas = exllama_ext.half_matmul_cublas
rms_norm = exllama_ext.rms_norm
rope_ = exllama_ext.rope_
rep_penalty = exllama_ext.rep_penalty
apply_rep_penalty = exllama_ext.apply_rep_penalty


# ---- Python wrappers (reference: cuda_ext.py:87-166) -------------------------------------------------------

def ext_make_q4(qweight, qzeros, scales, g_idx, device):
    """Construct Q4Matrix, return handle."""
    return make_q4(qweight, qzeros, scales, g_idx if g_idx is not None else none_tensor, device)


def ext_q4_matmul(x, q4, q4_width, lora_A=None, lora_B=None):
    """Matrix multiplication, returns x @ q4."""
    outshape = x.shape[:-1] + (q4_width,)
    x = x.view(-1, x.shape[-1])
    output = torch.empty((x.shape[0], q4_width), dtype=torch.float16, device=x.device)
    if lora_A is None:
        q4_matmul(x, q4, output)
    else:
        lora_temp = torch.empty((x.shape[0], lora_A.shape[1]), dtype=torch.float16, device=x.device)
        q4_matmul_lora(x, q4, output, lora_A, lora_B, lora_temp)
    return output.view(outshape)


def ext_half_matmul(x, w, cublas=False):
    """Matrix multiplication, returns x @ w, both half-precision tensors."""
    outshape = x.shape[:-1] + (w.shape[1],)
    x = x.view(-1, x.shape[-1])
    if cublas:
        output = torch.empty((x.shape[0], w.shape[1]), dtype=torch.float16, device=x.device)
        half_matmul_cublas(x, w, output)
    else:
        output = torch.zeros((x.shape[0], w.shape[1]), dtype=torch.float16, device=x.device)
        half_matmul(x, w, output)
    return output.view(outshape)


def ext_rope_(x, sin, cos, past_len, num_heads, head_dim):
    """RoPE embeddings, in place."""
    rope_(x, sin, cos, past_len, num_heads, head_dim)


def ext_rms_norm(x, w, epsilon):
    """RMS norm: x * w / sqrt(row_mean(x * x) + epsilon)."""
    outshape = x.shape
    x = x.view(-1, x.shape[-1])
    output = torch.empty_like(x)
    rms_norm(x, w, output, epsilon)
    return output.view(outshape)


def ext_rms_norm_(x, w, epsilon):
    x = x.view(-1, x.shape[-1])
    rms_norm(x, w, x, epsilon)


def ext_rep_penalty_mask_cpu(vocab_size, sequence, penalty_max, sustain, decay):
    rep_mask = torch.empty(vocab_size, dtype=torch.float32)
    rep_penalty(sequence, rep_mask, penalty_max, sustain, decay)
    return rep_mask


def ext_apply_rep_penalty_mask_cpu(sequence, penalty_max, sustain, decay, logits):
    apply_rep_penalty(sequence, penalty_max, sustain, decay, logits)
