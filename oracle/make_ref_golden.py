"""TEST INFRASTRUCTURE ONLY -- runs ON THE GPU BOX:  python -m oracle.make_ref_golden [out.npz]

Drives the REFERENCE's own GPU kernels (oracle/_ref/libexl_ref_kernels.so: /root/reference/exllama_ext/cuda_func/*.cu
hipified and compiled by oracle/build_ref_kernels.sh, loaded through the plain C ABI of oracle/ref_shim.cpp) on seeded
inputs and records inputs + outputs.  The resulting file is committed as tests/golden/ref_ops.npz and
tests/test_oracle_ref.py (CPU) checks oracle/exl_oracle.py against it: that is what pins the oracle to the reference
for the floating-point ops.  torch is used only to hold device memory.

Nothing under exllama_amd/ is imported here except `synth` (the seeded input generator); the product library is not loaded.
"""

import ctypes as C
import os
import sys

import numpy as np
import torch

from exllama_amd import synth
from oracle import exl_oracle as O

HERE = os.path.dirname(os.path.abspath(__file__))
DEV = "cuda:0"


def _load():
    lib = C.CDLL(os.path.join(HERE, "_ref", "libexl_ref_kernels.so"))
    lib.ref_make_q4.restype = C.c_void_p
    lib.ref_q4_x_map.restype = C.c_void_p
    vp, ci, cf = C.c_void_p, C.c_int, C.c_float
    lib.ref_make_q4.argtypes = [ci, ci, ci, vp, vp, vp, vp, ci]
    lib.ref_q4_x_map.argtypes = [vp]
    lib.ref_copy_x_map.argtypes = [vp, vp]
    lib.ref_prepare_buffers.argtypes = [ci, vp, ci, vp, vp, vp, ci]
    lib.ref_reconstruct.argtypes = [vp, vp]
    lib.ref_q4_matmul.argtypes = [vp, vp, ci, vp, ci]
    lib.ref_q4_matmul_recons.argtypes = [vp, vp, ci, vp, ci]
    lib.ref_column_remap.argtypes = [vp, vp, ci, ci, vp]
    lib.ref_rms_norm.argtypes = [vp, vp, vp, cf, ci, ci, ci]
    lib.ref_rope.argtypes = [vp, vp, vp, ci, ci, ci, ci, ci]
    lib.ref_half_matmul.argtypes = [vp, vp, vp, ci, ci, ci]
    lib.ref_half_matmul_blas.argtypes = [vp, vp, vp, ci, ci, ci]
    lib.ref_q4_mlp.argtypes = [vp, vp, cf, vp, vp, vp, ci, ci, ci]
    lib.ref_q4_attn.argtypes = [vp, vp, cf, vp, vp, vp, vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, ci, ci, vp, vp, ci, ci]
    lib.ref_q4_attn_2.argtypes = [vp, vp, vp, ci]
    lib.ref_set_tuning.argtypes = [ci] * 9
    return lib


def _ok(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what}: hip error {rc}")


class RefLinear:
    """One GPTQ linear handed to the reference's Q4Matrix (exllama_ext.cpp:156-194)."""

    def __init__(self, lib, lin):
        self.lib = lib
        self.host = {k: v.clone() for k, v in lin.items()}
        self.qweight = lin["qweight"].to(DEV).contiguous()
        self.qzeros = lin["qzeros"].to(DEV).contiguous()
        self.scales = lin["scales"].to(DEV).contiguous()
        self.K, self.N = self.qweight.shape[0] * 8, self.qweight.shape[1]
        self.G = self.qzeros.shape[0]
        g = lin.get("g_idx")
        self.g_idx = None if g is None else np.ascontiguousarray(g.numpy().astype(np.uint32))
        gp = None if self.g_idx is None else self.g_idx.ctypes.data_as(C.c_void_p)
        self.h = lib.ref_make_q4(self.K, self.N, self.G, self.qweight.data_ptr(), self.qzeros.data_ptr(), self.scales.data_ptr(), gp, 0)
        if not self.h:
            raise RuntimeError("ref_make_q4 failed")

    def x_map(self):
        p = self.lib.ref_q4_x_map(self.h)
        if not p:
            return None
        out = torch.empty(self.K, dtype=torch.int32, device=DEV)
        _ok(self.lib.ref_copy_x_map(self.h, out.data_ptr()), "copy_x_map")
        return out.cpu().numpy().view(np.uint32)


def main(out_path):
    lib = _load()
    gen = torch.Generator().manual_seed(20260922)
    rs = np.random.RandomState(99)
    out = {}
    max_kn = 1024 * 1024
    bufs = {
        "temp_state": torch.zeros(64 * 2048, dtype=torch.float16, device=DEV),
        "temp_mlp": torch.zeros(4 * 2048, dtype=torch.float16, device=DEV),
        "zeros": torch.zeros(65536, dtype=torch.float32, device=DEV),
        "temp_dq": torch.zeros(max_kn, dtype=torch.float16, device=DEV),
    }
    _ok(lib.ref_prepare_buffers(0, bufs["temp_state"].data_ptr(), bufs["temp_state"].numel(), bufs["temp_mlp"].data_ptr(),
                                bufs["zeros"].data_ptr(), bufs["temp_dq"].data_ptr(), 65536), "prepare_buffers")

    def tune(no_half2):
        """All four *_no_half2 flags together: the reference picks the rms_norm / rope / silu kernel VARIANT by matmul_no_half2 but
        sizes their grids by the per-op flags (rms_norm.cu:157, rope.cu:93,118, q4_mlp.cu:93,179), so mixed settings are not a
        configuration the reference supports."""
        lib.ref_set_tuning(8, 2, 8, 0, no_half2, no_half2, no_half2, no_half2, 0)

    def f16(*shape, scale=1.0):
        return (torch.randn(*shape, generator=gen) * scale).half()

    keep = []

    def dv(t):
        """Device copy that stays alive until main() returns: a temporary whose .data_ptr() is handed to a C call is freed
        right after the expression, and the caching allocator gives the same block to the next temporary."""
        if isinstance(t, np.ndarray):
            t = torch.from_numpy(t)
        d = t.to(DEV).clone()
        keep.append(d)
        return d

    # ---- q4 linears: make_sequential, reconstruct, column_remap, both matmul forms --------------------------------------
    shapes = {"a": (256, 128, 64, True), "b": (512, 96, 128, False), "c": (256, 64, 32, True), "d": (1024, 256, 128, True)}
    for tag, (K, N, gs, act) in shapes.items():
        lin = synth.make_q4_linear(K, N, gs, act, gen, "cpu", zeros="rand", std=0.05)
        r = RefLinear(lib, lin)
        for k, v in lin.items():
            out[f"lin_{tag}_{k}"] = v.numpy()
        out[f"lin_{tag}_seq_qweight"] = r.qweight.cpu().numpy()           # after the in-place act-order repack
        xm = r.x_map()
        if xm is not None:
            out[f"lin_{tag}_x_map"] = xm
        w16 = torch.empty((K, N), dtype=torch.float16, device=DEV)
        _ok(lib.ref_reconstruct(r.h, w16.data_ptr()), "reconstruct")
        if K * N <= 512 * 256:
            out[f"lin_{tag}_w16"] = w16.cpu().numpy()
        else:                                                             # large: keep a strided sample (rows, all columns)
            out[f"lin_{tag}_w16_rows"] = np.arange(0, K, 37)
            out[f"lin_{tag}_w16"] = w16.cpu().numpy()[::37]
        for rows in (1, 3):
            x = f16(rows, K)
            xd = x.to(DEV)
            res = f16(rows, N, scale=0.5)
            for half2 in (1, 0):
                tune(1 - half2)
                y = torch.zeros((rows, N), dtype=torch.float16, device=DEV)
                _ok(lib.ref_q4_matmul(r.h, xd.data_ptr(), rows, y.data_ptr(), 1), "q4_matmul")     # pre-zeroed + no_zero: no zeroing race
                out[f"lin_{tag}_gemv{rows}_h{half2}"] = y.cpu().numpy()
                y2 = res.to(DEV).clone()
                _ok(lib.ref_q4_matmul(r.h, xd.data_ptr(), rows, y2.data_ptr(), 1), "q4_matmul residual")
                out[f"lin_{tag}_gemv{rows}_res_h{half2}"] = y2.cpu().numpy()
            out[f"lin_{tag}_x{rows}"] = x.numpy()
            out[f"lin_{tag}_res{rows}"] = res.numpy()
        tune(0)
        for rows in (8, 19):
            x = f16(rows, K)
            xd = x.to(DEV)
            y = torch.zeros((rows, N), dtype=torch.float16, device=DEV)
            _ok(lib.ref_q4_matmul_recons(r.h, xd.data_ptr(), rows, y.data_ptr(), 0), "q4_matmul_recons")
            out[f"lin_{tag}_xr{rows}"] = x.numpy()
            out[f"lin_{tag}_recons{rows}"] = y.cpu().numpy()
        if xm is not None:
            x = f16(5, K)
            xn = torch.empty_like(x, device=DEV)
            _ok(lib.ref_column_remap(dv(x).data_ptr(), xn.data_ptr(), 5, K, lib.ref_q4_x_map(r.h)), "column_remap")
            out[f"lin_{tag}_remap_x"] = x.numpy()
            out[f"lin_{tag}_remap_y"] = xn.cpu().numpy()
    out["lin_tags"] = np.array(list(shapes.keys()))

    # ---- rms_norm (both kernel variants) ---------------------------------------------------------------------------------
    n = 0
    for rows, dim, eps in ((1, 512, 1e-6), (3, 4096, 1e-5), (7, 1024, 1e-6), (2, 5120, 1e-6), (1, 8192, 1e-6)):
        x = f16(rows, dim, scale=float(rs.uniform(0.3, 3.0)))
        w = (1.0 + 0.1 * torch.randn(dim, generator=gen)).half()
        for no_half2 in (0, 1):
            tune(no_half2)
            y = torch.empty((rows, dim), dtype=torch.float16, device=DEV)
            _ok(lib.ref_rms_norm(dv(x).data_ptr(), dv(w).data_ptr(), y.data_ptr(), eps, rows, dim, 0), "rms_norm")
            out[f"rms_{n}_y{no_half2}"] = y.cpu().numpy()
        out[f"rms_{n}_x"], out[f"rms_{n}_w"], out[f"rms_{n}_eps"] = x.numpy(), w.numpy(), np.float32(eps)
        n += 1
    out["rms_n"] = np.int64(n)

    # ---- rope (both kernel variants) -------------------------------------------------------------------------------------
    n = 0
    # head_dim 128 only: the reference's half2 RoPE grid is head_dim / 32 / 2 / 2 blocks wide (rope.cu:100-125), i.e. zero for 64
    for hd, heads, tokens, past, bsz in ((128, 4, 1, 0, 1), (128, 8, 3, 17, 1), (128, 4, 5, 40, 2), (128, 2, 1, 2047, 1), (256, 2, 2, 9, 1)):
        sin, cos = O.rope_tables(2048, hd)
        x = f16(bsz, tokens * heads * hd)
        for no_half2 in (0, 1):
            tune(no_half2)
            xd = dv(x)
            _ok(lib.ref_rope(xd.data_ptr(), dv(sin).data_ptr(), dv(cos).data_ptr(),
                             bsz, tokens * heads, hd, heads, past), "rope")
            out[f"rope_{n}_y{no_half2}"] = xd.cpu().numpy()
        out[f"rope_{n}_x"] = x.numpy()
        out[f"rope_{n}_params"] = np.array([hd, heads, tokens, past, bsz], dtype=np.int64)
        n += 1
    out["rope_n"] = np.int64(n)

    # ---- half_matmul ------------------------------------------------------------------------------------------------------
    x, w = f16(3, 256), f16(256, 96, scale=0.1)
    for name, fn in (("hm_kernel", lib.ref_half_matmul), ("hm_blas", lib.ref_half_matmul_blas)):
        y = torch.zeros((3, 96), dtype=torch.float16, device=DEV)
        _ok(fn(dv(x).data_ptr(), dv(w).data_ptr(), y.data_ptr(), 3, 256, 96), name)
        out[name] = y.cpu().numpy()
    out["hm_x"], out["hm_w"] = x.numpy(), w.numpy()

    # ---- fused decode ops: q4_attn (norm + qkv + rope + cache scatter), q4_attn_2, q4_mlp -------------------------------------
    tune(0)
    n = 0
    for dim, inter, heads, kvh, gs, act, past in ((512, 512, 4, 4, 128, False, 5), (512, 768, 4, 4, 64, True, 30)):
        hd = dim // heads
        kvd = kvh * hd
        mats = {}
        for name, (K, N) in (("q", (dim, dim)), ("k", (dim, kvd)), ("v", (dim, kvd)), ("o", (dim, dim)), ("gate", (dim, inter)),
                             ("up", (dim, inter)), ("down", (inter, dim))):
            lin = synth.make_q4_linear(K, N, gs, act, gen, "cpu", zeros="rand")
            mats[name] = RefLinear(lib, lin)
            for k, v in lin.items():
                out[f"fused_{n}_{name}_{k}"] = v.numpy()
        sin, cos = O.rope_tables(64, hd)
        sd, cd = dv(sin), dv(cos)
        x = f16(1, 1, dim)
        w1 = (1.0 + 0.1 * torch.randn(dim, generator=gen)).half()
        w2 = (1.0 + 0.1 * torch.randn(dim, generator=gen)).half()
        kc = torch.zeros((1, kvh, 64, hd), dtype=torch.float16, device=DEV)
        vc = torch.zeros_like(kc)
        qs = torch.zeros((1, 1, dim), dtype=torch.float16, device=DEV)
        ks = torch.zeros((1, 1, kvd), dtype=torch.float16, device=DEV)
        vs = torch.zeros((1, 1, kvd), dtype=torch.float16, device=DEV)
        xd = x.to(DEV).clone()
        _ok(lib.ref_q4_attn(xd.data_ptr(), dv(w1).data_ptr(), 1e-6, qs.data_ptr(), ks.data_ptr(), vs.data_ptr(), mats["q"].h,
                            mats["k"].h, mats["v"].h, sd.data_ptr(), cd.data_ptr(), 1, 1, dim, hd, heads, kvh, past, kc.data_ptr(),
                            vc.data_ptr(), 64, 0), "q4_attn")
        out[f"fused_{n}_q"], out[f"fused_{n}_k"], out[f"fused_{n}_v"] = qs.cpu().numpy(), ks.cpu().numpy(), vs.cpu().numpy()
        out[f"fused_{n}_kc"], out[f"fused_{n}_vc"] = kc.cpu().numpy(), vc.cpu().numpy()
        ao = f16(1, 1, dim, scale=0.5)
        x2 = x.to(DEV).clone()
        _ok(lib.ref_q4_attn_2(x2.data_ptr(), dv(ao).data_ptr(), mats["o"].h, 1), "q4_attn_2")
        out[f"fused_{n}_attn_out"], out[f"fused_{n}_x_after_o"] = ao.numpy(), x2.cpu().numpy()
        x3 = x.to(DEV).clone()
        _ok(lib.ref_q4_mlp(x3.data_ptr(), dv(w2).data_ptr(), 1e-6, mats["gate"].h, mats["up"].h, mats["down"].h, 1, dim, 0), "q4_mlp")
        out[f"fused_{n}_x_after_mlp"] = x3.cpu().numpy()
        out[f"fused_{n}_x"], out[f"fused_{n}_w1"], out[f"fused_{n}_w2"] = x.numpy(), w1.numpy(), w2.numpy()
        out[f"fused_{n}_params"] = np.array([dim, inter, heads, kvh, gs, int(act), past], dtype=np.int64)
        n += 1
    out["fused_n"] = np.int64(n)
    out["provenance"] = np.array("outputs of /root/reference/exllama_ext/cuda_func/*.cu (hipify-perl + hipcc gfx950, "
                                 "oracle/build_ref_kernels.sh) run on MI355X by oracle/make_ref_golden.py")
    os.makedirs(os.path.dirname(os.path.abspath(out_path)), exist_ok=True)
    np.savez_compressed(out_path, **out)
    print(f"wrote {out_path}: {len(out)} arrays, {os.path.getsize(out_path) / 1e6:.2f} MB")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/ref_golden/ref_ops.npz")
