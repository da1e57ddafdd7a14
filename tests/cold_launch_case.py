"""Child process of tests/test_cold_launch_gpu.py: ONE hand-counted prompt-pass kernel as the first GEMM launch of a fresh process,
on memory the allocator hands out poisoned (NaN), with a freshly copied activation tensor and a sentinel-filled output.

    python tests/cold_launch_case.py <case> <data.npz>        (environment: the EXL_GEMM_* switch that selects the kernel)

Prints one line: "OK <case> ..." or "BAD <case> ..." (exit code 0 / 1).  The reference inside data.npz comes from the CPU oracle
(the parent computes it once per case)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from exllama_amd import cuda_ext as ce   # noqa: E402

DEV = "cuda:0"
SENT = -65504.0


def handle(d, tag):
    t = {k: torch.from_numpy(d[f"{tag}_{k}"]).to(DEV).contiguous() for k in ("qweight", "qzeros", "scales")}
    return ce.ext_make_q4(t["qweight"], t["qzeros"], t["scales"], None, 0), t


def verdict(case, name, got, ref, ulps):
    got = got.float().cpu().numpy().astype(np.float64)
    ref = ref.astype(np.float64)
    sent = int((got == SENT).sum())
    nonfinite = int((~np.isfinite(got)).sum())
    scale = max(float(np.abs(ref).max()), 1e-3)
    bad = ~np.isfinite(got) | (np.abs(got - ref) > ulps * scale * 2.0 ** -10)
    if bad.any() or sent:
        r, c = np.argwhere(bad | (got == SENT))[0]
        print(f"BAD {case} {name}: sentinel {sent} nonfinite {nonfinite} bad {int(bad.sum())} first at ({r}, {c}) got {got[r, c]} ref {ref[r, c]}", flush=True)
        return False
    return True


def main():
    case, path = sys.argv[1], sys.argv[2]
    d = np.load(path)
    junk = torch.full((1 << 28,), float("nan"), dtype=torch.float16, device=DEV)     # 512 MB of NaN back to the caching allocator
    del junk
    ext = ce.exllama_ext
    x_host = torch.from_numpy(d["x"])
    rows, K = x_host.shape
    tmp = torch.empty((rows * 2, K), dtype=torch.float16, device=DEV)
    z = torch.zeros(64, dtype=torch.float16, device=DEV)
    ext.prepare_buffers(torch.device(DEV), tmp, z, torch.zeros((1, 64), dtype=torch.float32, device=DEV), z)
    ok = True
    if case in ("t16m128", "t16m128k", "t16m256", "t16w0", "t16s"):
        h, keep = handle(d, "w")
        N = int(d["w_scales"].shape[1])
        torch.cuda.synchronize()
        out = torch.full((rows, N), SENT, dtype=torch.float16, device=DEV)
        ext.q4_matmul_gemm(x_host.to(DEV), h, out)                                  # temporary activation tensor: the cold launch
        out_e = torch.empty((rows, N), dtype=torch.float16, device=DEV)            # poisoned output memory, second launch
        ext.q4_matmul_gemm(x_host.to(DEV), h, out_e)
        torch.cuda.synchronize()
        ok &= verdict(case, "cold/sentinel", out, d["ref"], 1.5)
        ok &= verdict(case, "second/empty", out_e, d["ref"], 1.5)
    elif case == "t16d2":
        h1, k1 = handle(d, "w")
        h2, k2 = handle(d, "v")
        N = int(d["w_scales"].shape[1])
        torch.cuda.synchronize()
        act = torch.full((rows, N), SENT, dtype=torch.float16, device=DEV)
        assert ext.q4_matmul_dual(x_host.to(DEV), h1, h2, act, None, silu=True)
        torch.cuda.synchronize()
        ok &= verdict(case, "cold/sentinel", act, d["ref"], 4.0)
    elif case == "t16w1":
        hq, kq = handle(d, "q")
        hk, kk = handle(d, "k")
        hv, kv = handle(d, "v")
        heads, kvh, hd = int(d["heads"]), int(d["kvh"]), 128
        max_seq = rows + 7
        sin, cos = torch.from_numpy(d["sin"]).to(DEV), torch.from_numpy(d["cos"]).to(DEV)
        torch.cuda.synchronize()
        q = torch.full((1, rows, heads * hd), SENT, dtype=torch.float16, device=DEV)
        kc = torch.full((1, kvh, max_seq, hd), SENT, dtype=torch.float16, device=DEV)
        vc = torch.full_like(kc, SENT)
        assert ext.q4_qkv_rope_cache(x_host.to(DEV), hq, hk, hv, q.view(-1, heads * hd), sin, cos, kc, vc, rows, 0, heads, kvh, hd, max_seq)
        torch.cuda.synchronize()
        ok &= verdict(case, "q", q[0], d["ref_q"], 2.0)
        ok &= verdict(case, "k", kc[0, :, :rows].permute(1, 0, 2).reshape(rows, kvh * hd), d["ref_k"], 2.0)
        ok &= verdict(case, "v", vc[0, :, :rows].permute(1, 0, 2).reshape(rows, kvh * hd), d["ref_v"], 2.0)
    elif case in ("t16g_dual", "t16g_dual_small", "t16r"):
        # round 6: the short-prompt GEMMs on fragment-order activations (csrc/q4_gemm_frag.hip), through exl_q4_matmul_frag
        h1, k1 = handle(d, "w")
        N = int(d["w_scales"].shape[1])
        torch.cuda.synchronize()
        if case == "t16r":
            out = torch.full((rows, N), SENT, dtype=torch.float16, device=DEV)
            assert ext.q4_matmul_frag(x_host.to(DEV), [h1], [out], kernel=1) is not None       # the narrow kernel, cold
            out_e = torch.empty((rows, N), dtype=torch.float16, device=DEV)
            assert ext.q4_matmul_frag(x_host.to(DEV), [h1], [out_e], kernel=1) is not None
            torch.cuda.synchronize()
            ok &= verdict(case, "cold/sentinel", out, d["ref"], 1.5)
            ok &= verdict(case, "second/empty", out_e, d["ref"], 1.5)
        else:
            h2, k2 = handle(d, "v")
            got = ext.q4_matmul_frag(x_host.to(DEV), [h1, h2], dual=True, kernel=0)             # the launcher's choice: the wide kernel
            assert got is not None
            torch.cuda.synchronize()
            ok &= verdict(case, "cold", ext.unfrag(got, rows, N), d["ref"], 4.0)
    else:
        raise SystemExit(f"unknown case {case}")
    if ok:
        print(f"OK {case}", flush=True)
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
