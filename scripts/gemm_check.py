"""Debug aid: the prefill GEMM against reconstruct() x torch.matmul and against itself (determinism)."""
import sys, torch, numpy as np
sys.path.insert(0, ".")
from exllama_amd import cuda_ext as ce, synth
ext = ce.exllama_ext
dev = "cuda:0"
for (M, K, N, gs) in [(20, 512, 512, 128), (20, 512, 1408, 128), (20, 1408, 512, 128), (128, 512, 512, 128), (200, 4096, 4096, 128), (2048, 4096, 4096, 128)]:
    gen = torch.Generator().manual_seed(K + N)
    lin = synth.make_q4_linear(K, N, gs, False, gen, "cpu", zeros="rand", std=0.05)
    d = {k: v.to(dev).contiguous() for k, v in lin.items() if k != "g_idx"}
    h = ce.ext_make_q4(d["qweight"], d["qzeros"], d["scales"], None, 0)
    w16 = torch.empty((K, N), dtype=torch.float16, device=dev)
    ext.q4_reconstruct(h, w16)
    x = torch.randn(M, K, generator=gen).half().to(dev)
    ref = (x.float() @ w16.float())
    outs = []
    for rep in range(3):
        # dirty the LDS / caches between runs with an unrelated kernel
        junk = torch.randn(1 << 20, device=dev).sin().sum()
        out = torch.full((M, N), float("nan"), dtype=torch.float16, device=dev)
        ext.q4_matmul_gemm(x, h, out) if hasattr(ext, "q4_matmul_gemm") else None
        outs.append(out.float())
    err = [(o - ref).abs().max().item() for o in outs]
    same = all(torch.equal(outs[0], o) for o in outs[1:])
    bad = (outs[0] - ref).abs() > 0.05 * ref.abs().max()
    print(M, K, N, "max err", [round(e, 4) for e in err], "scale", round(ref.abs().max().item(), 3), "deterministic", same,
          "bad elems", int(bad.sum()), "nan", int(torch.isnan(outs[0]).sum()))
    if bad.any():
        idx = bad.nonzero()
        print("   first bad (row, col):", idx[:8].tolist(), "rows", sorted(set(idx[:, 0].tolist()))[:20], "cols%128", sorted(set((idx[:, 1] % 128).tolist()))[:40])
