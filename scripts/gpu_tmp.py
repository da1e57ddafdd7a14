import sys, os, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from exllama_amd import synth
from exllama_amd import cuda_ext as ce
DEV = "cuda:0"
# poison the allocator's free memory with NaNs, as a long test run would
junk = torch.full((1 << 28,), float("nan"), dtype=torch.float16, device=DEV); del junk
def run(K, N, gs, act, rows, seed):
    gen = torch.Generator().manual_seed(seed)
    lin = synth.make_q4_linear(K, N, gs, act, gen, "cpu", zeros="rand", std=0.02 * (4096 / K) ** 0.5)
    d = {k: v.to(DEV).contiguous() for k, v in lin.items() if k != "g_idx"}
    h = ce.ext_make_q4(d["qweight"], d["qzeros"], d["scales"], lin.get("g_idx"), 0)
    x = torch.randn(rows, K, generator=gen).half()
    tmp = torch.empty((rows * 2, K), dtype=torch.float16, device=DEV)
    z = torch.zeros(64, dtype=torch.float16, device=DEV)
    ce.exllama_ext.prepare_buffers(torch.device(DEV), tmp, z, torch.zeros((1, 64), dtype=torch.float32, device=DEV), z)
    out = torch.empty((rows, N), dtype=torch.float16, device=DEV)
    ce.exllama_ext.q4_matmul_gemm(x.to(DEV), h, out)
    w16 = torch.empty((K, N), dtype=torch.float16, device=DEV)
    ce.exllama_ext.q4_reconstruct(h, w16)
    ref = x.to(DEV).float() @ w16.float()
    bad = ~torch.isfinite(out) | ((out.float() - ref).abs() > 0.02 * ref.abs().max())
    rb, cb = bad.any(1).nonzero().flatten(), bad.any(0).nonzero().flatten()
    print(K, N, gs, act, rows, "nonfinite", int((~torch.isfinite(out)).sum()), "bad", int(bad.sum()),
          "rows", (int(rb.min()), int(rb.max()), rb.numel()) if rb.numel() else None, "cols", (int(cb.min()), int(cb.max()), cb.numel()) if cb.numel() else None,
          "w16 finite", bool(torch.isfinite(w16).all()), flush=True)
run(4096, 11008, 32, False, 400, 2)
run(4096, 11008, 128, False, 512, 3)
run(4096, 4096, 32, False, 300, 4)
run(11008, 4096, 128, False, 384, 5)
run(4096, 11008, 32, False, 257, 6)
