"""Times exl_q4_matmul_dual (gate/up + SiLU*mul of the prompt pass) at the 7B shape, M = 2048: HIP events around `reps` launches."""
import sys
import torch
import os as _os
sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
from exllama_amd import synth
from exllama_amd import cuda_ext as ce

DEV = "cuda:0"
K, N, M = 4096, 11008, 2048
gen = torch.Generator().manual_seed(0)
hs, keep = [], []
for i in range(4):
    lin = synth.make_q4_linear(K, N, 128, False, gen, "cpu", zeros="sym")
    d = {k: v.to(DEV).contiguous() for k, v in lin.items() if k != "g_idx"}
    keep.append(d)
    hs.append(ce.ext_make_q4(d["qweight"], d["qzeros"], d["scales"], None, 0))
import os
mode = os.environ.get("DUAL_DATA", "random")            # random | zero_x | const_x: same instruction stream, different switching activity
x = (torch.randn(M, K, generator=gen) * 1.0).half().to(DEV)
if mode == "zero_x":
    x.zero_()
elif mode == "const_x":
    x.fill_(0.5)
out = torch.empty((M, N), dtype=torch.float16, device=DEV)
z = torch.zeros((1, 64), dtype=torch.float16, device=DEV)
ce.exllama_ext.prepare_buffers(torch.device(DEV), torch.zeros((M, N), dtype=torch.float16, device=DEV), z, torch.zeros((1, 64), dtype=torch.float32, device=DEV), z)
reps = int(os.environ.get("DUAL_REPS", "2000"))
for _ in range(3):
    assert ce.exllama_ext.q4_matmul_dual(x, hs[0], hs[1], out, None, True)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(reps):
    ce.exllama_ext.q4_matmul_dual(x, hs[2 * (i & 1)], hs[2 * (i & 1) + 1], out, None, True)
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / reps
print(mode, f"dual M {M} K {K} N {N}: {us:.1f} us  {2 * 2.0 * M * K * N / us / 1e6:.1f} TFLOP/s")
