"""One-off: what exactly goes wrong in the 128 x 128 tile GEMM under torch's poisoned allocator (the default route for 257 - 512 rows since round 3)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from exllama_amd import synth
from exllama_amd import cuda_ext as ce
DEV = "cuda:0"
junk = torch.full((1 << 28,), float("nan"), dtype=torch.float16, device=DEV); del junk
K, N, gs, rows = 4096, 11008, 32, 400
gen = torch.Generator().manual_seed(2)
lin = synth.make_q4_linear(K, N, gs, False, gen, "cpu", zeros="rand", std=0.02)
d = {k: v.to(DEV).contiguous() for k, v in lin.items() if k != "g_idx"}
h = ce.ext_make_q4(d["qweight"], d["qzeros"], d["scales"], None, 0)
x = torch.randn(rows, K, generator=gen).half()
tmp = torch.empty((rows * 2, K), dtype=torch.float16, device=DEV)
z = torch.zeros(64, dtype=torch.float16, device=DEV)
ce.exllama_ext.prepare_buffers(torch.device(DEV), tmp, z, torch.zeros((1, 64), dtype=torch.float32, device=DEV), z)
def mark(s):
    torch.cuda.synchronize(); print("ok:", s, flush=True)
mark("make_q4 + prepare_buffers")
if os.environ.get("GEMM_FIRST"):
    o0 = torch.empty((rows, N), dtype=torch.float16, device=DEV)
    ce.exllama_ext.q4_matmul_gemm(x.to(DEV), h, o0)
    mark("first gemm, before anything else")
w16 = torch.empty((K, N), dtype=torch.float16, device=DEV)
ce.exllama_ext.q4_reconstruct(h, w16)
mark("reconstruct")
xd = x.to(DEV)
ref = xd.float() @ w16.float()
mark("reference matmul")
SENT = -65504.0
def report(tag, out):
    torch.cuda.synchronize()
    sent = out == SENT
    bad = ~torch.isfinite(out) | sent | ((out.float() - ref).abs() > 0.02 * ref.abs().max())
    rb, cb = bad.any(1).nonzero().flatten().tolist(), bad.any(0).nonzero().flatten().tolist()
    print(tag, "sentinel", int(sent.sum()), "nonfinite", int((~torch.isfinite(out)).sum()), "bad", int(bad.sum()), "rows", rb[:40], "cols", cb[:40], flush=True)
    if rb:
        r, c = rb[0], cb[0]
        print("   got", out[r, c:c + 8].tolist(), "ref", [round(v, 3) for v in ref[r, c:c + 8].tolist()], flush=True)
for rep in range(2):
    out = torch.empty((rows, N), dtype=torch.float16, device=DEV)
    ce.exllama_ext.q4_matmul_gemm(x.to(DEV), h, out)                      # temporary activation tensor, uninitialised output
    mark("gemm A")
    report(f"A{rep} temp x, empty out  x%512={x.to(DEV).data_ptr() % 4096} out%4096={out.data_ptr() % 4096}", out)
for rep in range(2):
    out = torch.full((rows, N), SENT, dtype=torch.float16, device=DEV)
    ce.exllama_ext.q4_matmul_gemm(xd, h, out)                             # activation kept alive, sentinel output
    report(f"B{rep} kept x, sentinel out", out)
out = torch.full((rows, N), SENT, dtype=torch.float16, device=DEV)
ce.exllama_ext.q4_matmul_gemm(x.to(DEV), h, out)
report("C temp x, sentinel out", out)
torch.cuda.synchronize()
out = torch.empty((rows, N), dtype=torch.float16, device=DEV)
t = x.to(DEV); torch.cuda.synchronize()
ce.exllama_ext.q4_matmul_gemm(t, h, out); torch.cuda.synchronize(); del t
report("D x synced before and after, empty out", out)
