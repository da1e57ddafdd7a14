"""Phase stamps of q4_gemm_t16g's pipelined step (a -DEXL_T16G_PROBE build of csrc/q4_gemm_frag.hip in build/t16g_probe/libexl_amd.so -- the
product library carries no stamps): s_memtime of the first wave of each K-group of blocks 0 and grid / 2 at: step start, first group issued,
slab requests out, all groups issued, requests landed (the counted vmcnt wait), barrier passed.  Prints cycles per phase, averaged over steps.

    EXL_NO_FAST_BINDING=1 python scripts/probe_t16g.py
"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["EXL_NO_FAST_BINDING"] = "1"
import exllama_amd._lib as L   # noqa: E402
L.LIB_PATH = os.path.join(ROOT, "build", "t16g_probe", "libexl_amd.so")
from exllama_amd import cuda_ext, synth   # noqa: E402

lib = L.load()
lib.exl_debug_t16g_probe.restype = C.c_int
lib.exl_debug_t16g_probe.argtypes = [C.c_void_p]
ext = cuda_ext.exllama_ext
DEV = "cuda:0"


def handles(K, widths, seed):
    gen = torch.Generator().manual_seed(seed)
    keep = []
    for N in widths:
        lin = synth.make_q4_linear(K, N, 128, False, gen, "cpu", zeros="sym", nibbles="centered")
        d = {k: v.to(DEV).contiguous() for k, v in lin.items()}
        keep.append((cuda_ext.ext_make_q4(d["qweight"], d["qzeros"], d["scales"], None, 0), d))
    return keep


def run(tag, K, widths, rows, dual, kernel):
    keep = handles(K, widths, 1)
    hs = [h for h, _ in keep]
    x = torch.randn(rows, K, device=DEV).half()
    outs = None if dual else [torch.empty((rows, N), dtype=torch.float16, device=DEV) for N in widths]
    for _ in range(3):
        assert ext.q4_matmul_frag(x, hs, outs, dual=dual, kernel=kernel) is not None
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * 256)()
    assert lib.exl_debug_t16g_probe(buf) == 0
    names = ["first group", "slab requests out", "rest of the groups", "counted wait", "barrier"]
    print("==", tag)
    for slot, who in enumerate(("block 0, K-group 0", "block 0, K-group 1", "block grid/2, K-group 0", "block grid/2, K-group 1")):
        t = [buf[slot * 64 + i] for i in range(64)]
        if t[0] == 0:
            continue
        steps = []
        for s in range(10):
            b = 2 + 6 * s
            if t[b + 5] == 0:
                break
            steps.append([t[b + 1] - t[b], t[b + 2] - t[b + 1], t[b + 3] - t[b + 2], t[b + 4] - t[b + 3], t[b + 5] - t[b + 4], t[b + 5] - t[b]])
        if not steps:
            continue
        avg = [sum(c) / len(steps) for c in zip(*steps[1:])] if len(steps) > 1 else steps[0]
        print("  %s: prologue %d cycles; per step (mean of steps 1..%d): " % (who, t[1] - t[0], len(steps) - 1)
              + ", ".join("%s %d" % (n, v) for n, v in zip(names, avg)) + " | step %d" % avg[5])
        print("     steps:", [c[5] for c in steps])


run("gate / up, 128 rows, <8, 4, 1>", 4096, (11008, 11008), 128, True, 5)
run("q / k / v, 128 rows, <4, 4, 0>", 4096, (4096, 4096, 4096), 128, False, 3)
run("gate / up, 16 rows, <1, 4, 1>", 4096, (11008, 11008), 16, True, 7)
run("q / k / v, 16 rows, <1, 2, 0>", 4096, (4096, 4096, 4096), 16, False, 8)
