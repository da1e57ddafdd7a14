"""Dynamic guard for the hand-counted prefill kernels (reference path replaced: the reconstruct + cuBLAS matmul of
/root/reference/exllama_ext/cuda_func/q4_matmul.cu:301-344).

Round 2's defect (profiles/HISTORY.md 9.5): q4_gemm_t16m_kernel<2,2,4,4> wrote garbage into whole accumulator tiles ONLY on the first launch
after an idle period, with a freshly copied activation tensor -- the compiler had recycled registers an inline-asm load was still
writing.  The whole GPU suite was green with the bug present.  The static guard is scripts/isa_lint.py; this is the dynamic one:
every kernel whose vector-memory waits are counted by hand is launched COLD -- in a fresh process, as the first GEMM after a 512 MB
NaN fill went back to the allocator, on a temporary activation copy, into a sentinel-filled output -- several times each, and
compared with the CPU oracle (oracle.exl_oracle.q4_matmul_recons: the reference's reconstruct bits, fp32 products)."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from exllama_amd import synth
from oracle import exl_oracle as O

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REPEATS = 4            # processes per kernel: 10 kernels x 4 = 40 cold launches

#        case       K     N      gs   rows  environment that routes the launch to the kernel
CASES = {
    "t16m128": (4096, 11008, 32, 400, {"EXL_GEMM_NO_SPLITK": "1"}),         # q4_gemm_t16m_kernel<2,2,4,4>: the round-2 failure, exactly
    "t16m128k": (4096, 11008, 32, 400, {}),                                  # the same tile with K cut in two (fp32 slices + reduce kernel; the default at 257 .. 512 rows)
    "t16m256": (4224, 4096, 96, 700, {}),                                    # q4_gemm_t16m_kernel<4,2,4,4>: the 256-row tile without loader waves (group size 96: no power of two)
    "t16w0": (4096, 11008, 128, 600, {}),                                    # q4_gemm_t16w_kernel<0>: loader waves
    "t16s": (4096, 11008, 128, 128, {}),                                     # q4_gemm_t16s_kernel: short prompts
    "t16d2": (4096, 11008, 128, 700, {}),                                    # q4_gemm_t16d2_kernel: gate / up + SiLU
    "t16w1": (4096, 4096, 128, 600, {}),                                     # q4_gemm_t16w_kernel<1>: q / k / v + RoPE + cache
    "t16g_dual": (4096, 11008, 128, 128, {}),                                # q4_gemm_t16g_kernel<8,4,1>: short-prompt gate / up + SiLU (LDS-DMA ring, counted vmcnt)
    "t16g_dual_small": (4096, 11008, 128, 9, {}),                            # q4_gemm_t16g_kernel<1,4,1>: one row tile per block
    "t16r": (11008, 4096, 128, 100, {}),                                     # q4_gemm_t16r_kernel<4,2,0>: short-prompt down_proj (activations in registers)
}


def _lin(K, N, gs, seed):
    gen = torch.Generator().manual_seed(seed)
    lin = synth.make_q4_linear(K, N, gs, False, gen, "cpu", zeros="rand", std=0.02)
    return lin, gen


def _ow(lin):
    return dict(qweight=lin["qweight"].numpy().view(np.uint32), qzeros=lin["qzeros"].numpy().view(np.uint32), scales=lin["scales"].numpy(), x_map=None)


def _save(path, arrays, **lins):
    out = dict(arrays)
    for tag, lin in lins.items():
        for k in ("qweight", "qzeros", "scales"):
            out[f"{tag}_{k}"] = lin[k].numpy()
    np.savez(path, **out)


@pytest.mark.parametrize("case", list(CASES))
def test_hand_counted_gemm_kernels_cold(case, tmp_path):
    K, N, gs, rows, env = CASES[case]
    path = str(tmp_path / f"{case}.npz")
    lin, gen = _lin(K, N, gs, seed=2)
    x = torch.randn(rows, K, generator=gen).half()
    if case in ("t16d2", "t16g_dual", "t16g_dual_small"):
        lin2, _ = _lin(K, N, gs, seed=3)
        ref = O.silu_mul(O.q4_matmul_recons(x.numpy(), **_ow(lin)), O.q4_matmul_recons(x.numpy(), **_ow(lin2)))
        _save(path, {"x": x.numpy(), "ref": ref}, w=lin, v=lin2)
    elif case == "t16w1":
        heads = kvh = 32
        lk, _ = _lin(K, kvh * 128, gs, seed=3)
        lv, _ = _lin(K, kvh * 128, gs, seed=4)
        sin, cos = O.rope_tables(rows + 7, 128)
        q = O.rope(O.q4_matmul_recons(x.numpy(), **_ow(lin))[None], sin, cos, 0, heads, 128)[0]
        k = O.rope(O.q4_matmul_recons(x.numpy(), **_ow(lk))[None], sin, cos, 0, kvh, 128)[0]
        v = O.q4_matmul_recons(x.numpy(), **_ow(lv))
        _save(path, {"x": x.numpy(), "ref_q": q, "ref_k": k, "ref_v": v, "sin": sin, "cos": cos, "heads": heads, "kvh": kvh}, q=lin, k=lk, v=lv)
    else:
        _save(path, {"x": x.numpy(), "ref": O.q4_matmul_recons(x.numpy(), **_ow(lin))}, w=lin)
    e = dict(os.environ)
    e.update(env)
    lines = []
    for rep in range(REPEATS):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "cold_launch_case.py"), case, path], env=e, cwd=ROOT,
                           capture_output=True, text=True, timeout=300)
        tail = [ln for ln in (r.stdout + r.stderr).splitlines() if ln.startswith(("OK", "BAD")) or "fault" in ln.lower() or "Error" in ln]
        lines.append((rep, r.returncode, tail[-3:]))
    assert all(rc == 0 and t and t[-1].startswith("OK") for _, rc, t in lines), lines
