"""Occupancy assumptions of the hot kernels, held against the built gfx950 code objects (scripts/isa_lint.py:kernel_resources reads
the metadata notes of exllama_amd/libexl_amd.so).  DESIGN.md 3 states how many blocks of each kernel a CU is meant to hold; that is
a statement about REGISTERS and scratch, which a compiler update or an innocent-looking edit changes silently -- and the first symptom
would be a slower bench line.  CPU test: hipcc cross-compiles, nothing runs."""
import os
import re
import sys
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import isa_lint  # noqa: E402

LIB = os.path.join(ROOT, "exllama_amd", "libexl_amd.so")


@pytest.fixture(scope="module")
def res():
    if not os.path.exists(LIB):
        pytest.skip("libexl_amd.so not built")
    with tempfile.TemporaryDirectory() as d:
        return isa_lint.kernel_resources(LIB, d)


def _ints(name):
    """template arguments of a mangled kernel name, in order (ILi3ELi8ELi1E... -> [3, 8, 1, ...]; booleans count as 0 / 1)."""
    return [int(x) for x in re.findall(r"L[ib](\d+)E", name.split("Ev")[0])]


def test_decode_ring_kernels_fit_the_blocks_per_cu_they_are_launched_for(res):
    """dec_ring_kernel<U, UL, PNORM, EMODE, NV, NW, PRE, GM>: two 8-wave blocks per CU (4 waves per SIMD: <= 128 VGPRs) or one 16-wave
    block (same bound); the longest unrolled units (UL >= 24 at 8 waves) and the merge prologue (PNORM 3) are declared for ONE block
    of 8 waves per CU beside another (<= 256).  No scratch in any of them: a spill inside a hand-counted stream shifts every count."""
    ring = {n: r for n, r in res.items() if "dec_ring_kernel" in n}
    assert len(ring) >= 300, len(ring)
    for n, r in ring.items():
        U, UL, PNORM, EMODE, NV, NW = _ints(n)[:6]
        assert r["threads"] == NW * 64, n
        assert r["scratch"] == 0, (n, r)
        cap = 256 if ((UL >= 24 and NW == 8) or PNORM == 3) else 128
        assert r["vgpr"] + r["agpr"] <= cap, (n, r, cap)
        assert r["sgpr"] <= 112, (n, r)                                 # 4-6 waves per SIMD stay admissible (MI355X_MICROARCH.md: residency)


def test_compiler_scheduled_decode_kernels(res):
    for n, r in res.items():
        if "dec_attn_kernel" in n or "dec_attn_short_kernel" in n:
            assert r["scratch"] == 0 and r["vgpr"] <= 128, (n, r)        # 2 blocks of 4-8 waves per CU
        if "dec_head_kernel" in n:
            assert r["scratch"] == 0 and r["vgpr"] <= 64, (n, r)
        if "dec_stream_kernel" in n:                                   # the fallback stream: a few capped instantiations carry <= 32 bytes
            assert r["scratch"] <= 32 and r["vgpr"] <= 256, (n, r)     # of scratch OUTSIDE the inner loop (profiles/HISTORY.md 3)


def test_prompt_kernels(res):
    seen = set()
    for n, r in res.items():
        if "q4_gemm_t16w_kernel" in n:                                 # 8 MFMA + 4 loader waves = 3 waves per SIMD
            seen.add("t16w")
            assert r["threads"] == 768 and r["scratch"] == 0 and r["vgpr"] + r["agpr"] <= 170, (n, r)
        if "q4_gemm_t16d2_kernel" in n:                                # 8 waves, 128 accumulators + 72 fragment registers: one block per CU
            seen.add("t16d2")
            assert r["threads"] == 512 and r["scratch"] == 0 and r["vgpr"] + r["agpr"] <= 256, (n, r)
        if "q4_gemm_t16s_kernel" in n or "q4_gemm_t16m_kernel" in n:
            assert r["scratch"] == 0 and r["vgpr"] + r["agpr"] <= 264, (n, r)
        if "flash_prefill8_kernel" in n:                               # 8 waves, one block per CU, 2 waves per SIMD
            seen.add("flash8")
            assert r["threads"] == 512 and r["scratch"] == 0 and r["vgpr"] + r["agpr"] <= 256, (n, r)
        if "half_gemm_nt_kernel" in n:
            assert r["scratch"] == 0, (n, r)
        if "q4_gemm_t16g_kernel" in n or "q4_gemm_t16r_kernel" in n:     # short prompts: 8 waves, hand-counted requests -- one block per CU
            seen.add("t16g" if "t16g" in n else "t16r")                  # (2 waves per SIMD: <= 256), never a spill
            assert r["threads"] == 512 and r["scratch"] == 0 and r["vgpr"] + r["agpr"] <= 256 and r["lds"] == 0, (n, r)
        if "to_frag_kernel" in n:
            seen.add("to_frag")
            assert r["scratch"] == 0 and r["vgpr"] <= 256, (n, r)
    assert seen == {"t16w", "t16d2", "flash8", "t16g", "t16r", "to_frag"}, seen
