"""The in-flight-register lint (scripts/isa_lint.py) over the built gfx950 code objects: no instruction may touch a VGPR that an
outstanding (inline-asm, hand-counted) global load has not delivered yet.  Guards the defect class of profiles/HISTORY.md 9.5."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import isa_lint  # noqa: E402

LIB = os.path.join(ROOT, "exllama_amd", "libexl_amd.so")


def _k(lines):
    """[(mnemonic, operands)] -> instruction list at 4-byte spacing."""
    return [(0x100 + 4 * i, mn, ops) for i, (mn, ops) in enumerate(lines)]


def test_lint_flags_a_register_reused_under_an_outstanding_load():
    # the round-2 defect in miniature: a "dead" load destination used to move an accumulator before the covering wait
    bad = _k([("global_load_dwordx4", "v[4:7], v[4:5], off"),
              ("v_accvgpr_read_b32", "v4, a28"),
              ("v_accvgpr_write_b32", "a0, v4"),
              ("s_waitcnt", "vmcnt(0)"),
              ("s_endpgm", "")])
    hz = isa_lint.lint_kernel("bad", bad)
    assert len(hz) == 2 and all(h[5] == [4] for h in hz)
    good = _k([("global_load_dwordx4", "v[4:7], v[4:5], off"),
               ("v_accvgpr_read_b32", "v20, a28"),
               ("s_waitcnt", "vmcnt(0)"),
               ("v_accvgpr_write_b32", "a0, v4"),
               ("s_endpgm", "")])
    assert isa_lint.lint_kernel("good", good) == []


def test_lint_counts_the_queue_like_the_hardware():
    # vmcnt(N) leaves the youngest N outstanding; LDS-DMA and stores take a slot but deliver no register
    k = _k([("global_load_dwordx4", "v[8:11], v[0:1], off"),
            ("global_load_lds_dwordx4", "v[2:3], off"),
            ("global_store_dwordx2", "v[12:13], v[14:15], off"),
            ("s_waitcnt", "vmcnt(2)"),
            ("v_add_u32_e32", "v8, v8, v9"),                       # the dwordx4 is the third-youngest: delivered
            ("global_load_dword", "v16, v[0:1], off"),
            ("s_waitcnt", "vmcnt(1)"),
            ("v_mov_b32_e32", "v17, v16"),                         # one op may still be out, and v16 is the youngest
            ("s_endpgm", "")])
    hz = isa_lint.lint_kernel("k", k)
    assert [h[5] for h in hz] == [[16]]


def test_lint_follows_loops_and_joins():
    # loop-carried: the load issued at the bottom of the body is still out when the top of the next iteration reads v5;
    # the exit path waits first and is clean
    k = _k([("v_mov_b32_e32", "v5, 0"),
            ("v_add_u32_e32", "v6, v5, v6"),                       # 0x104: loop head
            ("global_load_dword", "v5, v[0:1], off"),
            ("s_cbranch_scc1", str((0x104 - (0x10c + 4)) // 4 & 0xFFFF)),
            ("s_waitcnt", "vmcnt(0)"),
            ("v_add_u32_e32", "v7, v5, v6"),
            ("s_endpgm", "")])
    hz = isa_lint.lint_kernel("loop", k)
    assert [(h[1], h[5]) for h in hz] == [(0x104, [5])]


def test_lint_flags_an_lds_dma_that_outlives_the_block():
    bad = _k([("global_load_lds_dwordx4", "v[2:3], off"),
              ("global_load_dword", "v9, v[0:1], off"),
              ("s_waitcnt", "vmcnt(1)"),                           # covers the DMA ...
              ("s_endpgm", "")])
    assert isa_lint.lint_kernel("ok", bad) == []
    bad = _k([("global_load_dword", "v9, v[0:1], off"),
              ("global_load_lds_dwordx4", "v[2:3], off"),
              ("s_waitcnt", "vmcnt(1)"),                           # ... this one only the register load
              ("s_endpgm", "")])
    hz = isa_lint.lint_kernel("bad", bad)
    assert len(hz) == 1 and hz[0][2] == "s_endpgm"


@pytest.mark.skipif(not os.path.exists(LIB), reason="library not built (python __graft_entry__.py build)")
@pytest.mark.skipif(not os.path.exists(os.path.join(isa_lint.LLVM, "llvm-objdump")), reason="no ROCm LLVM tools")
def test_built_library_has_no_in_flight_register_hazard():
    count, hazards = isa_lint.lint_library(LIB)
    assert count > 100                                             # every translation unit's kernels were found
    assert hazards == [], "\n".join(f"{h[0][:80]} {h[1]:#x}: {h[2]}  <- {h[4]}" for h in hazards[:10])


def test_lint_flags_accumulators_copied_inside_an_mfma_loop():
    # what hipcc made of the prompt attention kernel while a 64-register accumulator lived across a branch: whole accumulators moved per
    # iteration.  16 MFMAs with 32 v_mov_b64 (64 registers) in the loop is flagged, the same loop with the handful of moves every loop has is not
    def loop(nmov):
        body = [("v_mfma_f32_32x32x16_f16", "v[0:15], v[64:67], v[68:71], v[0:15]")] * 16 + [("v_mov_b64_e32", "v[100:101], v[0:1]")] * nmov
        return _k([("s_mov_b32", "s0, 0")] + body + [("s_cbranch_scc1", str((0x104 - (0x104 + 4 * len(body) + 4)) // 4 & 0xFFFF)), ("s_endpgm", "")])
    bad = isa_lint.accumulator_copy_hazards("bad", loop(32))
    assert len(bad) == 1 and "64 registers copied" in bad[0][4]
    assert isa_lint.accumulator_copy_hazards("good", loop(4)) == []
    few = _k([("v_mfma_f32_16x16x32_f16", "v[0:3], v[8:11], v[12:15], v[0:3]")] * 4 + [("v_mov_b64_e32", "v[20:21], v[0:1]")] * 30 +
             [("s_cbranch_scc1", str((0x100 - (0x100 + 4 * 34 + 4)) // 4 & 0xFFFF)), ("s_endpgm", "")])
    assert isa_lint.accumulator_copy_hazards("few MFMAs: not a matrix loop", few) == []


def test_lint_flags_a_foreign_operation_in_a_hand_counted_lds_queue():
    # flash_prefill8_kernel issues its K fragment reads by hand and waits with counted lgkmcnt: another kind of LDS read lifted into the
    # stretch, or a scalar memory read (returns out of order), breaks the count
    name = "_Z21flash_prefill8_kernelPKDF16_"
    reads = [("ds_read_b128", f"v[{8 * i}:{8 * i + 3}], v100") for i in range(4)]
    ok = _k(reads + [("s_waitcnt", "lgkmcnt(3)"), ("v_mfma_f32_32x32x16_f16", "v[64:79], v[0:3], v[40:43], v[64:79]"), ("s_waitcnt", "lgkmcnt(0)"), ("s_endpgm", "")])
    assert isa_lint.lgkm_count_hazards(name, ok) == []
    lifted = _k(reads[:2] + [("ds_read_b64_tr_b16", "v[50:51], v101")] + reads[2:] + [("s_waitcnt", "lgkmcnt(3)"), ("s_endpgm", "")])
    hz = isa_lint.lgkm_count_hazards(name, lifted)
    assert len(hz) == 1 and "ds_read_b64_tr_b16" in hz[0][4]
    scalar = _k([("s_memtime", "s[4:5]")] + reads + [("s_waitcnt", "lgkmcnt(3)"), ("s_endpgm", "")])
    assert len(isa_lint.lgkm_count_hazards(name, scalar)) == 1
    assert isa_lint.lgkm_count_hazards("some_other_kernel", lifted) == []               # compiler-counted kernels are correct by construction
    # behind a full drain the queue starts again
    again = _k([("ds_bpermute_b32", "v1, v2, v3"), ("s_waitcnt", "lgkmcnt(0)")] + reads + [("s_waitcnt", "lgkmcnt(2)"), ("s_endpgm", "")])
    assert isa_lint.lgkm_count_hazards(name, again) == []
