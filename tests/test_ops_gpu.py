"""GPU parity tests: every HIP kernel, called through the C ABI (exllama_amd.cuda_ext -> ctypes -> libexl_amd.so),
against the CPU oracle on the same seeded inputs.

Tolerances (stated per test): integer / bit-copy work is bit-exact; fp16 results of fp32-accumulated kernels are
compared with `|got - ref| <= atol` where atol is a small multiple of one fp16 ulp at the output's scale.
"""
import os

import numpy as np
import pytest
import torch

from exllama_amd import synth
from oracle import exl_oracle as O

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


@pytest.fixture(scope="module")
def ce():
    assert torch.cuda.is_available(), "GPU tests selected but no HIP device is visible"
    from exllama_amd import cuda_ext
    return cuda_ext


def _lin(K, N, gs, act, seed, zeros="rand", std=0.05):
    gen = torch.Generator().manual_seed(seed)
    lin = synth.make_q4_linear(K, N, gs, act, gen, "cpu", zeros=zeros, std=std)
    return lin, gen


def _to_dev(lin):
    return {k: v.to(DEV).contiguous() for k, v in lin.items() if k != "g_idx"}


def _handle(ce, lin):
    d = _to_dev(lin)
    h = ce.ext_make_q4(d["qweight"], d["qzeros"], d["scales"], lin.get("g_idx"), 0)
    return h, d


def _oracle_w(lin):
    qw = lin["qweight"].numpy().view(np.uint32)
    qz = lin["qzeros"].numpy().view(np.uint32)
    sc = lin["scales"].numpy()
    x_map = None
    if "g_idx" in lin:
        x_map, qw = O.make_sequential(qw, lin["g_idx"].numpy(), qz.shape[0])
    return dict(qweight=qw, qzeros=qz, scales=sc, x_map=x_map)


def _close(got, ref, ulps=2.0):
    """fp16 comparison: atol = ulps * (fp16 ulp at the largest |ref|) ; also reports the worst element."""
    got = np.asarray(got).astype(np.float64)
    ref = np.asarray(ref).astype(np.float64)
    assert got.shape == ref.shape
    assert np.isfinite(got).all()
    scale = max(np.abs(ref).max(), 1e-3)
    atol = ulps * scale * 2.0 ** -10
    err = np.abs(got - ref).max()
    assert err <= atol, f"max |diff| {err:.3e} > atol {atol:.3e} (scale {scale:.3e})"
    # relative check: the absolute bound above is set by the LARGEST element, so a wrong low-magnitude region could hide
    # under it.  Correct fp16 rounding alone gives rms(err) / rms(ref) ~ 2^-11 / sqrt(3); anything systematically wrong in
    # the small elements shows up here, over the whole array and over every 16-column / 16-element block of it.
    rms_ref = float(np.sqrt(np.mean(ref ** 2)))
    if rms_ref > 0 and ref.size >= 64:
        rel = float(np.sqrt(np.mean((got - ref) ** 2))) / rms_ref
        assert rel <= ulps * 2.0 ** -11, f"rms(diff) / rms(ref) = {rel:.3e} > {ulps * 2.0 ** -11:.3e}"
        flat_g, flat_r = got.reshape(-1), ref.reshape(-1)
        nblk = flat_r.size // 16
        if nblk >= 4:
            eb = np.sqrt(np.mean((flat_g[:nblk * 16] - flat_r[:nblk * 16]).reshape(nblk, 16) ** 2, axis=1))
            rb = np.sqrt(np.mean(flat_r[:nblk * 16].reshape(nblk, 16) ** 2, axis=1))
            bad = eb > 8.0 * ulps * 2.0 ** -11 * np.maximum(rb, rms_ref / 8.0)
            assert not bad.any(), f"{int(bad.sum())} of {nblk} 16-element blocks off (first: block {int(np.argmax(bad))})"


def _same_up_to_fp32_order(a, b):
    """Two launches of the same product whose K loops are cut differently (a tile of a partly filled last round is split along K,
    q4_gemm.hip: plan_gemm_tail) add the same fp32 terms in another order: the fp16 results are equal except where the fp32 sums
    straddle a rounding boundary -- at most one fp16 step at the sum's magnitude (rotated / cancelled values: at the output scale),
    and only for a small share of the elements."""
    if torch.equal(a, b):
        return
    af, bf = a.float(), b.float()
    assert torch.isfinite(af).all() and torch.isfinite(bf).all()
    scale = float(torch.maximum(af.abs(), bf.abs()).max())
    d = (af - bf).abs()
    assert float(d.max()) <= scale * 2.0 ** -10, (float(d.max()), scale)
    assert float((d > 0).float().mean()) <= 0.05, float((d > 0).float().mean())


# ---------------------------------------------------------------------------------------------------------
# make_q4 / act-order / reconstruct / column_remap: integer + bit-exact fp16
# ---------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("K,N,gs,act", [(256, 128, 64, True), (512, 96, 128, False), (256, 64, 32, True),
                                        (4096, 4096, 128, True), (11008, 4096, 128, False), (704, 256, 64, True)])
def test_make_q4_and_reconstruct_bit_exact(ce, K, N, gs, act):
    lin, _ = _lin(K, N, gs, act, seed=K + N)
    h, d = _handle(ce, lin)
    ow = _oracle_w(lin)
    info = ce.exllama_ext.q4_info(h)
    assert (info["height"], info["width"], info["groups"], info["groupsize"]) == (K, N, K // gs, gs)
    assert bool(info["x_map"]) == act, "act-order weights (and only those) own a device x_map"
    # the caller's qweight tensor is rewritten in place: act-order row repack (reference: q4_matrix.cu:159), then the
    # product's T16 re-tiling for every shape it covers -- both integer-exact against the oracle
    expect = ow["qweight"]
    assert info["layout"] == int(O.t16_eligible(K, N, gs))
    got = d["qweight"].cpu().numpy().view(np.uint32)
    if info["layout"] == 1:
        assert np.array_equal(got.reshape(-1), O.retile_t16(expect))
    else:
        assert np.array_equal(got, expect)
    w16 = torch.empty((K, N), dtype=torch.float16, device=DEV)
    ce.exllama_ext.q4_reconstruct(h, w16)
    ref = O.dequant_w16(ow["qweight"], ow["qzeros"], ow["scales"])
    assert np.array_equal(w16.cpu().numpy().view(np.uint16), ref.view(np.uint16))      # bit-exact fp16 weights


def test_make_q4_refuses_a_tensor_it_already_rewrote(ce):
    """make_q4 re-tiles qweight in place (the reference rewrites it too for act-order, q4_matrix.cu:159): a second handle on the
    SAME tensor would re-tile re-tiled words and compute garbage without a sound.  The library keeps a fingerprint of every tensor
    it rewrote and refuses; a fresh copy of the checkpoint tensor -- also one the allocator puts at the same address -- is fine."""
    lin, _ = _lin(512, 256, 128, False, seed=31)
    d = _to_dev(lin)
    h1 = ce.ext_make_q4(d["qweight"], d["qzeros"], d["scales"], None, 0)
    with pytest.raises(RuntimeError, match="already rewritten"):
        ce.ext_make_q4(d["qweight"], d["qzeros"], d["scales"], None, 0)
    addr = d["qweight"].data_ptr()
    d["qweight"].copy_(lin["qweight"])                                   # the checkpoint words again, at the same address
    h2 = ce.ext_make_q4(d["qweight"], d["qzeros"], d["scales"], None, 0)
    assert d["qweight"].data_ptr() == addr and h2 != h1
    w16 = torch.empty((512, 256), dtype=torch.float16, device=DEV)
    ce.exllama_ext.q4_reconstruct(h2, w16)
    ow = _oracle_w(lin)
    assert np.array_equal(w16.cpu().numpy().view(np.uint16), O.dequant_w16(ow["qweight"], ow["qzeros"], ow["scales"]).view(np.uint16))


def test_make_q4_golden_fixture(ce, golden_dir):
    g = np.load(os.path.join(golden_dir, "ops_small.npz"))
    for tag in ("a", "c"):
        qw = torch.from_numpy(g[f"q4{tag}_qweight"].view(np.int32).copy()).to(DEV)
        qz = torch.from_numpy(g[f"q4{tag}_qzeros"].view(np.int32).copy()).to(DEV)
        sc = torch.from_numpy(g[f"q4{tag}_scales"].copy()).to(DEV)
        gi = torch.from_numpy(g[f"q4{tag}_g_idx"].copy())
        h = ce.ext_make_q4(qw, qz, sc, gi, 0)
        seq = g[f"q4{tag}_qweight_seq"]
        got = qw.cpu().numpy().view(np.uint32)
        if ce.exllama_ext.q4_info(h)["layout"] == 1:
            assert np.array_equal(got.reshape(-1), O.retile_t16(seq))
        else:
            assert np.array_equal(got, seq)
        x = torch.from_numpy(g[f"q4{tag}_x"].copy()).to(DEV)
        out = ce.ext_q4_matmul(x[:3], h, qw.shape[1])
        _close(out.cpu().numpy(), g[f"q4{tag}_out_gemv"])


def test_empty_g_idx_is_rejected_by_caller_contract(ce):
    """All-zero g_idx never reaches make_q4 (model.py:147-149 drops it); a g_idx with an out-of-range group raises."""
    lin, _ = _lin(256, 64, 64, False, seed=3)
    d = _to_dev(lin)
    bad = torch.full((256,), 99, dtype=torch.int32)
    with pytest.raises(RuntimeError, match="out of range"):
        ce.ext_make_q4(d["qweight"], d["qzeros"], d["scales"], bad, 0)


def test_column_remap_bit_exact(ce):
    gen = torch.Generator().manual_seed(0)
    x = torch.randn(37, 512, generator=gen).half()
    perm = torch.randperm(512, generator=gen).to(torch.int32)
    out = torch.empty_like(x, device=DEV)
    ce.exllama_ext.column_remap(x.to(DEV), out, perm.to(DEV))
    assert np.array_equal(out.cpu().numpy().view(np.uint16), O.column_remap(x.numpy(), perm.numpy()).view(np.uint16))


# ---------------------------------------------------------------------------------------------------------
# q4 matmul: decode GEMV and prefill MFMA GEMM
# ---------------------------------------------------------------------------------------------------------
GEMV_SHAPES = [(256, 128, 64, True), (512, 96, 128, False), (256, 64, 32, True), (704, 256, 64, False),
               (4096, 4096, 128, False), (4096, 11008, 128, False), (11008, 4096, 128, False),
               (5120, 5120, 128, True), (6656, 6656, 32, True), (4096, 32000 // 32 * 32, 4096, False),
               (28672, 256, 128, False),       # Llama-2-70B down_proj: 28 row-blocks per wave, > 64 KiB of LDS for 2+ rows
               (40960, 128, 128, True)]        # beyond the decode kernel's reach: routed to the MFMA GEMM (q4_gemv_covers)


@pytest.mark.parametrize("K,N,gs,act", GEMV_SHAPES)
@pytest.mark.parametrize("rows", [1, 3, 8])
def test_q4_gemv_vs_oracle(ce, K, N, gs, act, rows):
    keep = None
    if K > 36864:
        keep = _prep_buffers(ce, 8, K, K)                            # the GEMM fallback gathers act-order activations into temp_state (borrowed: keep it alive)
    lin, gen = _lin(K, N, gs, act, seed=K * 7 + N + rows, std=0.02 * (4096 / K) ** 0.5)
    h, d = _handle(ce, lin)
    ow = _oracle_w(lin)
    x = torch.randn(rows, K, generator=gen).half()
    out = torch.empty((rows, N), dtype=torch.float16, device=DEV)
    ce.exllama_ext.q4_matmul_gemv(x.to(DEV), h, out)
    ref = O.q4_matmul_gemv_f32(x.numpy(), **ow)
    _close(out.cpu().numpy(), ref, ulps=1.5)          # fp32 accumulation, one final rounding: <= ~1 ulp at scale
    # residual-fusing epilogue (no_zero): out += x @ W
    res = (torch.randn(rows, N, generator=gen) * 0.5).half()
    out2 = res.to(DEV).clone()
    ce.exllama_ext.q4_matmul_gemv(x.to(DEV), h, out2, no_zero=True)
    _close(out2.cpu().numpy(), O.q4_matmul_gemv_f32(x.numpy(), out=res.numpy(), **ow), ulps=1.5)


def test_q4_gemv_is_deterministic(ce):
    """No atomics: bit-identical across runs (the reference's fp16 atomicAdd split-K is not)."""
    lin, gen = _lin(4096, 4096, 128, False, seed=1)
    h, d = _handle(ce, lin)
    x = torch.randn(1, 4096, generator=gen).half().to(DEV)
    outs = []
    for _ in range(5):
        o = torch.empty((1, 4096), dtype=torch.float16, device=DEV)
        ce.exllama_ext.q4_matmul_gemv(x, h, o)
        outs.append(o.cpu().numpy().view(np.uint16).copy())
    assert all(np.array_equal(outs[0], o) for o in outs[1:])


GEMM_SHAPES = [(256, 128, 64, True, 16), (512, 96, 128, False, 33), (704, 256, 64, True, 130), (256, 64, 32, True, 8),
               (4096, 4096, 128, False, 128), (4096, 11008, 128, False, 96), (11008, 4096, 128, True, 200),
               (6656, 6656, 32, True, 64),
               # <= 256 rows: the short-prompt kernel (q4_gemm_skinny.hip) -- two 64-row groups, a ragged row group and a ragged
               # column group, K shorter than the 8 waves, one group for the whole K (odd K), an odd group size (tile-kernel fallback)
               (4096, 4096, 128, False, 256), (512, 96, 128, False, 250), (256, 352, 32, True, 70), (1408, 256, 1408, False, 50),
               (1408, 128, 352, False, 50), (5120, 13824, 128, True, 17),
               # 257 .. 512 rows: the 128-row tile kernel (q4_gemm_t16m_kernel<2, 2, 4, 4>)
               (4096, 4096, 128, False, 300), (11008, 4096, 128, True, 512), (4096, 11008, 32, True, 400), (2048, 512, 64, False, 257),
               # > 512 rows: the 256-row pipelined tile, ragged last m-tile
               (4096, 4096, 128, False, 600), (1408, 512, 64, True, 530), (512, 11008, 32, False, 777),
               # more tiles than CUs with a partly filled last round: its tiles are cut along K (plan_gemm_tail): 13B o_proj (320 tiles,
               # act-order: the LDS-staged gather), 7B gate_proj with a ragged last m-tile (430 tiles)
               (5120, 5120, 128, True, 2048), (4096, 11008, 128, False, 1100),
               # BASELINE configs[4] (65B) at the full prompt length: gate / up (172 column tiles) and down_proj (K = 22016)
               (8192, 22016, 128, False, 2048), (22016, 8192, 128, False, 2048)]


@pytest.mark.parametrize("K,N,gs,act,rows", GEMM_SHAPES)
def test_q4_gemm_vs_oracle(ce, K, N, gs, act, rows):
    lin, gen = _lin(K, N, gs, act, seed=K + 3 * N + rows, std=0.02 * (4096 / K) ** 0.5)
    h, d = _handle(ce, lin)
    ow = _oracle_w(lin)
    x = torch.randn(rows, K, generator=gen).half()
    tmp = torch.empty((rows * 2, K), dtype=torch.float16, device=DEV)      # act-order gather scratch = "temp_state"
    z = torch.zeros(64, dtype=torch.float16, device=DEV)
    ce.exllama_ext.prepare_buffers(torch.device(DEV), tmp, z, torch.zeros((1, 64), dtype=torch.float32, device=DEV), z)
    out = torch.empty((rows, N), dtype=torch.float16, device=DEV)
    ce.exllama_ext.q4_matmul_gemm(x.to(DEV), h, out)
    ref = O.q4_matmul_recons(x.numpy(), **ow)
    _close(out.cpu().numpy(), ref, ulps=1.5)          # same W16 bits, fp32 accumulate in a different order
    res = (torch.randn(rows, N, generator=gen) * 0.5).half()
    out2 = res.to(DEV).clone()
    ce.exllama_ext.q4_matmul_gemm(x.to(DEV), h, out2, no_zero=True)
    _close(out2.cpu().numpy(), O.q4_matmul_recons(x.numpy(), out=res.numpy(), **ow), ulps=1.5)


# Short-prompt products on fragment-order activations (csrc/q4_gemm_frag.hip; exl_q4_matmul_frag = one of the four GEMM launches of
# exl_q4_layer_prompt): (K, widths of the matrices of ONE launch, group size, act-order, dual = gate / up + SiLU * mul, RMSNorm prologue)
FRAG_CASES = [
    (4096, (4096, 4096, 4096), 128, False, False, True),        # 7B q / k / v
    (4096, (11008, 11008), 128, False, True, True),             # 7B gate / up
    (4096, (4096,), 128, False, False, False),                  # 7B o_proj
    (11008, (4096,), 128, False, False, False),                 # 7B down_proj (86 row-blocks: K-groups of unequal length)
    (5120, (5120, 5120, 5120), 128, True, False, True),         # 13B act-order: one shared map, applied by the producer
    (5120, (13824, 13824), 128, True, True, True),
    (6656, (6656, 6656, 6656), 32, True, False, True),          # 33B g32: four groups per row-block
    (6656, (17920, 17920), 32, True, True, False),
    (8192, (8192, 1024, 1024), 128, False, False, True),        # 70B GQA: matrices of different widths in one launch
    (512, (640, 640), 64, False, True, True),                   # 4 row-blocks: one step per K-group of four (a consumer's K is a multiple of 128)
    (384, (352,), 32, False, False, False),                     # 3 row-blocks (an empty last step), 22 tiles
    (9216, (256,), 128, True, False, True),                     # K > 8192: the norm's whole-row pass re-reads instead of holding the row in registers
]


FRAG_KERNELS = {0: "auto", 1: "t16r", 2: "t16g", 3: "t16g_4x4", 4: "t16g_4x2", 5: "t16g_8x4", 6: "t16g_k_cut", 7: "t16g_1x4", 8: "t16g_1x2",
                9: "t16g_2x4", 10: "t16g_2x2"}


@pytest.mark.parametrize("K,widths,gs,act,dual,norm", FRAG_CASES)
def test_q4_matmul_frag_vs_oracle(ce, K, widths, gs, act, dual, norm):
    """Every kernel / block shape of the short-prompt GEMM against the oracle's reconstruct + matmul (q4_matmul.cu:301-344), with the
    RMSNorm and act-order gather of the fragment-order producer in front, the residual accumulation and the SiLU * mul epilogue
    (fragment-order output, un-permuted here); rows around the 16- / 64- / 128-row tile edges."""
    ext = ce.exllama_ext
    lins, g_idx = [], None
    gen = torch.Generator().manual_seed(K + 7 * widths[0] + len(widths))
    for N in widths:
        lin = synth.make_q4_linear(K, N, gs, act, gen, "cpu", zeros="rand", std=0.02 * (4096 / K) ** 0.5, g_idx=g_idx)
        g_idx = lin.get("g_idx")                                     # one launch = one act-order map (GPTQ: the matrices share their input)
        lins.append(lin)
    keep = [_handle(ce, lin) for lin in lins]                        # (handle, device tensors: the handle points INTO them)
    hs = [h for h, _ in keep]
    ows = [_oracle_w(lin) for lin in lins]
    nw = (torch.rand(K, generator=gen) + 0.5).half()
    nwd = nw.to(DEV) if norm else None
    eps = 1e-6
    covered = {k: 0 for k in FRAG_KERNELS}
    for rows in (2, 17, 64, 70, 128, 250, 256):
        x = torch.randn(rows, K, generator=gen).half()
        xd = x.to(DEV)
        xn = O.rms_norm(x.numpy(), nw.numpy(), eps) if norm else x.numpy()
        refs = [O.q4_matmul_recons(xn, **ow) for ow in ows]
        res = [(torch.randn(rows, N, generator=gen) * 0.5).half() for N in widths]
        refs_acc = None if dual else [O.q4_matmul_recons(xn, out=r.numpy().copy(), **ow) for r, ow in zip(res, ows)]
        ref_act = O.silu_mul(refs[0], refs[1]) if dual else None
        for kernel, kname in FRAG_KERNELS.items():
            tag = f"{kname}, {rows} rows"
            if dual:
                got = ext.q4_matmul_frag(xd, hs, norm_weight=nwd, eps=eps, dual=True, kernel=kernel)
                if got is None:
                    continue
                full = ext.unfrag(got, got.numel() // (2 * widths[0]), widths[0])
                try:
                    _close(full[:rows].cpu().numpy(), ref_act, ulps=2.0)      # (one more rounding: the product of two rounded halves)
                except AssertionError as e:
                    raise AssertionError(f"{tag}: {e}") from None
                assert not full[rows:].any(), f"{tag}: padding rows of the fragment-order output must be zero"
            else:
                outs = [torch.full((rows, N), float("nan"), dtype=torch.float16, device=DEV) for N in widths]
                if ext.q4_matmul_frag(xd, hs, outs, norm_weight=nwd, eps=eps, kernel=kernel) is None:
                    continue
                acc = [r.to(DEV).clone() for r in res]
                assert ext.q4_matmul_frag(xd, hs, acc, norm_weight=nwd, eps=eps, no_zero=True, kernel=kernel) is not None
                # (with the RMSNorm in front 2 ulps: the row's scale 1 / rms is rounded to fp16 once on either side -- where the two fp32
                # sums of squares straddle a rounding boundary every element of the normed row moves one step: measured 1.56 ulps on one
                # element of the 13B act-order case, r06j)
                try:
                    for o, r in zip(outs, refs):
                        _close(o.cpu().numpy(), r, ulps=2.0 if norm else 1.5)
                    for o, r in zip(acc, refs_acc):
                        _close(o.cpu().numpy(), r, ulps=2.0 if norm else 1.5)
                except AssertionError as e:
                    raise AssertionError(f"{tag}: {e}") from None
            covered[kernel] += 1
    assert covered[0] == 7 and covered[1] == 7, covered               # the launcher's choice and the narrow kernel take every case of the list
    assert covered[2] == 7 and covered[3] == 7 and covered[4] == 7 and covered[5] == 4, str(covered)   # (<8, 4>: from 65 rows on)
    assert covered[6] == (7 if len(widths) == 1 and K >= 4096 else 0), str(covered)   # K cut over blocks: one matrix, a K worth cutting
    assert all(covered[k] == 7 for k in (7, 8, 9, 10)), str(covered)  # one / two row tiles per block: any row count (more row groups)


@pytest.mark.parametrize("K,N,rows", [(4096, 4096, 128), (11008, 4096, 70), (5120, 5120, 250), (4096, 4096, 2)])
def test_q4_matmul_frag_row_sums_of_squares_feed_the_next_norm(ce, K, N, rows):
    """The RMSNorm behind o_proj / down_proj takes its sums of squares from the GEMM's epilogue (GrArgs::rowsq: one partial sum per row and
    column group, of the FINAL fp16 values, residual included) instead of reading the rows again: the partial sums add up to the squares of
    what the launch wrote, and a norm fed with them gives what the norm that sums for itself gives (one fp32 sum in another order: at
    most the last bit of the scale, i.e. one fp16 step on few elements) -- for every kernel / block shape that can write x."""
    ext = ce.exllama_ext
    gen = torch.Generator().manual_seed(K + N + rows)
    lin = synth.make_q4_linear(K, N, 128, False, gen, "cpu", zeros="rand", std=0.02 * (4096 / K) ** 0.5)
    lin2 = synth.make_q4_linear(N, 512, 128, False, gen, "cpu", zeros="rand", std=0.02)
    (h, keep), (h2, keep2) = _handle(ce, lin), _handle(ce, lin2)
    x = torch.randn(rows, K, generator=gen).half().to(DEV)
    res = (torch.randn(rows, N, generator=gen) * 0.5).half()
    nw = (torch.rand(N, generator=gen) + 0.5).half().to(DEV)
    took = 0
    for kernel in range(11):
        out = res.to(DEV).clone()
        sq = torch.full((rows * (N // 16 + 4),), float("nan"), dtype=torch.float32, device=DEV)
        if ext.q4_matmul_frag(x, [h], [out], no_zero=True, kernel=kernel, rowsq_out=sq) is None:
            continue
        slots = ext.last_rowsq_slots
        assert 0 < slots <= N // 16 + 4, (kernel, slots)
        part = sq[:rows * slots].view(rows, slots)
        want = (out.float() ** 2).sum(1)
        assert torch.isfinite(part).all(), kernel
        assert torch.allclose(part.sum(1), want, rtol=2e-5, atol=0), (kernel, float((part.sum(1) - want).abs().max()))
        # the next launch's norm: fed with the partial sums / summing for itself
        a = torch.empty((rows, 512), dtype=torch.float16, device=DEV)
        b = torch.empty_like(a)
        assert ext.q4_matmul_frag(out, [h2], [a], norm_weight=nw, eps=1e-6, kernel=1) is not None
        assert ext.q4_matmul_frag(out, [h2], [b], norm_weight=nw, eps=1e-6, kernel=1, rowsq_in=part.contiguous()) is not None
        _same_up_to_fp32_order(a, b)
        took += 1
    assert took >= 9


def test_q4_matmul_zero_rows_is_a_no_op(ce):
    lin, gen = _lin(512, 256, 128, False, seed=5)
    h, d = _handle(ce, lin)
    x = torch.empty((0, 512), dtype=torch.float16, device=DEV)
    for fn in (ce.exllama_ext.q4_matmul, ce.exllama_ext.q4_matmul_gemv, ce.exllama_ext.q4_matmul_gemm):
        out = torch.empty((0, 256), dtype=torch.float16, device=DEV)
        fn(x, h, out)
    torch.cuda.synchronize()


def test_q4_matmul_threshold_dispatch(ce):
    """rows < matmul_recons_thd -> GEMV, else GEMM (reference: exllama_ext.cpp:217); both agree within tolerance."""
    lin, gen = _lin(512, 256, 128, False, seed=9)
    h, d = _handle(ce, lin)
    ow = _oracle_w(lin)
    for rows in (1, 7, 8, 40):
        x = torch.randn(rows, 512, generator=gen).half()
        out = ce.ext_q4_matmul(x.to(DEV), h, 256)
        ref = O.q4_matmul_recons(x.numpy(), **ow)
        _close(out.cpu().numpy(), ref, ulps=3.0)
    x3 = torch.randn(2, 5, 512, generator=gen).half()                       # leading dims are flattened (cuda_ext.py:100)
    assert ce.ext_q4_matmul(x3.to(DEV), h, 256).shape == (2, 5, 256)


def test_full_size_prefill_property_gemm_equals_reconstruct_times_blas(ce):
    """BASELINE size (7B gate_proj, M = 2048): the fused-dequant MFMA GEMM must equal torch's fp16 GEMM on the
    reconstructed W16 (an independent implementation of the same product) -- size-independent equivalence."""
    K, N, M = 4096, 11008, 2048
    lin, gen = _lin(K, N, 128, False, seed=77, zeros="sym", std=0.02)
    h, d = _handle(ce, lin)
    x = torch.randn(M, K, generator=gen).half()
    x = torch.where(x.abs() < 2.0 ** -10, torch.full_like(x, 2.0 ** -10), x).to(DEV)     # keep fp16 subnormals out of the exactness check
    out = torch.empty((M, N), dtype=torch.float16, device=DEV)
    ce.exllama_ext.q4_matmul_gemm(x, h, out)
    again = torch.empty_like(out)
    ce.exllama_ext.q4_matmul_gemm(x, h, again)
    assert torch.equal(out, again)                                  # deterministic: no atomics, fixed tile order
    w16 = torch.empty((K, N), dtype=torch.float16, device=DEV)
    ce.exllama_ext.q4_reconstruct(h, w16)
    ref = (x.float() @ w16.float())
    err = (out.float() - ref).abs().max().item()
    scale = ref.abs().max().item()
    assert err <= 2.0 * scale * 2.0 ** -10, (err, scale)
    # linearity: (2x) @ W == 2 (x @ W) exactly in fp16 (power-of-two scaling commutes with rounding)
    out2 = torch.empty_like(out)
    ce.exllama_ext.q4_matmul_gemm((x * 2).contiguous(), h, out2)
    normal = out.abs() >= 2.0 ** -13                               # fp16-subnormal outputs round on a coarser grid
    nbad = int(((out2 != out * 2) & normal).sum())
    assert nbad == 0, f"{nbad} of {out.numel()} elements break exact linearity"


@pytest.mark.parametrize("K,N,gs,rows", [(512, 256, 128, 513), (4096, 11008, 128, 2048), (1024, 1408, 32, 700), (256, 128, 64, 1000)])
def test_q4_matmul_dual_equals_separate_products(ce, K, N, gs, rows):
    """exl_q4_matmul_dual (long-prompt gate/up fusion) against the three separate ops the reference issues
    (model.py:266-273) and against the oracle: same tile order and the same fp16 roundings, so bit-identical."""
    lin1, gen = _lin(K, N, gs, False, seed=K + N + rows, std=0.02 * (4096 / K) ** 0.5)
    lin2, _ = _lin(K, N, gs, False, seed=K + N + rows + 1, std=0.02 * (4096 / K) ** 0.5)
    h1, d1 = _handle(ce, lin1)
    h2, d2 = _handle(ce, lin2)
    x = torch.randn(rows, K, generator=gen).half().to(DEV)
    g = torch.empty((rows, N), dtype=torch.float16, device=DEV)
    u = torch.empty_like(g)
    ce.exllama_ext.q4_matmul_gemm(x, h1, g)
    ce.exllama_ext.q4_matmul_gemm(x, h2, u)
    dg = torch.full_like(g, float("nan"))
    du = torch.full_like(g, float("nan"))
    assert ce.exllama_ext.q4_matmul_dual(x, h1, h2, dg, du, silu=False)
    assert torch.equal(dg, g) and torch.equal(du, u)
    act = torch.full_like(g, float("nan"))
    assert ce.exllama_ext.q4_matmul_dual(x, h1, h2, act, None, silu=True)
    ce.exllama_ext.silu_mul(g, u)
    assert torch.equal(act, g)
    if K * N <= 1 << 21 or rows == 2048:                                        # (2048 rows: the 7B shape, whose last round is cut along K)
        ref = O.silu_mul(O.q4_matmul_recons(x.cpu().numpy(), **_oracle_w(lin1)), O.q4_matmul_recons(x.cpu().numpy(), **_oracle_w(lin2)))
        _close(act.cpu().numpy(), ref, ulps=4.0)


def test_q4_matmul_dual_declines_what_it_does_not_cover(ce):
    """Short prompts, act-order and mismatched shapes are left to the separate kernels: nothing launched, False returned."""
    lin1, gen = _lin(512, 256, 128, False, seed=1)
    lin2, _ = _lin(512, 256, 128, True, seed=2)
    lin3, _ = _lin(512, 128, 128, False, seed=3)
    h1, _d1 = _handle(ce, lin1)
    h2, _d2 = _handle(ce, lin2)
    h3, _d3 = _handle(ce, lin3)
    x = torch.randn(600, 512, generator=gen).half().to(DEV)
    out = torch.zeros((600, 256), dtype=torch.float16, device=DEV)
    assert not ce.exllama_ext.q4_matmul_dual(x[:100].contiguous(), h1, h1, out[:100], None, silu=True)
    assert not ce.exllama_ext.q4_matmul_dual(x, h1, h2, out, None, silu=True)
    assert not ce.exllama_ext.q4_matmul_dual(x, h1, h3, out, None, silu=True)
    assert float(out.abs().max()) == 0.0


@pytest.mark.parametrize("hidden,heads,kvh,gs,bsz,q_len,past", [(512, 4, 4, 128, 1, 600, 0), (1024, 8, 2, 64, 2, 300, 5), (4096, 32, 32, 128, 1, 2048, 0)])
def test_q4_qkv_rope_cache_equals_separate_ops(ce, hidden, heads, kvh, gs, bsz, q_len, past):
    """exl_q4_qkv_rope_cache (long-prompt fusion of the attention front half) against the calls the reference issues
    (model.py:431-445: three q4_matmul, two rope_, the cache scatter): q and both caches bit-identical, the rest of the
    cache untouched."""
    hd, max_seq = 128, past + q_len + 7
    gen = torch.Generator().manual_seed(hidden + q_len)
    std = 0.02 * (4096 / hidden) ** 0.5
    lq, _ = _lin(hidden, heads * hd, gs, False, seed=1 + hidden, std=std)
    lk, _ = _lin(hidden, kvh * hd, gs, False, seed=2 + hidden, std=std)
    lv, _ = _lin(hidden, kvh * hd, gs, False, seed=3 + hidden, std=std)
    (hq, _dq), (hk, _dk), (hv, _dv) = _handle(ce, lq), _handle(ce, lk), _handle(ce, lv)
    x = torch.randn(bsz * q_len, hidden, generator=gen).half().to(DEV)
    pos = torch.arange(max_seq, dtype=torch.float32)[:, None] * (10000.0 ** (-torch.arange(0, hd, 2, dtype=torch.float32) / hd))[None, :]
    emb = torch.cat([pos, pos], dim=-1)
    sin, cos = emb.sin().half().to(DEV), emb.cos().half().to(DEV)
    ext = ce.exllama_ext
    # reference sequence
    q = torch.empty((bsz, q_len, heads * hd), dtype=torch.float16, device=DEV)
    k = torch.empty((bsz, q_len, kvh * hd), dtype=torch.float16, device=DEV)
    v = torch.empty_like(k)
    ext.q4_matmul_gemm(x, hq, q.view(-1, heads * hd))
    ext.q4_matmul_gemm(x, hk, k.view(-1, kvh * hd))
    ext.q4_matmul_gemm(x, hv, v.view(-1, kvh * hd))
    ext.rope_(q, sin, cos, past, heads, hd)
    ext.rope_(k, sin, cos, past, kvh, hd)
    kc = torch.full((bsz, kvh, max_seq, hd), 7.0, dtype=torch.float16, device=DEV)
    vc = torch.full_like(kc, -3.0)
    ext.update_cache(k, v, kc, vc, past)
    # fused
    q2 = torch.full_like(q, float("nan"))
    kc2 = torch.full_like(kc, 7.0)
    vc2 = torch.full_like(vc, -3.0)
    assert ext.q4_qkv_rope_cache(x, hq, hk, hv, q2.view(-1, heads * hd), sin, cos, kc2, vc2, q_len, past, heads, kvh, hd, max_seq)
    assert torch.equal(q2, q)
    assert torch.equal(kc2, kc) and torch.equal(vc2, vc)
    assert float(kc2[:, :, past + q_len:].float().min()) == 7.0 and float(vc2[:, :, past + q_len:].float().max()) == -3.0     # beyond the prompt: untouched
    # declines short prompts and act-order
    assert not ext.q4_qkv_rope_cache(x[:256].contiguous(), hq, hk, hv, q2.view(-1, heads * hd)[:256], sin, cos, kc2, vc2, 256 // bsz, 0, heads, kvh, hd, max_seq)


@pytest.mark.parametrize("hidden,heads,kvh,inter,gs,rows", [(1024, 8, 2, 2816, 64, 600), (5120, 40, 40, 13824, 128, 2048)])
def test_prompt_fusions_take_act_order_matrices_that_share_a_map(ce, hidden, heads, kvh, inter, gs, rows):
    """GPTQ with act-order quantises q / k / v (and gate / up) against the same input, so their g_idx tensors are identical
    (synth.make_checkpoint(act_order="gptq")).  The fused prompt launches then gather ONCE -- inside the RMSNorm kernel when the
    norm is their prologue (exl_q4_attn_prompt, exl_q4_mlp_prompt) -- and must reproduce, bit for bit, what the reference issues:
    rms_norm, column_remap + matmul per matrix, rope_ x 2, the cache scatter; rms_norm, two matmuls, silu_mul, matmul + residual
    (model.py:431-445, :266-273; q4_matmul.cu:320-325).  Matrices with DIFFERENT maps are declined (separate ops)."""
    hd, max_seq = 128, rows + 5
    ext = ce.exllama_ext
    keep = _prep_buffers(ce, rows, hidden, inter)
    gen = torch.Generator().manual_seed(hidden + rows)
    std = 0.02 * (4096 / hidden) ** 0.5
    lq, _ = _lin(hidden, heads * hd, gs, True, seed=1 + hidden, std=std)
    share = lambda n, seed, K, g: synth.make_q4_linear(K, n, gs, True, torch.Generator().manual_seed(seed), "cpu", zeros="rand", std=std, g_idx=g)
    lk, lv = share(kvh * hd, 2 + hidden, hidden, lq["g_idx"]), share(kvh * hd, 3 + hidden, hidden, lq["g_idx"])
    lg, _ = _lin(hidden, inter, gs, True, seed=4 + hidden, std=std)
    lu = share(inter, 5 + hidden, hidden, lg["g_idx"])
    ld, _ = _lin(inter, hidden, gs, True, seed=6 + hidden, std=0.02 * (4096 / inter) ** 0.5)
    lk_own, _ = _lin(hidden, kvh * hd, gs, True, seed=7 + hidden, std=std)            # its own permutation
    (hq, _0), (hk, _1), (hv, _2), (hg, _3), (hu, _4), (hdn, _5), (hko, _6) = (_handle(ce, l) for l in (lq, lk, lv, lg, lu, ld, lk_own))
    x = torch.randn(rows, hidden, generator=gen).half().to(DEV)
    w = (1 + 0.1 * torch.randn(hidden, generator=gen)).half().to(DEV)
    pos = torch.arange(max_seq, dtype=torch.float32)[:, None] * (10000.0 ** (-torch.arange(0, hd, 2, dtype=torch.float32) / hd))[None, :]
    emb = torch.cat([pos, pos], dim=-1)
    sin, cos = emb.sin().half().to(DEV), emb.cos().half().to(DEV)
    # ---- attention front half: the reference's sequence
    normed = ce.ext_rms_norm(x, w, 1e-6)
    q = torch.empty((1, rows, heads * hd), dtype=torch.float16, device=DEV)
    k = torch.empty((1, rows, kvh * hd), dtype=torch.float16, device=DEV)
    v = torch.empty_like(k)
    for h_, o_ in ((hq, q), (hk, k), (hv, v)):
        ext.q4_matmul_gemm(normed, h_, o_.view(rows, -1))                       # column_remap inside, per matrix
    ext.rope_(q, sin, cos, 0, heads, hd)
    ext.rope_(k, sin, cos, 0, kvh, hd)
    kc = torch.full((1, kvh, max_seq, hd), 7.0, dtype=torch.float16, device=DEV)
    vc = torch.full_like(kc, -3.0)
    ext.update_cache(k, v, kc, vc, 0)
    for norm_w, src in ((w, x), (None, normed)):                                # norm as the prologue / input already normalised
        q2, kc2, vc2 = torch.full_like(q, float("nan")), torch.full_like(kc, 7.0), torch.full_like(vc, -3.0)
        assert ext.q4_qkv_rope_cache(src, hq, hk, hv, q2.view(rows, -1), sin, cos, kc2, vc2, rows, 0, heads, kvh, hd, max_seq,
                                     norm_weight=norm_w, eps=1e-6)
        # (the separate q / k / v launches of the 13B shape split their last round along K, the fused launch does not)
        _same_up_to_fp32_order(q2, q); _same_up_to_fp32_order(kc2, kc); _same_up_to_fp32_order(vc2, vc)
    assert not ext.q4_qkv_rope_cache(x, hq, hko, hv, q2.view(rows, -1), sin, cos, kc2, vc2, rows, 0, heads, kvh, hd, max_seq, norm_weight=w, eps=1e-6)
    # ---- MLP half
    g = torch.empty((rows, inter), dtype=torch.float16, device=DEV)
    u = torch.empty_like(g)
    ext.q4_matmul_gemm(normed, hg, g)
    ext.q4_matmul_gemm(normed, hu, u)
    ext.silu_mul(g, u)
    want = x.clone()
    ext.q4_matmul_gemm(g, hdn, want, no_zero=True)                              # residual added in the epilogue
    got = x.clone()
    act = torch.full_like(g, float("nan"))
    assert ext.q4_mlp_prompt(got, w, 1e-6, hg, hu, hdn, act)
    assert torch.equal(act, g) and torch.equal(got, want)
    # different maps for gate and up: declined, nothing written
    lu_own, _ = _lin(hidden, inter, gs, True, seed=8 + hidden, std=std)
    huo, _7 = _handle(ce, lu_own)
    untouched = x.clone()
    assert not ext.q4_mlp_prompt(untouched, w, 1e-6, hg, huo, hdn, act)
    assert torch.equal(untouched, x)
    if hidden <= 1024:                                                          # and the oracle, at the size it finishes in seconds
        xn = O.rms_norm(x.cpu().numpy(), w.cpu().numpy(), 1e-6)
        with np.errstate(over="ignore"):
            a_ref = O.silu_mul(O.q4_matmul_recons(xn, **_oracle_w(lg)), O.q4_matmul_recons(xn, **_oracle_w(lu)))
            ref = O.q4_matmul_recons(a_ref, **_oracle_w(ld), out=x.cpu().numpy())
        _close(got.cpu().numpy(), ref, ulps=4.0)
        _close(q.cpu().numpy().reshape(rows, -1), O.rope(O.q4_matmul_recons(xn, **_oracle_w(lq)).reshape(1, -1), sin.cpu().numpy(), cos.cpu().numpy(), 0, heads, hd).reshape(rows, -1), ulps=2.0)


@pytest.mark.parametrize("hidden,vocab,rows", [(512, 640, 1), (4096, 32000, 1), (4096, 32000, 5), (1024, 777, 8)])
def test_embedding_and_head_matmul(ce, hidden, vocab, rows):
    """The HIP forms of the two torch ops on the token path (reference: model.py:1002 embedding, :1077 lm_head): the gather is a
    bit copy; the head is an fp16 GEMV with fp32 accumulation rounded to fp16 (nn.Linear in fp16, then .float())."""
    gen = torch.Generator().manual_seed(hidden + rows)
    table = (torch.randn(vocab, hidden, generator=gen) * 0.05).half()
    ids = torch.randint(0, vocab, (2, 7), generator=gen)
    out = torch.full((2, 7, hidden), float("nan"), dtype=torch.float16, device=DEV)
    ce.exllama_ext.embedding(ids.to(DEV), table.to(DEV), out)
    assert np.array_equal(out.cpu().numpy().view(np.uint16), table[ids].numpy().view(np.uint16))
    x = torch.randn(rows, hidden, generator=gen).half()
    logits = torch.full((rows, vocab), float("nan"), dtype=torch.float32, device=DEV)
    assert ce.exllama_ext.head_matmul(x.to(DEV), table.to(DEV), logits)
    ref = O.half_matmul(x.numpy(), table.t().contiguous().numpy())          # fp32 products and sums, one rounding to fp16
    _close(logits.cpu().numpy(), ref, ulps=1.5)
    big = torch.randn(9, hidden, generator=gen).half()                      # more rows than the GEMV stages: the MFMA GEMM, or -- a vocabulary
    sentinel = torch.full((9, vocab), 7.0, dtype=torch.float32, device=DEV)  # that is no multiple of 4 -- declined with nothing written
    if vocab % 4 == 0:
        assert ce.exllama_ext.head_matmul(big.to(DEV), table.to(DEV), sentinel)
        _close(sentinel.cpu().numpy(), O.half_matmul(big.numpy(), table.t().contiguous().numpy()), ulps=1.5)
    else:
        assert not ce.exllama_ext.head_matmul(big.to(DEV), table.to(DEV), sentinel)
        assert float(sentinel.min()) == 7.0


@pytest.mark.parametrize("hidden,vocab,rows", [(128, 512, 9), (512, 640, 300), (4096, 32000, 2048), (5120, 32000, 777), (1024, 1000, 130)])
def test_whole_sequence_head_gemm(ce, hidden, vocab, rows):
    """exl_head_matmul for more than 8 rows (half_gemm_nt.hip): the lm_head of a whole sequence -- the `-ppl` leg of the reference
    (model.py:1077-1078 with last_id_only = False; perplexity.py:121-138) -- as an fp16 MFMA GEMM whose two tiles travel by LDS-DMA.
    fp32 products and sums, ONE rounding to fp16, widened to fp32: against the oracle's half_matmul; ragged row / vocabulary tiles
    (300 = 2 x 128 + 44 rows, 1000 = 7 x 128 + 104 columns), the shortest K the ring takes (two K steps), poisoned output."""
    gen = torch.Generator().manual_seed(hidden + rows)
    w = (torch.randn(vocab, hidden, generator=gen) * 0.05).half()
    x = torch.randn(rows, hidden, generator=gen).half()
    out = torch.full((rows, vocab), float("nan"), dtype=torch.float32, device=DEV)
    assert ce.exllama_ext.head_matmul(x.to(DEV), w.to(DEV), out)
    got = out.cpu().numpy()
    assert np.array_equal(got, got.astype(np.float16).astype(np.float32))   # every value is an fp16 value
    ref = (x.float() @ w.float().t()).half().float().numpy() if rows * vocab > 1 << 22 else O.half_matmul(x.numpy(), w.t().contiguous().numpy())
    _close(got, ref, ulps=1.5)


def test_q4_matmul_lora(ce):
    lin, gen = _lin(512, 256, 128, False, seed=21)
    h, d = _handle(ce, lin)
    ow = _oracle_w(lin)
    x = torch.randn(5, 512, generator=gen).half()
    a = (torch.randn(512, 16, generator=gen) * 0.05).half()
    b = (torch.randn(16, 256, generator=gen) * 0.05).half()
    out = ce.ext_q4_matmul(x.to(DEV), h, 256, a.to(DEV), b.to(DEV))
    t = O.half_matmul(x.numpy(), a.numpy())
    ref = O.q4_matmul_gemv_f32(x.numpy(), out=O.half_matmul(t, b.numpy()), **ow)
    _close(out.cpu().numpy(), ref, ulps=3.0)


@pytest.mark.parametrize("M,K,N", [(1, 512, 16), (3, 64, 200), (70, 130 // 2 * 2, 96), (130, 256, 64),
                                   (1, 4096, 16), (600, 4096, 64), (600, 64, 4096), (77, 200, 72), (5, 136, 8),     # LoRA down / up shapes, ragged K
                                   # few rows x long K (half_skinny_partial_kernel: K cut over up to 64 blocks): ranks 8 / 16 / 32 / 64, 7 rows, K = 11008
                                   (1, 11008, 64), (7, 4096, 32), (8, 5120, 8), (2, 1024, 16),
                                   # many rows x long K x <= 64 columns (half_tall_partial_kernel: one wave per 16 rows and K part, MFMA): ragged
                                   # last row tile, 48 columns, the down_proj K
                                   (2048, 4096, 16), (777, 11008, 32), (70, 1024, 48), (2048, 5120, 64)])
def test_half_matmul(ce, M, K, N):
    gen = torch.Generator().manual_seed(M * K + N)
    x = torch.randn(M, K, generator=gen).half()
    w = (torch.randn(K, N, generator=gen) * 0.1).half()
    ref = O.half_matmul(x.numpy(), w.numpy())
    _close(ce.ext_half_matmul(x.to(DEV), w.to(DEV), cublas=True).cpu().numpy(), ref, ulps=1.5)
    _close(ce.ext_half_matmul(x.to(DEV), w.to(DEV), cublas=False).cpu().numpy(), ref, ulps=1.5)


# ---------------------------------------------------------------------------------------------------------
# glue kernels
# ---------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("rows,dim", [(1, 4096), (6, 320), (2048, 4096), (3, 8192), (5, 22016 // 8 * 8)])
def test_rms_norm(ce, rows, dim, golden_dir):
    gen = torch.Generator().manual_seed(rows + dim)
    x = (torch.randn(rows, dim, generator=gen) * 2).half()
    w = (1 + 0.1 * torch.randn(dim, generator=gen)).half()
    out = ce.ext_rms_norm(x.to(DEV), w.to(DEV), 1e-6)
    ref = O.rms_norm(x.numpy(), w.numpy(), 1e-6)
    got = out.cpu().numpy()
    # two fp16 multiplies after an fp32 reduction: identical up to 1 ulp where the fp16 rounding of rsqrt flips
    diff = np.abs(got.view(np.int16).astype(np.int32) - ref.view(np.int16).astype(np.int32))
    assert diff.max() <= 2, diff.max()
    assert (diff > 0).mean() < 0.02
    xin = x.to(DEV).clone()
    ce.ext_rms_norm_(xin, w.to(DEV), 1e-6)                                   # in-place variant
    assert torch.equal(xin, out)


def test_rms_norm_golden(ce, golden_dir):
    g = np.load(os.path.join(golden_dir, "ops_small.npz"))
    out = ce.ext_rms_norm(torch.from_numpy(g["rms_x"].copy()).to(DEV), torch.from_numpy(g["rms_w"].copy()).to(DEV), 1e-6)
    diff = np.abs(out.cpu().numpy().view(np.int16).astype(np.int32) - g["rms_out"].view(np.int16).astype(np.int32))
    assert diff.max() <= 1


@pytest.mark.parametrize("bsz,q_len,heads,hd,past", [(1, 1, 32, 128, 0), (1, 1, 32, 128, 1919), (2, 3, 4, 32, 5),
                                                     (1, 2048, 32, 128, 0), (1, 7, 8, 64, 100),
                                                     # head_dim that is no multiple of 16 (OpenLLaMA-3B: 100; rope.cu:27-87 takes any even
                                                     # width): the element-pair kernel
                                                     (1, 1, 32, 100, 77), (2, 9, 4, 100, 3), (1, 300, 8, 36, 0)])
def test_rope_bit_exact(ce, bsz, q_len, heads, hd, past):
    gen = torch.Generator().manual_seed(q_len + heads)
    sin, cos = O.rope_tables(max(2048, past + q_len), hd)
    x = torch.randn(bsz, q_len, heads * hd, generator=gen).half()
    xd = x.to(DEV).clone()
    ce.ext_rope_(xd, torch.from_numpy(sin).to(DEV)[None, None], torch.from_numpy(cos).to(DEV)[None, None], past, heads, hd)
    ref = O.rope(x.numpy().reshape(bsz, -1), sin, cos, past, heads, hd).reshape(x.shape)
    assert np.array_equal(xd.cpu().numpy().view(np.uint16), ref.view(np.uint16))       # fp16 mul + fma: exact contract
    # device-side position (graph replay path) gives the same bits
    xd2 = x.to(DEV).clone()
    pos = torch.tensor([past], dtype=torch.int32, device=DEV)
    ce.exllama_ext.rope_(xd2, torch.from_numpy(sin).to(DEV), torch.from_numpy(cos).to(DEV), 0, heads, hd, past_len_dev=pos)
    assert torch.equal(xd2, xd)


def test_silu_mul(ce):
    gen = torch.Generator().manual_seed(4)
    x = (torch.randn(4, 11008, generator=gen) * 3).half()
    y = torch.randn(4, 11008, generator=gen).half()
    xd = x.to(DEV).clone()
    ce.exllama_ext.silu_mul(xd, y.to(DEV))
    with np.errstate(over="ignore"):
        ref = O.silu_mul(x.numpy(), y.numpy())
    got = xd.cpu().numpy()
    diff = np.abs(got.view(np.int16).astype(np.int32) - ref.view(np.int16).astype(np.int32))
    assert diff.max() <= 2          # hardware exp / rcp approximations: <= 2 fp16 ulp (oracle docstring)
    np.testing.assert_allclose(got.astype(np.float32), ref.astype(np.float32), rtol=3e-3, atol=1e-4)


def test_update_cache_bit_exact(ce):
    gen = torch.Generator().manual_seed(5)
    bsz, q_len, kvh, hd, max_seq, past = 2, 3, 4, 32, 16, 7
    k = torch.randn(bsz, q_len, kvh * hd, generator=gen).half()
    v = torch.randn(bsz, q_len, kvh * hd, generator=gen).half()
    kc = torch.zeros(bsz, kvh, max_seq, hd, dtype=torch.float16, device=DEV)
    vc = torch.zeros_like(kc)
    ce.exllama_ext.update_cache(k.to(DEV), v.to(DEV), kc, vc, past)
    rk = np.zeros((bsz, kvh, max_seq, hd), dtype=np.float16)
    rv = np.zeros_like(rk)
    O.update_cache(k.numpy(), v.numpy(), rk, rv, past)
    assert np.array_equal(kc.cpu().numpy().view(np.uint16), rk.view(np.uint16))
    assert np.array_equal(vc.cpu().numpy().view(np.uint16), rv.view(np.uint16))
    with pytest.raises(RuntimeError, match="exceeds max_seq_len"):
        ce.exllama_ext.update_cache(k.to(DEV), v.to(DEV), kc, vc, max_seq - 1)
    # head_dim 100 (no multiple of 8: the 4-byte copy kernel), also with the position on the device
    bsz, q_len, kvh, hd, max_seq, past = 2, 5, 32, 100, 40, 11
    k = torch.randn(bsz, q_len, kvh * hd, generator=gen).half()
    v = torch.randn(bsz, q_len, kvh * hd, generator=gen).half()
    rk = np.zeros((bsz, kvh, max_seq, hd), dtype=np.float16)
    rv = np.zeros_like(rk)
    O.update_cache(k.numpy(), v.numpy(), rk, rv, past)
    for dev_pos in (False, True):
        kc = torch.zeros(bsz, kvh, max_seq, hd, dtype=torch.float16, device=DEV)
        vc = torch.zeros_like(kc)
        if dev_pos:
            ce.exllama_ext.update_cache(k.to(DEV), v.to(DEV), kc, vc, 0, past_len_dev=torch.tensor([past], dtype=torch.int32, device=DEV))
        else:
            ce.exllama_ext.update_cache(k.to(DEV), v.to(DEV), kc, vc, past)
        assert np.array_equal(kc.cpu().numpy().view(np.uint16), rk.view(np.uint16))
        assert np.array_equal(vc.cpu().numpy().view(np.uint16), rv.view(np.uint16))


# ---------------------------------------------------------------------------------------------------------
# attention
# ---------------------------------------------------------------------------------------------------------
def _attn_case(ce, bsz, q_len, heads, kvh, hd, past, max_seq, seed, with_mask=False, dev_pos=False, poison=False, spikes=()):
    """poison: the cache rows behind the last visible key hold NaN (they may hold anything: a kernel must not let them in, not even under
    a zero weight).  spikes: key positions whose K row is a large multiple of a query row -- a row maximum that grows by far more than
    2^8 LATE in the key loop, the case the lazy reference maximum of the prompt kernels has to rescale for."""
    gen = torch.Generator().manual_seed(seed)
    kv_len = past + q_len
    q = torch.randn(bsz, q_len, heads * hd, generator=gen).half()
    kc = torch.zeros(bsz, kvh, max_seq, hd, dtype=torch.float16)
    vc = torch.zeros_like(kc)
    kc[:, :, :kv_len] = torch.randn(bsz, kvh, kv_len, hd, generator=gen).half()
    vc[:, :, :kv_len] = torch.randn(bsz, kvh, kv_len, hd, generator=gen).half()
    for pos_k in spikes:                                                       # scores ~ +-|q|^2 * 3 / sqrt(hd) ~ 30-40 at this key for the rows
        qrow = min(q_len - 1, max(0, pos_k - past + 7))                        # that see it: e^35 times everything before
        kc[:, :, pos_k] = 3.0 * q.view(bsz, q_len, heads, hd)[:, qrow, ::heads // kvh]
    if poison:
        kc[:, :, kv_len:] = float("nan")
        vc[:, :, kv_len:] = float("nan")
    mask = None
    if with_mask:
        mask = torch.zeros(bsz, 1, q_len, kv_len, dtype=torch.float16)
        mask[:, :, :, :2] = -65504.0                                          # left padding
    out = torch.empty_like(q, device=DEV)
    pos = torch.tensor([past], dtype=torch.int32, device=DEV) if dev_pos else None
    ce.exllama_ext.attention(q.to(DEV), kc.to(DEV), vc.to(DEV), out, past, heads,
                             mask=None if mask is None else mask.to(DEV), past_len_dev=pos)
    qn = q.numpy().reshape(bsz, q_len, heads, hd).transpose(0, 2, 1, 3)
    ref = O.attention(qn, kc.numpy()[:, :, :kv_len], vc.numpy()[:, :, :kv_len], causal_past_len=past,
                      mask=None if mask is None else mask.numpy())
    ref = ref.transpose(0, 2, 1, 3).reshape(bsz, q_len, heads * hd)
    _close(out.cpu().numpy(), ref, ulps=4.0)        # fp16 probabilities inside the MFMA path: a few ulp at |o| ~ 1


@pytest.mark.parametrize("past", [0, 1, 63, 64, 300, 1919, 2047])
def test_attention_decode(ce, past):
    _attn_case(ce, 1, 1, 32, 32, 128, past, 2048, seed=past)


def test_attention_masked_long_prompt_in_workspace_chunks(ce):
    """bsz 2 x 1024 masked tokens x 32 heads: 17 M floats of split partials against a 16 M-float workspace -- the launcher
    walks the query rows in chunks that fit (it used to fail with 'workspace too small').  Checked on a few heads."""
    keep = _prep_buffers(ce, 2, 256, 256)
    gen = torch.Generator().manual_seed(77)
    bsz, q_len, heads, hd = 2, 1024, 32, 128
    q = torch.randn(bsz, q_len, heads * hd, generator=gen).half()
    kc = torch.randn(bsz, heads, q_len, hd, generator=gen).half()
    vc = torch.randn(bsz, heads, q_len, hd, generator=gen).half()
    mask = torch.zeros(bsz, 1, q_len, q_len, dtype=torch.float16)
    mask[0, :, :, :5] = -65504.0                                       # left padding of the first sequence
    mask += torch.triu(torch.full((q_len, q_len), -65504.0), diagonal=1).half()
    mask = mask.clamp(min=-65504.0)
    out = torch.empty_like(q, device=DEV)
    ce.exllama_ext.attention(q.to(DEV), kc.to(DEV), vc.to(DEV), out, 0, heads, mask=mask.to(DEV))
    o = out.cpu().numpy().reshape(bsz, q_len, heads, hd)
    for hsel in (0, 13, 31):
        qn = q.numpy().reshape(bsz, q_len, heads, hd)[:, :, hsel][:, None]
        ref = O.attention(qn, kc.numpy()[:, hsel:hsel + 1], vc.numpy()[:, hsel:hsel + 1], causal_past_len=0, mask=mask.numpy())
        got = o[:, 8:, hsel]                                           # (rows whose every visible key is padding are undefined)
        _close(got, ref[:, 0, 8:], ulps=4.0)


def test_attention_decode_variants(ce):
    _attn_case(ce, 2, 1, 8, 4, 64, 37, 64, seed=1)                 # GQA, bsz 2, hd 64
    _attn_case(ce, 1, 3, 8, 8, 128, 10, 64, seed=2)                # short q_len: causal inside the new tokens
    _attn_case(ce, 1, 1, 4, 4, 128, 500, 2048, seed=3, dev_pos=True)
    _attn_case(ce, 2, 5, 4, 2, 32, 6, 32, seed=4, with_mask=True)  # additive padding mask (model.py:1016-1026)
    # head_dim % 8 != 0 (OpenLLaMA-3B: 100): 8-byte accesses, 32 lanes per key row; decode, a short prompt, a long one (no flash kernel for
    # this width: the split-KV kernel serves every query row), GQA, NaN behind the last key
    _attn_case(ce, 1, 1, 32, 32, 100, 700, 1024, seed=5, poison=True)
    _attn_case(ce, 2, 6, 8, 4, 100, 19, 64, seed=6, with_mask=True)
    _attn_case(ce, 1, 150, 4, 4, 100, 0, 256, seed=7, poison=True)
    _attn_case(ce, 1, 1, 4, 2, 36, 90, 128, seed=8, dev_pos=True)     # 9 lanes of 4 halves -> 16 lanes per key row
    _attn_case(ce, 1, 2, 2, 2, 200, 33, 64, seed=9)                   # 50 lanes -> 64 lanes per key row


@pytest.mark.parametrize("q_len,past,heads,kvh", [(16, 0, 4, 4), (128, 0, 4, 4), (200, 0, 8, 2), (333, 45, 4, 4), (64, 1000, 2, 2)])
def test_attention_prefill_flash(ce, q_len, past, heads, kvh):
    _attn_case(ce, 1, q_len, heads, kvh, 128, past, 2048, seed=q_len + past)


@pytest.mark.parametrize("bsz,q_len,past,heads,kvh", [(1, 129, 0, 4, 4), (2, 257, 63, 4, 2), (1, 192, 0, 8, 8), (1, 320, 1, 4, 1), (1, 64, 193, 4, 4),
                                                      (2, 128, 129, 2, 2), (1, 1000, 1047, 2, 1)])
def test_attention_prefill_8wave_kernel_edges(ce, bsz, q_len, past, heads, kvh):
    """flash_prefill8_kernel (everything beyond one query block of <= 256 keys): one query row in the last block, odd and even numbers of
    key tiles (the odd-tile waves with and without a last tile of their own), key counts that are no multiple of the 64-key tile with NaN
    behind the last key (the DMA of the last tile must re-read valid rows instead), batch 2, GQA down to one kv head, a prompt chunk
    behind a cache, the cache full to its last row."""
    _attn_case(ce, bsz, q_len, heads, kvh, 128, past, 2048 if past + q_len <= 2048 else past + q_len, seed=1000 + q_len + past, poison=True)


@pytest.mark.parametrize("q_len,past,spikes", [(128, 0, (100,)), (512, 0, (70, 300, 301, 500)), (384, 100, (120, 470)), (700, 0, (690,))])
def test_attention_prefill_reference_point_moves_late(ce, q_len, past, spikes):
    """Both prompt kernels keep a LAZY reference maximum (it only moves when a row maximum grows by more than 2^8, then O and l are
    rescaled once).  Random scores never take that branch after the first tiles: here single keys dominate everything before them by
    e^30 and more, at the start, in the middle and in the last tile, for one and for both of the 8-wave kernel's tile sets."""
    _attn_case(ce, 1, q_len, 4, 2, 128, past, 1024, seed=77 + q_len, spikes=spikes)


def test_attention_prefill_full_size_row_checks(ce):
    """S = 2048, 32 heads (BASELINE config 2): spot-check rows of a few heads against the oracle, and the causal
    property: the first row equals v[0], and output rows are independent of later keys."""
    gen = torch.Generator().manual_seed(8)
    heads, hd, S = 32, 128, 2048
    q = torch.randn(1, S, heads * hd, generator=gen).half().to(DEV)
    kc = torch.randn(1, heads, S, hd, generator=gen).half().to(DEV)
    vc = torch.randn(1, heads, S, hd, generator=gen).half().to(DEV)
    out = torch.empty_like(q)
    ce.exllama_ext.attention(q, kc, vc, out, 0, heads)
    o = out.view(S, heads, hd)
    assert torch.equal(o[0], vc[0, :, 0, :])                        # softmax over one key
    for hsel in (0, 17, 31):
        qn = q.view(S, heads, hd)[:, hsel].cpu().numpy()[None, None]
        ref = O.attention(qn, kc[:, hsel:hsel + 1].cpu().numpy(), vc[:, hsel:hsel + 1].cpu().numpy(), causal_past_len=0)
        _close(o[:, hsel].cpu().numpy(), ref[0, 0], ulps=4.0)
    kc2 = kc.clone()
    kc2[:, :, 1024:] = 0                                             # perturb the future: rows < 1024 must not move
    out2 = torch.empty_like(q)
    ce.exllama_ext.attention(q, kc2, vc, out2, 0, heads)
    assert torch.equal(out2[:, :1024], out[:, :1024])


# ---------------------------------------------------------------------------------------------------------
# fused decode ops (compositions exactly as the reference launches them)
# ---------------------------------------------------------------------------------------------------------
def _prep_buffers(ce, max_rows, dim, inter):
    ts = torch.zeros((max(2 * max_rows, 16), max(dim, inter)), dtype=torch.float16, device=DEV)
    tm = torch.zeros((2 * max_rows, inter), dtype=torch.float16, device=DEV)
    tz = torch.zeros((1, 65536), dtype=torch.float32, device=DEV)
    td = torch.zeros((1, 64), dtype=torch.float16, device=DEV)
    ce.exllama_ext.prepare_buffers(torch.device(DEV), ts, tm, tz, td)
    return ts, tm, tz, td


@pytest.mark.parametrize("dim,heads,kvh,act,gs", [(256, 4, 4, False, 64), (512, 8, 4, True, 128), (4096, 32, 32, False, 128)])
def test_q4_attn_fused(ce, dim, heads, kvh, act, gs):
    hd = dim // heads
    keep = _prep_buffers(ce, 2, dim, dim)
    gen = torch.Generator().manual_seed(dim)
    lq, _ = _lin(dim, heads * hd, gs, act, seed=dim + 1, std=0.03)
    lk, _ = _lin(dim, kvh * hd, gs, act, seed=dim + 2, std=0.03)
    lv, _ = _lin(dim, kvh * hd, gs, act, seed=dim + 3, std=0.03)
    hq, dq = _handle(ce, lq)
    hk, dk = _handle(ce, lk)
    hv, dv = _handle(ce, lv)
    max_seq, past = 64, 9
    sin, cos = O.rope_tables(max_seq, hd)
    x = torch.randn(1, 1, dim, generator=gen).half()
    w = (1 + 0.1 * torch.randn(dim, generator=gen)).half()
    q = torch.empty((1, 1, heads * hd), dtype=torch.float16, device=DEV)
    k = torch.empty((1, 1, kvh * hd), dtype=torch.float16, device=DEV)
    v = torch.empty_like(k)
    kc = torch.zeros(1, kvh, max_seq, hd, dtype=torch.float16, device=DEV)
    vc = torch.zeros_like(kc)
    nt = ce.none_tensor
    ce.exllama_ext.q4_attn(x.to(DEV), w.to(DEV), 1e-6, q, k, v, hq, hk, hv, torch.from_numpy(sin).to(DEV)[None, None],
                           torch.from_numpy(cos).to(DEV)[None, None], 1, past, heads, kvh, hd, kc, vc, max_seq,
                           nt, nt, nt, nt, nt, nt, nt)
    rkc = np.zeros((1, kvh, max_seq, hd), dtype=np.float16)
    rvc = np.zeros_like(rkc)
    rq, rk, rv = O.q4_attn(x.numpy(), w.numpy(), 1e-6, _oracle_w(lq), _oracle_w(lk), _oracle_w(lv), sin, cos, past, heads,
                           kvh, hd, rkc, rvc)
    _close(q.cpu().numpy(), rq, ulps=3.0)
    _close(k.cpu().numpy(), rk, ulps=3.0)
    _close(v.cpu().numpy(), rv, ulps=3.0)
    # the scatter is a bit copy of the states the kernel itself produced
    assert torch.equal(kc[0, :, past, :].reshape(-1), k.view(-1))
    assert torch.equal(vc[0, :, past, :].reshape(-1), v.view(-1))
    assert not kc[:, :, :past].any() and not kc[:, :, past + 1:].any()


@pytest.mark.parametrize("dim,inter,gs,act", [(512, 1408, 128, True), (4096, 11008, 128, False), (5120, 13824, 32, False)])
def test_q4_attn_2_and_mlp_fused(ce, dim, inter, gs, act):
    """q4_attn_2 / q4_mlp (reference: q4_attn.cu:206-228, q4_mlp.cu:100-199) at a small shape and at the 7B / 13B layer shapes."""
    keep = _prep_buffers(ce, 2, dim, inter)
    gen = torch.Generator().manual_seed(2)
    std = 0.03 * (512.0 / dim) ** 0.5
    lo, _ = _lin(dim, dim, gs, act, seed=31, std=std)
    lg, _ = _lin(dim, inter, gs, act, seed=32, std=std)
    lu, _ = _lin(dim, inter, gs, False, seed=33, std=std)
    ld, _ = _lin(inter, dim, gs, act, seed=34, std=0.02 * (1408.0 / inter) ** 0.5)
    ho, _d0 = _handle(ce, lo)
    hg, _d1 = _handle(ce, lg)
    hu, _d2 = _handle(ce, lu)
    hdn, _d3 = _handle(ce, ld)
    nt = ce.none_tensor
    for rows in (1, 2):
        x = torch.randn(rows, 1, dim, generator=gen).half()
        attn = torch.randn(rows, 1, dim, generator=gen).half()
        xd = x.to(DEV).clone()
        ce.exllama_ext.q4_attn_2(xd, attn.to(DEV), ho, nt, nt, nt)
        ref = O.q4_attn_2(x.numpy(), attn.numpy(), _oracle_w(lo))
        _close(xd.cpu().numpy(), ref, ulps=2.0)
        w = (1 + 0.1 * torch.randn(dim, generator=gen)).half()
        xm = x.to(DEV).clone().view(-1, dim)
        ce.exllama_ext.q4_mlp(xm, w.to(DEV), 1e-6, hg, hu, hdn, nt, nt, nt, nt, nt, nt, nt)
        with np.errstate(over="ignore"):
            refm = O.q4_mlp(x.numpy().reshape(-1, dim), w.numpy(), 1e-6, _oracle_w(lg), _oracle_w(lu), _oracle_w(ld))
        _close(xm.cpu().numpy(), refm, ulps=4.0)


@pytest.mark.parametrize("dim,heads,inter,ranks", [(4096, 32, 11008, (16, 16, 16, 16, 16, 16, 16)), (4096, 32, 11008, (16, 8, 0, 64, 12, 0, 24)),
                                                   (5120, 40, 13824, (0, 32, 4, 0, 0, 16, 0))])
def test_one_row_adapter_launches_equal_the_separate_products(ce, dim, heads, inter, ranks):
    """q4_attn / q4_attn_2 / q4_mlp with LoRA operands at ONE row (the reference's per-token call, model.py:254-289) run the executor's
    fused launches + its two adapter kernels (csrc/decode_fused.hip: dec_op_gemv, dec_op_lora); at two rows the same entry points
    run norm, the two half GEMMs per adapter and the q4 products one after the other (the path test_q4_matmul_lora and the LoRA
    prefill test pin against the oracle).  Same arithmetic up to the order of the fp32 sums: row 0 of the two-row call is the yardstick.
    Ranks in the order q k v o gate up down (0 = no adapter on that projection), mixed, not multiples of 8, up to 64."""
    hd, gs, max_seq, past = dim // heads, 128, 32, 5
    keep = _prep_buffers(ce, 2, dim, inter)
    gen = torch.Generator().manual_seed(dim + sum(ranks))
    std = 0.03 * (512.0 / dim) ** 0.5
    shapes = [(dim, dim), (dim, dim), (dim, dim), (dim, dim), (dim, inter), (dim, inter), (inter, dim)]
    hs = [_handle(ce, _lin(K, N, gs, False, seed=300 + i, std=std if i < 6 else 0.02 * (1408.0 / inter) ** 0.5)[0]) for i, (K, N) in enumerate(shapes)]
    nt = ce.none_tensor
    ab = []
    for (K, N), r in zip(shapes, ranks):
        if r == 0:
            ab.append((nt, nt))
        else:
            ab.append(((torch.randn(K, r, generator=gen) * 0.04 * (16.0 / r) ** 0.5).half().to(DEV), (torch.randn(r, N, generator=gen) * 0.04).half().to(DEV)))
    sin, cos = (torch.from_numpy(t).to(DEV)[None, None] for t in O.rope_tables(max_seq, hd))
    w = (1 + 0.1 * torch.randn(dim, generator=gen)).half().to(DEV)
    x1 = torch.randn(1, 1, dim, generator=gen).half()
    attn1 = torch.randn(1, 1, dim, generator=gen).half()
    out = {}
    for rows in (1, 2):
        x = x1.repeat(rows, 1, 1).to(DEV)
        attn = attn1.repeat(rows, 1, 1).to(DEV)
        lt = torch.zeros((rows, 64), dtype=torch.float16, device=DEV)
        q = torch.empty((rows, 1, dim), dtype=torch.float16, device=DEV)
        k, v = torch.empty_like(q), torch.empty_like(q)
        kc = torch.zeros(rows, heads, max_seq, hd, dtype=torch.float16, device=DEV)
        vc = torch.zeros_like(kc)
        ce.exllama_ext.q4_attn(x, w, 1e-6, q, k, v, hs[0][0], hs[1][0], hs[2][0], sin, cos, 1, past, heads, heads, hd, kc, vc, max_seq,
                               ab[0][0], ab[0][1], ab[1][0], ab[1][1], ab[2][0], ab[2][1], lt)
        h = x.clone()
        ce.exllama_ext.q4_attn_2(h, attn, hs[3][0], ab[3][0], ab[3][1], lt)
        m = x.clone().view(-1, dim)
        ce.exllama_ext.q4_mlp(m, w, 1e-6, hs[4][0], hs[5][0], hs[6][0], ab[4][0], ab[4][1], ab[5][0], ab[5][1], ab[6][0], ab[6][1], lt)
        out[rows] = {"q": q[0], "k": k[0], "v": v[0], "kc": kc[0, :, past], "vc": vc[0, :, past], "attn_2": h[0], "mlp": m[0]}
    for name, ref in out[2].items():
        _close(out[1][name].float().cpu().numpy(), ref.float().cpu().numpy(), ulps=3.0)
    # and the adapters are not a no-op on the outputs they touch
    base = torch.empty((1, 1, dim), dtype=torch.float16, device=DEV)
    b2, b3 = torch.empty_like(base), torch.empty_like(base)
    kc = torch.zeros(1, heads, max_seq, hd, dtype=torch.float16, device=DEV)
    ce.exllama_ext.q4_attn(x1.to(DEV), w, 1e-6, base, b2, b3, hs[0][0], hs[1][0], hs[2][0], sin, cos, 1, past, heads, heads, hd, kc, torch.zeros_like(kc), max_seq,
                           nt, nt, nt, nt, nt, nt, nt)
    for name, plain, r in (("q", base, ranks[0]), ("k", b2, ranks[1]), ("v", b3, ranks[2])):
        same = torch.equal(plain[0], out[1][name])
        assert same == (r == 0), (name, r)


def test_compiled_binding_and_ctypes_path_launch_the_same_work(ce):
    """exllama_amd/_exl_fast.so (the compiled binding of the per-token entry points, csrc/binding/exl_fast.cpp) against the ctypes
    methods it replaces (kept in the class as the A/B reference): the same C-ABI calls with the same arguments, so every output is
    bit-identical -- q4_matmul, rms_norm, rope_ (host and device-side position), q4_attn, attention, q4_attn_2, q4_mlp with LoRA
    operands present and absent."""
    fast, slow_cls, ext = ce.FAST_BINDING, type(ce.exllama_ext), ce.exllama_ext
    assert fast is not None and ext.q4_attn is fast.q4_attn
    slow = lambda name: (lambda *a, **kw: getattr(slow_cls, name)(ext, *a, **kw))
    dim, heads, kvh, hd, inter, gs, r = 512, 4, 2, 128, 1408, 128, 16
    keep = _prep_buffers(ce, 2, dim, inter)
    gen = torch.Generator().manual_seed(77)
    lins = [_lin(K, N, gs, act, seed=90 + i, std=0.03)[0] for i, (K, N, act) in
            enumerate([(dim, heads * hd, True), (dim, kvh * hd, True), (dim, kvh * hd, False), (dim, dim, True), (dim, inter, False), (dim, inter, True), (inter, dim, True)])]
    (hq, _0), (hk, _1), (hv, _2), (ho, _3), (hg, _4), (hu, _5), (hdn, _6) = (_handle(ce, l) for l in lins)
    max_seq, past = 64, 11
    sin, cos = (torch.from_numpy(t).to(DEV)[None, None] for t in O.rope_tables(max_seq, hd))
    x = torch.randn(1, 1, dim, generator=gen).half().to(DEV)
    w = (1 + 0.1 * torch.randn(dim, generator=gen)).half().to(DEV)
    nt = ce.none_tensor
    lora = lambda n_out, n_in=dim: ((torch.randn(n_in, r, generator=gen) * 0.05).half().to(DEV), (torch.randn(r, n_out, generator=gen) * 0.05).half().to(DEV))
    pos = torch.tensor([past], dtype=torch.int32, device=DEV)
    adapters = {}
    for with_lora in (False, True):
        la = {k: (lora(n) if with_lora else (nt, nt)) for k, n in (("q", heads * hd), ("k", kvh * hd), ("v", kvh * hd), ("o", dim), ("g", inter), ("u", inter))}
        adapters[with_lora] = (la, lora(dim, inter) if with_lora else (nt, nt), torch.zeros((1, r), dtype=torch.float16, device=DEV) if with_lora else nt)
    results = {}
    for tag, pick in (("fast", lambda n: getattr(fast, n)), ("ctypes", slow)):
        out = {}
        # q4_matmul / rms_norm / rope_
        xn = torch.empty((3, dim), dtype=torch.float16, device=DEV)
        x3 = torch.randn(3, dim, generator=torch.Generator().manual_seed(5)).half().to(DEV)
        pick("rms_norm")(x3, w, xn, 1e-6)
        y = torch.empty((3, heads * hd), dtype=torch.float16, device=DEV)
        pick("q4_matmul")(xn, hq, y)
        y2 = y.clone().view(1, 3, heads * hd)
        pick("rope_")(y2, sin, cos, 3, heads, hd)
        y3 = y.clone().view(1, 3, heads * hd)
        pick("rope_")(y3, sin, cos, 0, heads, hd, past_len_dev=pos)
        out.update(xn=xn, y=y, y2=y2, y3=y3)
        for with_lora in (False, True):
            la, ld, lt = adapters[with_lora]
            q = torch.empty((1, 1, heads * hd), dtype=torch.float16, device=DEV)
            k = torch.empty((1, 1, kvh * hd), dtype=torch.float16, device=DEV)
            v = torch.empty_like(k)
            kc = torch.randn(1, kvh, max_seq, hd, generator=torch.Generator().manual_seed(6)).half().to(DEV)
            vc = torch.randn(1, kvh, max_seq, hd, generator=torch.Generator().manual_seed(7)).half().to(DEV)
            h = x.clone()
            pick("q4_attn")(h, w, 1e-6, q, k, v, hq, hk, hv, sin, cos, 1, past, heads, kvh, hd, kc, vc, max_seq,
                            la["q"][0], la["q"][1], la["k"][0], la["k"][1], la["v"][0], la["v"][1], lt)
            attn = torch.empty_like(q)
            pick("attention")(q, kc, vc, attn, past, heads)
            pick("q4_attn_2")(h, attn, ho, la["o"][0], la["o"][1], lt)
            m = h.clone().view(1, dim)
            pick("q4_mlp")(m, w, 1e-6, hg, hu, hdn, la["g"][0], la["g"][1], la["u"][0], la["u"][1], ld[0], ld[1], lt)
            out.update({f"q{with_lora}": q, f"kc{with_lora}": kc, f"attn{with_lora}": attn, f"h{with_lora}": h, f"m{with_lora}": m})
        torch.cuda.synchronize()
        results[tag] = out
    for key, a in results["fast"].items():
        assert torch.isfinite(a.float()).all(), key
        assert torch.equal(a, results["ctypes"][key]), key
    assert not torch.equal(results["fast"]["mFalse"], results["fast"]["mTrue"])      # the adapters did something
