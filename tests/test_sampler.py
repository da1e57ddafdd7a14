"""The on-device sampler (exllama_amd/csrc/sampler.hip) against oracle/sampler_oracle.py, the numpy
restatement of the reference's generator.py:91-170 / :344-381 + rep_penalty.cpp:36-74.  Token ids are integer results:
they must be EQUAL given the same logits, history and uniform draw (draws within float rounding of a cumulative boundary of
the final distribution are the only excuse, and are counted)."""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import exl_oracle as O
from oracle import sampler_oracle as S


# ---- CPU: the oracle itself ------------------------------------------------------------------------------------------
def test_philox_known_answer_and_uniform_range():
    assert S.philox4x32_10(0, 0) == 0x6627E8D5                      # Random123 known-answer test, counter 0 key 0
    us = [float(S.uniform_from_philox(1234, p)) for p in range(2000)]
    assert 0.0 <= min(us) and max(us) < 1.0 and abs(np.mean(us) - 0.5) < 0.03


def test_sampler_oracle_follows_the_reference_loops():
    rs = np.random.RandomState(0)
    logits = (rs.randn(300) * 3).astype(np.float32)
    hist = rs.randint(0, 300, size=40)
    # top_k = 1 is greedy on the penalised, banned logits
    lg = logits.reshape(1, -1).copy()
    O.apply_rep_penalty(hist.reshape(1, -1), 1.15, 256, 128, lg)
    lg[0, 7] = -10000.0
    tok, p, idx, probs = S.sample(logits, hist, top_k=1, banned_token=7, u=0.3)
    assert tok == int(np.argmax(lg[0])) and p == 1.0 and len(idx) == 1
    # the top-p loop by hand (generator.py:121-134): probabilities 0.5, 0.3, 0.2; the token whose addition CROSSES top_p is
    # dropped -- top_p 0.65 keeps one token (0.5 + 0.3 > 0.65), top_p 0.85 keeps two (0.8 <= 0.85, 1.0 > 0.85)
    l3 = np.log(np.array([0.5, 0.3, 0.2], dtype=np.float64)).astype(np.float32)
    assert list(S.sample(l3, [], temperature=1.0, top_k=3, top_p=0.65, rep_penalty_max=1.0, u=0.7)[2]) == [0]
    tok, p, idx, probs = S.sample(l3, [], temperature=1.0, top_k=3, top_p=0.85, rep_penalty_max=1.0, u=0.7)
    assert list(idx) == [0, 1] and abs(probs[0] - 0.625) < 1e-6 and tok == 1
    # min_p cuts inside the loop; top_p = 0 disables the loop altogether
    tok, p, idx, probs = S.sample(l3, [], temperature=1.0, top_k=3, top_p=0.99, min_p=0.25, rep_penalty_max=1.0, u=0.0)
    assert list(idx) == [0, 1] and tok == 0
    assert len(S.sample(l3, [], temperature=1.0, top_k=3, top_p=0.0, rep_penalty_max=1.0)[2]) == 3
    # typical sampling reorders by |entropy - surprise| and cuts
    tok, p, idx, probs = S.sample(l3, [], temperature=1.0, top_k=3, top_p=0.0, typical=0.4, rep_penalty_max=1.0, u=0.0)
    assert len(idx) >= 1 and abs(float(probs.sum()) - 1.0) < 1e-6


# ---- GPU -------------------------------------------------------------------------------------------------------------
def _device_sample(lib, logits, hist, settings, u, max_len):
    from exllama_amd import _lib
    dev = "cuda:0"
    V = logits.size
    lg = torch.from_numpy(logits.copy()).to(dev)
    probs = torch.empty(V, dtype=torch.float32, device=dev)
    history = torch.zeros(max_len + 2, dtype=torch.int64, device=dev)
    n = len(hist)
    if n:
        history[:n] = torch.as_tensor(np.asarray(hist), dtype=torch.int64)
    pos = torch.tensor([n], dtype=torch.int32, device=dev)
    unif = torch.zeros(max_len + 2, dtype=torch.float32, device=dev)
    unif[n] = float(u)
    tok = torch.zeros(1, dtype=torch.int64, device=dev)
    pr = torch.zeros(1, dtype=torch.float32, device=dev)
    _lib.check(lib.exl_sample(0, lg.data_ptr(), probs.data_ptr(), V, history.data_ptr(), tok.data_ptr(), pos.data_ptr(), unif.data_ptr(),
                              pr.data_ptr(), C.byref(settings), torch.cuda.current_stream().cuda_stream), "sample")
    torch.cuda.synchronize()
    assert int(history[n]) == int(tok)
    return int(tok), float(pr)



def test_sampler_oracle_reproduces_the_references_own_sample():
    """Pinned by the reference: tests/golden/sampler_ref.npz holds what /root/reference/generator.py's ExLlamaGenerator.sample
    (:91-170) hands to torch.multinomial -- the surviving tokens and their probabilities after temperature, softmax, top-k,
    top-p / min-p and typical sampling -- for seeded logits (oracle/make_sampler_golden.py, run where the reference lives).  The
    oracle must reproduce the survivors, their probabilities and the inverse-CDF token."""
    import os
    path = os.path.join(os.path.dirname(__file__), "golden", "sampler_ref.npz")
    z = np.load(path)
    n = int(z["n"])
    assert n >= 10
    for i in range(n):
        temp, top_k, top_p, min_p, typical, u = (float(v) for v in z[f"params_{i}"])
        logits = z[f"logits_{i}"]
        tok, prob, idx, probs = S.sample(logits, [], temperature=temp, top_k=int(top_k), top_p=top_p, min_p=min_p, typical=typical,
                                         rep_penalty_max=1.0, u=u)
        ref_p = z[f"probs_{i}"]
        assert probs.shape == ref_p.shape, (i, probs.shape, ref_p.shape)                  # the same number of survivors
        assert np.array_equal(np.sort(np.asarray(idx)), z[f"ids_sorted_{i}"]), i           # ... the same tokens
        np.testing.assert_allclose(probs, ref_p, rtol=3e-6, atol=1e-9, err_msg=str(i))     # ... in the same order (torch vs numpy softmax: 1 ulp)
        assert tok == int(z[f"token_{i}"]), (i, tok, int(z[f"token_{i}"]))
        assert abs(prob - float(z[f"tokprob_{i}"])) <= 3e-6 * max(1.0, prob)

@pytest.mark.gpu
def test_device_sampler_matches_the_oracle_token_for_token():
    from exllama_amd import _lib
    lib = _lib.load()
    rs = np.random.RandomState(42)
    cases = [
        dict(),                                                                     # the reference's default Settings
        dict(temperature=0.7, top_k=100, top_p=0.9),
        dict(temperature=1.3, top_k=1024, top_p=0.0, rep_penalty_max=1.0),         # largest list, no top-p, no penalty
        dict(top_k=50, top_p=0.8, min_p=0.02),
        dict(top_k=64, top_p=0.0, typical=0.6),
        dict(top_k=40, top_p=0.65, typical=0.9, rep_penalty_max=1.3, rep_sustain=-1, rep_decay=0),
        dict(top_k=5, top_p=0.95, rep_sustain=8, rep_decay=16, banned_token=1),
        dict(top_k=1),
        # the whole-vocabulary sort (dec_sample_big_kernel): top_k = 0 as the reference means it (generator.py:110-111: torch.sort, no
        # renormalisation) and top_k beyond the 1024 entries of the LDS network
        dict(top_k=0),
        dict(top_k=0, top_p=0.0, rep_penalty_max=1.0),                              # nothing cut: the draw runs over the sorted vocabulary
        dict(top_k=0, temperature=1.6, top_p=0.92, min_p=0.0002),                   # flat: thousands survive the top-p loop
        dict(top_k=0, top_p=0.0, typical=0.5),                                      # typical sampling over everything
        dict(top_k=3000, top_p=0.9, typical=0.8),
        dict(top_k=5000, top_p=0.0, rep_penalty_max=1.0),
    ]
    near, total = 0, 0
    for ci, kw in enumerate(cases):
        for V, hist_len in ((32000, 600), (32000, 3), (512, 90), (33, 0)):
            for rep in range(3):
                logits = (rs.randn(V) * rs.uniform(1.0, 6.0)).astype(np.float32)
                if rep == 2:
                    logits = np.round(logits)                                       # many exact ties
                hist = rs.randint(0, V, size=hist_len)
                u = float(rs.rand())
                k = dict(kw)
                if k.get("banned_token", -1) >= V:
                    k["banned_token"] = 0
                settings = _lib.ExlSampler(**k)
                got, got_p = _device_sample(lib, logits, hist, settings, u, 700)
                want, want_p, idx, probs = S.sample(logits, hist, u=u, **{**dict(temperature=0.95, top_k=40, top_p=0.65), **k})
                total += 1
                if got != want:
                    assert S.boundary_distance(probs, u) < 1e-5, (ci, V, hist_len, rep, got, want, u)
                    near += 1
                else:
                    assert abs(got_p - want_p) <= 1e-5 * max(1.0, want_p)
    assert near <= max(2, total // 50), (near, total)
    # Philox draws when no uniform numbers are supplied: reproducible, and equal to the oracle's generator
    logits = (rs.randn(32000) * 3).astype(np.float32)
    hist = rs.randint(0, 32000, size=77)
    settings = _lib.ExlSampler(seed=987654321, top_k=200, top_p=0.95)
    dev = "cuda:0"
    outs = []
    for _ in range(2):
        lg = torch.from_numpy(logits.copy()).to(dev)
        probs = torch.empty(32000, dtype=torch.float32, device=dev)
        history = torch.zeros(128, dtype=torch.int64, device=dev)
        history[:77] = torch.as_tensor(hist, dtype=torch.int64)
        pos = torch.tensor([77], dtype=torch.int32, device=dev)
        tok = torch.zeros(1, dtype=torch.int64, device=dev)
        _lib.check(lib.exl_sample(0, lg.data_ptr(), probs.data_ptr(), 32000, history.data_ptr(), tok.data_ptr(), pos.data_ptr(), None, None,
                                  C.byref(settings), torch.cuda.current_stream().cuda_stream), "sample")
        outs.append(int(tok))
    want = S.sample(logits, hist, top_k=200, top_p=0.95, u=S.uniform_from_philox(987654321, 77))[0]
    assert outs[0] == outs[1] == want
    with pytest.raises(RuntimeError, match="top_k"):
        bad = _lib.ExlSampler(top_k=-1)
        _lib.check(lib.exl_sample(0, lg.data_ptr(), probs.data_ptr(), 32000, history.data_ptr(), tok.data_ptr(), pos.data_ptr(), None, None,
                                  C.byref(bad), torch.cuda.current_stream().cuda_stream), "sample")


@pytest.mark.gpu
def test_device_sampler_whole_vocabulary_sort_beyond_65536_entries():
    """Round 5's whole-vocabulary sort (top_k = 0 / > 1024) held 65536 entries; its workspace now grows with the vocabulary: Llama-3's
    128256 tokens and a ragged 70001, against the oracle of the reference's sample() (generator.py:91-170), and a second, smaller
    vocabulary afterwards (the workspace is only ever grown)."""
    from exllama_amd import _lib
    lib = _lib.load()
    rs = np.random.RandomState(7)
    for V in (128256, 70001, 32000):
        for kw in (dict(top_k=0), dict(top_k=0, temperature=1.5, top_p=0.9, min_p=0.0001), dict(top_k=5000, top_p=0.0, rep_penalty_max=1.0),
                   dict(top_k=0, top_p=0.0, typical=0.5)):
            logits = (rs.randn(V) * rs.uniform(1.0, 5.0)).astype(np.float32)
            hist = rs.randint(0, V, size=300)
            u = float(rs.rand())
            got, got_p = _device_sample(lib, logits, hist, _lib.ExlSampler(**kw), u, 400)
            want, want_p, idx, probs = S.sample(logits, hist, u=u, **{**dict(temperature=0.95, top_k=40, top_p=0.65), **kw})
            if got != want:
                assert S.boundary_distance(probs, u) < 1e-5, (V, kw, got, want, u)
            else:
                assert abs(got_p - want_p) <= 1e-5 * max(1.0, want_p)


@pytest.mark.gpu
def test_generate_sample_inside_the_graph_equals_the_host_loop():
    """model.generate_sample: decode kernels + sampler in one replayed graph per token.  The tokens must be the ones the oracle
    sampler picks from the logits an ordinary forward pass produces for the same history and the same draws."""
    from exllama_amd import _lib, synth
    from exllama_amd.model import ExLlama, ExLlamaCache, ExLlamaConfig
    dims = synth.LLAMA_TINY_HD128
    tensors = synth.make_checkpoint(dims, groupsize=128, act_order=False, seed=2, device="cpu", zeros="rand")
    cfg = ExLlamaConfig(synth.config_dict(dims))
    cfg.max_seq_len = 256
    model = ExLlama(cfg, tensors=tensors)
    cache = ExLlamaCache(model)
    prompt = torch.randint(1, dims.vocab_size, (1, 150), generator=torch.Generator().manual_seed(8)).to("cuda:0")
    n = 20
    settings = _lib.ExlSampler(temperature=1.1, top_k=30, top_p=0.9, rep_penalty_max=1.2, rep_sustain=32, rep_decay=16, banned_token=1)
    unif = torch.rand(cfg.max_seq_len + 1, generator=torch.Generator().manual_seed(9)).to("cuda:0")
    model.forward(prompt[:, :-1], cache, preprocess_only=True)
    model.enable_decode_graph(cache)
    got = model.generate_sample(prompt, cache, n, settings=settings, uniforms=unif).cpu().tolist()
    assert cache.current_seq_len == 149 + n and len(got) == n
    # host loop on a fresh cache: forward (same executor kernels) -> oracle sampler with the same history / draws
    cache2 = ExLlamaCache(model)
    model.forward(prompt[:, :-1], cache2, preprocess_only=True)
    model.enable_decode_graph(cache2)
    seq = prompt[0].cpu().tolist()
    near = 0
    for i in range(n):
        logits = model.forward(torch.tensor([[seq[-1]]], device="cuda:0"), cache2)[0, 0].float().cpu().numpy()
        u = float(unif[len(seq)])
        want, _, idx, probs = S.sample(logits, seq, temperature=1.1, top_k=30, top_p=0.9, rep_penalty_max=1.2, rep_sustain=32, rep_decay=16,
                                       banned_token=1, u=u)
        if want != got[i]:
            assert S.boundary_distance(probs, u) < 1e-5, (i, want, got[i])
            near += 1
        seq.append(got[i])                                            # teacher forcing keeps the two histories identical
    assert near <= 1
    # a second call continues from the new position with the extended sequence; changed settings re-capture
    model.enable_decode_graph(cache)                              # (the executor was bound to cache2 for the host loop)
    more = model.generate_sample(torch.tensor([prompt[0].cpu().tolist() + got]), cache, 3, settings=_lib.ExlSampler(top_k=1, rep_penalty_max=1.0))
    assert more.numel() == 3 and cache.current_seq_len == 149 + n + 3
    # generate_greedy on the SAME decoder after generate_sample (round-2 defect: the greedy graphs were only captured when no
    # sampler had created the shared history buffer first -> KeyError 'ggraphs'); top_k = 1 sampling is greedy decoding
    start = cache.current_seq_len
    last_tok = more[-1].view(1, 1)
    greedy = model.generate_greedy(last_tok, cache, 4)
    cache.current_seq_len = start
    again = model.generate_sample(torch.tensor([prompt[0].cpu().tolist() + got + more.cpu().tolist()]), cache, 4,
                                  settings=_lib.ExlSampler(top_k=1, rep_penalty_max=1.0))
    assert greedy.tolist() == again.tolist()
    # top_k = 0 inside the captured graph (the sampler's whole-vocabulary workspace was allocated when the decoder was created)
    start = cache.current_seq_len
    seq0 = prompt[0].cpu().tolist() + got + more.cpu().tolist() + greedy.cpu().tolist()
    s0 = _lib.ExlSampler(temperature=1.2, top_k=0, top_p=0.8, rep_penalty_max=1.1, rep_sustain=16, rep_decay=8)
    wide = model.generate_sample(torch.tensor([seq0]), cache, 5, settings=s0, uniforms=unif).cpu().tolist()
    cache.current_seq_len = start
    model.enable_decode_graph(cache)
    seq, near = list(seq0), 0
    for i in range(5):
        logits = model.forward(torch.tensor([[seq[-1]]], device="cuda:0"), cache)[0, 0].float().cpu().numpy()
        u = float(unif[len(seq)])
        want, _, idx, probs = S.sample(logits, seq, temperature=1.2, top_k=0, top_p=0.8, rep_penalty_max=1.1, rep_sustain=16, rep_decay=8, u=u)
        if want != wide[i]:
            assert S.boundary_distance(probs, u) < 1e-5, (i, want, wide[i])
            near += 1
        seq.append(wide[i])
    assert near <= 1
    model.free_unmanaged()
