"""CPU restatement of the reference's token sampler -- TEST INFRASTRUCTURE ONLY (see exl_oracle.py header).

Follows /root/reference/generator.py:91-170 (`ExLlamaGenerator.sample`), :344-381 (`gen_single_token`: repetition penalty,
then the BOS ban, then sample) and /root/reference/exllama_ext/cpu_func/rep_penalty.cpp:36-74 (`apply_rep_penalty_cpu`)
statement by statement, in numpy, with ONE deviation that is part of the product's contract (exllama_amd/csrc/sampler.hip):
the final draw.  The reference calls torch.multinomial, whose exponential-race algorithm cannot be reproduced from a stream of
uniform numbers; here, as on the device, the token is drawn by inverse CDF over the surviving list in its order from one
uniform number u in [0, 1).  Given the same logits, history and u the device kernel must return the same token id
(bit-exact integer result) except when u falls within float rounding of a cumulative boundary.

PARITY PINNING: tests/golden/sampler_ref.npz holds what the reference's own ExLlamaGenerator.sample hands to torch.multinomial
(surviving tokens + probabilities) for seeded logits, produced by importing /root/reference/generator.py on CPU
(oracle/make_sampler_golden.py); tests/test_sampler.py requires this module to reproduce survivors, probabilities (3e-6:
torch vs numpy softmax) and the inverse-CDF token.  The repetition-penalty part is pinned by the reference itself: tests compare `rep_penalty` with oracle/_ref/
librep_penalty_ref.so (the reference's rep_penalty.cpp compiled unmodified) through exl_oracle.apply_rep_penalty.
"""

import numpy as np

from . import exl_oracle as O

f32 = np.float32


def philox4x32_10(seed, counter):
    """First output word of Philox4x32-10 (Salmon et al., SC'11; Random123) for counter (counter, 0, 0, 0), key (seed lo,
    seed hi).  Known answer: philox4x32_10(0, 0) = 0x6627e8d5."""
    m0, m1, w0, w1 = 0xD2511F53, 0xCD9E8D57, 0x9E3779B9, 0xBB67AE85
    c = [counter & 0xFFFFFFFF, 0, 0, 0]
    k = [seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF]
    for _ in range(10):
        p0, p1 = m0 * c[0], m1 * c[2]
        c = [((p1 >> 32) ^ c[1] ^ k[0]) & 0xFFFFFFFF, p1 & 0xFFFFFFFF, ((p0 >> 32) ^ c[3] ^ k[1]) & 0xFFFFFFFF, p0 & 0xFFFFFFFF]
        k = [(k[0] + w0) & 0xFFFFFFFF, (k[1] + w1) & 0xFFFFFFFF]
    return c[0]


def uniform_from_philox(seed, position):
    return f32((philox4x32_10(seed, position) >> 8) * (1.0 / 16777216.0))


def _normalize(p):
    """F.normalize(p = 1, dim = -1): x / max(||x||_1, 1e-12), fp32."""
    s = f32(0.0)
    for v in p:
        s = f32(s + v)
    return (p / max(s, f32(1e-12))).astype(f32)


def sample(logits, history, temperature=0.95, top_k=40, top_p=0.65, min_p=0.0, typical=0.0, rep_penalty_max=1.15,
           rep_sustain=256, rep_decay=128, banned_token=-1, u=0.5):
    """logits fp32 [vocab] (not modified), history: the token sequence so far.  Returns (token, probability in the final
    distribution, surviving token ids, their probabilities)."""
    lg = np.array(logits, dtype=f32).reshape(1, -1).copy()
    seq = np.asarray(history, dtype=np.int64).reshape(1, -1)
    if rep_penalty_max != 1.0 and seq.shape[1] > 0:
        O.apply_rep_penalty(seq, rep_penalty_max, rep_sustain, rep_decay, lg)          # rep_penalty.cpp:36-74, generator.py:353
    lg = lg[0]
    if banned_token >= 0:
        lg[banned_token] = f32(-10000.0)                                              # generator.py:355
    lg = (lg / f32(temperature)).astype(f32)                                           # generator.py:104
    lg = (lg + f32(1e-8)).astype(f32)                                                  # generator.py:105
    e = np.exp((lg - lg.max()).astype(f32)).astype(f32)
    probs = (e * f32(1.0 / float(e.astype(np.float64).sum()))).astype(f32)            # generator.py:106 softmax
    order = np.lexsort((np.arange(probs.size), -probs.astype(np.float64)))             # ties: lower id first
    if top_k == 0:                                                                     # generator.py:110-111: torch.sort of the whole
        top_probs = probs[order]                                                       # vocabulary, NOT renormalised
    else:                                                                              # generator.py:112-114: topk + F.normalize
        order = order[:top_k]
        top_probs = _normalize(probs[order])
    top_idx = order
    if top_p > 0.0:                                                                    # generator.py:118-134
        num = 0
        cum = float(top_probs[0])
        while True:
            num += 1
            if num == top_probs.shape[-1]:
                break
            if top_probs[num] < f32(min_p):
                break
            cum += float(top_probs[num])
            if cum > float(f32(top_p)):
                break
        top_probs = _normalize(top_probs[:num])
        top_idx = top_idx[:num]
    if typical > 0.0:                                                                  # generator.py:138-161
        log_probs = np.log((top_probs + f32(1e-10)).astype(f32)).astype(f32)
        neg_entropy = f32((top_probs.astype(np.float64) * log_probs.astype(np.float64)).sum())
        dev = np.abs((neg_entropy - log_probs).astype(f32)).astype(f32)
        o2 = np.lexsort((np.arange(dev.size), dev))
        top_probs, top_idx = top_probs[o2], top_idx[o2]
        num = 0
        cum = float(top_probs[0])
        while True:
            num += 1
            if num == top_probs.shape[-1]:
                break
            cum += float(top_probs[num])
            if cum > float(f32(typical)):
                break
        top_probs = _normalize(top_probs[:num])
        top_idx = top_idx[:num]
    cum, pick = 0.0, top_probs.size - 1
    for i, p in enumerate(top_probs):                                                  # the draw: inverse CDF in list order
        cum += float(p)
        if float(f32(u)) < cum:
            pick = i
            break
    return int(top_idx[pick]), float(top_probs[pick]), top_idx, top_probs


def boundary_distance(top_probs, u):
    """Distance of u to the nearest cumulative boundary of the final distribution (tests skip the token-equality check for
    draws closer than fp32 rounding to a boundary)."""
    c = np.cumsum(top_probs.astype(np.float64))
    return float(np.min(np.abs(c - float(u)))) if c.size else 1.0
