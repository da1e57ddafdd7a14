"""TEST INFRASTRUCTURE.  Golden vectors for the token sampler from the REFERENCE'S OWN code: imports
/root/reference/generator.py in this container (CPU; its `cuda_ext` / `model` / `lora` imports are stubbed -- `sample` is pure
torch) and runs ExLlamaGenerator.sample (generator.py:91-170) on seeded logits.  torch.multinomial is intercepted: the list
of surviving tokens and their probabilities it is handed (everything the reference computes BEFORE the draw: temperature, softmax,
top-k, top-p / min-p, typical) is recorded, and the draw itself is replaced by the inverse-CDF draw from a recorded uniform
number -- the one documented deviation of this repository's sampler (oracle/sampler_oracle.py header).

    python oracle/make_sampler_golden.py      ->  tests/golden/sampler_ref.npz

tests/test_sampler.py (CPU) then requires oracle/sampler_oracle.py to reproduce ids, probabilities and token."""
import importlib
import os
import sys
import types

import numpy as np
import torch

REF = os.environ.get("EXL_REFERENCE", "/root/reference")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load_reference_generator():
    sys.path.insert(0, REF)
    for name, attrs in (("cuda_ext", {}), ("model", {"ExLlama": object, "ExLlamaCache": object}), ("lora", {"ExLlamaLora": object})):
        m = types.ModuleType(name)
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m
    return importlib.import_module("generator")


CASES = [
    # vocab, seed, spread, temperature, top_k, top_p, min_p, typical, u
    (4096, 1, 3.0, 0.95, 40, 0.65, 0.0, 0.0, 0.37),         # the reference's default settings
    (4096, 2, 3.0, 0.7, 1, 0.0, 0.0, 0.0, 0.5),             # greedy
    (4096, 3, 6.0, 1.3, 200, 0.9, 0.0, 0.0, 0.91),
    (4096, 4, 2.0, 1.0, 50, 0.0, 0.0, 0.0, 0.12),           # top-p off
    (4096, 5, 4.0, 0.95, 40, 0.8, 0.05, 0.0, 0.66),         # min-p cut
    (4096, 6, 3.0, 0.95, 40, 0.65, 0.0, 0.4, 0.29),         # typical sampling
    (4096, 7, 5.0, 1.1, 100, 0.95, 0.0, 0.9, 0.77),
    (4096, 8, 1.0, 2.0, 1024, 0.99, 0.0, 0.0, 0.999),       # wide, flat distribution, draw at the very end
    (32000, 9, 3.5, 0.95, 40, 0.65, 0.0, 0.0, 0.05),        # real vocabulary size
    (32000, 10, 3.5, 0.8, 300, 0.85, 0.01, 0.6, 0.58),
    (512, 11, 8.0, 0.5, 10, 0.5, 0.0, 0.0, 0.0),            # peaked, u = 0
    # top_k = 0 (generator.py:110-111: the whole vocabulary is sorted, no renormalisation) and top_k beyond 1024
    (32000, 12, 3.5, 0.95, 0, 0.65, 0.0, 0.0, 0.41),        # the default settings with top_k = 0
    (32000, 13, 1.0, 1.5, 0, 0.9, 0.0, 0.0, 0.73),          # flat: thousands of survivors of the top-p loop
    (4096, 14, 3.0, 1.0, 0, 0.0, 0.0, 0.0, 0.62),           # nothing cut at all: the draw runs over the whole sorted vocabulary
    (4096, 15, 2.0, 1.0, 0, 0.0, 0.0, 0.5, 0.33),           # typical sampling over the whole vocabulary
    (32000, 16, 2.5, 1.0, 5000, 0.95, 0.001, 0.0, 0.88),    # top_k = 5000 (renormalised), min-p cut
    (4096, 17, 4.0, 0.9, 2000, 0.9, 0.0, 0.7, 0.15),        # top_k = 2000, top-p, typical
]


def main():
    gen = load_reference_generator()

    class Host:                                             # the one attribute sample() reads from self
        disallowed_tokens = None
    captured = {}
    real_multinomial = torch.multinomial

    def multinomial(p, n, *a, **k):
        captured["p"] = p.detach().clone()
        c = torch.cumsum(p.double(), 0)
        pick = int(torch.searchsorted(c, torch.tensor(float(np.float32(captured["u"])), dtype=torch.float64), right=True).clamp(max=p.numel() - 1))
        return torch.tensor([pick], dtype=torch.int64)

    out = {"n": np.array(len(CASES))}
    torch.multinomial = multinomial
    try:
        for i, (V, seed, spread, temp, top_k, top_p, min_p, typical, u) in enumerate(CASES):
            g = torch.Generator().manual_seed(seed)
            logits = (torch.randn(V, generator=g) * spread).float()
            captured["u"] = u
            # generator.py:91: sample(self, logits, temperature, top_k, top_p, min_p, typical, num = 1); logits are modified in place
            tok, prob = gen.ExLlamaGenerator.sample(Host(), logits.clone().view(1, 1, V), temp, top_k, top_p, min_p, typical)
            p = captured["p"].float().numpy()
            out[f"logits_{i}"] = logits.numpy()
            out[f"params_{i}"] = np.array([temp, top_k, top_p, min_p, typical, u], dtype=np.float64)
            out[f"probs_{i}"] = p
            out[f"token_{i}"] = np.array(int(tok))
            out[f"tokprob_{i}"] = np.array(float(prob))
            # the surviving token ids in the reference's order: recover them by running sample() again asking for ALL of them
            captured["u"] = u
            torch.multinomial = lambda pp, n, *a, **k: torch.arange(pp.numel())
            ids, _ = gen.ExLlamaGenerator.sample(Host(), logits.clone().view(1, 1, V), temp, top_k, top_p, min_p, typical, num=-1)
            torch.multinomial = multinomial
            out[f"ids_sorted_{i}"] = ids.view(-1).numpy().astype(np.int64)          # (num = -1 returns them sorted by id: generator.py:166-168)
    finally:
        torch.multinomial = real_multinomial
    path = os.path.join(ROOT, "tests", "golden", "sampler_ref.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
