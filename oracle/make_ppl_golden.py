"""TEST INFRASTRUCTURE.  Golden values for the perplexity harness (BASELINE's `-ppl` leg) from the REFERENCE'S OWN
code: imports /root/reference/perplexity.py in this container (CPU; `cuda_ext` stubbed) and drives its Perplexity.load / .test
(perplexity.py:56-138) with a deterministic stand-in model (logits are a fixed function of token id, position and vocabulary index;
it advances the cache position like ExLlama.forward) and a byte tokenizer.  Recorded: the chunking of a raw text and of a .jsonl
dataset for several (chunk_size, truncate, overlap, minlength) settings, and the perplexity the reference prints for each, in
whole-chunk and in token-by-token mode.

    python oracle/make_ppl_golden.py      ->  tests/golden/ppl_ref.json
"""
import contextlib
import importlib
import io
import json
import os
import re
import sys
import tempfile
import types

import torch

REF = os.environ.get("EXL_REFERENCE", "/root/reference")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
V = 97


class StandInCache:
    def __init__(self):
        self.current_seq_len = 0


class StandInModel:
    """logits[b, t, v] = 3 sin(0.37 id + 0.11 v + 0.05 position): depends on the cache position, like a real model."""

    def __init__(self):
        self.calls = 0

    def forward(self, input_ids, cache, last_id_only=True, lora=None, **kw):
        self.calls += 1
        b, t = input_ids.shape
        pos = cache.current_seq_len + torch.arange(t, dtype=torch.float64)
        v = torch.arange(V, dtype=torch.float64)
        lg = 3.0 * torch.sin(0.37 * input_ids.double().unsqueeze(-1) + 0.11 * v.view(1, 1, V) + 0.05 * pos.view(1, t, 1))
        cache.current_seq_len += t
        lg = lg.float()
        return lg[:, -1:, :] if last_id_only else lg


class ByteTokenizer:
    def encode(self, text):
        return torch.tensor([[b % V for b in text.encode("utf-8")]], dtype=torch.long)


TEXT = "".join(chr(32 + (i * 7 + (i // 13) * 3) % 90) for i in range(700))
JSONL = [{"text": "".join(chr(40 + (i * 5 + j) % 60) for i in range(n))} for j, n in enumerate([3, 80, 45, 12, 130, 60])]
RAW_CASES = [dict(chunk_size=128, chunk_truncate=None, overlap=0), dict(chunk_size=100, chunk_truncate=None, overlap=20),
             dict(chunk_size=64, chunk_truncate=40, overlap=8), dict(chunk_size=50, chunk_truncate=None, overlap=500)]
JSON_CASES = [dict(chunk_size=100, chunk_truncate=None, minlength=10), dict(chunk_size=50, chunk_truncate=30, minlength=50)]


def write_datasets(d):
    raw = os.path.join(d, "data.txt")
    with open(raw, "w", encoding="utf-8") as f:
        f.write(TEXT)
    js = os.path.join(d, "data.jsonl")
    with open(js, "w") as f:
        for row in JSONL:
            f.write(json.dumps(row) + "\n")
    return raw, js


def main():
    sys.path.insert(0, REF)
    sys.modules["cuda_ext"] = types.ModuleType("cuda_ext")
    ppl = importlib.import_module("perplexity")
    out = []
    with tempfile.TemporaryDirectory() as d:
        raw, js = write_datasets(d)
        for kind, path, cases in (("raw", raw, RAW_CASES), ("jsonl", js, JSON_CASES)):
            for case in cases:
                rec = {"kind": kind, "args": case}
                for mode in (False, True):
                    p = ppl.Perplexity("default", StandInModel(), StandInCache(), ByteTokenizer())
                    p.load(path, **case)
                    buf = io.StringIO()
                    with contextlib.redirect_stdout(buf):
                        p.test(ppl_token=mode)
                    rec["chunks"] = [[int(c.shape[1]), int(c[0, 0]), int(c[0, -1])] for c in p.dataset_chunks]
                    rec["ppl_token" if mode else "ppl_chunk"] = float(re.search(r"Perplexity: ([0-9.]+)", buf.getvalue()).group(1))
                out.append(rec)
    path = os.path.join(ROOT, "tests", "golden", "ppl_ref.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("wrote", path, len(out), "cases", [(r["ppl_chunk"], r["ppl_token"]) for r in out])


if __name__ == "__main__":
    main()
