#!/usr/bin/env python3
"""tests/golden/ppl_full_depth_7b.npz: the CPU oracle's per-token negative log-likelihoods of ONE text over all 32 layers of the
synthetic Llama-7B g128 checkpoint of tests/parity.py:perplexity_three_ways (seed 17), so that the GPU suite's full-depth perplexity
test (tests/test_model_gpu.py::test_perplexity_full_depth_7b) does not have to spend five minutes of host time on the oracle's prompt
pass when the text the HIP decode path samples is exactly this one (the sampled ids are compared first; any difference -- a kernel
change that moves a logit across a sampling threshold -- sends the test down the slow path, oracle included).

The text itself comes from the GPU (it is what the HIP path sampled): `scripts/ppl_full_depth.py --model 7b --seeds 17 --hip-only DIR`
on an MI355X leaves DIR/ppl_7b_17.pt; this script (any host, no GPU) reads it, runs the oracle, and writes the fixture.

    python oracle/make_ppl_full_depth_golden.py gpurun_out/r05e/ppl/ppl_7b_17.pt
    python oracle/make_ppl_full_depth_golden.py gpurun_out/r06f/ppl/ppl_13b_act_17.pt      (round 6: BASELINE configs[2], 40 layers, act-order)
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def main():
    from exllama_amd import synth
    from parity import perplexity_oracle
    blob = torch.load(sys.argv[1])
    rec, ids = blob["rec"], blob["ids"]
    dims = synth.PRESETS[rec["model"]]
    assert rec["layers"] == dims.num_hidden_layers and rec["seed"] == 17
    done, nll = perplexity_oracle(rec, ids, dims, log=lambda *a: print(*a, flush=True), return_nll=True)
    out = os.path.join(ROOT, "tests", "golden", "ppl_full_depth_%s%s.npz" % (rec["model"], "_act" if rec.get("act_order") else ""))
    np.savez_compressed(out, ids=ids.numpy().astype(np.int32), oracle_nll=nll.numpy().astype(np.float64),
                        meta=np.array([rec["layers"], rec["groupsize"], rec["seed"], rec["ckpt_seed"], int(rec["head_scale"] * 1000)], dtype=np.int64))
    print("wrote", out, "oracle perplexity", done["values"][2], "HIP whole (at generation time)", done["values"][0])


if __name__ == "__main__":
    main()
