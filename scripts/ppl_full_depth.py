#!/usr/bin/env python3
"""Perplexity parity at FULL depth (north_star: "perplexity equal to 2 dp"; reference: perplexity.py:92-138, README.md:139-148):
tests/parity.py:perplexity_three_ways on whole models -- 32-layer 7B g128 by default, 13B g128 act-order with --model 13b --act-order
-- one JSON record per text, appended to --out (profiles/r05_model_tolerance_stats.jsonl).  The oracle's prompt pass over all
layers takes minutes of host time per text (and ~4 bytes of host memory per weight), which is why the GPU suite runs ONE text
(tests/test_model_gpu.py::test_perplexity_full_depth_7b) and the other texts are produced here.

    python scripts/ppl_full_depth.py --model 7b --seeds 17,18 --out profiles/r05_model_tolerance_stats.jsonl
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="7b")
    ap.add_argument("--layers", type=int, default=None)
    ap.add_argument("--groupsize", type=int, default=128)
    ap.add_argument("--act-order", action="store_true")
    ap.add_argument("--tokens", type=int, default=1536)
    ap.add_argument("--seeds", default="17")
    ap.add_argument("--head-scale", type=float, default=4.6)
    ap.add_argument("--zeros", default="sym", choices=["sym", "rand"], help="stored zero points: symmetric (bench.py's models) or random per group and column")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "ppl_full_depth.jsonl"))
    ap.add_argument("--hip-only", metavar="DIR", default=None,
                    help="GPU box: sample + score on the HIP paths only and leave <DIR>/ppl_<model>_<seed>.pt (record + token ids) for --oracle-from")
    ap.add_argument("--oracle-from", metavar="DIR", default=None,
                    help="any host, no GPU needed: finish the records a --hip-only run left in DIR with the CPU oracle's perplexity of the same texts")
    args = ap.parse_args()
    from exllama_amd import synth
    import torch
    from parity import perplexity_hip, perplexity_oracle, perplexity_three_ways
    say = lambda *a: print(*a, flush=True)
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    if args.oracle_from:
        import glob
        for path in sorted(glob.glob(os.path.join(args.oracle_from, "ppl_*.pt"))):
            blob = torch.load(path)
            rec = perplexity_oracle(blob["rec"], blob["ids"], synth.PRESETS[blob["rec"]["model"]], log=say)
            rec["oracle_host"] = "container CPU (the HIP half ran on the MI355X box: scripts/ppl_full_depth.py --hip-only)"
            print(json.dumps(rec), flush=True)
            with open(args.out, "a") as f:
                f.write(json.dumps(rec) + "\n")
        return
    dims = synth.PRESETS[args.model]
    L = dims.num_hidden_layers if args.layers is None else args.layers
    act = "gptq" if args.act_order else False
    for seed in [int(s) for s in args.seeds.split(",")]:
        if args.hip_only:
            os.makedirs(args.hip_only, exist_ok=True)
            rec, ids = perplexity_hip(dims, L, args.groupsize, act, tokens=args.tokens, seed=seed, head_scale=args.head_scale, log=say, zeros=args.zeros)
            rec["model"] = args.model
            torch.save({"rec": rec, "ids": ids}, os.path.join(args.hip_only, f"ppl_{args.model}{'_act' if act else ''}_{seed}.pt"))
            print(json.dumps(rec), flush=True)
            continue
        rec = perplexity_three_ways(dims, L, args.groupsize, act, tokens=args.tokens, seed=seed, head_scale=args.head_scale, log=say, zeros=args.zeros)
        rec["model"] = args.model
        print(json.dumps(rec), flush=True)
        with open(args.out, "a") as f:
            f.write(json.dumps(rec) + "\n")


if __name__ == "__main__":
    main()
