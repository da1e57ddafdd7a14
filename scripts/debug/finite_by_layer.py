"""Debug aid: where do the activations of a full-depth synthetic model leave the fp16 range?  Prints max |hidden| after every layer for a
few prompt lengths (op path) -- python scripts/debug/finite_by_layer.py --model 13b --act-order --rows 4,64,600"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from exllama_amd import synth                                              # noqa: E402
from exllama_amd.model import ExLlama, ExLlamaBuffer, ExLlamaCache, ExLlamaConfig   # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="13b")
    ap.add_argument("--act-order", action="store_true")
    ap.add_argument("--groupsize", type=int, default=128)
    ap.add_argument("--seed", type=int, default=23)
    ap.add_argument("--zeros", default="sym")
    ap.add_argument("--rows", default="4,64,600")
    ap.add_argument("--layers", type=int, default=None)
    ap.add_argument("--head-scales", default="")
    ap.add_argument("--nibbles", default="centered")
    a = ap.parse_args()
    dims = synth.PRESETS[a.model]
    L = a.layers or dims.num_hidden_layers
    t = synth.make_checkpoint(dims, groupsize=a.groupsize, act_order="gptq" if a.act_order else False, seed=a.seed, device="cuda:0", zeros=a.zeros, num_layers=L, nibbles=a.nibbles)
    cfg = ExLlamaConfig(synth.config_dict(dims, L))
    cfg.max_seq_len = cfg.max_input_len = 2048
    model = ExLlama(cfg, tensors=t)
    for rows in [int(r) for r in a.rows.split(",")]:
        ids = torch.randint(1, dims.vocab_size, (1, rows), generator=torch.Generator().manual_seed(17)).to("cuda:0")
        cache = ExLlamaCache(model)
        hidden = model.embed(ids)
        buf = ExLlamaBuffer(cfg)
        line = []
        for i, layer in enumerate(model.layers):
            hidden = layer.forward(hidden, cache, buf, None)
            m = float(hidden.float().abs().max())
            line.append(f"{m:.3g}")
            if not torch.isfinite(hidden).all():
                line.append(f"<- not finite at layer {i}")
                break
        print(f"rows {rows}: " + " ".join(line), flush=True)
        if a.head_scales and torch.isfinite(hidden).all():
            import math
            lg = model.head(hidden, last_id_only=False).float()
            for hs in [float(x) for x in a.head_scales.split(",")]:
                p = torch.log_softmax(lg * hs, -1)
                ent = float(-(p.exp() * p).sum(-1).mean())
                print(f"   head x{hs}: mean next-token entropy {ent:.3f} nats (perplexity of own samples ~ {math.exp(ent):.2f})", flush=True)


if __name__ == "__main__":
    main()
