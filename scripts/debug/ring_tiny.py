"""Which GEMV class of the rolling-ring stream differs from the compiler-scheduled one on the tiny head-dim-128 preset (the
model test that failed in round 3, GPU call 2)?  Eager decode steps after a 200-token prompt, per-class ring masks."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from exllama_amd import synth, cuda_ext
from exllama_amd.model import ExLlama, ExLlamaCache, ExLlamaConfig

name, gs, prompt, max_seq = sys.argv[1] if len(sys.argv) > 1 else "tiny_hd128", 128, int(sys.argv[2]) if len(sys.argv) > 2 else 200, 320
dims = synth.PRESETS[name]
tensors = synth.make_checkpoint(dims, groupsize=gs, act_order=False, seed=21, device="cpu", zeros="rand", num_layers=None)
cfg = ExLlamaConfig(synth.config_dict(dims, None))
cfg.max_seq_len = max_seq
cfg.max_input_len = max_seq
model = ExLlama(cfg, tensors={k: v.clone() for k, v in tensors.items()})
lib = cuda_ext.exllama_ext._lib
ids = torch.randint(1, dims.vocab_size, (1, prompt), generator=torch.Generator().manual_seed(4)).to("cuda:0")
toks = [3, 17, 99, 250, 7, 11]

def run(mask, fence=1, use_graph=False):
    c = ExLlamaCache(model)
    model.disable_decode_graph()
    model.forward(ids, c, preprocess_only=True)
    if mask is not None:
        model.enable_decode_graph(c, use_graph=False)
        for sg in model._decoder["stages"]:
            cuda_ext.check(lib.exl_decoder_set_option(sg["handle"], 0, mask), "opt")
            cuda_ext.check(lib.exl_decoder_set_option(sg["handle"], 1, fence), "opt")
    outs = []
    for t in toks:
        outs.append(model.forward(torch.tensor([[t]], device="cuda:0"), c)[0, 0].float().cpu().numpy().copy())
    return np.stack(outs)

ops = run(None)
base = run(0)
print("scale", float(np.abs(ops).max()), "stream vs ops per token:", [round(float(np.abs(base[i] - ops[i]).max()), 4) for i in range(len(toks))])
for rep in range(2):
    for mask in (1, 2, 4, 8, 15, 0):
        for fence in (1, 0):
            r = run(mask, fence)
            print(f"rep {rep} mask {mask:2d} fence {fence}: vs stream per token", [round(float(np.abs(r[i] - base[i]).max()), 4) for i in range(len(toks))], flush=True)
