"""Per-step error of the decode paths against the oracle on a truncated real-shape model (round 3: 13B act-order showed 7e-4,
6e-4 and then 5.4e-3 x scale on three consecutive eager steps)."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from exllama_amd import synth, cuda_ext
from exllama_amd.model import ExLlama, ExLlamaCache, ExLlamaConfig
from oracle.model_oracle import OracleLlama

key, gs, act, L = sys.argv[1], int(sys.argv[2]), sys.argv[3] == "1", int(sys.argv[4])
P, n_new = int(sys.argv[5]) if len(sys.argv) > 5 else 20, 8
dims = synth.PRESETS[key]
tensors = synth.make_checkpoint(dims, groupsize=gs, act_order=act, seed=3, device="cpu", zeros="rand", num_layers=L)
cfg = ExLlamaConfig(synth.config_dict(dims, L))
cfg.max_seq_len = 1408
cfg.max_input_len = 2048
model = ExLlama(cfg, tensors=tensors)
ref = OracleLlama(synth.config_dict(dims, L), tensors, max_seq_len=cfg.max_seq_len)
ref.prepare()
lib = cuda_ext.exllama_ext._lib
rs = np.random.RandomState(5)
ids = torch.from_numpy(rs.randint(1, dims.vocab_size, size=(1, 2060))).to("cuda:0")
cache = ExLlamaCache(model)
model.forward(ids[:, :P], cache, preprocess_only=True)
for i in range(L):
    ref.kc[i][0, :, :P] = cache.key_states[i][0, :, :P].cpu().numpy()
    ref.vc[i][0, :, :P] = cache.value_states[i][0, :, :P].cpu().numpy()
ref.past = P
toks = ids[0, P:P + n_new].tolist()
ref_steps = [ref.forward(np.array([[t]]))[0, 0] for t in toks]
scale = float(np.abs(np.stack(ref_steps)).max())

def stats(lg, want):
    d = lg.astype(np.float64) - want.astype(np.float64)
    return f"{np.abs(d).max() / scale:.2e}/{np.sqrt((d ** 2).mean()) / np.sqrt((want.astype(np.float64) ** 2).mean()):.2e}"

def run(tag, executor, graph, ring):
    c = ExLlamaCache(model, copy_from=cache)
    c.current_seq_len = P
    model.disable_decode_graph()
    if executor:
        model.enable_decode_graph(c, use_graph=graph)
        for sg in model._decoder["stages"]:
            cuda_ext.check(lib.exl_decoder_set_option(sg["handle"], 0, ring), "opt")
    out = []
    for t in toks:
        out.append(model.forward(torch.tensor([[t]], device="cuda:0"), c)[0, 0].float().cpu().numpy())
    print(f"{tag:28s}", " ".join(stats(o, w) for o, w in zip(out, ref_steps)), flush=True)
    return out

print("scale", scale, "(max err / scale)/(rms rel) per step")
a = run("op path", False, False, 0)
b = run("executor eager, stream", True, False, 0)
c_ = run("executor eager, ring", True, False, 15)
d = run("executor graph, ring", False or True, True, 15)
print("op path vs executor(stream):", " ".join(f"{np.abs(x - y).max() / scale:.2e}" for x, y in zip(a, b)))
print("stream vs ring bitwise:", [bool(np.array_equal(x, y)) for x, y in zip(b, c_)])
