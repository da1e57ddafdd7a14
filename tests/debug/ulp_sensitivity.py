"""CPU only: how much do the oracle's logits move when the rope'd q / k of a decode step change by one fp16 ulp on a
quarter of their elements?  (Round 3: three HIP paths agree with each other to one logit ulp on a 13B act-order layer but
one decode step in eight sits 4.8e-3 x scale away from the oracle; the others 6e-4.)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from exllama_amd import synth
from oracle import exl_oracle as O
from oracle.model_oracle import OracleLlama

key, gs, act, L = sys.argv[1], int(sys.argv[2]), sys.argv[3] == "1", int(sys.argv[4])
P, n_new = 20, 8
dims = synth.PRESETS[key]
tensors = synth.make_checkpoint(dims, groupsize=gs, act_order=act, seed=3, device="cpu", zeros="rand", num_layers=L)
rs = np.random.RandomState(5)
ids = rs.randint(1, dims.vocab_size, size=(1, 2060))
ref = OracleLlama(synth.config_dict(dims, L), tensors, max_seq_len=64)
ref.prepare()
ref.forward(ids[:, :P])
kc0, vc0 = [k.copy() for k in ref.kc], [v.copy() for v in ref.vc]

def steps(noise):
    ref.kc, ref.vc, ref.past = [k.copy() for k in kc0], [v.copy() for v in vc0], P
    rope0 = O.rope
    nrs = np.random.RandomState(11)
    def rope(*a, **kw):
        y = rope0(*a, **kw)
        if noise:
            bump = nrs.rand(*y.shape) < 0.25
            y = np.where(bump, np.nextafter(y, np.where(nrs.rand(*y.shape) < 0.5, np.float16(np.inf), np.float16(-np.inf)).astype(np.float16)), y)
        return y
    O.rope = rope
    try:
        return [ref.forward(ids[:, P + i:P + i + 1])[0, 0] for i in range(n_new)]
    finally:
        O.rope = rope0

base, pert = steps(False), steps(True)
scale = float(np.abs(np.stack(base)).max())
print("per step  max|d|/scale  rms rel  (q, k moved by one ulp on 25 % of the elements)")
for b, p in zip(base, pert):
    d = p.astype(np.float64) - b
    print(f"  {np.abs(d).max() / scale:.2e}  {np.sqrt((d ** 2).mean() / (b.astype(np.float64) ** 2).mean()):.2e}")
# attention peakedness per step: largest score spread
