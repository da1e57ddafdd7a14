"""One 2048-token prompt pass of a truncated Llama (7B shapes by default) -- the workload of the prefill PMC passes of
scripts/gpu_r04_profiles.sh (rocprofv3 --pmc ... -- python scripts/prefill_once.py): few launches, every prefill kernel once per layer."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from exllama_amd import synth                                      # noqa: E402
from exllama_amd.model import ExLlama, ExLlamaCache, ExLlamaConfig  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="7b")
ap.add_argument("--layers", type=int, default=4)
ap.add_argument("--act-order", default="")
ap.add_argument("--reps", type=int, default=2)
ap.add_argument("--prompt", type=int, default=2048)
ap.add_argument("--time", action="store_true", help="print the wall time per pass (events) over --reps passes after one warm-up")
a = ap.parse_args()
dims = synth.PRESETS[a.model]
act = {"": False, "gptq": "gptq", "independent": True}[a.act_order]
t = synth.make_checkpoint(dims, groupsize=128, act_order=act, seed=0, device="cuda:0", zeros="sym", num_layers=a.layers, nibbles="centered")
cfg = ExLlamaConfig(synth.config_dict(dims, a.layers))
cfg.max_seq_len = 2048 + 8
cfg.max_input_len = 2048
m = ExLlama(cfg, tensors=t)
c = ExLlamaCache(m)
ids = torch.randint(0, 31999, (1, a.prompt), device="cuda:0")
for _ in range(a.reps):
    c.current_seq_len = 0
    m.forward(ids, c)
torch.cuda.synchronize()
if a.time:
    import time
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(a.reps):
        c.current_seq_len = 0
        m.forward(ids, c)
    e1.record()
    t1 = time.perf_counter()                                       # host time to ENQUEUE the passes (the stream may still be running)
    torch.cuda.synchronize()
    print("prompt %d tokens, %d layers: %.1f us per layer and pass on the stream, %.1f us per layer of host time to enqueue"
          % (a.prompt, a.layers, e0.elapsed_time(e1) * 1e3 / a.reps / a.layers, (t1 - t0) * 1e6 / a.reps / a.layers))
print("ok")
