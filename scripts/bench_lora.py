"""N3 with numbers: what a LoRA adapter on all 7 projections costs on Llama-7B GPTQ g128 (synthetic), one MI355X.

    python scripts/bench_lora.py [--layers 32] [--out gpurun_out/lora.json]

Per rank (16, 64): 2048-token prefill tokens/s and single-token decode tokens/s at context 2048 with the adapter (q4_matmul_lora in the
prompt pass; q4_attn / q4_attn_2 / q4_mlp with LoRA operands per token -- the reference's path, exllama_ext.cpp:245-324, :424-602),
next to the same model without an adapter on the same op-by-op path and on the native executor (which has no LoRA operands)."""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from exllama_amd import synth                                         # noqa: E402
from exllama_amd.lora import ExLlamaLora                              # noqa: E402
from exllama_amd.model import ExLlama, ExLlamaCache, ExLlamaConfig    # noqa: E402


def adapter(dims, L, r, seed=0):
    g = torch.Generator().manual_seed(seed)
    h, I, kvd = dims.hidden_size, dims.intermediate_size, dims.num_key_value_heads * dims.head_dim
    shapes = {"self_attn.q_proj": (h, h), "self_attn.k_proj": (h, kvd), "self_attn.v_proj": (h, kvd), "self_attn.o_proj": (h, h),
              "mlp.gate_proj": (h, I), "mlp.up_proj": (h, I), "mlp.down_proj": (I, h)}
    t = {}
    for i in range(L):
        for name, (k, n) in shapes.items():
            t[f"base_model.model.model.layers.{i}.{name}.lora_A.weight"] = (torch.randn(r, k, generator=g) * 0.02).half()
            t[f"base_model.model.model.layers.{i}.{name}.lora_B.weight"] = (torch.randn(n, r, generator=g) * 0.02).half()
    return t


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layers", type=int, default=32)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    dims, L, S, G = synth.PRESETS["7b"], a.layers, 2048, 64
    dev = "cuda:0"
    tensors = synth.make_checkpoint(dims, groupsize=128, act_order=False, seed=0, device=dev, zeros="sym", num_layers=L, nibbles="centered")
    cfg = ExLlamaConfig(synth.config_dict(dims, L))
    cfg.max_seq_len, cfg.max_input_len = S + G + 8, S
    model = ExLlama(cfg, tensors=tensors)
    del tensors
    cache = ExLlamaCache(model)
    ids = torch.randint(0, 31999, (1, S), device=dev, generator=torch.Generator(device=dev).manual_seed(1))

    def run(lora):
        for _ in range(2):
            cache.current_seq_len = 0
            lg = model.forward(ids, cache, lora=lora)
        torch.cuda.synchronize()
        cache.current_seq_len = 0
        t = time.perf_counter()
        lg = model.forward(ids, cache, lora=lora)
        torch.cuda.synchronize()
        pre = time.perf_counter() - t
        for _ in range(8):
            lg = model.forward(lg[0, -1].argmax().view(1, 1), cache, lora=lora)
        cache.current_seq_len = S
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(G):
            lg = model.forward(lg[0, -1].argmax().view(1, 1), cache, lora=lora)
        torch.cuda.synchronize()
        return {"prefill_tokens_per_s": round(S / pre, 1), "decode_tokens_per_s": round(G / (time.perf_counter() - t), 2)}

    res = {"what": __doc__.split("\n")[0], "layers": L, "prompt_tokens": S, "decode_context": S, "decode_tokens": G}
    res["no_adapter_op_by_op"] = run(None)
    for r in (16, 64):
        lora = ExLlamaLora(model, {"r": r, "lora_alpha": 2 * r}, None, tensors=adapter(dims, L, r))
        res[f"rank_{r}_all_7_projections"] = run(lora)                       # op-by-op: q4_attn / q4_attn_2 / q4_mlp with LoRA operands per token
        model.enable_decode_graph(cache, lora=lora)                           # the adapter inside the executor's graph (exl_decoder_set_lora)
        res[f"rank_{r}_all_7_projections_executor_graph"] = run(lora)
        model.disable_decode_graph()
        del lora
    model.enable_decode_graph(cache)
    res["no_adapter_executor_graph"] = run(None)
    base = res["no_adapter_op_by_op"]["decode_tokens_per_s"]
    ex = res["no_adapter_executor_graph"]["decode_tokens_per_s"]
    for r in (16, 64):
        d = res[f"rank_{r}_all_7_projections"]
        d["decode_vs_no_adapter_same_path"] = round(d["decode_tokens_per_s"] / base, 3)
        g = res[f"rank_{r}_all_7_projections_executor_graph"]
        g["decode_vs_no_adapter_same_path"] = round(g["decode_tokens_per_s"] / ex, 3)
    line = json.dumps(res)
    print(line)
    if a.out:
        with open(a.out, "w") as f:
            f.write(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
