"""TEST INFRASTRUCTURE.  Golden values for the LoRA adapter loader from the REFERENCE'S OWN code: imports /root/reference/lora.py in
this container (CPU; `cuda_ext` stubbed, the target modules are bare instances of the reference's Ex4bitLinear with the shapes
set) and records what ExLlamaLora.__init__ (lora.py:18-125) makes of a seeded synthetic adapter: r / alpha / scaling, which bias
was ignored, and for every tensor its key, shape, dtype and SHA-256 (transposed, B pre-scaled by alpha / r, converted to fp16).

    python oracle/make_lora_golden.py      ->  tests/golden/lora_ref.json
"""
import hashlib
import importlib
import json
import os
import sys
import tempfile
import types

import torch

REF = os.environ.get("EXL_REFERENCE", "/root/reference")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
H, INTER, LAYERS, R, ALPHA = 32, 48, 3, 4, 6


def adapter_state():
    """The synthetic adapter (seeded): fp32, bf16 and fp16 tensors, attention and MLP targets, one all-zero bias."""
    g = torch.Generator().manual_seed(123)
    pre = "base_model.model.model.layers."
    sd = {}
    for i, (part, name, k, n, dt) in enumerate([("self_attn", "q_proj", H, H, torch.float32), ("self_attn", "v_proj", H, H, torch.bfloat16),
                                                  ("self_attn", "o_proj", H, H, torch.float16), ("mlp", "gate_proj", H, INTER, torch.float32),
                                                  ("mlp", "down_proj", INTER, H, torch.bfloat16)]):
        layer = i % LAYERS
        sd[f"{pre}{layer}.{part}.{name}.lora_A.weight"] = torch.randn(R, k, generator=g).to(dt)
        sd[f"{pre}{layer}.{part}.{name}.lora_B.weight"] = torch.randn(n, R, generator=g).to(dt)
    sd[f"{pre}1.mlp.down_proj.bias"] = torch.zeros(H)
    return sd


def digest(t):
    return hashlib.sha256(t.contiguous().view(torch.uint8).numpy().tobytes()).hexdigest()


def describe(lora):
    return {"lora_r": lora.lora_r, "lora_alpha": lora.lora_alpha, "lora_scaling": lora.lora_scaling, "bias_ignored": lora.bias_ignored,
            "tensors": {k: {"shape": list(v.shape), "dtype": str(v.dtype), "sha256": digest(v)} for k, v in sorted(lora.tensors.items())}}


def main():
    sys.path.insert(0, REF)
    sys.modules["cuda_ext"] = types.ModuleType("cuda_ext")
    ref_model = importlib.import_module("model")
    ref_lora = importlib.import_module("lora")

    def lin(k, n):
        m = ref_model.Ex4bitLinear.__new__(ref_model.Ex4bitLinear)
        m.in_features, m.out_features = k, n
        return m

    class Blk:
        pass
    model = Blk()
    model.config = Blk()
    model.config.device_map = ref_model.ExLlamaDeviceMap(LAYERS)
    model.config.device_map.layers = ["cpu"] * LAYERS
    model.layers = []
    for _ in range(LAYERS):
        l = Blk(); l.self_attn = Blk(); l.mlp = Blk()
        for n in ("q_proj", "k_proj", "v_proj", "o_proj"):
            setattr(l.self_attn, n, lin(H, H))
        l.mlp.gate_proj, l.mlp.up_proj, l.mlp.down_proj = lin(H, INTER), lin(H, INTER), lin(INTER, H)
        model.layers.append(l)
    from safetensors.torch import save_file
    with tempfile.TemporaryDirectory() as d:
        cfg = os.path.join(d, "adapter_config.json")
        with open(cfg, "w") as f:
            json.dump({"r": R, "lora_alpha": ALPHA, "fan_in_fan_out": False}, f)
        st = os.path.join(d, "adapter_model.safetensors")
        save_file({k: v.contiguous() for k, v in adapter_state().items()}, st)
        out = describe(ref_lora.ExLlamaLora(model, cfg, st))
    path = os.path.join(ROOT, "tests", "golden", "lora_ref.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("wrote", path, len(out["tensors"]), "tensors")


if __name__ == "__main__":
    main()
