"""TEST INFRASTRUCTURE.  Golden values for the host-side classes from the REFERENCE'S OWN code: imports /root/reference/model.py
in this container (CPU; `cuda_ext` stubbed -- ExLlamaConfig and ExLlamaDeviceMap are pure Python, model.py:39-127, :636-668) and
records every public attribute of a config built from a config.json, its rotary base after calculate_rotary_embedding_base, and
where the device map sends every kind of tensor key.

    python oracle/make_host_golden.py      ->  tests/golden/host_ref.json
"""
import importlib
import json
import os
import sys
import tempfile
import types

REF = os.environ.get("EXL_REFERENCE", "/root/reference")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CONFIGS = {
    "llama7b": {"bos_token_id": 1, "eos_token_id": 2, "pad_token_id": 0, "hidden_size": 4096, "initializer_range": 0.02, "intermediate_size": 11008,
                "num_attention_heads": 32, "num_hidden_layers": 32, "rms_norm_eps": 1e-06, "vocab_size": 32000},
    "llama2_70b": {"bos_token_id": 1, "eos_token_id": 2, "pad_token_id": 0, "hidden_size": 8192, "initializer_range": 0.02, "intermediate_size": 28672,
                   "num_attention_heads": 64, "num_key_value_heads": 8, "num_hidden_layers": 80, "rms_norm_eps": 1e-05, "vocab_size": 32000,
                   "rope_theta": 10000.0},
    "codellama": {"bos_token_id": 1, "eos_token_id": 2, "hidden_size": 5120, "initializer_range": 0.02, "intermediate_size": 13824,
                  "num_attention_heads": 40, "num_hidden_layers": 40, "rms_norm_eps": 1e-05, "vocab_size": 32016, "rope_theta": 1000000},
}
KEYS = ["lm_head.weight", "model.norm.weight", "model.embed_tokens.weight", "model.layers.0.input_layernorm.weight",
        "model.layers.3.self_attn.q_proj.qweight", "model.layers.3.mlp.down_proj.scales", "model.layers.7.post_attention_layernorm.weight",
        "model.layers.7.self_attn.o_proj.g_idx"]


def main():
    sys.path.insert(0, REF)
    sys.modules["cuda_ext"] = types.ModuleType("cuda_ext")
    ref = importlib.import_module("model")
    out = {"configs": {}, "device_map": {}}
    skip = {"device_map", "model_path", "auto_map"}
    for name, cfg in CONFIGS.items():
        with tempfile.NamedTemporaryFile("w", suffix=".json", delete=False) as f:
            json.dump(cfg, f)
        c = ref.ExLlamaConfig(f.name)
        os.unlink(f.name)
        attrs = {k: v for k, v in vars(c).items() if k not in skip and isinstance(v, (int, float, bool, str, type(None)))}
        c.alpha_value = 2.5
        c.calculate_rotary_embedding_base()
        attrs["rotary_embedding_base_after_alpha_2.5"] = c.rotary_embedding_base
        c.set_auto_map("10.5,24")
        attrs["auto_map_of_10.5,24"] = c.auto_map
        out["configs"][name] = attrs
    m = ref.ExLlamaDeviceMap(8)
    m.layers = ["cuda:0"] * 4 + ["cuda:1"] * 4
    m.norm = m.lm_head = "cuda:1"
    out["device_map"]["layers"] = m.layers
    out["device_map"]["map"] = {k: m.map(k) for k in KEYS}
    out["device_map"]["layers_devs"] = m.get_layers_devs()
    out["device_map"]["all_devs"] = m.get_all_devs()
    fresh = ref.ExLlamaDeviceMap(3)
    out["device_map"]["defaults"] = {"embed_tokens": fresh.embed_tokens, "lm_head": fresh.lm_head, "norm": fresh.norm, "layers": fresh.layers}
    path = os.path.join(ROOT, "tests", "golden", "host_ref.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("wrote", path)


if __name__ == "__main__":
    main()
