"""TEST INFRASTRUCTURE.  Golden values for the command-line -> config mapping from the REFERENCE'S OWN code: imports
/root/reference/model_init.py in this container (CPU; `cuda_ext` stubbed) and records, for a list of argument vectors, the
scalar attributes of the ExLlamaConfig its add_args / post_parse / get_model_files / make_config produce (model_init.py:7-160).

    python oracle/make_init_golden.py      ->  tests/golden/init_ref.json
"""
import argparse
import importlib
import json
import os
import sys
import tempfile
import types

REF = os.environ.get("EXL_REFERENCE", "/root/reference")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CONFIG = {"bos_token_id": 1, "eos_token_id": 2, "pad_token_id": 0, "hidden_size": 4096, "initializer_range": 0.02, "intermediate_size": 11008,
          "num_attention_heads": 32, "num_hidden_layers": 32, "rms_norm_eps": 1e-06, "vocab_size": 32000}
# {DIR} = a directory holding config.json, tokenizer.model and ONE model.safetensors
ARGVS = [
    ["-d", "{DIR}"],
    ["-d", "{DIR}", "-l", "4096", "-cpe", "2.0"],
    ["-d", "{DIR}", "-a", "2.5", "-gs", "20,7.5,7"],
    ["-d", "{DIR}", "-theta", "500000"],
    ["-d", "{DIR}", "-mmrt", "16", "-fmt", "0", "-sdpt", "1", "-mmfr", "-nfa"],
    ["-d", "{DIR}", "-flash"],
    ["-d", "{DIR}", "-flash", "1024"],
    ["-d", "{DIR}", "-fh2"],
    ["-d", "{DIR}", "-fh2", "-rnnh2", "-snh2"],
    ["-d", "{DIR}", "-nh2", "-cs", "-gpfix"],
    ["-t", "{DIR}/tokenizer.model", "-c", "{DIR}/config.json", "-m", "{DIR}/model.safetensors", "-l", "1024"],
]


def make_dir(d):
    os.makedirs(d, exist_ok=True)
    with open(os.path.join(d, "config.json"), "w") as f:
        json.dump(CONFIG, f)
    for name in ("tokenizer.model", "model.safetensors"):
        open(os.path.join(d, name), "wb").close()


def scalars(c, d):
    out = {}
    for k, v in vars(c).items():
        if k == "device_map":
            continue
        if isinstance(v, str):
            v = v.replace(d, "{DIR}")
        if isinstance(v, list):
            v = [x.replace(d, "{DIR}") if isinstance(x, str) else x for x in v]
        if isinstance(v, (int, float, bool, str, type(None), list)):
            out[k] = v
    return out


def main():
    sys.path.insert(0, REF)
    sys.modules["cuda_ext"] = types.ModuleType("cuda_ext")
    mi = importlib.import_module("model_init")
    out = []
    with tempfile.TemporaryDirectory() as d:
        make_dir(d)
        for argv in ARGVS:
            parser = argparse.ArgumentParser()
            mi.add_args(parser)
            args = parser.parse_args([a.replace("{DIR}", d) for a in argv])
            mi.post_parse(args)
            mi.get_model_files(args)
            out.append({"argv": argv, "config": scalars(mi.make_config(args), d)})
    path = os.path.join(ROOT, "tests", "golden", "init_ref.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("wrote", path, len(out), "cases")


if __name__ == "__main__":
    main()
