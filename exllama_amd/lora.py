"""PEFT LoRA adapter loader for the q4 linears of exllama_amd.model (mirror of the reference's lora.py:7-125).

What `Ex4bitLinear.forward(..., lora=...)` and the fused ops consume is `lora.tensors`: a dict keyed
"model.layers.<i>.<self_attn|mlp>.<proj>.lora_A.weight" / "...lora_B.weight" holding the adapter halves TRANSPOSED
(A: [in_features, r], B: [r, out_features]) in fp16 on the device of the target layer, with alpha / r already folded
into B -- so the run-time product is x @ A @ B with no further scaling (exl_q4_matmul_lora, include/exl_amd.h).
"""
import json

import torch

_ATTN = ("q_proj", "k_proj", "v_proj", "o_proj")
_MLP = ("gate_proj", "up_proj", "down_proj")


class ExLlamaLora:
    """tensors / lora_r / lora_alpha / lora_scaling / bias_ignored as in the reference (lora.py:9-16)."""

    def __init__(self, model, lora_config_path, lora_path, tensors=None):
        """`tensors`: optional in-memory state dict instead of reading `lora_path` (tests, synthetic adapters)."""
        self.lora_config_path = lora_config_path
        self.lora_path = lora_path
        self.model = model
        self.config = model.config
        self.tensors = {}
        self.bias_ignored = False

        if isinstance(lora_config_path, dict):
            cfg = lora_config_path
        else:
            with open(lora_config_path) as f:
                cfg = json.load(f)
        self.lora_r = cfg["r"]
        self.lora_alpha = float(cfg["lora_alpha"])
        self.lora_scaling = self.lora_alpha / self.lora_r
        if cfg.get("fan_in_fan_out"):
            raise ValueError(" ## Error: fan_in_fan_out mode not supported.")

        if tensors is None:
            if str(lora_path).endswith(".safetensors"):
                from safetensors.torch import load_file
                tensors = load_file(lora_path, device="cpu")
            else:
                tensors = torch.load(lora_path, map_location="cpu")

        for key, tensor in tensors.items():
            self._add(key, tensor)

    def _target(self, key):
        """'…model.layers.3.mlp.up_proj.lora_A.weight' -> (normalised key, Ex4bitLinear, 'lora_A')."""
        at = key.find("model.layers.")
        if at < 0:
            raise ValueError(f" ## Error: unsupported layer in {self.lora_path}: {key}")
        norm = key[at:]
        parts = norm.split(".")
        if len(parts) < 6:
            raise ValueError(f" ## Error: unsupported layer in {self.lora_path}: {key}")
        index, block, proj, half = int(parts[2]), parts[3], parts[4], parts[5]
        legal = _ATTN if block == "self_attn" else _MLP if block == "mlp" else ()
        if proj not in legal or not 0 <= index < len(self.model.layers):
            raise ValueError(f" ## Error: unsupported layer in {self.lora_path}: {key}")
        return norm, getattr(getattr(self.model.layers[index], block), proj), half

    def _add(self, key, tensor):
        norm, linear, half = self._target(key)
        if half == "bias":                                   # PEFT may save all-zero biases: ignore those, reject real ones
            if float(tensor.abs().max()) > 1e-6:
                raise ValueError(f" ## Error: unsupported bias target {self.lora_path}: {key}")
            self.bias_ignored = True
            return
        if half == "lora_A":
            ok = tensor.shape[1] == linear.in_features
        elif half == "lora_B":
            ok = tensor.shape[0] == linear.out_features
        else:
            raise ValueError(f" ## Error: unsupported layer in {self.lora_path}: {key}")
        if not ok:
            raise ValueError(f" ## Error: incompatible tensor shape in {self.lora_path}: {key}")
        if tensor.dtype not in (torch.float16, torch.bfloat16, torch.float32):
            raise ValueError(f" ## Error: unsupported tensor dtype in {self.lora_path}")

        t = tensor.T.contiguous()                            # x @ A @ B at run time: store both halves transposed
        if half == "lora_B" and self.lora_scaling != 1.0:
            t = t * self.lora_scaling                        # in the source dtype, then one rounding to fp16 (lora.py:104-115)
        t = t.to(torch.float16)
        # an act-order down_proj folded into its producers at load (exllama_amd.model._fold_act_order_down_proj): the intermediate
        # activations travel in down_proj's row order, so the adapter halves that touch them are put into the same order
        parts = norm.split(".")
        if parts[3] == "mlp":
            fold = getattr(self.model.layers[int(parts[2])].mlp, "fold_map", None)
            if fold is not None:
                if parts[4] in ("gate_proj", "up_proj") and half == "lora_B":
                    t = t[:, fold].contiguous()
                elif parts[4] == "down_proj" and half == "lora_A":
                    t = t[fold, :].contiguous()
        self.tensors[norm] = t.to(self.config.device_map.map(norm), non_blocking=True)
