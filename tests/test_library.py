"""CPU tests of the C-ABI library: it loads, exports every symbol include/exl_amd.h declares, follows the error
convention, and its host-side functions (repetition penalty) are bit-exact against the reference binary's vectors.
No GPU compute is launched here."""
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from exllama_amd import _lib
    return _lib.load()


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "exl_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(exl_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported(lib):
    from exllama_amd import _lib
    names = _declared_symbols()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/exl_amd.h but not exported by libexl_amd.so"
        assert n in _lib.SIGNATURES, f"{n} has no ctypes signature in exllama_amd/_lib.py"
    assert lib.exl_version() >= 100


def test_op_surface_names_match_reference():
    """The names the reference's model.py / generator.py use on `cuda_ext`."""
    from exllama_amd import cuda_ext
    for n in ("ext_make_q4", "ext_q4_matmul", "ext_half_matmul", "ext_rope_", "ext_rms_norm", "ext_rms_norm_",
              "ext_rep_penalty_mask_cpu", "ext_apply_rep_penalty_mask_cpu", "none_tensor", "exllama_ext"):
        assert hasattr(cuda_ext, n)
    for n in ("set_tuning_params", "prepare_buffers", "cleanup", "make_q4", "q4_matmul", "q4_matmul_lora", "q4_attn",
              "q4_attn_2", "q4_mlp", "column_remap", "rms_norm", "rope_", "half_matmul", "half_matmul_cublas",
              "rep_penalty", "apply_rep_penalty"):
        assert callable(getattr(cuda_ext.exllama_ext, n)), n
    assert cuda_ext.none_tensor.device.type == "meta"
    import cuda_ext as shim                      # the root-level drop-in module
    assert shim.exllama_ext is cuda_ext.exllama_ext


def test_error_convention(lib):
    from exllama_amd import cuda_ext
    with pytest.raises(RuntimeError, match="invalid"):
        cuda_ext.exllama_ext.q4_info(12345678)            # not a handle
    x = torch.zeros((1, 8), dtype=torch.float32)
    with pytest.raises(RuntimeError, match="incorrect datatype"):
        cuda_ext.exllama_ext.rms_norm(x, x, x, 1e-6)
    xh = torch.zeros((1, 8), dtype=torch.float16)
    with pytest.raises(RuntimeError, match="HIP device"):
        cuda_ext.exllama_ext.rms_norm(xh, xh[0], xh, 1e-6)       # CPU tensors: no fallback path
    t = cuda_ext._lib.ExlTuning()
    cuda_ext.exllama_ext.set_tuning_params(4, 2, 8, False, True, True, True, True, False)
    assert lib.exl_get_tuning(t) == 0 and t.matmul_recons_thd == 4 and t.rope_no_half2 == 1
    cuda_ext.exllama_ext.set_tuning_params(8, 2, 8, False, False, False, False, False, False)


def test_model_refuses_to_run_without_gpu():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from exllama_amd import synth
    from exllama_amd.model import ExLlama, ExLlamaConfig
    cfg = ExLlamaConfig(synth.config_dict(synth.LLAMA_TINY))
    with pytest.raises(RuntimeError, match="no CPU execution path"):
        ExLlama(cfg, tensors={})


def test_rep_penalty_bit_exact_vs_reference_vectors(golden_dir):
    from exllama_amd import cuda_ext
    g = np.load(os.path.join(golden_dir, "rep_penalty.npz"))
    for n in range(int(g["ncases"])):
        vocab, seq_len, sustain, decay = (int(v) for v in g[f"c{n}_params"])
        pmax = float(g[f"c{n}_pmax"])
        seq = torch.from_numpy(g[f"c{n}_seq"].astype(np.int64))[None]
        mask = cuda_ext.ext_rep_penalty_mask_cpu(vocab, seq, pmax, sustain, decay)
        assert np.array_equal(mask.numpy().view(np.uint32), g[f"c{n}_mask"].view(np.uint32)), n
        logits = torch.from_numpy(g[f"c{n}_logits"].copy())[None]
        cuda_ext.ext_apply_rep_penalty_mask_cpu(seq, pmax, sustain, decay, logits)
        assert np.array_equal(logits[0].numpy().view(np.uint32), g[f"c{n}_applied"].view(np.uint32)), n


def test_rep_penalty_batch_and_bounds():
    from exllama_amd import cuda_ext
    seq = torch.tensor([[1, 2, 3], [3, 3, 0]], dtype=torch.long)
    logits = torch.tensor([[1.0, -1.0, 2.0, -2.0], [1.0, -1.0, 2.0, -2.0]])
    cuda_ext.ext_apply_rep_penalty_mask_cpu(seq, 2.0, -1, 0, logits)
    assert logits.tolist() == [[1.0, -2.0, 1.0, -4.0], [0.5, -1.0, 2.0, -4.0]]
    with pytest.raises(RuntimeError, match="outside the vocabulary"):
        cuda_ext.ext_rep_penalty_mask_cpu(2, torch.tensor([[5]], dtype=torch.long), 1.2, -1, 0)


REFERENCE = "/root/reference"


@pytest.mark.skipif(not os.path.isdir(REFERENCE), reason="reference checkout not present (GPU box)")
def test_every_reference_call_site_binds():
    """Drop-in check at the API level: every call the reference's own model.py / generator.py / alt_generator.py make into
    `cuda_ext` and `cuda_ext.exllama_ext` is parsed out of their source and bound (argument count and keyword names) against
    the signature of the function this repository exports under the same name."""
    import ast
    import inspect
    from exllama_amd import cuda_ext

    def dotted(node):
        parts = []
        while isinstance(node, ast.Attribute):
            parts.append(node.attr)
            node = node.value
        if isinstance(node, ast.Name):
            parts.append(node.id)
        return ".".join(reversed(parts))

    seen = set()
    for fname in ("model.py", "generator.py", "alt_generator.py"):
        tree = ast.parse(open(os.path.join(REFERENCE, fname)).read())
        for node in ast.walk(tree):
            if not isinstance(node, ast.Call):
                continue
            name = dotted(node.func)
            if name.startswith("cuda_ext.exllama_ext."):
                target = getattr(cuda_ext.exllama_ext, name.split(".")[-1], None)
            elif name.startswith("cuda_ext."):
                target = getattr(cuda_ext, name.split(".")[-1], None)
            else:
                continue
            assert callable(target), f"{fname}:{node.lineno}: {name} is not provided"
            sig = inspect.signature(target)
            try:
                sig.bind(*[None] * len(node.args), **{k.arg: None for k in node.keywords})
            except TypeError as e:
                raise AssertionError(f"{fname}:{node.lineno}: {name}(...) does not bind: {e}")
            seen.add(name)
    # the fused decode ops, the loader and the sampler hooks are all among them
    for must in ("cuda_ext.exllama_ext.q4_attn", "cuda_ext.exllama_ext.q4_attn_2", "cuda_ext.exllama_ext.q4_mlp",
                 "cuda_ext.exllama_ext.prepare_buffers", "cuda_ext.ext_make_q4", "cuda_ext.ext_q4_matmul",
                 "cuda_ext.ext_apply_rep_penalty_mask_cpu"):
        assert must in seen, must


@pytest.mark.skipif(not os.path.isdir(REFERENCE), reason="reference checkout not present (GPU box)")
def test_reference_generator_and_harness_calls_bind_to_model_api():
    """Same check one level up: what the reference's generator.py / alt_generator.py / perplexity.py /
    test_benchmark_inference.py call on the model and the cache (`self.model.forward(...)`, `cache.copy_states(...)`, ...) binds
    to exllama_amd.model's classes, and the config fields they read exist."""
    import ast
    import inspect
    from exllama_amd.model import ExLlama, ExLlamaCache, ExLlamaConfig
    from exllama_amd import synth

    def dotted(node):
        parts = []
        while isinstance(node, ast.Attribute):
            parts.append(node.attr)
            node = node.value
        if isinstance(node, ast.Name):
            parts.append(node.id)
        return ".".join(reversed(parts))

    cfg = ExLlamaConfig(synth.config_dict(synth.LLAMA_TINY))
    owners = {"model": ExLlama, "cache": ExLlamaCache}
    calls, fields = set(), set()
    for fname in ("generator.py", "alt_generator.py", "perplexity.py", "test_benchmark_inference.py"):
        tree = ast.parse(open(os.path.join(REFERENCE, fname)).read())
        for node in ast.walk(tree):
            if isinstance(node, ast.Call):
                parts = dotted(node.func).split(".")
                if parts[0] == "self":
                    parts = parts[1:]
                if len(parts) == 2 and parts[0] in owners:
                    fn = getattr(owners[parts[0]], parts[1], None)
                    assert callable(fn), f"{fname}:{node.lineno}: {'.'.join(parts)} is not provided"
                    try:
                        inspect.signature(fn).bind(None, *[None] * len(node.args), **{k.arg: None for k in node.keywords})
                    except TypeError as e:
                        raise AssertionError(f"{fname}:{node.lineno}: {'.'.join(parts)}(...) does not bind: {e}")
                    calls.add(".".join(parts))
            elif isinstance(node, ast.Attribute):
                parts = dotted(node).split(".")
                if "config" in parts[:-1] and parts[-2] == "config":
                    assert hasattr(cfg, parts[-1]), f"{fname}:{node.lineno}: config.{parts[-1]} missing"
                    fields.add(parts[-1])
    assert {"model.forward", "cache.clone", "cache.roll_left", "cache.copy_states"} <= calls, calls
    assert {"max_seq_len", "vocab_size", "matmul_recons_thd"} <= fields, fields
    # constructor forms used by the reference: ExLlamaCache(model), ExLlamaCache(model, batch_size = n), ExLlamaCache(model, copy_from = c)
    sig = inspect.signature(ExLlamaCache.__init__)
    for kw in ({}, {"batch_size": 2}, {"copy_from": None}, {"max_seq_len": 8}):
        sig.bind(None, None, **kw)


@pytest.mark.skipif(not os.path.isdir(REFERENCE), reason="reference checkout not present (GPU box)")
def test_reference_modules_import_against_the_shim():
    """With this repository ahead of the reference on PYTHONPATH, the reference's OWN model.py, generator.py, lora.py,
    perplexity.py and model_init.py import (their module-level `import cuda_ext` resolves to our shim) and their config /
    argument plumbing runs; only constructing the model needs a HIP device."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import json, tempfile, os, argparse\n"
        "import cuda_ext, model, lora, generator, perplexity, model_init\n"
        "assert cuda_ext.__file__.startswith(%r) and model.__file__.startswith(%r)\n"
        "from exllama_amd import synth\n"
        "d = tempfile.mkdtemp(); p = os.path.join(d, 'config.json')\n"
        "json.dump(synth.config_dict(synth.LLAMA_TINY), open(p, 'w'))\n"
        "ap = argparse.ArgumentParser(); model_init.add_args(ap); perplexity.add_args(ap)\n"
        "a = ap.parse_args(['-t', 'tok', '-c', p, '-m', 'w.safetensors', '-l', '512']); model_init.post_parse(a)\n"
        "c = model_init.make_config(a)\n"
        "assert c.max_seq_len == 512 and c.hidden_size == synth.LLAMA_TINY.hidden_size\n"
        "c.set_tuning_params()\n"                       # reference code path: cuda_ext.exllama_ext.set_tuning_params(...) -> exl_set_tuning
        "print('OK')\n" % (root, REFERENCE))
    env = dict(os.environ, PYTHONPATH=root + os.pathsep + REFERENCE)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith("OK"), r.stderr[-2000:]


def test_fused_ops_name_the_offending_tensor_without_a_gpu():
    """The shim checks device / contiguity of the nine tensors of q4_attn in one sweep and, when the sweep fails, once more one by one to
    NAME the offender (exllama_amd/cuda_ext.py) -- the reference's TORCH_CHECK messages name theirs (exllama_ext.cpp:53-75).  No kernel
    runs: the check fails first."""
    import torch
    from exllama_amd import cuda_ext
    ext = cuda_ext.exllama_ext
    h = torch.zeros((1, 1, 64), dtype=torch.float16)
    w = torch.zeros((64,), dtype=torch.float16)
    none = cuda_ext.none_tensor
    assert cuda_ext._is_none(none) and cuda_ext._is_none(None) and not cuda_ext._is_none(h) and cuda_ext._ptr(none) is None
    with pytest.raises(RuntimeError, match="x must be on a HIP device"):
        ext.q4_attn(h, w, 1e-6, h, h, h, 0, 0, 0, w, w, 1, 0, 1, 1, 64, h, h, 16, none, none, none, none, none, none, none)
    with pytest.raises(RuntimeError, match="x must be on a HIP device"):
        ext.q4_attn_2(h, h, 0, none, none, none)
    with pytest.raises(RuntimeError, match="x must be on a HIP device"):
        ext.q4_mlp(h.view(1, 64), w, 1e-6, 0, 0, 0, none, none, none, none, none, none, none)
    with pytest.raises(RuntimeError, match="incorrect datatype"):
        ext.q4_attn(h, w, 1e-6, h.float(), h, h, 0, 0, 0, w, w, 1, 0, 1, 1, 64, h, h, 16, none, none, none, none, none, none, none)


def test_compiled_binding_mirrors_the_ctypes_methods():
    """exllama_amd/_exl_fast.so (csrc/binding/exl_fast.cpp) replaces the per-token methods of cuda_ext.exllama_ext: it must load, be
    what the module dispatches to, and expose the same parameter lists (names, order, defaults) as the ctypes methods it replaces --
    they stay in the class as the A/B reference."""
    import inspect
    from exllama_amd import cuda_ext
    fast = cuda_ext.FAST_BINDING
    assert fast is not None, "the compiled binding is not active (EXL_NO_FAST_BINDING set?)"
    for name in ("q4_matmul", "rms_norm", "rope_", "q4_attn", "q4_attn_2", "q4_mlp", "attention"):
        f = getattr(fast, name)
        assert getattr(cuda_ext.exllama_ext, name) is f
        py = inspect.signature(getattr(type(cuda_ext.exllama_ext), name))
        want = [(p.name, p.default) for p in list(py.parameters.values())[1:]]          # without self
        got = [(p.name, p.default) for p in inspect.signature(f).parameters.values()]
        assert got == want, (name, got, want)
    assert cuda_ext.rms_norm is fast.rms_norm and cuda_ext.q4_matmul is fast.q4_matmul and cuda_ext.rope_ is fast.rope_


def test_compiled_binding_raises_what_the_ctypes_path_raises():
    """Same RuntimeError / TypeError behaviour on bad arguments, checked without a GPU (every check precedes the native call)."""
    import torch
    from exllama_amd import cuda_ext
    fast, ext_cls = cuda_ext.FAST_BINDING, type(cuda_ext.exllama_ext)
    slow = cuda_ext.exllama_ext
    h = torch.zeros((2, 64), dtype=torch.float16)
    none = cuda_ext.none_tensor
    cases = [
        ("rms_norm", (h, h[0], h, 1e-6)),                      # CPU tensors
        ("rms_norm", (h.float(), h[0], h, 1e-6)),              # wrong dtype
        ("rope_", (h, h, h, 0, 1, 64)),
        ("q4_matmul", (h, 0, h)),
        ("q4_attn_2", (h, h, 0, none, none, none)),
        ("q4_mlp", (h, h[0], 1e-6, 0, 0, 0, none, none, none, none, none, none, none)),
        ("attention", (h.view(1, 2, 64), h.view(1, 1, 2, 64), h.view(1, 1, 2, 64), h.view(1, 2, 64), 0, 1)),
    ]
    for name, args in cases:
        errs = []
        for fn in (getattr(fast, name), lambda *a, _n=name: getattr(ext_cls, _n)(slow, *a)):
            with pytest.raises(RuntimeError) as e:
                fn(*args)
            errs.append(str(e.value))
        key = "incorrect datatype" if "incorrect datatype" in errs[1] else "HIP device" if "HIP device" in errs[1] else "handle"
        assert key in errs[0] and key in errs[1], (name, errs)
    with pytest.raises(TypeError):
        fast.rms_norm(h, h)
    with pytest.raises(TypeError):
        fast.attention(h, h, h, h, 0, 1, bogus=1)


def test_fragment_order_is_what_the_header_says(lib):
    """include/exl_amd.h (exl_q4_matmul_frag): out_frag[((mt * (K / 32) + 4 rb + j) * 64 + lane) * 16 .. + 16] = act[16 mt + (lane & 15)][128 rb +
    32 (lane >> 4) + 8 j .. + 8]; exl_frag_bytes pads the rows to 64, from 65 rows on to a multiple of 128.  cuda_ext.unfrag is that
    formula backwards (the GPU tests read fragment-order outputs through it): held here against the formula itself, on the host."""
    from exllama_amd import cuda_ext
    assert lib.exl_frag_bytes(1, 128) == 64 * 128 * 2 and lib.exl_frag_bytes(64, 256) == 64 * 256 * 2
    assert lib.exl_frag_bytes(65, 128) == 128 * 128 * 2 and lib.exl_frag_bytes(129, 128) == 256 * 128 * 2 and lib.exl_frag_bytes(0, 128) == 0
    rows, K = 70, 384
    pad = lib.exl_frag_bytes(rows, K) // (2 * K)
    act = np.zeros((pad, K), dtype=np.float16)
    act[:rows] = np.random.RandomState(3).randn(rows, K).astype(np.float16)
    frag = np.zeros(pad * K, dtype=np.float16)
    for mt in range(pad // 16):
        for rb in range(K // 128):
            for j in range(4):
                for lane in range(64):
                    at = ((mt * (K // 32) + 4 * rb + j) * 64 + lane) * 8
                    k0 = 128 * rb + 32 * (lane >> 4) + 8 * j
                    frag[at:at + 8] = act[16 * mt + (lane & 15), k0:k0 + 8]
    back = cuda_ext._ExllamaExt.unfrag(torch.from_numpy(frag).view(torch.uint8), rows, K)
    assert back.shape == (rows, K) and np.array_equal(back.numpy(), act[:rows])
