"""The drop-in path, timed: the reference's UNMODIFIED model.py (unpacked from the git-ignored archive oracle/_ref/refpy.tgz into a
temporary directory, as tests/test_reference_dropin_gpu.py does) builds ITS ExLlama on a synthetic Llama-7B GPTQ g128 checkpoint
with `import cuda_ext` resolving to this repository's shim, and runs the protocol of the reference's own benchmark
(test_benchmark_inference.py:155-197, `-p -l 2048`): two warm-up passes, a timed 1920-token prompt pass, then 128 greedy tokens
(torch.argmax on the host between forward passes) at full context and 128 more from context 4.  Device-synchronised wall times.

    python scripts/bench_dropin.py [--layers 32] [--out gpurun_out/dropin.json]

This is measurement infrastructure (it executes reference code): nothing under exllama_amd/ imports it."""
import argparse
import importlib
import json
import os
import sys
import tarfile
import tempfile
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from exllama_amd import synth   # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layers", type=int, default=32)
    ap.add_argument("--length", type=int, default=2048)
    ap.add_argument("--out", default=None)
    ap.add_argument("--profile", default=None, help="write a cProfile listing of 64 more decode steps (host time by function) to this file")
    args = ap.parse_args()
    archive = os.path.join(ROOT, "oracle", "_ref", "refpy.tgz")
    if not os.path.exists(archive):
        raise SystemExit("reference model.py not staged (scripts/stage_reference_py.sh)")
    work = tempfile.mkdtemp(prefix="dropin_")
    with tarfile.open(archive) as tf:
        tf.extractall(work)
    for p in (work, ROOT):                                     # ROOT first: `import cuda_ext` is this repository's shim
        if p in sys.path:
            sys.path.remove(p)
        sys.path.insert(0, p)
    ref = importlib.import_module("model")
    import cuda_ext
    assert os.path.samefile(os.path.dirname(ref.__file__), work) and os.path.samefile(os.path.dirname(cuda_ext.__file__), ROOT)

    dims = synth.PRESETS["7b"]
    t0 = time.time()
    cfg_path, st_path = synth.save_checkpoint(os.path.join(work, "ckpt"), dims, groupsize=128, act_order=False, seed=0, zeros="sym", nibbles="centered",
                                              num_layers=args.layers)
    t_ckpt = time.time() - t0
    cfg = ref.ExLlamaConfig(cfg_path)
    cfg.model_path = st_path
    cfg.max_seq_len = args.length
    model = ref.ExLlama(cfg)
    cache = ref.ExLlamaCache(model)
    gen_tokens = 128
    ids = torch.randint(0, 31999, (1, args.length - gen_tokens), generator=torch.Generator().manual_seed(0)).cuda()

    def sync():
        torch.cuda.synchronize()

    for _ in range(2):                                         # "Warming up apparently makes a huge difference" (reference comment)
        cache.current_seq_len = 0
        logits = model.forward(ids, cache, True)
    sync()
    cache.current_seq_len = 0
    t = time.time()
    logits = model.forward(ids, cache, True)
    sync()
    prefill_s = time.time() - t
    speeds = []
    for _ in range(2):
        sync()
        t = time.time()
        for _ in range(gen_tokens):
            token = torch.argmax(logits[0, -1, :])
            logits = model.forward(token.view(1, 1), cache, True)
        sync()
        speeds.append(gen_tokens / (time.time() - t))
        cache.current_seq_len = 4
    out = {"what": "the reference's unmodified model.py on the cuda_ext shim, the reference's own -p protocol (host argmax, op-by-op "
                   "fused ops q4_attn / attention / q4_attn_2 / q4_mlp, no hipGraph), Llama-7B GPTQ g128 synthetic, one MI355X",
           "layers": args.layers, "prompt_tokens": int(ids.shape[-1]),
           "prefill_tokens_per_s": round(ids.shape[-1] / prefill_s, 1), "prefill_ms": round(prefill_s * 1e3, 2),
           "decode_worst_tokens_per_s": round(speeds[0], 1), "decode_best_tokens_per_s": round(speeds[1], 1),
           "checkpoint_build_s": round(t_ckpt, 1)}
    print(json.dumps(out))
    if args.out:
        with open(args.out, "w") as f:
            json.dump(out, f, indent=1)
    if args.profile:                                           # where the HOST time of a token goes (the path is host-bound)
        import cProfile, io, pstats
        pr = cProfile.Profile()
        sync()
        pr.enable()
        for _ in range(64):
            token = torch.argmax(logits[0, -1, :])
            logits = model.forward(token.view(1, 1), cache, True)
        sync()
        pr.disable()
        buf = io.StringIO()
        pstats.Stats(pr, stream=buf).sort_stats("tottime").print_stats(30)
        with open(args.profile, "w") as f:
            f.write(buf.getvalue())
    model.free_unmanaged()


if __name__ == "__main__":
    main()
