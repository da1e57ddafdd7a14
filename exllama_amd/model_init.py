"""Command-line surface -> ExLlamaConfig (mirror of the reference's model_init.py; flags, defaults and meanings kept so the
reference's launch lines work unchanged).  Flags that only select CUDA code paths (-rnnh2 … -fh2, -cs, -flash) are accepted
and recorded on the config for parity; the HIP kernels have a single code path, so they change nothing here."""
import glob
import os
import sys

from .model import ExLlamaConfig

# (short, long, kwargs) -- reference: model_init.py:7-38
_FLAGS = [
    ("-t", "--tokenizer", dict(type=str, help="Tokenizer model path")),
    ("-c", "--config", dict(type=str, help="Model config path (config.json)")),
    ("-m", "--model", dict(type=str, help="Model weights path (.pt or .safetensors file)")),
    ("-d", "--directory", dict(type=str, help="Path to directory containing config.json, model.tokenizer and * .safetensors")),
    ("-gs", "--gpu_split", dict(type=str, help="Comma-separated list of VRAM (in GB) to use per GPU device for model layers, e.g. -gs 20,7,7")),
    ("-l", "--length", dict(type=int, default=2048, help="Maximum sequence length")),
    ("-cpe", "--compress_pos_emb", dict(type=float, default=1.0, help="Compression factor for positional embeddings")),
    ("-a", "--alpha", dict(type=float, default=1.0, help="alpha for context size extension via embedding extension")),
    ("-theta", "--theta", dict(type=float, help="theta (base) for RoPE embeddings")),
    ("-gpfix", "--gpu_peer_fix", dict(action="store_true", help="Prevent direct copies of data between GPUs")),
    ("-flash", "--flash_attn", dict(nargs="?", const="default", metavar="METHOD", help="Accepted for parity: the in-tree flash kernel is always used; an integer sets max_input_len")),
    ("-mmrt", "--matmul_recons_thd", dict(type=int, default=8, help="No. rows at which the q4 matmul switches from the GEMV to the MFMA GEMM. 0 = never, 1 = always")),
    ("-fmt", "--fused_mlp_thd", dict(type=int, default=2, help="Maximum no. of rows for which to use fused MLP. 0 = never")),
    ("-sdpt", "--sdp_thd", dict(type=int, default=8, help="Accepted for parity (attention kernel choice is by shape)")),
    ("-mmfr", "--matmul_fused_remap", dict(action="store_true", help="Fuse column remapping in Q4 matmul kernel")),
    ("-nfa", "--no_fused_attn", dict(action="store_true", help="Disable fused attention")),
    ("-rnnh2", "--rmsnorm_no_half2", dict(action="store_true", help="Accepted for parity")),
    ("-rpnh2", "--rope_no_half2", dict(action="store_true", help="Accepted for parity")),
    ("-mmnh2", "--matmul_no_half2", dict(action="store_true", help="Accepted for parity")),
    ("-snh2", "--silu_no_half2", dict(action="store_true", help="Accepted for parity")),
    ("-nh2", "--no_half2", dict(action="store_true", help="Accepted for parity")),
    ("-fh2", "--force_half2", dict(action="store_true", help="Accepted for parity")),
    ("-cs", "--concurrent_streams", dict(action="store_true", help="Accepted for parity")),
    ("-aff", "--affinity", dict(type=str, help="Comma-separated list, sets processor core affinity. E.g.: -aff 0,1,2,3")),
]
_NO_HALF2 = ("rmsnorm_no_half2", "rope_no_half2", "matmul_no_half2", "silu_no_half2")


def add_args(parser):
    for short, long_, kw in _FLAGS:
        parser.add_argument(short, long_, **kw)


def post_parse(args):
    """The reference turns every half2 path off on ROCm unless forced (model_init.py:41-47)."""
    if args.no_half2 or not args.force_half2:
        for name in _NO_HALF2:
            setattr(args, name, True)


def get_model_files(args):
    """-d DIR fills in tokenizer / config / model paths (reference: model_init.py:52-69)."""
    if args.directory is not None:
        args.tokenizer = os.path.join(args.directory, "tokenizer.model")
        args.config = os.path.join(args.directory, "config.json")
        pattern = os.path.join(args.directory, "*.safetensors")
        found = sorted(glob.glob(pattern))
        if not found:
            print(f" !! No files matching {pattern}")
            sys.exit()
        args.model = found
    elif args.tokenizer is None or args.config is None or args.model is None:
        print(" !! Please specify either -d or all of -t, -c and -m")
        sys.exit()


def _wildcard_name(names):
    """One display name for a shard list: differing characters become '*'."""
    longest = max(names, key=len)
    out = list(longest)
    for name in names:
        for i, ch in enumerate(name):
            if out[i] != "*" and out[i] != ch:
                out[i] = "*"
    return "".join(out)


def print_options(args, extra_options=None):
    opts = []
    if args.gpu_split is not None:
        opts.append(f"gpu_split: {args.gpu_split}")
    if args.gpu_peer_fix:
        opts.append("gpu_peer_fix")
    if args.affinity:
        opts.append(f" --affinity: {args.affinity}")
    if extra_options is not None:
        opts += extra_options
    print(f" -- Tokenizer: {args.tokenizer}")
    print(f" -- Model config: {args.config}")
    print(f" -- Model: {args.model if isinstance(args.model, str) else _wildcard_name(args.model)}")
    print(f" -- Sequence length: {args.length}")
    if args.compress_pos_emb != 1.0:
        print(f" -- RoPE compression factor: {args.compress_pos_emb}")
    if args.alpha != 1.0:
        print(f" -- RoPE alpha factor: {args.alpha}")
    print(" -- Tuning:")
    for name in ("matmul_recons_thd", "fused_mlp_thd"):
        v = getattr(args, name)
        print(f" -- --{name}: {v}" + (" (disabled)" if v == 0 else ""))
    for name in ("matmul_fused_remap", "no_fused_attn"):
        if getattr(args, name):
            print(f" -- --{name}")
    print(f" -- Options: {opts}")


def make_config(args):
    """reference: model_init.py:123-157."""
    config = ExLlamaConfig(args.config)
    config.model_path = args.model
    config.max_seq_len = args.length
    config.compress_pos_emb = args.compress_pos_emb
    config.set_auto_map(args.gpu_split)
    config.gpu_peer_fix = args.gpu_peer_fix
    config.alpha_value = args.alpha
    config.calculate_rotary_embedding_base()
    if args.flash_attn:
        config.use_flash_attn_2 = True
        try:
            config.max_input_len = int(args.flash_attn)
        except ValueError:
            pass
    for name in ("matmul_recons_thd", "fused_mlp_thd", "sdp_thd", "matmul_fused_remap", "concurrent_streams") + _NO_HALF2:
        setattr(config, name, getattr(args, name))
    config.fused_attn = not args.no_fused_attn
    if args.theta:
        config.rotary_embedding_base = args.theta
    return config


def set_globals(args):
    """-aff 0,1,2,3 pins the process (reference: globals.py:3-22); used when timing the CPU baseline."""
    if args.affinity:
        os.sched_setaffinity(0, {int(c) for c in args.affinity.split(",")})


def print_stats(model):
    print(f" -- Groupsize (inferred): {model.config.groupsize if model.config.groupsize is not None else 'None'}")
    print(f" -- Act-order (inferred): {'yes' if model.config.act_order else 'no'}")
    if model.config.empty_g_idx:
        print(" !! Model has empty group index (discarded)")
