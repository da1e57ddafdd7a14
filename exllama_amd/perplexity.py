"""Chunked perplexity evaluation (mirror of the reference's perplexity.py:16-138): the `-ppl` leg of its harness, and the
only end-to-end accuracy number the reference publishes (README perplexities, 2 decimals).

Token ids in, one number out: every chunk is pushed through `model.forward(ids[:, :-1], cache, last_id_only=False)` from an
empty cache, the log-probability of each next token is gathered, and exp(-mean) over all chunks is the perplexity.
`ppl_token=True` feeds the chunk one token at a time instead (the decode kernels): both modes must agree, which is the
reference's own `-v` check (test_benchmark_inference.py:237-246) and what tests/test_model_gpu.py asserts.
"""
import json
import math
import os
import sys

import torch
import torch.nn.functional as F

from .model import ExLlamaCache


class Perplexity:
    def __init__(self, method="default", model=None, cache=None, tokenizer=None):
        self.method = method
        self.model = model
        self.cache = cache
        self.tokenizer = tokenizer                       # anything with encode(str) -> LongTensor[1, n]; optional
        self.dataset_chunks = []
        self._begin()

    def _begin(self):
        if self.cache is None:
            self.cache = ExLlamaCache(self.model)
        else:
            self.cache.current_seq_len = 0

    # ---- datasets ------------------------------------------------------------------------------------------------------
    def add_tokens(self, tokens, chunk_size, chunk_truncate=None, overlap=0):
        """Sliding windows over one long token row (what `load` does with a raw-text file after tokenising it)."""
        tokens = tokens.view(1, -1)
        if overlap >= chunk_size:                        # the window must advance, and needs one token to predict
            overlap = chunk_size - 2
        for start in range(0, tokens.size(1), chunk_size - overlap):
            chunk = tokens[:, start:start + chunk_size]
            if chunk_truncate is not None:
                chunk = chunk[:, :chunk_truncate]
            self.dataset_chunks.append(chunk)

    def load(self, dataset_path, chunk_size, chunk_truncate=None, overlap=0, minlength=0, json_key="text"):
        """.json / .jsonl: one (truncated) chunk per record longer than `minlength` characters; anything else: raw text cut
        into windows of `chunk_size` tokens (reference: perplexity.py:55-91)."""
        if self.tokenizer is None:
            raise ValueError("Perplexity.load needs a tokenizer; use add_tokens() for pre-tokenised data")
        if os.path.splitext(dataset_path)[1] in (".jsonl", ".json"):
            with open(dataset_path) as f:
                for line in f:
                    text = json.loads(line)[json_key]
                    if len(text) <= minlength:
                        continue
                    chunk = self.tokenizer.encode(text)[:, :chunk_size]
                    if chunk_truncate is not None:
                        chunk = chunk[:, :chunk_truncate]
                    self.dataset_chunks.append(chunk)
        else:
            with open(dataset_path, encoding="utf-8") as f:
                self.add_tokens(self.tokenizer.encode(f.read()), chunk_size, chunk_truncate, overlap)

    # ---- evaluation ----------------------------------------------------------------------------------------------------
    def _chunk_logits(self, input_ids, lora, ppl_token):
        if not ppl_token:
            return self.model.forward(input_ids, self.cache, last_id_only=False, lora=lora)
        steps = [self.model.forward(input_ids[:, i:i + 1], self.cache, last_id_only=False, lora=lora)
                 for i in range(input_ids.shape[-1])]
        return torch.cat(steps, dim=1)

    def test(self, chunk_limit=sys.maxsize, lora=None, tag="", ppl_token=False, quiet=False):
        """Returns the perplexity (the reference only prints it)."""
        if not self.dataset_chunks:
            sys.exit(" xx ERROR: Empty dataset!")
        if not quiet:
            print(f" -- Testing {min(len(self.dataset_chunks), chunk_limit)} chunks", end="", flush=True)
        logprob_sum, logprob_count = 0.0, 0
        for n, chunk in enumerate(self.dataset_chunks):
            if chunk_limit and n >= chunk_limit:
                break
            if chunk.shape[-1] < 2:
                continue
            self._begin()
            logits = self._chunk_logits(chunk[:, :-1], lora, ppl_token)
            targets = chunk[:, 1:].to(logits.device)
            token_lp = F.log_softmax(logits.float(), dim=-1).gather(-1, targets.unsqueeze(-1)).squeeze(-1)
            logprob_sum += token_lp.sum().item()
            logprob_count += targets.numel()
            if not quiet and n % 10 == 0:
                print(".", end="", flush=True)
        ppl = math.exp(-logprob_sum / logprob_count)
        if not quiet:
            print("")
            print(f" ** Perplexity{tag}: {ppl:.4f}")
        return ppl


_PPL_FLAGS = [
    ("-ppl", "--perplexity", dict(nargs="?", const="default", metavar="METHOD", help="Perplexity benchmark. Optionally specify method: gptq-for-llama")),
    ("-ppl_ds", "--perplexity_dataset", dict(metavar="DATAPATH", type=str, help="Load dataset for perplexity (JSONL if .jsonl, otherwise parses it as raw text)")),
    ("-ppl_cn", "--perplexity_chunk_num", dict(nargs="?", type=int, default=100, help="Number of chunks for perplexity benchmark")),
    ("-ppl_cs", "--perplexity_chunk_size", dict(type=int, default=2048, help="Size of chunks for perplexity benchmark")),
    ("-ppl_ct", "--perplexity_chunk_truncate", dict(type=int, default=2048, help="Truncated size of chunks for perplexity benchmark")),
    ("-ppl_co", "--perplexity_chunk_overlap", dict(type=int, default=0, help="Chunk overlap")),
    ("-ppl_cm", "--perplexity_chunk_min", dict(type=int, default=50, help="Minimum chunk length")),
    ("-ppl_key", "--perplexity_json_key", dict(type=str, default="text", help="Key to extract from JSON dataset, default: 'text'")),
    ("-ppl_t", "--perplexity_token", dict(action="store_true", help="Run perplexity test on individual tokens, for debug purposes (slow)")),
]


def add_args(parser):
    for short, long_, kw in _PPL_FLAGS:
        parser.add_argument(short, long_, **kw)


def post_parse(args):
    """Method presets (reference: perplexity.py:153-180)."""
    if not args.perplexity:
        return
    if args.perplexity == "gptq-for-llama":
        args.perplexity_dataset = "datasets/wikitext2.txt"
        args.perplexity_chunk_num = 128
        args.perplexity_chunk_size = 2048
        args.perplexity_chunk_truncate = 2048
        args.perplexity_chunk_overlap = 0
        args.perplexity_chunk_min = 0
    if args.perplexity_dataset is None:
        args.perplexity_dataset = "datasets/wikitext2_val_sample.jsonl"
