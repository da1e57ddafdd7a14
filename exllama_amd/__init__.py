"""exllama_amd -- MI355X-native (gfx950 / CDNA4) 4-bit GPTQ Llama inference path.

Only what the hot path needs lives here:
  csrc/         hand-written HIP kernels + the C ABI (include/exl_amd.h) -> libexl_amd.so
  _lib.py       ctypes binding of the C ABI (fails loudly if the library is not built)
  cuda_ext.py   host-side mirror of the reference's `cuda_ext` / `exllama_ext` operator surface
  model.py      ExLlamaConfig / ExLlama / ExLlamaCache (the reference's Python API) driving those ops
  model_init.py the reference's command-line flags -> ExLlamaConfig
  perplexity.py chunked perplexity evaluation (the -ppl leg of the reference's harness)
  lora.py       PEFT LoRA adapter loader feeding the q4 linears
  pipeline.py   layer split of one model across processes (RCCL send/recv of the hidden state)
  synth.py      seeded synthetic GPTQ checkpoints (no real weights exist in this environment)
"""

__version__ = "0.1.0"
