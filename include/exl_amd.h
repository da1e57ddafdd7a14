/*
 * exl_amd.h -- C ABI of libexl_amd.so: the MI355X (gfx950 / CDNA4) implementation of the
 * 4-bit GPTQ Llama hot path of turboderp/exllama (v1).
 *
 * This is the drop-in boundary.  Every entry point replaces one function
 * of the reference's pybind module `exllama_ext` (/root/reference/exllama_ext/exllama_ext.cpp:743-762)
 * -- the reference-side binding a maintainer would add is shown in INTEGRATION.md and shipped
 * as `exllama_amd/cuda_ext.py` (ctypes).  No torch types appear here: plain pointers and sizes.
 *
 * Conventions
 *   - all tensor pointers are DEVICE pointers to contiguous row-major data unless the name ends in
 *     `_host`;  `half` data is IEEE fp16 passed as `void*` / `uint16_t*`.
 *   - `stream` is a `hipStream_t` passed as `void*` (NULL = the legacy default stream).  All kernels
 *     are enqueued on it and nothing synchronises unless stated; this makes every compute entry point
 *     capturable in a hipGraph.
 *   - return value: 0 on success, otherwise a negative EXL_E_* code or a positive hipError_t;
 *     `exl_last_error()` returns a human-readable message (thread-local).
 *   - weight / scale / zero tensors stay OWNED by the caller; a Q4 handle BORROWS them
 *     (reference: exllama_ext/cuda_func/q4_matrix.cu:46-48) and, for act-order weights, REWRITES
 *     `qweight` in place at construction (reference: q4_matrix.cu:159).
 *   - optional device-side position: functions taking `past_len` also take `past_len_dev`
 *     (const int32_t* on the device, may be NULL).  When non-NULL the kernels read the position
 *     from it instead of the host integer, so a captured graph can be replayed at any position.
 */
#ifndef EXL_AMD_H
#define EXL_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EXL_OK               0
#define EXL_E_INVALID       (-1)   /* bad argument (shape / alignment / null)        */
#define EXL_E_NO_BUFFERS    (-2)   /* exl_prepare_buffers not called for this device  */
#define EXL_E_TOO_SMALL     (-3)   /* a borrowed scratch buffer is too small          */
#define EXL_E_UNSUPPORTED   (-4)   /* shape outside what the kernels support          */

#define EXL_MAX_DEVICES      16    /* reference: exllama_ext/cuda_buffers.cuh:9 (CUDA_MAX_DEVICES) */

const char* exl_last_error(void);
int  exl_version(void);

/* ---- tuning (reference: exllama_ext/tuning.h:4-16, exllama_ext.cpp:89-112 set_tuning_params) ------- */
typedef struct ExlTuning {
    int32_t matmul_recons_thd;    /* rows at which q4_matmul switches from the GEMV to the MFMA GEMM; 0 = never */
    int32_t fused_mlp_thd;
    int32_t sdp_thd;
    int32_t matmul_fused_remap;   /* accepted for compatibility; the act-order gather is always fused here       */
    int32_t rmsnorm_no_half2;     /* the five half2/stream flags are accepted and ignored: there is one          */
    int32_t rope_no_half2;        /* kernel per op on CDNA4, not a half/half2 pair                               */
    int32_t matmul_no_half2;
    int32_t silu_no_half2;
    int32_t concurrent_streams;
} ExlTuning;

int exl_set_tuning(const ExlTuning* t);
int exl_get_tuning(ExlTuning* t);

/* ---- per-device scratch (reference: exllama_ext.cpp:126-152 prepare_buffers, cuda_buffers.cu:66-89) -- */
/* temp_state: half [temp_state_numel]; temp_mlp: half [2*fused_mlp_thd*intermediate];
 * temp_zeros_float: float [max_zeros_float] (unused here: no atomics, kept for signature parity);
 * temp_dq: half [max K*N] (unused here: the prefill GEMM dequantises in registers).
 * Also allocates the library's own fp32 workspace for split-K slabs and attention partials
 * (hipMalloc, once per device; therefore NOT capturable -- call before any graph capture). */
int exl_prepare_buffers(int device, void* temp_state, size_t temp_state_numel, void* temp_mlp,
                        size_t temp_mlp_numel, void* temp_zeros_float, size_t max_zeros_float, void* temp_dq,
                        size_t temp_dq_numel);
/* reference: exllama_ext.cpp:117-121 cleanup(): frees every Q4 handle and every per-device buffer set. */
int exl_cleanup(void);

/* ---- Q4 matrix handle (reference: exllama_ext.cpp:157-194 make_q4, q4_matrix.cu:26-53, :104-168) ---- */
/* height = in_features K, width = out_features N, groups = K / groupsize.
 * g_idx_host: int32 [K] in HOST memory or NULL (reference passes the CPU tensor, model.py:144).
 * Synchronises `stream` when g_idx_host != NULL (as the reference's cudaDeviceSynchronize, q4_matrix.cu:163).
 * OWNERSHIP: the handle borrows the three tensors (the caller keeps them alive, as Ex4bitLinear does, model.py:141-143) and
 * REWRITES `qweight` IN PLACE: act-order repack (the reference does the same, q4_matrix.cu:159) and, for every Llama shape
 * (K % 128 == 0, N % 16 == 0, groupsize % 32 == 0), the re-tiling into the streaming layout -- so a tensor can back ONE handle,
 * once: calling make_q4 again on a tensor a live handle has already rewritten is refused (EXL_E_INVALID; the handle keeps a
 * 32-byte fingerprint of the rewritten tensor, so a recycled address holding a fresh checkpoint tensor is accepted). */
int exl_make_q4(int device, int height, int width, int groups, uint32_t* qweight, uint32_t* qzeros,
                uint16_t* scales, const uint32_t* g_idx_host, void* stream, void** out_handle);
int exl_free_q4(void* handle);
/* Introspection used by tests: any out pointer may be NULL. x_map is the DEVICE pointer (uint32 [K]) or NULL. */
int exl_q4_info(void* handle, int* device, int* height, int* width, int* groups, int* groupsize,
                const uint32_t** x_map_dev);
/* Weight layout the handle ended up with: EXL_LAYOUT_T16 (1) when height % 128 == 0, width % 16 == 0 and
 * groupsize % 32 == 0 -- make_q4 then RE-TILES the caller's qweight tensor in place (same bytes, same buffer) into the
 * 16-column-tile streaming layout documented in DESIGN.md ("T16"); otherwise EXL_LAYOUT_GPTQ (0), served by slower
 * generic kernels.  Either way the tensor must not be interpreted by the caller after make_q4 (the reference already
 * rewrites it for act-order weights, q4_matrix.cu:159). */
#define EXL_LAYOUT_GPTQ 0
#define EXL_LAYOUT_T16  1
int exl_q4_layout(void* handle, int* layout);

/* ---- q4 matmul: out[M,N] (+)= x[M,K] @ dequant(W)  (reference: exllama_ext.cpp:199-240 q4_matmul) ---- */
/* Dispatch on the tuning threshold exactly like the reference: rows < matmul_recons_thd (or thd == 0)
 * -> wave64 GEMV, else fused-dequant MFMA GEMM.  no_zero != 0 accumulates into `out` (residual fusion,
 * reference q4_matmul.cu:78-82 / cublas beta = 1 at :338). */
int exl_q4_matmul(void* w, const void* x, int x_height, void* out, int no_zero, void* stream);
/* The two kernels individually (reference: q4_matmul.cu:239-299 q4_matmul_cuda, :301-344 q4_matmul_recons_cuda). */
int exl_q4_matmul_gemv(void* w, const void* x, int x_height, void* out, int no_zero, void* stream);
int exl_q4_matmul_gemm(void* w, const void* x, int x_height, void* out, int no_zero, void* stream);
/* Prompt-pass fusion with no counterpart in the reference (it runs gate_proj, up_proj and the SiLU kernel one after the
 * other, model.py:266-273): out1 = silu(x @ W1) * (x @ W2) when silu != 0, else out1 = x @ W1 and out2 = x @ W2, in ONE
 * kernel that shares the activation tile between the two products.  Only for two T16 matrices of identical shape and
 * group size (act-order only when both carry the same row permutation: gathered once) and more than 512 rows: otherwise nothing is launched and *launched = 0 -- the caller then
 * issues the separate calls (exl_q4_matmul twice + exl_silu_mul). */
int exl_q4_matmul_dual(void* w1, void* w2, const void* x, int x_height, void* out1, void* out2, int silu, void* stream,
                       int* launched);
/* Prompt-pass fusion of the attention front half (reference: model.py:431-445 -- three q4_matmul, two rope_, the cache
 * scatter): q_out [bsz*q_len, heads*128] = rope(x @ Wq); key_cache[b, h, past_len + t, :] = rope(x @ Wk);
 * value_cache[b, h, past_len + t, :] = x @ Wv, in ONE kernel (RoPE and the cache write are the GEMM's epilogue).  Same
 * eligibility rule and `launched` protocol as exl_q4_matmul_dual, plus head_dim == 128; results are bit-identical to the
 * separate calls. */
int exl_q4_qkv_rope_cache(void* wq, void* wk, void* wv, const void* x, int bsz, int q_len, void* q_out, const void* sin,
                          const void* cos, void* key_cache, void* value_cache, int heads, int kv_heads, int head_dim,
                          int past_len, int max_seq_len, void* stream, int* launched);
/* The same launch with the layer's input RMSNorm as its prologue (x is the residual stream, norm_w != NULL; reference:
 * model.py:524-530 -- rms_norm, then the ops above).  Act-order q / k / v matrices are accepted when they share one row
 * permutation (GPTQ quantises them against the same input, so their g_idx tensors are identical): the norm kernel then writes
 * its rows through that map -- the reference's three column_remap passes (q4_matmul.cu:320-325) cost nothing.  norm_w == NULL:
 * exactly exl_q4_qkv_rope_cache (x already normalised; a shared map is gathered once).  Borrows temp_state. */
int exl_q4_attn_prompt(void* wq, void* wk, void* wv, const void* x, const void* norm_w, float eps, int bsz, int q_len, void* q_out,
                       const void* sin, const void* cos, void* key_cache, void* value_cache, int heads, int kv_heads, int head_dim,
                       int past_len, int max_seq_len, void* stream, int* launched);
/* The MLP half of a layer for a prompt of more than 512 rows, in place on the residual stream x [rows, hidden]:
 * x += down( silu(gate(n)) * up(n) ), n = RMSNorm(x) (reference: model.py:266-273 + :546-552: rms_norm, q4_matmul x 2, silu_mul,
 * q4_matmul with the residual) as norm(+ act-order gather) -> exl_q4_matmul_dual's kernel -> down_proj GEMM.  `act`: scratch of
 * rows * intermediate halves.  Same `launched` protocol; gate / up must be dual-eligible (act-order: one shared map). */
int exl_q4_mlp_prompt(void* x, const void* norm_w, float eps, void* gate, void* up, void* down, int rows, void* act, void* stream,
                      int* launched);
/* One decoder layer of a SHORT prompt (2 .. 256 rows) in place on the residual stream: the launches that replace what the reference's
 * model.py:421-552 drives op by op at this height (rms_norm, q / k / v q4_matmul, rope_ x 2, cache copy, attention, o_proj q4_matmul,
 * rms_norm, gate / up q4_matmul, silu_mul, down_proj q4_matmul), enqueued by ONE call; the GEMMs read their activations in the MFMA's
 * fragment order, written by their producers (csrc/q4_gemm_frag.hip).  *launched = 0: not covered (row count, a LoRA / bias is the
 * caller's test, layouts, act-order maps that differ inside q / k / v or gate / up, a map on down_proj) -- run the ops one by one. */
int exl_q4_layer_prompt(void* x, int bsz, int q_len, int past_len, const void* in_norm_w, const void* post_norm_w, float eps,
                        void* wq, void* wk, void* wv, void* wo, void* wgate, void* wup, void* wdown, const void* sin, const void* cos,
                        void* key_cache, void* value_cache, int heads, int kv_heads, int head_dim, int max_seq_len, void* stream,
                        float* rowsq, size_t rowsq_floats, int rowsq_in_slots, int* rowsq_out_slots, int* launched);
/* rowsq (optional, rows * (hidden / 16 + 4) floats; rowsq_floats = its size): the sums of squares RMSNorm needs travel from the GEMM that
 * wrote the residual stream to the norm that reads it, as `slots` partial sums per row.  On return *rowsq_out_slots > 0 says: rowsq holds
 * them for the x this call left behind.  Pass that number as rowsq_in_slots to the NEXT layer's call if -- and only if -- nothing has
 * written x in between (same pointer, same rows); pass 0 otherwise (the first layer, a layer not taken by this entry point before): the
 * norm then reads whole rows itself (slower, same result up to the order of an fp32 sum). */
size_t exl_frag_bytes(int rows, int K);
/* The short-prompt product by itself (the building block of exl_q4_layer_prompt; what the reference runs at this height as
 * q4_matmul -> reconstruct + cuBLAS, exllama_ext.cpp:199-240, q4_matmul.cu:301-344): x [rows, K] row-major is turned into fragment
 * order (with RMSNorm when norm_w != NULL, gathered through w[0]'s act-order map), then outs[i] (+)= x @ W_i for 1 .. 3 matrices of one
 * launch, or (dual, nmat = 2) out_frag = silu(x @ W_0) * (x @ W_1) in the fragment order of a consumer with K = width:
 *     out_frag[((mt * (K / 32) + 4 rb + j) * 64 + lane) * 16 .. + 16] = act[16 mt + (lane & 15)][128 rb + 32 (lane >> 4) + 8 j .. + 8]
 * (exl_frag_bytes(rows, width) bytes: rows padded to 64, from 65 rows on to a multiple of 128).  kernel: 0 = the launcher's choice,
 * 1 = activations in registers (narrow matrices), 2 .. 5 = activations shared through LDS (block shapes, csrc/q4_gemm_frag.hip),
 * 6 = that kernel with K cut over several blocks per output tile (one matrix: o_proj / down_proj), 7 .. 10 = its one- and two-row-tile
 * shapes (prompts of up to 16 / 32 rows).
 * *launched = 0: not covered (more than 256 rows, layout, group size, maps that differ).
 * rowsq_in / rowsq_in_slots: partial sums of squares of x for the RMSNorm (NULL / 0: the norm sums whole rows itself); rowsq_out: receives
 * rowsq_out[row * *rowsq_out_slots + slot] for ONE output matrix (not dual; rows * (width / 16 + 4) floats), what the next norm adds up. */
int exl_q4_matmul_frag(void* const* w, int nmat, const void* x, int rows, const void* norm_w, float eps, void* const* outs,
                       int no_zero, int dual, void* out_frag, int kernel, void* stream, const float* rowsq_in, int rowsq_in_slots,
                       float* rowsq_out, int* rowsq_out_slots, int* launched);
/* out = x @ W + (x @ lora_A) @ lora_B   (reference: exllama_ext.cpp:245-324 q4_matmul_lora) */
int exl_q4_matmul_lora(void* w, const void* x, int x_height, void* out, const void* lora_a, const void* lora_b,
                       int rank, void* lora_temp, void* stream);
/* Full dequantisation W16[K,N] = half(q - (z+1)) * scale  (reference: q4_matrix.cu:170-224 reconstruct). */
int exl_q4_reconstruct(void* w, void* out_w16, void* stream);

/* ---- act-order activation gather x_new[:,c] = x[:,x_map[c]] (reference: exllama_ext.cpp:328-358, column_remap.cu) */
int exl_column_remap(const void* x, void* x_new, int height, int width, const uint32_t* x_map_dev, void* stream);

/* ---- fp16 GEMM out (+)= x[M,K] @ w[K,N] (reference: exllama_ext.cpp:362-422 half_matmul / half_matmul_cublas) */
int exl_half_matmul(const void* x, const void* w, void* out, int height, int dim, int width, int no_zero,
                    void* stream);

/* ---- RMSNorm (reference: exllama_ext.cpp:606-643 rms_norm, rms_norm.cu:178-213). In place if out == x. -- */
int exl_rms_norm(const void* x, const void* w, void* out, float epsilon, int rows, int dim, void* stream);

/* ---- RoPE, in place (reference: exllama_ext.cpp:647-680 rope_, rope.cu:100-125) ----------------------- */
/* x: half [bsz, rows_per_batch, head_dim]; sin/cos: half [max_seq_len, head_dim]; position of row r is
 * past_len + r / num_heads. */
int exl_rope(void* x, const void* sin, const void* cos, int bsz, int rows_per_batch, int head_dim,
             int num_heads, int past_len, const int32_t* past_len_dev, void* stream);

/* ---- SiLU(x) * y, in place on x (reference: q4_mlp.cu:47-88 silu_mul_cuda_kernel) --------------------- */
int exl_silu_mul(void* x, const void* y, int height, int width, void* stream);

/* ---- KV-cache scatter (reference: q4_attn.cu:19-72 update_cache_kernel; model.py:440-443) ------------- */
/* states: half [bsz, q_len, num_kv_heads*head_dim]; caches: half [cache_bsz, num_kv_heads, max_seq_len, head_dim] */
int exl_update_cache(const void* key_states, const void* value_states, void* key_cache, void* value_cache,
                     int bsz, int q_len, int num_kv_heads, int head_dim, int max_seq_len, int past_len,
                     const int32_t* past_len_dev, void* stream);

/* ---- fp16 attention over the KV cache (replaces the ATen calls at model.py:376-409 and :463-495) ------ */
/* q: half [bsz, q_len, num_heads*head_dim] (already RoPE'd); caches as above; out: half [bsz, q_len, num_heads*head_dim].
 * Keys 0 .. past_len+q_len-1 are attended; query i sees keys j <= past_len + i (causal).
 * mask: optional additive half mask [bsz, 1, q_len, past_len+q_len] (model.py:1016-1026) or NULL.
 * GQA is handled by indexing (no repeat_kv).  Small q_len uses the split-KV decode kernel, large q_len the
 * MFMA flash kernel. */
int exl_attention(const void* q, const void* key_cache, const void* value_cache, void* out, const void* mask,
                  int bsz, int q_len, int num_heads, int num_kv_heads, int head_dim, int max_seq_len,
                  int past_len, const int32_t* past_len_dev, void* stream);

/* ---- fused decode ops ------------------------------------------------------------------------------- */
/* reference: exllama_ext.cpp:424-528 q4_attn -> q4_attn.cu:74-204:
 *   rms_norm(x) -> q/k/v projections (+LoRA) -> RoPE(q), RoPE(k) -> KV scatter at past_len.
 * x: half [bsz, q_len, dim]; query/key/value_states: outputs half [bsz, q_len, *].  LoRA pointers may be NULL. */
int exl_q4_attn(int device, const void* x, const void* rms_norm_weight, float epsilon, void* query_states,
                void* key_states, void* value_states, void* q_proj, void* k_proj, void* v_proj, const void* sin,
                const void* cos, int bsz, int q_len, int dim, int head_dim, int num_heads, int num_kv_heads,
                int past_len, const int32_t* past_len_dev, void* key_cache, void* value_cache, int max_seq_len,
                const void* q_a, const void* q_b, int q_rank, const void* k_a, const void* k_b, int k_rank,
                const void* v_a, const void* v_b, int v_rank, void* lora_temp, void* stream);
/* reference: exllama_ext.cpp:530-563 q4_attn_2 -> q4_attn.cu:206-228:  x += attn_output @ o_proj (+LoRA). */
int exl_q4_attn_2(void* x, const void* attn_output, void* o_proj, int height, const void* o_a, const void* o_b,
                  int o_rank, void* lora_temp, void* stream);
/* reference: exllama_ext.cpp:567-602 q4_mlp -> q4_mlp.cu:100-199:
 *   x += (silu(norm(x) @ gate) * (norm(x) @ up)) @ down. */
int exl_q4_mlp(int device, void* x, const void* rms_norm_weight, float epsilon, void* gate, void* up, void* down,
               int height, int dim, const void* gate_a, const void* gate_b, int gate_rank, const void* up_a,
               const void* up_b, int up_rank, const void* down_a, const void* down_b, int down_rank,
               void* lora_temp, void* stream);

/* ---- native single-token decode executor (no counterpart in the reference: it replaces the Python layer loop
 * model.py:1053-1058 + the three fused ops + the ATen attention for bsz = 1, q_len = 1) -------------------------- */
/* All pointers are device pointers that must stay valid for the decoder's lifetime: embed [vocab, hidden] fp16,
 * final_norm [hidden] fp16, lm_head [vocab, hidden] fp16, sin/cos [max_seq_len, head_dim] fp16. head_dim must be 128.
 * A decoder may be ONE STAGE of a layer-split model (reference: ExLlamaDeviceMap, model.py:636-668, hop at :1053-1058):
 * n_layers is then the stage's own layer count; embed == NULL -> the step starts from the residual stream the caller wrote to
 * exl_decoder_hidden(); final_norm == lm_head == NULL -> no head: the step leaves the residual stream there for the next stage. */
int exl_decoder_create(int device, int n_layers, int hidden, int inter, int heads, int kv_heads, int head_dim,
                       int vocab, int max_seq_len, float eps, const void* embed, const void* final_norm,
                       const void* lm_head, const void* sin, const void* cos, void** out_decoder);
/* q..down: Q4 handles of layer `index`; norms fp16 [hidden]; caches fp16 [1, kv_heads, max_seq_len, head_dim]. */
int exl_decoder_set_layer(void* decoder, int index, void* q, void* k, void* v, void* o, void* gate, void* up,
                          void* down, const void* in_norm, const void* post_norm, void* key_cache, void* value_cache);
/* One token: reads the token id (int64) and the position (int32) from DEVICE memory, appends K/V at that position,
 * writes fp32 logits [vocab]; when advance != 0 the device position is incremented at the end of the step.
 * 5 kernels per layer + 1 (hidden sizes above 4096 with several KV splits: 6), no allocation, no synchronisation: capturable
 * in a hipGraph. */
int exl_decoder_step(void* decoder, const int64_t* token_dev, int32_t* pos_dev, float* logits_out, int advance,
                     void* stream);
/* Number of KV splits the attention kernel of subsequent steps uses (1 .. the value chosen at creation; 0 = that value).
 * One split writes the attention output directly; several are merged in the o_proj kernel (hidden <= 4096) or by a small
 * kernel of their own.  *max_context (may be NULL) receives the longest context
 * (position) a step with this setting can serve.  A captured graph keeps the setting it was captured with, so a caller
 * can hold one graph per context bucket and pick by position (exllama_amd/model.py does). */
int exl_decoder_set_kv_splits(void* decoder, int nsplit, int* max_context);
/* One greedy generation step, all on the device: exl_decoder_step(..., advance = 1) followed by an argmax kernel that
 * writes the next token to *token_io_dev (where the following step reads its input) and, if history_dev is not NULL, to
 * history_dev[position of that token] (history_dev must hold max_seq_len + 1 entries).  Ties go to the lowest index.
 * Captured into a hipGraph, N replays generate N tokens with no host work in between (the reference's loop runs
 * torch.argmax and two copies on the host side per token, test_benchmark_inference.py:188-191). */
int exl_decoder_step_greedy(void* decoder, int64_t* token_io_dev, int32_t* pos_dev, float* logits_out, int64_t* history_dev,
                            void* stream);

/* ---- on-device sampler (reference: generator.py:91-170 sample(), :344-381 gen_single_token(),
 * cpu_func/rep_penalty.cpp:36-74) ------------------------------------------------------------------------------ */
typedef struct ExlSampler {
    float   temperature;        /* > 0                                                            (Settings.temperature, 0.95) */
    int32_t top_k;              /* 0 = the whole vocabulary, sorted, not renormalised (generator.py:110-111); > 1024 also sorts it (top_k, 40).
                                   The whole-vocabulary sort (top_k = 0 or > 1024) works in a per-device workspace that grows with the
                                   vocabulary (24 bytes per entry of the next power of two; up to 2^22 entries, EXL_E_UNSUPPORTED beyond);
                                   1 <= top_k <= 1024 needs none. */
    float   top_p;              /* 0 disables                                                                  (top_p, 0.65) */
    float   min_p;              /* cut inside the top-p loop (generator.py:128)                                 (min_p, 0.0) */
    float   typical;            /* 0 disables locally typical sampling                                        (typical, 0.0) */
    float   rep_penalty_max;    /* 1 disables                                          (token_repetition_penalty_max, 1.15) */
    int32_t rep_sustain;        /* -1 = whole sequence                             (token_repetition_penalty_sustain, 256) */
    int32_t rep_decay;          /*                                                   (token_repetition_penalty_decay, 128) */
    int32_t banned_token;       /* logit forced to -10000 (gen_single_token bans BOS, generator.py:355); -1 = none          */
    int32_t reserved;
    uint64_t seed;              /* Philox4x32-10 key when no uniform numbers are supplied                                    */
} ExlSampler;
/* Samples ONE token from fp32 logits in device memory: repetition penalty over history[0 .. *pos_new - 1], ban, temperature,
 * softmax, top-k, top-p / min-p, typical, then an inverse-CDF draw over the surviving list with u = uniforms[*pos_new]
 * (uniforms_dev != NULL) or Philox(seed, *pos_new).  Writes the token to *token_out_dev and history_dev[*pos_new]; logits
 * are modified in place (penalty, ban, temperature -- as the reference's in-place tensor ops do); probs_scratch_dev: vocab
 * floats; prob_out_dev (may be NULL): the token's probability in the final distribution.  One kernel, no synchronisation. */
int exl_sample(int device, float* logits_dev, float* probs_scratch_dev, int vocab, int64_t* history_dev, int64_t* token_out_dev,
               const int32_t* pos_new_dev, const float* uniforms_dev, float* prob_out_dev, const ExlSampler* s, void* stream);
/* exl_decoder_step(..., advance = 1) followed by the sampler: the sampled token lands in *token_io_dev (the next step's input)
 * and in history_dev[position of that token]; history_dev [max_seq_len + 1] must hold the sequence so far (prompt included)
 * at positions 0 .. *pos_dev.  Captured into a hipGraph, N replays generate N sampled tokens with no host work in between. */
int exl_decoder_step_sample(void* decoder, int64_t* token_io_dev, int32_t* pos_dev, float* logits_out, int64_t* history_dev,
                            const ExlSampler* s, const float* uniforms_dev, void* stream);

/* Measurement aid (bench.py): for each kernel class, `reps` passes over all layers' launches of that class back to
 * back between two hipEvents on `stream` (the weights stream from HBM as in a real step); class_ms_host[c] (HOST memory,
 * EXL_DEC_NCLASS floats) = mean time of one pass = that class' share of one token.  Overwrites the K/V slot at *pos_dev,
 * does not advance the position, synchronises. */
#define EXL_DEC_QKV     0
#define EXL_DEC_ATTN    1
#define EXL_DEC_MERGE   2
#define EXL_DEC_O       3
#define EXL_DEC_GATE_UP 4
#define EXL_DEC_DOWN    5
#define EXL_DEC_HEAD    6
#define EXL_DEC_NCLASS  7
int exl_decoder_step_timed(void* decoder, const int64_t* token_dev, int32_t* pos_dev, float* logits_out, int reps,
                           void* stream, float* class_ms_host);
/* The decoder's residual stream, fp16 [hidden] in device memory: input of a stage created without `embed`, output of a stage
 * created without `lm_head` (valid after the step's kernels have run on its stream). */
int exl_decoder_hidden(void* decoder, void** out_hidden_dev);
/* Makes the decoder use a CALLER-OWNED residual stream (fp16 [hidden], device memory that outlives the decoder and every graph
 * captured from it) instead of its own: the hand-off buffer between the stages of a layer split -- the caller copies it from
 * device to device or sends it to the next rank (RCCL point-to-point).  Call before the first step / capture. */
int exl_decoder_set_hidden(void* decoder, void* hidden_dev);
/* Test / measurement aid: which kernel configuration one step launches for kernel class `cls` with the current KV-split
 * setting, without launching anything.  out10: [0] launched (0 = this class is folded away), GEMV classes: [1] U (16-byte
 * loads in flight per lane), [2] NP (passes), [3] kernel kind (2 = rolling-ring stream; 1 / 0 = compiler-scheduled stream, group size % 128 == 0 / 32, 64), [4] PNORM, [5] EMODE, [6] NV (8-half
 * activation vectors per thread), [7] grid, [8] dynamic LDS bytes, [9] activation images; attention / merge: [1] KV splits. */
int exl_decoder_plan(void* decoder, int cls, int* out10);
/* Test / measurement aid: which weight-stream kernel the GEMV classes of this decoder launch.  EXL_DEC_OPT_RING: a bit per class (1 q/k/v,
 * 2 o_proj, 4 gate/up, 8 down_proj; default 15; environment EXL_DEC_RING): set = the hand-counted rolling-ring stream wherever
 * it covers the launch, 0 = the compiler-scheduled stream everywhere (same results bit for bit: tests compare the two).  EXL_DEC_OPT_RING_FENCE: 1 (default; EXL_DEC_RING_FENCE)
 * = a block queues the activation requests of all its waves before any weight request.  Graphs captured before a change keep
 * the kernels they were captured with. */
#define EXL_DEC_OPT_RING       0
#define EXL_DEC_OPT_RING_FENCE 1
#define EXL_DEC_OPT_RING_DEPTH 2   /* 16-byte loads a lane keeps in flight in the ring stream: 2 .. 4 (default 3, 4 where 3 does not fit the unit length; EXL_DEC_RING_DEPTH) */
#define EXL_DEC_OPT_RING_WIDE  3   /* 1 (default; EXL_DEC_RING_WIDE): 16-wave blocks where a plain-vector launch leaves one block per CU */
int exl_decoder_set_option(void* decoder, int option, int value);
/* Tensor parallelism (not in the reference: doc/TODO.md:19; exllama_amd/tp.py): a decoder built from ONE rank's shard --
 * heads * head_dim < hidden (its own heads), its own intermediate columns, the full residual stream.  exl_decoder_step_part
 * runs a token step in pieces so that the caller can all-reduce the residual stream (exl_decoder_hidden) between them:
 * part 0 = attention half of `layer` (RMSNorm + q/k/v (+ embedding lookup in layer 0 of a first stage), attention, o_proj),
 * part 1 = MLP half (RMSNorm + gate/up + SiLU*mul, down_proj), part 2 = final norm + head (+ position advance).  After parts
 * 0 and 1 the residual stream of a rank holds its PARTIAL sum, plus the incoming residual on the one rank for which
 * exl_decoder_set_tp(decoder, 1) was called (the default; call it with 0 on the others): the sum over ranks is the new
 * residual stream.  Every call only enqueues on `stream` (capturable together with the collectives). */
int exl_decoder_step_part(void* decoder, int layer, int part, const int64_t* token_dev, int32_t* pos_dev, float* logits_out,
                          int advance, void* stream);
int exl_decoder_set_tp(void* decoder, int residual_owner);
/* LoRA operands of one layer of the executor (reference: the lora_A / lora_B arguments of q4_attn / q4_attn_2 / q4_mlp,
 * exllama_ext.cpp:424-602, and q4_matmul_lora, :245-324): a7 / b7 / rank7 in the order q, k, v, o, gate, up, down (NULL / 0 = no
 * adapter on that projection; all NULL clears the layer).  A [in_features, r], B [r, out_features] fp16, alpha / r folded into B,
 * r <= 64.  out = W x + (x A) B then runs INSIDE the token step (two small launches behind each GEMV launch with adapters), so a
 * model with an adapter keeps the hipGraph path.  EXL_E_UNSUPPORTED for layers whose o_proj / down_proj gather through an act-order
 * map and on tensor-parallel shards: the caller keeps the op-by-op path.  Set before capturing; new adapters = capture again. */
int exl_decoder_set_lora(void* decoder, int layer, const void* const* a7, const void* const* b7, const int* rank7);
int exl_decoder_free(void* decoder);

/* ---- embedding lookup and the prompt pass' lm_head (reference: torch ops inside model.py, not exllama_ext functions:
 * model.py:1002 `self.embed_tokens(input_ids)`, :1077 `self.lm_head(hidden_states)`) -- so that no BLAS / ATen kernel is left on
 * the token path.  exl_embedding: out[i] = table[ids[i]] (fp16 rows, ids int64 in DEVICE memory, clamped to the table).
 * exl_head_matmul: out[r][v] = float(half(x[r] . w[v])): a GEMV for rows <= 8 (the last-token logits of a prompt, a short prompt),
 * an fp16 MFMA GEMM with both tiles staged by LDS-DMA for whole sequences (the `-ppl` leg, perplexity.py:121-138); returns 1
 * (nothing launched) only for shapes neither covers (hidden % 64 != 0, vocab % 4 != 0): the caller keeps its own GEMM then. */
int exl_embedding(const int64_t* ids_dev, const void* table, void* out, int n_ids, int hidden, int vocab, void* stream);
int exl_head_matmul(const void* x, const void* w, float* out, int rows, int hidden, int vocab, void* stream);

/* ---- repetition penalty, HOST memory, fp32 (reference: exllama_ext.cpp:684-741, cpu_func/rep_penalty.cpp) */
int exl_rep_penalty(int vocab_size, const uint64_t* sequence_host, float* rep_mask_host, float penalty_max,
                    int sustain, int decay, int seq_len);
int exl_apply_rep_penalty(int vocab_size, const uint64_t* sequence_host, float penalty_max, int sustain,
                          int decay, int seq_len, int bsz, float* logits_host);

#ifdef __cplusplus
}
#endif
#endif /* EXL_AMD_H */
