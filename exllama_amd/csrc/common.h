// Internal declarations shared by the HIP translation units of libexl_amd.so (gfx950 only).
#pragma once

#include <vector>
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/exl_amd.h"

typedef _Float16 f16;
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

// 128-bit streaming load (weights are read exactly once per token: keep them out of the way of L2-resident data)
#ifdef __HIPCC__
__device__ __forceinline__ uint4 nt_load16(const void* p)
{
    const u32x4 v = __builtin_nontemporal_load((const u32x4*) p);
    return make_uint4(v[0], v[1], v[2], v[3]);
}
#endif

// ---- error plumbing ---------------------------------------------------------------------------------
void exl_set_error(const char* fmt, ...);

#define EXL_FAIL(code, ...) do { exl_set_error(__VA_ARGS__); return (code); } while (0)
#define EXL_REQUIRE(cond, code, ...) do { if (!(cond)) { exl_set_error(__VA_ARGS__); return (code); } } while (0)
#define EXL_HIP(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { \
    exl_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); return (int) _e; } } while (0)
#define EXL_LAUNCH_CHECK() do { hipError_t _e = hipGetLastError(); if (_e != hipSuccess) { \
    exl_set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(_e), __FILE__, __LINE__); return (int) _e; } } while (0)
#define EXL_TRY(expr) do { int _r = (expr); if (_r != 0) return _r; } while (0)

// More than 64 KiB of dynamic LDS is an opt-in per kernel and PER DEVICE (hipFuncSetAttribute acts on the current device's
// copy of the function): `done` is the caller's static per-device flag array for that kernel.
#ifdef __HIPCC__
static inline int exl_lds_opt_in(const void* kfn, bool (&done)[EXL_MAX_DEVICES])
{
    int dev = 0;
    EXL_HIP(hipGetDevice(&dev));
    if (dev >= 0 && dev < EXL_MAX_DEVICES && done[dev]) return 0;
    EXL_HIP(hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    if (dev >= 0 && dev < EXL_MAX_DEVICES) done[dev] = true;
    return 0;
}
#endif

// ---- Q4 matrix handle (reference: exllama_ext/cuda_func/q4_matrix.cuh:8-46) ---------------------------
struct Q4Matrix {
    uint32_t magic;
    int device;
    int height;        // K
    int width;         // N
    int groups;
    int groupsize;
    uint32_t* qweight; // borrowed; layout 0: GPTQ [K/8, N]; layout 1: T16 pieces (gemv_t16.h), re-tiled IN PLACE by make_q4
    uint32_t* qzeros;  // borrowed [G, N/8]
    f16* scales;       // borrowed [G, N]
    uint32_t* x_map;   // owned   [K] or NULL (act-order)
    uint64_t xmap_hash;// FNV-1a of the map's K entries (0 without a map): matrices quantised against the same input (q / k / v, gate / up)
                       // carry the SAME map in GPTQ checkpoints; equal (height, hash) lets the fused prompt kernels gather once
    std::vector<uint32_t> xmap_host;   // the map's entries on the host: equal hashes are CONFIRMED entry by entry (q4_same_map) before a
                       // fused launch gathers k / v / up through q's / gate's map -- a hash collision must not pick a wrong permutation
    int layout;        // EXL_LAYOUT_GPTQ or EXL_LAYOUT_T16
    uint32_t fp[8];    // first and last 16 bytes of qweight AFTER the in-place rewrite (make_q4's double-call guard); fp_valid: rewritten
    bool fp_valid;
};
#define EXL_LAYOUT_GPTQ 0   // as loaded; kept only for shapes the T16 tiling cannot express (K % 128 or N % 16 != 0)
#define EXL_LAYOUT_T16  1   // the product layout
#define EXL_Q4_MAGIC 0x51344d58u

// Looks the pointer up in the registry of live handles (never dereferences an unknown pointer).
Q4Matrix* q4_from_handle(void* h);

// ---- per-device buffers (reference: exllama_ext/cuda_buffers.cuh) -------------------------------------
struct DeviceBuffers {
    bool prepared;
    f16* temp_state;  size_t temp_state_numel;
    f16* temp_mlp;    size_t temp_mlp_numel;
    float* temp_zeros_float; size_t max_zeros_float;
    f16* temp_dq;     size_t temp_dq_numel;
    float* workspace; size_t workspace_floats;   // owned: decode split-K slabs / attention partials
    void* sampler_ws; size_t sampler_ws_bytes;   // owned: whole-vocabulary sort of the device sampler (top_k = 0 / > 1024), allocated on first request
    float* gemm_ws;   size_t gemm_ws_floats;     // owned, grown on demand: fp32 slices of the prompt GEMMs' K splits (its own buffer: a caller
                                                 // running attention on another stream must not see them land in `workspace`)
};
DeviceBuffers* exl_buffers(int device);           // never NULL for 0 <= device < EXL_MAX_DEVICES
int exl_workspace(int device, size_t floats, float** out);   // fails if too small / not prepared
// The sampler's whole-vocabulary sort (top_k = 0 or > 1024) works in a per-device workspace sized for the vocabulary rounded up to a
// power of two: a 64-bit key, two probabilities and two indices per entry.  (Round 5 held 65536 entries, fixed; Llama-3's 128256
// tokens and anything else up to 2^22 fit now.)
#define SMP_BIG_LIMIT (1 << 22)
static inline size_t smp_big_np(int vocab) { size_t np = 1; while (np < (size_t) (vocab > 1 ? vocab : 1)) np <<= 1; return np; }
static inline size_t smp_big_bytes(int vocab) { return smp_big_np(vocab) * (8 + 4 + 4 + 4 + 4); }
int exl_sampler_workspace(int device, size_t bytes, void** out);  // allocated once per device (NOT capturable)
int exl_gemm_workspace(int device, size_t floats, float** out);   // grows (device-synchronising, NOT capturable) up to 512 MiB; non-zero: no room
extern ExlTuning g_tuning;

// ---- launchers implemented in the .hip files -----------------------------------------------------------
int launch_make_sequential(Q4Matrix* m, const uint32_t* x_map_host, hipStream_t s);
int launch_retile_t16(Q4Matrix* m, hipStream_t s);            // GPTQ -> T16 in place (through a temporary copy); sets m->layout
int launch_reconstruct(const Q4Matrix* m, f16* out, hipStream_t s);
int launch_column_remap(const f16* x, f16* x_new, int height, int width, const uint32_t* x_map, hipStream_t s);

// x_norm: optional fused prologue = RMSNorm(x) with weight `norm_w` (NULL = plain x)
struct GemvArgs {
    const Q4Matrix* w;
    const f16* x; int rows;
    f16* out; int no_zero;
    float* slabs; size_t slab_floats;
};
int launch_q4_gemv(const Q4Matrix* w, const f16* x, int rows, f16* out, int no_zero, float* ws, size_t ws_floats,
                   hipStream_t s);
bool q4_gemv_covers(const Q4Matrix* w);          // false: in_features beyond what the decode GEMV stages -> use launch_q4_gemm
int launch_q4_gemm(const Q4Matrix* w, const f16* x, int rows, f16* out, int no_zero, f16* remap_tmp,
                   size_t remap_tmp_numel, hipStream_t s);
// Optional prologue of the two fused prompt-pass launches below: `norm_w` != NULL -> x is the residual stream and the kernel input is
// RMSNorm(x) (written to `tmp`); act-order matrices that share one map get their input gathered ONCE into `tmp` (by the norm
// kernel when there is one, else by column_remap).  `tmp` must hold rows * K halves whenever either applies.
struct PromptPrologue { const f16* norm_w; float eps; f16* tmp; size_t tmp_numel; };
bool q4_same_map(const Q4Matrix* a, const Q4Matrix* b);      // both without a map, or maps of equal height whose entries are equal (hash first, then confirmed entry by entry)
int launch_q4_qkv_rope_cache(const Q4Matrix* wq, const Q4Matrix* wk, const Q4Matrix* wv, const f16* x, int rows, f16* q_out,
                             const f16* sin, const f16* cos, f16* kc, f16* vc, int q_len, int heads, int kv_heads, int head_dim,
                             int past_len, int max_seq, const PromptPrologue& pro, hipStream_t s);
int launch_q4_gemm_dual(const Q4Matrix* w1, const Q4Matrix* w2, const f16* x, int rows, f16* out1, f16* out2, int silu,
                        const PromptPrologue& pro, hipStream_t s);   // 1 = not eligible (run the products separately)
int launch_half_gemm(const f16* x, const f16* w, f16* out, int M, int K, int N, int no_zero, hipStream_t s);

int launch_dec_sample(float* logits, float* probs, int64_t* history, int64_t* token_io, const int32_t* pos_dev, const float* uniforms,
                      float* prob_out, int vocab, const ExlSampler* s, hipStream_t stream);
int launch_embedding(const int64_t* ids, const f16* table, f16* out, int n_ids, int hidden, int vocab, hipStream_t s);
int launch_head_rows(const f16* x, const f16* w, float* out, int rows, int hidden, int vocab, hipStream_t s);   // 1 = not covered
int launch_head_gemm(const f16* x, const f16* w, float* out, int rows, int hidden, int vocab, hipStream_t s);   // 1 = not covered; half_gemm_nt.hip
// flash_prefill.hip: the MFMA prompt attention (head_dim 128, q_len >= 16, no mask); frag = 1: the output in the fragment order of
// q4_gemm_frag.hip (rows = bsz * q_len, K = heads * 128; `out` holds frag_bytes(rows, heads * 128))
int launch_flash_prefill(const f16* q, const f16* kc, const f16* vc, f16* out, int bsz, int q_len, int heads, int kv_heads, int hd, int max_seq,
                         int past_len, hipStream_t s, int frag);
// q4_gemm_frag.hip: short-prompt GEMMs on fragment-order activations
size_t frag_bytes(int rows, int K);
int launch_to_frag(const f16* x, const f16* norm_w, float eps, const uint32_t* x_map, void* xf, int rows, int K, hipStream_t s,
                   const float* rowsq = nullptr, int nslots = 0);   // rowsq: the producer's per-row partial sums of squares (GrArgs::rowsq)
// 1 = not covered; force: which kernel (0 = the launcher's choice); rowsq / rowsq_slots: per-row partial sums of the squares of the output
// (one matrix, not dual) for the RMSNorm behind the launch: rowsq[row * *rowsq_slots + slot], rows x (width / 16 + 4) floats hold any shape
int launch_gemm_t16r(int nmat, const Q4Matrix* const* w, const void* xf, int rows, f16* const* outs, int no_zero, int dual, void* out_frag,
                     hipStream_t s, int force = 0, float* rowsq = nullptr, int* rowsq_slots = nullptr, float* kws = nullptr, size_t kws_floats = 0);
size_t gemm_frag_ksplit_floats(int rows, int N);                    // scratch for the K-cut form of a one-matrix launch (kws): 8 fp32 slices of the padded output
bool gemm_t16r_covers(int nmat, const Q4Matrix* const* w, int rows, int dual);
int launch_rms_norm(const f16* x, const f16* w, f16* out, float eps, int rows, int dim, hipStream_t s);
int launch_rms_norm_gather(const f16* x, const f16* w, f16* out, const uint32_t* x_map, float eps, int rows, int dim, hipStream_t s);   // x_map NULL: plain norm
// decode_fused.hip: one fused executor launch for the op-level entry points (0 done, 1 not covered, > 1 error)
int dec_op_gemv(int device, int cls, int pnorm, int emode, const f16* vec, const f16* norm_w, float eps, int nmat,
                Q4Matrix* const* mats, f16* const* outs, f16* hid_io, hipStream_t s);
// decode_fused.hip: the executor's two adapter launches behind an op-level fused launch at one row (rank <= 64): outs[i] (+)= (x A_i) B_i
int dec_op_lora(int device, int nmat, const f16* x, const f16* norm_w, float eps, int K, const f16* const* a3, const f16* const* b3, const int* r3,
                f16* const* outs, const int* widths, int silu, f16* act, hipStream_t s);
int launch_rope(f16* x, const f16* sin, const f16* cos, int bsz, int rows_per_batch, int head_dim, int num_heads,
                int past_len, const int32_t* past_len_dev, hipStream_t s);
int launch_silu_mul(f16* x, const f16* y, int height, int width, hipStream_t s);
int launch_rope_qk_cache(f16* q, f16* k, const f16* v, f16* kc, f16* vc, const f16* sin, const f16* cos, int bsz, int q_len,
                         int heads, int kvh, int hd, int max_seq, int past_len, const int32_t* past_len_dev, hipStream_t s);
int launch_update_cache(const f16* k, const f16* v, f16* kc, f16* vc, int bsz, int q_len, int kvh, int hd,
                        int max_seq, int past_len, const int32_t* past_len_dev, hipStream_t s);
int launch_attention(const f16* q, const f16* kc, const f16* vc, f16* out, const f16* mask, int bsz, int q_len,
                     int heads, int kv_heads, int hd, int max_seq, int past_len, const int32_t* past_len_dev,
                     float* ws, size_t ws_floats, hipStream_t s);
