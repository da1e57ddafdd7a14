// TEST INFRASTRUCTURE ONLY.  A plain C ABI over the REFERENCE's own GPU entry points (the *_cuda functions of
// /root/reference/exllama_ext/cuda_func/*.cu, hipified at build time by oracle/build_ref_kernels.sh), so that
// oracle/make_ref_golden.py can run the reference's kernels on the GPU box through ctypes -- without torch's JIT
// extension build -- and record their outputs as golden vectors for oracle/exl_oracle.py.
//
// This file contains no reference code: it only forwards raw pointers to the functions the reference's pybind layer
// (exllama_ext.cpp:126-761) forwards torch tensors to.  Never linked into libexl_amd.so.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <hipblas/hipblas.h>
#include <stdint.h>

#include "tuning.h"
#include "cuda_buffers.cuh"
#include "cuda_func/q4_matrix.cuh"
#include "cuda_func/q4_matmul.cuh"
#include "cuda_func/column_remap.cuh"
#include "cuda_func/rms_norm.cuh"
#include "cuda_func/rope.cuh"
#include "cuda_func/half_matmul.cuh"
#include "cuda_func/q4_attn.cuh"
#include "cuda_func/q4_mlp.cuh"

static ExLlamaTuning g_tuning = {8, 2, 8, false, false, false, false, false, false};
static hipblasHandle_t g_blas = nullptr;

static hipblasHandle_t blas()
{
    if (!g_blas) hipblasCreate(&g_blas);
    return g_blas;
}

static int finish()
{
    hipError_t e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipGetLastError();
    return (int) e;
}

extern "C" {

// exllama_ext.cpp:89-112
int ref_set_tuning(int matmul_recons_thd, int fused_mlp_thd, int sdp_thd, int matmul_fused_remap, int rmsnorm_no_half2,
                   int rope_no_half2, int matmul_no_half2, int silu_no_half2, int concurrent_streams)
{
    g_tuning.matmul_recons_thd = matmul_recons_thd; g_tuning.fused_mlp_thd = fused_mlp_thd; g_tuning.sdp_thd = sdp_thd;
    g_tuning.matmul_fused_remap = matmul_fused_remap != 0; g_tuning.rmsnorm_no_half2 = rmsnorm_no_half2 != 0;
    g_tuning.rope_no_half2 = rope_no_half2 != 0; g_tuning.matmul_no_half2 = matmul_no_half2 != 0;
    g_tuning.silu_no_half2 = silu_no_half2 != 0; g_tuning.concurrent_streams = concurrent_streams != 0;
    return 0;
}

// exllama_ext.cpp:126-152
int ref_prepare_buffers(int device, void* temp_state, int temp_state_size, void* temp_mlp, void* temp_zeros_float,
                        void* temp_dq, int max_zeros_float)
{
    prepare_buffers_cuda(device, (half*) temp_state, temp_state_size, (half*) temp_mlp, (float*) temp_zeros_float,
                         (half*) temp_dq, max_zeros_float);
    return finish();
}

// exllama_ext.cpp:156-194 (g_idx: HOST pointer or NULL, as the reference's make_q4 receives a CPU tensor)
void* ref_make_q4(int height, int width, int groups, void* qweight, void* qzeros, void* scales, const uint32_t* g_idx_host,
                  int device)
{
    Q4Matrix* m = new Q4Matrix(height, width, groups, (uint32_t*) qweight, (uint32_t*) qzeros, (half*) scales,
                               (uint32_t*) g_idx_host, device);
    g_q4_keep_matrix(m);
    return finish() == 0 ? (void*) m : nullptr;
}

const void* ref_q4_x_map(void* h) { return ((Q4Matrix*) h)->cuda_x_map; }

// device-to-device copy of the act-order map (uint32 [height]) into a caller-owned buffer
int ref_copy_x_map(void* h, void* dst)
{
    Q4Matrix* m = (Q4Matrix*) h;
    if (!m->cuda_x_map) return -1;
    hipError_t e = hipMemcpy(dst, m->cuda_x_map, (size_t) m->height * sizeof(uint32_t), hipMemcpyDeviceToDevice);
    return e == hipSuccess ? finish() : (int) e;
}

// q4_matrix.cu:170-224
int ref_reconstruct(void* h, void* out)
{
    ((Q4Matrix*) h)->reconstruct((half*) out);
    return finish();
}

// q4_matmul.cu:239-299 -- the decode kernel (fp16 accumulate, split-K atomics); `out` must be zeroed or hold the residual
int ref_q4_matmul(void* h, const void* x, int rows, void* out, int no_zero)
{
    q4_matmul_cuda(&g_tuning, (const half*) x, rows, (Q4Matrix*) h, (half*) out, no_zero != 0);
    return finish();
}

// q4_matmul.cu:301-344 -- column_remap + reconstruct + hipBLAS Hgemm
int ref_q4_matmul_recons(void* h, const void* x, int rows, void* out, int no_zero)
{
    q4_matmul_recons_cuda(&g_tuning, (const half*) x, rows, (Q4Matrix*) h, (half*) out, blas(), no_zero != 0);
    return finish();
}

// column_remap.cu:42-61
int ref_column_remap(const void* x, void* x_new, int height, int width, const void* x_map)
{
    column_remap_cuda((const half*) x, (half*) x_new, height, width, (const uint32_t*) x_map);
    return finish();
}

// rms_norm.cu:178-213
int ref_rms_norm(void* x, const void* w, void* out, float eps, int rows, int dim, int device)
{
    rms_norm_cuda(&g_tuning, (half*) x, (const half*) w, (half*) out, eps, rows, dim, device);
    return finish();
}

// rope.cu:100-125
int ref_rope(void* x, const void* sin, const void* cos, int bsz, int rows, int head_dim, int num_heads, int past_len)
{
    rope_cuda(&g_tuning, (half*) x, (const half*) sin, (const half*) cos, bsz, rows, head_dim, num_heads, past_len);
    return finish();
}

// half_matmul.cu: the plain kernel and the BLAS form
int ref_half_matmul(const void* x, const void* w, void* out, int height, int dim, int width)
{
    half_matmul_cuda((const half*) x, (const half*) w, (half*) out, height, dim, width);
    return finish();
}

int ref_half_matmul_blas(const void* x, const void* w, void* out, int height, int dim, int width)
{
    half_matmul_cublas_cuda(&g_tuning, (const half*) x, (const half*) w, (half*) out, height, dim, width, blas());
    return finish();
}

// q4_mlp.cu:100-199 (no LoRA operands)
int ref_q4_mlp(void* x, const void* rms_w, float eps, void* gate, void* up, void* down, int height, int dim, int device)
{
    q4_mlp_cuda(&g_tuning, (half*) x, (const half*) rms_w, eps, (Q4Matrix*) gate, (Q4Matrix*) up, (Q4Matrix*) down, height, dim,
                nullptr, nullptr, 0, nullptr, nullptr, 0, nullptr, nullptr, 0, nullptr, blas(), device);
    return finish();
}

// q4_attn.cu:74-204 (no LoRA operands)
int ref_q4_attn(void* x, const void* rms_w, float eps, void* q_states, void* k_states, void* v_states, void* q, void* k, void* v,
                void* sin, void* cos, int bsz, int q_len, int dim, int head_dim, int num_heads, int num_kv_heads, int past_len,
                void* key_cache, void* value_cache, int max_seq_len, int device)
{
    q4_attn_cuda(&g_tuning, (hipStream_t) 0, blas(), (half*) x, (const half*) rms_w, eps, (half*) q_states, (half*) k_states,
                 (half*) v_states, (Q4Matrix*) q, (Q4Matrix*) k, (Q4Matrix*) v, (half*) sin, (half*) cos, bsz, q_len, dim, head_dim,
                 num_heads, num_kv_heads, past_len, (half*) key_cache, (half*) value_cache, nullptr, nullptr, 0, nullptr, nullptr,
                 0, nullptr, nullptr, 0, nullptr, max_seq_len, device);
    return finish();
}

// q4_attn.cu:206-228
int ref_q4_attn_2(void* x, void* attn_output, void* o_proj, int height)
{
    q4_attn_2_cuda(&g_tuning, blas(), (half*) x, (half*) attn_output, (Q4Matrix*) o_proj, height, nullptr, nullptr, 0, nullptr);
    return finish();
}

int ref_cleanup()
{
    cleanup_buffers_cuda();
    g_q4_free_matrices();
    return finish();
}

}   // extern "C"
