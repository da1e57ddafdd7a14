// HBM-bound glue kernels: RMSNorm, RoPE, SiLU*mul, KV-cache scatter.  One pass each, 128-bit accesses,
// wave64 reductions, no atomics, no zeroed scratch.
//
// Behavioural reference (arithmetic contract):
//   /root/reference/exllama_ext/cuda_func/rms_norm.cu:21-152   (two kernels + fp32 atomics there)
//   /root/reference/exllama_ext/cuda_func/rope.cu:21-88
//   /root/reference/exllama_ext/cuda_func/q4_mlp.cu:16-88      silu / silu_mul_cuda_kernel
//   /root/reference/exllama_ext/cuda_func/q4_attn.cu:19-72     update_cache_kernel
#include "common.h"

__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// ---------------------------------------------------------------------------------------------------
// RMSNorm: out = h( h(x * h(rsqrt(mean(x^2) + eps))) * w ).  One 256-thread block per row; the row is kept
// in registers between the reduction and the scaling (dim <= 256*8*MAXV), so x is read from HBM once.
// ---------------------------------------------------------------------------------------------------
template <int MAXV>
__global__ __launch_bounds__(256) void rms_norm_kernel(const f16* __restrict__ x, const f16* __restrict__ w,
                                                       f16* __restrict__ out, float eps, int dim)
{
    __shared__ float red[4];
    const int row = blockIdx.x;
    const int tid = threadIdx.x;
    const f16* xr = x + (size_t) row * dim;
    f16* orow = out + (size_t) row * dim;
    const int nvec = dim >> 3;                              // dim % 8 == 0 (checked by the launcher)

    f16x8 v[MAXV];
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int idx = tid + i * 256;
        if (idx < nvec) {
            v[i] = *(const f16x8*) (xr + idx * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float f = (float) v[i][j]; acc = fmaf(f, f, acc); }
        }
    }
    acc = wave_sum(acc);
    if ((tid & 63) == 0) red[tid >> 6] = acc;
    __syncthreads();
    const float total = red[0] + red[1] + red[2] + red[3];
    const float rmf = 1.0f / sqrtf(total * (1.0f / (float) dim) + eps);
    const f16 rm = (f16) rmf;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int idx = tid + i * 256;
        if (idx < nvec) {
            const f16x8 wv = *(const f16x8*) (w + idx * 8);
            f16x8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const f16 m = v[i][j] * rm;              // fp16 multiply, rounded
                o[j] = m * wv[j];                        // second fp16 multiply, rounded
            }
            *(f16x8*) (orow + idx * 8) = o;
        }
    }
}

int launch_rms_norm(const f16* x, const f16* w, f16* out, float eps, int rows, int dim, hipStream_t s)
{
    if (rows <= 0) return 0;
    EXL_REQUIRE(dim % 8 == 0, EXL_E_UNSUPPORTED, "rms_norm: dim (%d) must be a multiple of 8", dim);
    EXL_REQUIRE(dim <= 256 * 8 * 16, EXL_E_UNSUPPORTED, "rms_norm: dim (%d) > 32768 unsupported", dim);
    const int nvec = dim / 8;
    const int per = (nvec + 255) / 256;
    if (per <= 2)      hipLaunchKernelGGL(rms_norm_kernel<2>,  dim3(rows), dim3(256), 0, s, x, w, out, eps, dim);
    else if (per <= 4) hipLaunchKernelGGL(rms_norm_kernel<4>,  dim3(rows), dim3(256), 0, s, x, w, out, eps, dim);
    else if (per <= 8) hipLaunchKernelGGL(rms_norm_kernel<8>,  dim3(rows), dim3(256), 0, s, x, w, out, eps, dim);
    else               hipLaunchKernelGGL(rms_norm_kernel<16>, dim3(rows), dim3(256), 0, s, x, w, out, eps, dim);
    EXL_LAUNCH_CHECK();
    return 0;
}

// RMSNorm whose output row is written in an act-order matrix' row order: out[row][c] = norm(x)[row][x_map[c]] -- the
// reference's rms_norm followed by column_remap (rms_norm.cu + column_remap.cu:7-36) as ONE pass over x: the normalised row goes
// through LDS (dim * 2 bytes), the gather reads LDS, the stores stay 16-byte coalesced.  Same bits as the two kernels.
template <int MAXV>
__global__ __launch_bounds__(256) void rms_norm_gather_kernel(const f16* __restrict__ x, const f16* __restrict__ w,
                                                              f16* __restrict__ out, const uint32_t* __restrict__ x_map,
                                                              float eps, int dim)
{
    extern __shared__ __attribute__((aligned(16))) f16 nrow[];
    __shared__ float red[4];
    const int row = blockIdx.x;
    const int tid = threadIdx.x;
    const f16* xr = x + (size_t) row * dim;
    f16* orow = out + (size_t) row * dim;
    const int nvec = dim >> 3;

    f16x8 v[MAXV];
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int idx = tid + i * 256;
        if (idx < nvec) {
            v[i] = *(const f16x8*) (xr + idx * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float f = (float) v[i][j]; acc = fmaf(f, f, acc); }
        }
    }
    acc = wave_sum(acc);
    if ((tid & 63) == 0) red[tid >> 6] = acc;
    __syncthreads();
    const float total = red[0] + red[1] + red[2] + red[3];
    const float rmf = 1.0f / sqrtf(total * (1.0f / (float) dim) + eps);
    const f16 rm = (f16) rmf;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int idx = tid + i * 256;
        if (idx < nvec) {
            const f16x8 wv = *(const f16x8*) (w + idx * 8);
            f16x8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const f16 m = v[i][j] * rm;
                o[j] = m * wv[j];
            }
            *(f16x8*) (nrow + idx * 8) = o;
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int idx = tid + i * 256;
        if (idx < nvec) {
            const uint4 m0 = *(const uint4*) (x_map + idx * 8);
            const uint4 m1 = *(const uint4*) (x_map + idx * 8 + 4);
            f16x8 o;
            o[0] = nrow[m0.x]; o[1] = nrow[m0.y]; o[2] = nrow[m0.z]; o[3] = nrow[m0.w];
            o[4] = nrow[m1.x]; o[5] = nrow[m1.y]; o[6] = nrow[m1.z]; o[7] = nrow[m1.w];
            *(f16x8*) (orow + idx * 8) = o;
        }
    }
}

int launch_rms_norm_gather(const f16* x, const f16* w, f16* out, const uint32_t* x_map, float eps, int rows, int dim, hipStream_t s)
{
    if (!x_map) return launch_rms_norm(x, w, out, eps, rows, dim, s);
    if (rows <= 0) return 0;
    EXL_REQUIRE(dim % 8 == 0 && dim <= 256 * 8 * 16, EXL_E_UNSUPPORTED, "rms_norm_gather: dim (%d) must be a multiple of 8, <= 32768", dim);
    const int per = (dim / 8 + 255) / 256;
    const size_t lds = (size_t) dim * 2;                                 // <= 64 KiB
    if (per <= 2)      hipLaunchKernelGGL(rms_norm_gather_kernel<2>,  dim3(rows), dim3(256), lds, s, x, w, out, x_map, eps, dim);
    else if (per <= 4) hipLaunchKernelGGL(rms_norm_gather_kernel<4>,  dim3(rows), dim3(256), lds, s, x, w, out, x_map, eps, dim);
    else if (per <= 8) hipLaunchKernelGGL(rms_norm_gather_kernel<8>,  dim3(rows), dim3(256), lds, s, x, w, out, x_map, eps, dim);
    else               hipLaunchKernelGGL(rms_norm_gather_kernel<16>, dim3(rows), dim3(256), lds, s, x, w, out, x_map, eps, dim);
    EXL_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------------
// RoPE (rotate-half), in place.  One thread = 8 columns of the left half + the matching 8 of the right half.
//   l' = h(fma(l, cos_l, h(r * h(-sin_l))))      r' = h(fma(r, cos_r, h(l * sin_r)))
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rope_kernel(f16* __restrict__ x, const f16* __restrict__ sin,
                                                   const f16* __restrict__ cos, int rows_per_batch, int head_dim,
                                                   int num_heads, int past_len, const int32_t* __restrict__ past_len_dev,
                                                   int total_rows)
{
    const int vec_per_row = head_dim >> 4;                  // (head_dim / 2) / 8
    const int gid = blockIdx.x * 256 + threadIdx.x;
    const int grow = gid / vec_per_row;                     // global row over bsz * rows_per_batch
    if (grow >= total_rows) return;
    const int v = gid - grow * vec_per_row;
    const int row = grow % rows_per_batch;
    const int past = past_len_dev ? *past_len_dev : past_len;
    const int pos = past + row / num_heads;
    const int hd2 = head_dim >> 1;

    f16* xp = x + (size_t) grow * head_dim + v * 8;
    const f16* sp = sin + (size_t) pos * head_dim + v * 8;
    const f16* cp = cos + (size_t) pos * head_dim + v * 8;
    const f16x8 l = *(const f16x8*) xp;
    const f16x8 r = *(const f16x8*) (xp + hd2);
    const f16x8 sl = *(const f16x8*) sp;
    const f16x8 sr = *(const f16x8*) (sp + hd2);
    const f16x8 cl = *(const f16x8*) cp;
    const f16x8 cr = *(const f16x8*) (cp + hd2);
    f16x8 nl, nr;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const f16 ls = r[j] * (-sl[j]);
        const f16 rs = l[j] * sr[j];
        nl[j] = __builtin_fmaf16(l[j], cl[j], ls);
        nr[j] = __builtin_fmaf16(r[j], cr[j], rs);
    }
    *(f16x8*) xp = nl;
    *(f16x8*) (xp + hd2) = nr;
}

// Any even head_dim (OpenLLaMA-3B: 100 -- the reference's rope.cu:27-87 takes any even width): one thread = ONE pair (l, r), the same
// arithmetic; 2-byte accesses, used only where the 16-byte form above cannot be (head_dim % 16 != 0).
__global__ __launch_bounds__(256) void rope_pair_kernel(f16* __restrict__ x, const f16* __restrict__ sin, const f16* __restrict__ cos,
                                                        int rows_per_batch, int head_dim, int num_heads, int past_len,
                                                        const int32_t* __restrict__ past_len_dev, long total)
{
    const long gid = (long) blockIdx.x * 256 + threadIdx.x;
    if (gid >= total) return;
    const int hd2 = head_dim >> 1;
    const long grow = gid / hd2;
    const int i = (int) (gid - grow * hd2);
    const int row = (int) (grow % rows_per_batch);
    const int past = past_len_dev ? *past_len_dev : past_len;
    const int pos = past + row / num_heads;
    f16* xp = x + (size_t) grow * head_dim;
    const f16* sp = sin + (size_t) pos * head_dim;
    const f16* cp = cos + (size_t) pos * head_dim;
    const f16 l = xp[i], r = xp[i + hd2];
    const f16 ls = r * (-sp[i]);
    const f16 rs = l * sp[i + hd2];
    xp[i] = __builtin_fmaf16(l, cp[i], ls);
    xp[i + hd2] = __builtin_fmaf16(r, cp[i + hd2], rs);
}

int launch_rope(f16* x, const f16* sin, const f16* cos, int bsz, int rows_per_batch, int head_dim, int num_heads,
                int past_len, const int32_t* past_len_dev, hipStream_t s)
{
    const int total_rows = bsz * rows_per_batch;
    if (total_rows <= 0) return 0;
    EXL_REQUIRE(head_dim > 0 && head_dim % 2 == 0, EXL_E_UNSUPPORTED, "rope: head_dim (%d) must be even", head_dim);
    EXL_REQUIRE(num_heads > 0, EXL_E_INVALID, "rope: num_heads must be > 0");
    if (head_dim % 16 != 0) {
        const long pairs = (long) total_rows * (head_dim / 2);
        hipLaunchKernelGGL(rope_pair_kernel, dim3((unsigned) ((pairs + 255) / 256)), dim3(256), 0, s, x, sin, cos, rows_per_batch,
                           head_dim, num_heads, past_len, past_len_dev, pairs);
        EXL_LAUNCH_CHECK();
        return 0;
    }
    const long total = (long) total_rows * (head_dim / 16);
    hipLaunchKernelGGL(rope_kernel, dim3((unsigned) ((total + 255) / 256)), dim3(256), 0, s, x, sin, cos, rows_per_batch,
                       head_dim, num_heads, past_len, past_len_dev, total_rows);
    EXL_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------------
// SiLU(x) * y in fp16 steps: e = h(exp(-x)); s = h(1 + e); rc = h(1 / s); out = h(h(x * rc) * y)
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ f16 silu_mul_h(f16 x, f16 y)
{
    const f16 e = (f16) __expf((float) (f16) (-x));
    const f16 sm = (f16) 1.0f + e;
    const f16 rc = (f16) (1.0f / (float) sm);
    const f16 v = x * rc;
    return v * y;
}

__global__ __launch_bounds__(256) void silu_mul_kernel(f16* __restrict__ x, const f16* __restrict__ y, long nvec)
{
    const long i = (long) blockIdx.x * 256 + threadIdx.x;
    if (i >= nvec) return;
    const f16x8 a = *(const f16x8*) (x + i * 8);
    const f16x8 b = *(const f16x8*) (y + i * 8);
    f16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = silu_mul_h(a[j], b[j]);
    *(f16x8*) (x + i * 8) = o;
}

int launch_silu_mul(f16* x, const f16* y, int height, int width, hipStream_t s)
{
    const long n = (long) height * width;
    if (n <= 0) return 0;
    EXL_REQUIRE(n % 8 == 0, EXL_E_UNSUPPORTED, "silu_mul: height*width must be a multiple of 8");
    const long nvec = n / 8;
    hipLaunchKernelGGL(silu_mul_kernel, dim3((unsigned) ((nvec + 255) / 256)), dim3(256), 0, s, x, y, nvec);
    EXL_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------------
// KV scatter: cache[b, h, past + t, :] = state[b, t, h, :]   (bit copy, 128-bit)
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void update_cache_kernel(const f16* __restrict__ k, const f16* __restrict__ v,
                                                           f16* __restrict__ kc, f16* __restrict__ vc, int q_len,
                                                           int kvh, int hd, int max_seq, int past_len,
                                                           const int32_t* __restrict__ past_len_dev, long total_vec)
{
    const long gid = (long) blockIdx.x * 256 + threadIdx.x;
    if (gid >= total_vec) return;
    const int vec_per_head = hd >> 3;
    const int d8 = (int) (gid % vec_per_head);
    long rest = gid / vec_per_head;
    const int h = (int) (rest % kvh); rest /= kvh;
    const int t = (int) (rest % q_len);
    const int b = (int) (rest / q_len);
    const int past = past_len_dev ? *past_len_dev : past_len;
    const size_t src = (((size_t) b * q_len + t) * kvh + h) * hd + d8 * 8;
    const size_t dst = (((size_t) b * kvh + h) * max_seq + (past + t)) * hd + d8 * 8;
    *(uint4*) (kc + dst) = *(const uint4*) (k + src);
    *(uint4*) (vc + dst) = *(const uint4*) (v + src);
}

// ---------------------------------------------------------------------------------------------------
// q4_attn's tail in ONE launch (reference: rope_cuda on q, rope_cuda on k, update_cache_kernel; q4_attn.cu:160-204): RoPE on
// the q heads in place, RoPE on the k heads in place AND into the cache row, v into the cache row.  One thread = the 8 + 8
// paired columns of one head of one token; same arithmetic as rope_kernel, same bytes as update_cache_kernel.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rope_qk_cache_kernel(f16* __restrict__ q, f16* __restrict__ k, const f16* __restrict__ v,
                                                            f16* __restrict__ kc, f16* __restrict__ vc,
                                                            const f16* __restrict__ sin, const f16* __restrict__ cos, int q_len,
                                                            int heads, int kvh, int hd, int max_seq, int past_len,
                                                            const int32_t* __restrict__ past_len_dev, long total)
{
    const long gid = (long) blockIdx.x * 256 + threadIdx.x;
    if (gid >= total) return;
    const int vec_per_row = hd >> 4;
    const int vv = (int) (gid % vec_per_row);
    long rest = gid / vec_per_row;
    const int slot = (int) (rest % (heads + kvh)); rest /= (heads + kvh);
    const int t = (int) (rest % q_len);
    const int b = (int) (rest / q_len);
    const int past = past_len_dev ? *past_len_dev : past_len;
    const int pos = past + t;
    const int hd2 = hd >> 1;
    const bool is_k = slot >= heads;
    const int h = is_k ? slot - heads : slot;
    f16* xp = (is_k ? k + (((size_t) b * q_len + t) * kvh + h) * hd : q + (((size_t) b * q_len + t) * heads + h) * hd) + vv * 8;
    const f16* sp = sin + (size_t) pos * hd + vv * 8;
    const f16* cp = cos + (size_t) pos * hd + vv * 8;
    const f16x8 l = *(const f16x8*) xp;
    const f16x8 r = *(const f16x8*) (xp + hd2);
    const f16x8 sl = *(const f16x8*) sp;
    const f16x8 sr = *(const f16x8*) (sp + hd2);
    const f16x8 cl = *(const f16x8*) cp;
    const f16x8 cr = *(const f16x8*) (cp + hd2);
    f16x8 nl, nr;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const f16 ls = r[j] * (-sl[j]);
        const f16 rs = l[j] * sr[j];
        nl[j] = __builtin_fmaf16(l[j], cl[j], ls);
        nr[j] = __builtin_fmaf16(r[j], cr[j], rs);
    }
    *(f16x8*) xp = nl;
    *(f16x8*) (xp + hd2) = nr;
    if (is_k) {
        const size_t src = (((size_t) b * q_len + t) * kvh + h) * hd + vv * 8;
        const size_t dst = (((size_t) b * kvh + h) * max_seq + pos) * hd + vv * 8;
        *(f16x8*) (kc + dst) = nl;
        *(f16x8*) (kc + dst + hd2) = nr;
        *(uint4*) (vc + dst) = *(const uint4*) (v + src);
        *(uint4*) (vc + dst + hd2) = *(const uint4*) (v + src + hd2);
    }
}

int launch_update_cache(const f16* k, const f16* v, f16* kc, f16* vc, int bsz, int q_len, int kvh, int hd,
                        int max_seq, int past_len, const int32_t* past_len_dev, hipStream_t s);

int launch_rope_qk_cache(f16* q, f16* k, const f16* v, f16* kc, f16* vc, const f16* sin, const f16* cos, int bsz, int q_len,
                         int heads, int kvh, int hd, int max_seq, int past_len, const int32_t* past_len_dev, hipStream_t s)
{
    EXL_REQUIRE(hd > 0 && hd % 2 == 0, EXL_E_UNSUPPORTED, "rope: head_dim (%d) must be even", hd);
    EXL_REQUIRE(heads > 0 && kvh > 0, EXL_E_INVALID, "rope: num_heads must be > 0");
    if (hd % 16 != 0) {                                               // the three reference steps (q4_attn.cu:160-204) with the any-width kernels
        EXL_TRY(launch_rope(q, sin, cos, bsz, q_len * heads, hd, heads, past_len, past_len_dev, s));
        EXL_TRY(launch_rope(k, sin, cos, bsz, q_len * kvh, hd, kvh, past_len, past_len_dev, s));
        return launch_update_cache(k, v, kc, vc, bsz, q_len, kvh, hd, max_seq, past_len, past_len_dev, s);
    }
    const long total = (long) bsz * q_len * (heads + kvh) * (hd / 16);
    if (total <= 0) return 0;
    hipLaunchKernelGGL(rope_qk_cache_kernel, dim3((unsigned) ((total + 255) / 256)), dim3(256), 0, s, q, k, v, kc, vc, sin, cos,
                       q_len, heads, kvh, hd, max_seq, past_len, past_len_dev, total);
    EXL_LAUNCH_CHECK();
    return 0;
}

__global__ __launch_bounds__(256) void update_cache_pair_kernel(const f16* __restrict__ k, const f16* __restrict__ v, f16* __restrict__ kc,
                                                                f16* __restrict__ vc, int q_len, int kvh, int hd, int max_seq, int past_len,
                                                                const int32_t* __restrict__ past_len_dev, long total)
{
    const long gid = (long) blockIdx.x * 256 + threadIdx.x;
    if (gid >= total) return;
    const int per_head = hd >> 1;
    const int d2 = (int) (gid % per_head);
    long rest = gid / per_head;
    const int h = (int) (rest % kvh); rest /= kvh;
    const int t = (int) (rest % q_len);
    const int b = (int) (rest / q_len);
    const int past = past_len_dev ? *past_len_dev : past_len;
    const size_t src = (((size_t) b * q_len + t) * kvh + h) * hd + d2 * 2;
    const size_t dst = (((size_t) b * kvh + h) * max_seq + (past + t)) * hd + d2 * 2;
    *(uint32_t*) (kc + dst) = *(const uint32_t*) (k + src);
    *(uint32_t*) (vc + dst) = *(const uint32_t*) (v + src);
}

int launch_update_cache(const f16* k, const f16* v, f16* kc, f16* vc, int bsz, int q_len, int kvh, int hd,
                        int max_seq, int past_len, const int32_t* past_len_dev, hipStream_t s)
{
    EXL_REQUIRE(hd > 0 && hd % 2 == 0, EXL_E_UNSUPPORTED, "update_cache: head_dim (%d) must be even", hd);
    if (hd % 8 != 0) {                                                // 4-byte copies (q4_attn.cu:19-72 copies element pairs too)
        const long pairs = (long) bsz * q_len * kvh * (hd / 2);
        if (pairs <= 0) return 0;
        hipLaunchKernelGGL(update_cache_pair_kernel, dim3((unsigned) ((pairs + 255) / 256)), dim3(256), 0, s, k, v, kc, vc, q_len,
                           kvh, hd, max_seq, past_len, past_len_dev, pairs);
        EXL_LAUNCH_CHECK();
        return 0;
    }
    const long total = (long) bsz * q_len * kvh * (hd / 8);
    if (total <= 0) return 0;
    hipLaunchKernelGGL(update_cache_kernel, dim3((unsigned) ((total + 255) / 256)), dim3(256), 0, s, k, v, kc, vc, q_len,
                       kvh, hd, max_seq, past_len, past_len_dev, total);
    EXL_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------------
// Embedding lookup and the prompt pass' lm_head for a handful of rows (reference: torch ops, model.py:1002 embed_tokens and
// :1077 lm_head -- ATen / cuBLAS there, ATen / hipBLASLt here until round 3; the decode executor has its own fused forms).
//   embedding: one block per token, 16-byte copies of the table row.
//   head: one wave per vocabulary row (128-bit streaming loads of the fp16 matrix: read exactly once), the activation rows
//         staged in LDS; fp32 accumulate, rounded to fp16 like nn.Linear in fp16, returned as fp32 (model.py:1077-1080).
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void embedding_kernel(const int64_t* __restrict__ ids, const f16* __restrict__ table, f16* __restrict__ out,
                                                        int hidden, int vocab)
{
    const int64_t id = ids[blockIdx.x];
    const int64_t row = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);          // (torch raises on the host; here: clamp, never fault)
    const uint4* src = (const uint4*) (table + (size_t) row * hidden);
    uint4* dst = (uint4*) (out + (size_t) blockIdx.x * hidden);
    for (int i = threadIdx.x; i < hidden / 8; i += 256) dst[i] = src[i];
}

int launch_embedding(const int64_t* ids, const f16* table, f16* out, int n_ids, int hidden, int vocab, hipStream_t s)
{
    if (n_ids <= 0) return 0;
    EXL_REQUIRE(hidden % 8 == 0, EXL_E_UNSUPPORTED, "embedding: hidden size (%d) must be a multiple of 8", hidden);
    hipLaunchKernelGGL(embedding_kernel, dim3(n_ids), dim3(256), 0, s, ids, table, out, hidden, vocab);
    EXL_LAUNCH_CHECK();
    return 0;
}

#define HEAD_MAX_ROWS 8
template <int ROWS>
__global__ __launch_bounds__(256) void head_rows_kernel(const f16* __restrict__ x, const f16* __restrict__ w, float* __restrict__ out,
                                                        int hidden, int vocab, int rows_per_block)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint4* xs = (uint4*) smem;                                               // [ROWS][hidden / 8]
    const int nvec = hidden >> 3;
    for (int i = threadIdx.x; i < ROWS * nvec; i += 256) xs[i] = ((const uint4*) x)[i];
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int row0 = blockIdx.x * rows_per_block;
    for (int r = wave; r < rows_per_block; r += 4) {
        const int v = row0 + r;
        if (v >= vocab) break;
        const f16* wr = w + (size_t) v * hidden;
        float acc[ROWS];
#pragma unroll
        for (int m = 0; m < ROWS; ++m) acc[m] = 0.f;
        for (int i = lane; i < nvec; i += 64) {
            const uint4 wv = nt_load16(wr + i * 8);
#pragma unroll
            for (int m = 0; m < ROWS; ++m) {
                const uint4 xv = xs[m * nvec + i];
                acc[m] = __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2, wv.x), __builtin_bit_cast(f16x2, xv.x), acc[m], false);
                acc[m] = __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2, wv.y), __builtin_bit_cast(f16x2, xv.y), acc[m], false);
                acc[m] = __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2, wv.z), __builtin_bit_cast(f16x2, xv.z), acc[m], false);
                acc[m] = __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2, wv.w), __builtin_bit_cast(f16x2, xv.w), acc[m], false);
            }
        }
#pragma unroll
        for (int m = 0; m < ROWS; ++m) {
            float a = acc[m];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) a += __shfl_xor(a, off, 64);
            if (lane == 0) out[(size_t) m * vocab + v] = (float) (f16) a;
        }
    }
}

// 1 = not covered (more rows than the kernel stages: the caller uses its BLAS path)
int launch_head_rows(const f16* x, const f16* w, float* out, int rows, int hidden, int vocab, hipStream_t s)
{
    if (rows <= 0) return 0;
    if (rows > HEAD_MAX_ROWS || hidden % 8 != 0 || (size_t) rows * hidden * 2 > 64 * 1024) return 1;     // the rows it stages must fit 64 KiB of LDS
    const int rpb = 32;
    const dim3 grid((vocab + rpb - 1) / rpb);
    const size_t smem = (size_t) rows * hidden * 2;
    switch (rows) {
#define HEAD_CASE(R) case R: hipLaunchKernelGGL(head_rows_kernel<R>, grid, dim3(256), smem, s, x, w, out, hidden, vocab, rpb); break;
        HEAD_CASE(1) HEAD_CASE(2) HEAD_CASE(3) HEAD_CASE(4) HEAD_CASE(5) HEAD_CASE(6) HEAD_CASE(7) HEAD_CASE(8)
#undef HEAD_CASE
    }
    EXL_LAUNCH_CHECK();
    return 0;
}
