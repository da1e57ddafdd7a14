// fp16 attention over the KV cache, CDNA4 HIP.  Replaces the ATen call chains of the reference:
//   /root/reference/model.py:376-409  (fused decode: matmul, /= sqrt(hd), fp16 softmax, matmul; repeat_kv copies)
//   /root/reference/model.py:463-495  (short q_len: same + additive mask; long q_len: F.scaled_dot_product_attention)
//
// Kernel A (this file): split-KV "flash-decoding" for small q_len.  Grid = (splits, heads, bsz*q_len); every
// block streams a contiguous slice of K then V for ONE (query row, head) with 128-bit loads (hd/8 lanes per key
// row), keeps scores in LDS, and emits an un-normalised partial (o[hd], m, l) in fp32; a second tiny kernel
// merges the partials.  GQA is an index (head -> head / group), no repeat_kv copies.  Softmax statistics and
// accumulation are fp32; the output is rounded once to fp16.
// Kernel B (flash_prefill.hip): MFMA flash attention for long q_len.
#include "common.h"

#define ATT_THREADS 256
#define ATT_MAX_SPLIT_KEYS 1024

// VW: halves per lane and access -- 8 (16-byte loads) where head_dim % 8 == 0, else 4 (head_dim % 4 == 0: OpenLLaMA-3B's 100, whose
// rows are only 8-byte aligned)
template <int VW> struct AttVec;
template <> struct AttVec<8> { typedef f16x8 T; };
template <> struct AttVec<4> { typedef f16x4 T; };
template <int LPK, int VW = 8>   // lanes per key row = head_dim / VW rounded up to a power of two (8, 16, 32, 64)
__global__ __launch_bounds__(ATT_THREADS) void attn_decode_kernel(
    const f16* __restrict__ q, const f16* __restrict__ kc, const f16* __restrict__ vc, const f16* __restrict__ mask,
    float* __restrict__ partial, f16* __restrict__ out, int q_len, int heads, int kv_heads, int hd, int max_seq,
    int past_len, const int32_t* __restrict__ past_len_dev, int nsplit, float scale, int bq_base)
{
    constexpr int KPI = ATT_THREADS / LPK;                    // keys per block iteration
    __shared__ float sc[ATT_MAX_SPLIT_KEYS];
    __shared__ float red[KPI][VW * LPK + 1];
    typedef typename AttVec<VW>::T vec_t;
    __shared__ float stat[8];

    const int split = blockIdx.x;
    const int h = blockIdx.y;
    const int bq = blockIdx.z + bq_base;                      // query rows are processed in chunks sized to the workspace
    const int b = bq / q_len;
    const int qi = bq - b * q_len;
    const int past = past_len_dev ? *past_len_dev : past_len;
    const int kv_len = past + q_len;
    const int vis = min(kv_len, past + qi + 1);               // causal: keys 0 .. past+qi
    int L = (vis + nsplit - 1) / nsplit;
    L = (L + 15) & ~15;
    const int s0 = min(vis, split * L);
    const int s1 = min(vis, s0 + L);
    const int nkeys = s1 - s0;

    const int tid = threadIdx.x;
    const int d8 = tid % LPK;
    const int ks = tid / LPK;
    const bool d_ok = d8 * VW < hd;
    const int kvh = h / (heads / kv_heads);
    const f16* kbase = kc + (((size_t) b * kv_heads + kvh) * max_seq) * hd + d8 * VW;
    const f16* vbase = vc + (((size_t) b * kv_heads + kvh) * max_seq) * hd + d8 * VW;
    const f16* mrow = mask ? mask + ((size_t) b * q_len + qi) * kv_len : nullptr;

    float qf[VW];
    {
        vec_t qv = {};
        if (d_ok) qv = *(const vec_t*) (q + ((size_t) bq * heads + h) * hd + d8 * VW);
#pragma unroll
        for (int j = 0; j < VW; ++j) qf[j] = (float) qv[j] * scale;
    }

    // ---- pass 1: scores -> LDS, block max ---------------------------------------------------------
    float mx = -INFINITY;
    for (int j0 = 0; j0 < nkeys; j0 += KPI) {
        const int j = j0 + ks;
        float dot = 0.f;
        if (j < nkeys && d_ok) {
            const vec_t kv = *(const vec_t*) (kbase + (size_t) (s0 + j) * hd);
#pragma unroll
            for (int e = 0; e < VW; ++e) dot = fmaf(qf[e], (float) kv[e], dot);
        }
#pragma unroll
        for (int off = 1; off < LPK; off <<= 1) dot += __shfl_xor(dot, off, 64);
        if (j < nkeys) {
            if (mrow) dot += (float) mrow[s0 + j];
            if (d8 == 0) sc[j] = dot;
            mx = fmaxf(mx, dot);
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
    if ((tid & 63) == 0) stat[tid >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(stat[0], stat[1]), fmaxf(stat[2], stat[3]));
    __syncthreads();

    // ---- pass 2: p = exp(s - m) in place, block sum ---------------------------------------------------
    float lsum = 0.f;
    for (int j = tid; j < nkeys; j += ATT_THREADS) {
        const float p = __expf(sc[j] - mx);
        sc[j] = p;
        lsum += p;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) lsum += __shfl_xor(lsum, off, 64);
    if ((tid & 63) == 0) stat[4 + (tid >> 6)] = lsum;
    __syncthreads();
    lsum = stat[4] + stat[5] + stat[6] + stat[7];

    // ---- pass 3: o[d] = sum_j p_j * v[j][d] -------------------------------------------------------------
    float o[VW];
#pragma unroll
    for (int e = 0; e < VW; ++e) o[e] = 0.f;
    if (d_ok) {
        for (int j = ks; j < nkeys; j += KPI) {
            const vec_t vv = *(const vec_t*) (vbase + (size_t) (s0 + j) * hd);
            const float p = sc[j];
#pragma unroll
            for (int e = 0; e < VW; ++e) o[e] = fmaf(p, (float) vv[e], o[e]);
        }
    }
#pragma unroll
    for (int e = 0; e < VW; ++e) red[ks][d8 * VW + e] = o[e];
    __syncthreads();
    for (int d = tid; d < hd; d += ATT_THREADS) {
        float v = 0.f;
#pragma unroll 4
        for (int s = 0; s < KPI; ++s) v += red[s][d];
        if (nsplit == 1) {
            out[((size_t) bq * heads + h) * hd + d] = (f16) (nkeys > 0 ? v / lsum : 0.f);
        } else {
            partial[(((size_t) blockIdx.z * heads + h) * nsplit + split) * (hd + 2) + d] = v;
        }
    }
    if (nsplit > 1 && tid == 0) {
        float* pp = partial + (((size_t) blockIdx.z * heads + h) * nsplit + split) * (hd + 2) + hd;
        pp[0] = nkeys > 0 ? mx : -INFINITY;
        pp[1] = nkeys > 0 ? lsum : 0.f;
    }
}

__global__ __launch_bounds__(128) void attn_combine_kernel(const float* __restrict__ partial, f16* __restrict__ out,
                                                           int hd, int nsplit)
{
    const size_t row = blockIdx.x;                            // (bq * heads + h)
    const float* pp = partial + row * nsplit * (hd + 2);
    float M = -INFINITY;
    for (int s = 0; s < nsplit; ++s) M = fmaxf(M, pp[s * (hd + 2) + hd]);
    float l = 0.f;
    for (int s = 0; s < nsplit; ++s) {
        const float ms = pp[s * (hd + 2) + hd];
        if (ms > -INFINITY) l += pp[s * (hd + 2) + hd + 1] * __expf(ms - M);
    }
    for (int d = threadIdx.x; d < hd; d += 128) {
        float o = 0.f;
        for (int s = 0; s < nsplit; ++s) {
            const float ms = pp[s * (hd + 2) + hd];
            if (ms > -INFINITY) o += pp[s * (hd + 2) + d] * __expf(ms - M);
        }
        out[row * hd + d] = (f16) (l > 0.f ? o / l : 0.f);
    }
}

int launch_flash_prefill(const f16* q, const f16* kc, const f16* vc, f16* out, int bsz, int q_len, int heads,
                         int kv_heads, int hd, int max_seq, int past_len, hipStream_t s, int frag);

int launch_attention(const f16* q, const f16* kc, const f16* vc, f16* out, const f16* mask, int bsz, int q_len,
                     int heads, int kv_heads, int hd, int max_seq, int past_len, const int32_t* past_len_dev,
                     float* ws, size_t ws_floats, hipStream_t s)
{
    if (bsz <= 0 || q_len <= 0) return 0;
    EXL_REQUIRE(hd % 4 == 0 && hd > 0 && hd <= 256, EXL_E_UNSUPPORTED, "attention: head_dim (%d) must be a multiple of 4, <= 256", hd);
    EXL_REQUIRE(kv_heads > 0 && heads % kv_heads == 0, EXL_E_INVALID, "attention: heads (%d) %% kv_heads (%d) != 0", heads, kv_heads);
    EXL_REQUIRE(past_len + q_len <= max_seq, EXL_E_INVALID, "attention: past_len + q_len (%d) exceeds max_seq_len (%d)",
                past_len + q_len, max_seq);

    // long prompts without an explicit mask: MFMA flash kernel
    if (q_len >= 16 && !mask && !past_len_dev && hd == 128)
        return launch_flash_prefill(q, kc, vc, out, bsz, q_len, heads, kv_heads, hd, max_seq, past_len, s, 0);

    // grid sizing uses the host position; with a device-side position it is an upper bound
    const int kv_max = past_len_dev ? max_seq : past_len + q_len;
    const long base = (long) bsz * q_len * heads;
    int nsplit = (int) ((512 + base - 1) / base);
    const int max_by_len = (kv_max + 63) / 64;
    if (nsplit > max_by_len) nsplit = max_by_len;
    const int min_by_lds = (kv_max + ATT_MAX_SPLIT_KEYS - 16 - 1) / (ATT_MAX_SPLIT_KEYS - 16);
    if (nsplit < min_by_lds) nsplit = min_by_lds;
    if (nsplit < 1) nsplit = 1;
    // The partials of all (row, head, split) triples of a launch live in the fixed workspace: long masked prompts (batched
    // generation with padding, model.py:1014-1033) are processed in chunks of query rows that fit it, instead of failing
    // (bsz 1 x 2048 masked tokens with 32 heads would need 25.5 M floats of a 16 M-float workspace in one go).
    const int total_rows = bsz * q_len;
    int chunk = total_rows;
    float* partial = nullptr;
    if (nsplit > 1) {
        const size_t per_row = (size_t) heads * nsplit * (hd + 2);
        EXL_REQUIRE(ws && ws_floats >= per_row, EXL_E_TOO_SMALL, "attention: workspace too small (%zu < %zu floats)", ws_floats, per_row);
        const size_t fit = ws_floats / per_row;
        if ((size_t) chunk > fit) chunk = (int) fit;
        partial = ws;
    }
    if (chunk > 65535) chunk = 65535;                             // gridDim.z limit
    const float scale = 1.0f / sqrtf((float) hd);
    const bool narrow = hd % 8 != 0;                                 // 8-byte accesses (VW = 4)
    const int lanes = narrow ? hd / 4 : hd / 8;
    const int lpk = lanes <= 8 ? 8 : lanes <= 16 ? 16 : lanes <= 32 ? 32 : 64;
    for (int r0 = 0; r0 < total_rows; r0 += chunk) {
        const int nr = total_rows - r0 < chunk ? total_rows - r0 : chunk;
        dim3 grid(nsplit, heads, nr);
#define ATT_ARGS q, kc, vc, mask, partial, out, q_len, heads, kv_heads, hd, max_seq, past_len, past_len_dev, nsplit, scale, r0
        if (narrow) {
            if (lpk <= 16)      hipLaunchKernelGGL((attn_decode_kernel<16, 4>), grid, dim3(ATT_THREADS), 0, s, ATT_ARGS);
            else if (lpk == 32) hipLaunchKernelGGL((attn_decode_kernel<32, 4>), grid, dim3(ATT_THREADS), 0, s, ATT_ARGS);
            else                hipLaunchKernelGGL((attn_decode_kernel<64, 4>), grid, dim3(ATT_THREADS), 0, s, ATT_ARGS);
        }
        else if (lpk == 8)  hipLaunchKernelGGL(attn_decode_kernel<8>,  grid, dim3(ATT_THREADS), 0, s, ATT_ARGS);
        else if (lpk == 16) hipLaunchKernelGGL(attn_decode_kernel<16>, grid, dim3(ATT_THREADS), 0, s, ATT_ARGS);
        else                hipLaunchKernelGGL(attn_decode_kernel<32>, grid, dim3(ATT_THREADS), 0, s, ATT_ARGS);
#undef ATT_ARGS
        EXL_LAUNCH_CHECK();
        if (nsplit > 1) {
            hipLaunchKernelGGL(attn_combine_kernel, dim3((unsigned) (nr * heads)), dim3(128), 0, s, partial, out + (size_t) r0 * heads * hd, hd, nsplit);
            EXL_LAUNCH_CHECK();
        }
    }
    return 0;
}
