// Native single-token decode executor: one C call (capturable into one hipGraph) runs a whole Llama token step
// as 5 kernels per layer + 1 head kernel, instead of the ~20 launches per layer of the op-by-op path.
//
// It is the MI355X answer to the reference's decode loop (/root/reference/model.py:1053-1058 driving
// q4_attn -> ATen attention -> q4_attn_2 -> q4_mlp, i.e. q4_attn.cu:74-228 + q4_mlp.cu:100-199 + model.py:376-409):
// same arithmetic contract per op (fp16 values at the same points: normed x, q/k/v,
// attention output, gate/up, activation, residual stream), different decomposition:
//
//   K1 norm_gemv   x = h(hid + sum(down slabs of the previous layer)) [layer 0: the embedding row];
//                  RMSNorm in LDS; q, k, v projections as ONE launch over the three matrices (fp16 out)
//   K2 attn        RoPE(q), RoPE(k_new) in registers, k_new/v_new appended to the cache, split-KV attention
//                  over the cache -> fp32 partials (o, m, l) per (head, split)
//   K3 vec_gemv    merges the attention partials while staging its activation slice, o_proj split-K -> fp32 slabs
//   K4 norm_gemv   x = h(hid + sum(o slabs)); RMSNorm; gate and up projections as one launch (fp16 out)
//   K5 vec_gemv    silu(g) * u computed while staging, down_proj split-K -> fp32 slabs
//   K6 head        x = h(hid + sum(down slabs)); final RMSNorm; fp16 lm_head GEMV -> fp32 logits; advances the position
//
// Split-K partial sums are never combined with atomics: each consumer adds the slabs in a fixed order in its
// prologue ("launch-boundary reduce"), so the result is bit-reproducible.  The position is read from device
// memory, so one captured graph serves every context length.
#include "gemv_core.h"

#include <vector>

#define DEC_MAX_MATS 3
#define DEC_ATT_MAX_KEYS 1024

struct ANormArgs {
    // source of the residual stream
    const f16* hid_in;            // [h] or the embedding table when tok != NULL
    const int64_t* tok;           // token id (layer 0) or NULL
    const float* slabs;           // [nslab][h] fp32 partial sums to add, or NULL
    int nslab;
    f16* hid_out;                 // written by block 0 (may alias nothing that is read in this launch)
    const f16* norm_w;
    float eps;
    int h;
    int nmat;
    GcMatrix mat[DEC_MAX_MATS];
    f16* out[DEC_MAX_MATS];
    int tile_end[DEC_MAX_MATS];   // cumulative 32-column tile counts
};

__device__ __forceinline__ float block_sum_256(float v, float* red4, int tid)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    if ((tid & 63) == 0) red4[tid >> 6] = v;
    __syncthreads();
    return red4[0] + red4[1] + red4[2] + red4[3];
}

// K1 / K4
// Load order matters: the few L2-resident prologue loads (x, slabs) are issued BEFORE the 16 streaming weight loads
// (loads retire in order, so anything queued behind the weight stream would wait for HBM), all in straight-line code
// so that hipcc can wait with a counted vmcnt while the weights are still in flight.
#define DEC_MAXS 4          // split-K slabs per consumer

template <int DEC_NV>       // 8-half vectors per thread: hidden <= 2048 * DEC_NV
__global__ __launch_bounds__(256) void dec_norm_gemv_kernel(const ANormArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    f16* xlin = (f16*) smem;                                  // [h]   (act-order only)
    uint4* xs = (uint4*) (smem + (size_t) a.h * 2);           // [h / 8]
    float* red = (float*) (smem + (size_t) a.h * 4);          // [4 * GC_BN] (+4 for the norm reduction)

    const int tid = threadIdx.x;
    const int tx = tid % GC_TX, ty = tid / GC_TX;
    int mi = 0, tile = blockIdx.x;
#pragma unroll
    for (int i = 0; i < DEC_MAX_MATS - 1; ++i)
        if (mi == i && i + 1 < a.nmat && (int) blockIdx.x >= a.tile_end[i]) { mi = i + 1; tile = blockIdx.x - a.tile_end[i]; }
    const GcMatrix& m = a.mat[mi];
    const int col = tile * GC_BN + tx * 4;
    const bool col_ok = col < m.N;

    // ---- 1. prologue loads: residual stream + slabs (clamped addresses instead of branches) -----------------
    const int nvec = a.h >> 3;
    const f16* src = a.tok ? a.hid_in + (size_t) (*a.tok) * a.h : a.hid_in;
    uint4 xraw[DEC_NV];
    float4 sl[DEC_NV][DEC_MAXS][2];
    const float* slab0 = a.slabs ? a.slabs : (const float*) src;       // never dereferenced as slabs when nslab == 0
#pragma unroll
    for (int i = 0; i < DEC_NV; ++i) {
        const int idx = tid + i * 256;
        const int ci = idx < nvec ? idx : 0;
        if (i * 256 < nvec) {                                            // uniform per kernel
            xraw[i] = *(const uint4*) (src + ci * 8);
#pragma unroll
            for (int s = 0; s < DEC_MAXS; ++s) {
                if (s < a.nslab) {                                       // uniform per kernel
                    sl[i][s][0] = *(const float4*) (slab0 + (size_t) s * a.h + ci * 8);
                    sl[i][s][1] = *(const float4*) (slab0 + (size_t) s * a.h + ci * 8 + 4);
                }
            }
        }
    }
    // ---- 2. weight stream ------------------------------------------------------------------------------------
    const GcPlan plan = gc_plan(0, m.K >> 3);
    uint4 wv[GC_MAXR];
    gc_issue(m, plan, 0, col, col_ok, ty, wv);

    // ---- 3. residual add, RMSNorm -----------------------------------------------------------------------------
    f16x8 xv[DEC_NV];
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < DEC_NV; ++i) {
        const int idx = tid + i * 256;
        if (i * 256 < nvec) {
            f16x8 v = __builtin_bit_cast(f16x8, xraw[i]);
            if (a.nslab > 0) {
                float f[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s = 0; s < DEC_MAXS; ++s) {
                    if (s < a.nslab) {
                        f[0] += sl[i][s][0].x; f[1] += sl[i][s][0].y; f[2] += sl[i][s][0].z; f[3] += sl[i][s][0].w;
                        f[4] += sl[i][s][1].x; f[5] += sl[i][s][1].y; f[6] += sl[i][s][1].z; f[7] += sl[i][s][1].w;
                    }
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = (f16) (f[j] + (float) v[j]);
            }
            if (idx < nvec) {
                if (blockIdx.x == 0 && a.hid_out) *(f16x8*) (a.hid_out + idx * 8) = v;
#pragma unroll
                for (int j = 0; j < 8; ++j) { const float f = (float) v[j]; ss = fmaf(f, f, ss); }
            }
            xv[i] = v;
        }
    }
    const float total = block_sum_256(ss, red + 4 * GC_BN, tid);
    const f16 rm = (f16) (1.0f / sqrtf(total * (1.0f / (float) a.h) + a.eps));
#pragma unroll
    for (int i = 0; i < DEC_NV; ++i) {
        const int idx = tid + i * 256;
        if (i * 256 < nvec && idx < nvec) {
            const f16x8 w = *(const f16x8*) (a.norm_w + idx * 8);
            f16x8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) { const f16 t = xv[i][j] * rm; o[j] = t * w[j]; }
            if (m.x_map) *(f16x8*) (xlin + idx * 8) = o;
            else         xs[idx] = gc_permute(__builtin_bit_cast(uint4, o));
        }
    }
    __syncthreads();
    if (m.x_map) {                                                       // act-order: gather through x_map
        gc_stage_from_lds(xlin, m.x_map, 0, m.K >> 3, xs, tid);
        __syncthreads();
    }

    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    gc_consume(m, plan, 0, col, col_ok, ty, wv, xs, acc);
    for (int pass = 1; pass < plan.npass; ++pass) {
        gc_issue(m, plan, pass, col, col_ok, ty, wv);
        gc_consume(m, plan, pass, col, col_ok, ty, wv, xs, acc);
    }
    const float v = gc_block_reduce(acc, red, tid);
    if (tid < GC_BN) {
        const int n = tile * GC_BN + tid;
        if (n < m.N) a.out[mi][n] = (f16) v;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// K2: RoPE + cache append + split-KV attention (head_dim 128, 16 lanes per key row)
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void dec_attn_kernel(const f16* __restrict__ q, const f16* __restrict__ k_new,
                                                       const f16* __restrict__ v_new, f16* __restrict__ kc,
                                                       f16* __restrict__ vc, const f16* __restrict__ sin,
                                                       const f16* __restrict__ cos, float* __restrict__ partial,
                                                       const int32_t* __restrict__ pos_dev, int heads, int kv_heads,
                                                       int max_seq, int nsplit, float scale)
{
    constexpr int HD = 128, LPK = 16, KPI = 16;
    __shared__ float sc[DEC_ATT_MAX_KEYS];
    __shared__ float red[KPI][HD + 1];
    __shared__ float stat[8];

    const int split = blockIdx.x, h = blockIdx.y;
    const int tid = threadIdx.x;
    const int d8 = tid & 15, ks = tid >> 4;
    const int past = *pos_dev;
    const int vis = past + 1;
    int L = (vis + nsplit - 1) / nsplit;
    L = (L + 15) & ~15;
    const int s0 = min(vis, split * L), s1 = min(vis, s0 + L);
    const int nkeys = s1 - s0;
    const int kvh = h / (heads / kv_heads);

    // RoPE on q and on the new key: element d pairs with d +- 64, i.e. lane d8 with lane d8 ^ 8
    const f16x8 sn = *(const f16x8*) (sin + (size_t) past * HD + d8 * 8);
    const f16x8 cs = *(const f16x8*) (cos + (size_t) past * HD + d8 * 8);
    const bool left = d8 < 8;
    auto rope8 = [&](f16x8 own) {
        const uint4 oi = __builtin_bit_cast(uint4, own);
        uint4 pi;
        pi.x = __shfl_xor((int) oi.x, 8, 64); pi.y = __shfl_xor((int) oi.y, 8, 64);
        pi.z = __shfl_xor((int) oi.z, 8, 64); pi.w = __shfl_xor((int) oi.w, 8, 64);
        const f16x8 oth = __builtin_bit_cast(f16x8, pi);
        f16x8 r;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const f16 s = left ? (f16) (-sn[j]) : sn[j];
            const f16 t = oth[j] * s;
            r[j] = __builtin_fmaf16(own[j], cs[j], t);
        }
        return r;
    };
    const f16x8 qr = rope8(*(const f16x8*) (q + (size_t) h * HD + d8 * 8));
    const f16x8 kr = rope8(*(const f16x8*) (k_new + (size_t) kvh * HD + d8 * 8));
    const f16x8 vn = *(const f16x8*) (v_new + (size_t) kvh * HD + d8 * 8);
    f16* kbase = kc + (size_t) kvh * max_seq * HD + d8 * 8;
    f16* vbase = vc + (size_t) kvh * max_seq * HD + d8 * 8;
    if (split == 0 && (h % (heads / kv_heads)) == 0 && ks == 0) {
        *(f16x8*) (kbase + (size_t) past * HD) = kr;
        *(f16x8*) (vbase + (size_t) past * HD) = vn;
    }
    float qf[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) qf[j] = (float) qr[j] * scale;

    constexpr int UN = 8;                                       // key rows in flight per thread
    float mx = -INFINITY;
    for (int j0 = 0; j0 < nkeys; j0 += KPI * UN) {
        f16x8 kv[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int j = j0 + u * KPI + ks;
            const int key = s0 + (j < nkeys ? j : 0);           // clamped: always a valid row
            kv[u] = *(const f16x8*) (kbase + (size_t) key * HD);
            if (key == past) kv[u] = kr;                        // the new key never comes from memory
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int j = j0 + u * KPI + ks;
            float dot = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) dot = fmaf(qf[e], (float) kv[u][e], dot);
#pragma unroll
            for (int off = 1; off < LPK; off <<= 1) dot += __shfl_xor(dot, off, 64);
            if (j < nkeys) {
                if (d8 == 0) sc[j] = dot;
                mx = fmaxf(mx, dot);
            }
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
    if ((tid & 63) == 0) stat[tid >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(stat[0], stat[1]), fmaxf(stat[2], stat[3]));
    __syncthreads();
    float lsum = 0.f;
    for (int j = tid; j < nkeys; j += 256) {
        const float p = __expf(sc[j] - mx);
        sc[j] = p;
        lsum += p;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) lsum += __shfl_xor(lsum, off, 64);
    if ((tid & 63) == 0) stat[4 + (tid >> 6)] = lsum;
    __syncthreads();
    lsum = stat[4] + stat[5] + stat[6] + stat[7];

    float o[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = 0.f;
    for (int j0 = 0; j0 < nkeys; j0 += KPI * UN) {
        f16x8 vv[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int j = j0 + u * KPI + ks;
            const int key = s0 + (j < nkeys ? j : 0);
            vv[u] = *(const f16x8*) (vbase + (size_t) key * HD);
            if (key == past) vv[u] = vn;
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int j = j0 + u * KPI + ks;
            const float p = j < nkeys ? sc[j] : 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = fmaf(p, (float) vv[u][e], o[e]);
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) red[ks][d8 * 8 + e] = o[e];
    __syncthreads();
    float* pp = partial + ((size_t) h * nsplit + split) * (HD + 2);
    if (tid < HD) {
        float v = 0.f;
#pragma unroll
        for (int s = 0; s < KPI; ++s) v += red[s][tid];
        pp[tid] = v;
    }
    if (tid == 0) {
        pp[HD] = nkeys > 0 ? mx : -INFINITY;
        pp[HD + 1] = nkeys > 0 ? lsum : 0.f;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// K3 / K5: activation built on the fly (attention merge | silu*mul | plain vector) -> split-K GEMV -> fp32 slabs
// ---------------------------------------------------------------------------------------------------------------
struct BVecArgs {
    int mode;                     // 0: plain fp16 vector, 1: attention partial merge, 2: silu(g) * u
    const f16* vec;               // mode 0
    const float* partial;         // mode 1: [heads][nsplit][130]
    int nsplit;
    const f16* g;                 // mode 2
    const f16* u;
    GcMatrix mat;
    float* slabs;                 // [splitk][N]
    int prows_per_block;
};

__device__ __forceinline__ f16 silu_mul_f16(f16 x, f16 y)
{
    const f16 e = (f16) __expf((float) (f16) (-x));
    const f16 sm = (f16) 1.0f + e;
    const f16 rc = (f16) (1.0f / (float) sm);
    const f16 v = x * rc;
    return v * y;
}

__device__ __forceinline__ f16 attn_merge_elem(const float* partial, int nsplit, int e)
{
    const int head = e >> 7, d = e & 127;
    const float* pp = partial + (size_t) head * nsplit * 130;
    float M = -INFINITY;
    for (int s = 0; s < nsplit; ++s) M = fmaxf(M, pp[s * 130 + 128]);
    float l = 0.f, o = 0.f;
    for (int s = 0; s < nsplit; ++s) {
        const float ms = pp[s * 130 + 128];
        if (ms > -INFINITY) {
            const float w = __expf(ms - M);
            l = fmaf(pp[s * 130 + 129], w, l);
            o = fmaf(pp[s * 130 + d], w, o);
        }
    }
    return (f16) (o / l);
}

#define DEC_MAX_NSPLIT 8

template <int FAST, int NG>  // FAST: 0 general path, 1 attention merge, 2 silu*mul; NG: 8-element groups per thread
__global__ __launch_bounds__(256) void dec_vec_gemv_kernel(const BVecArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint4* xs = (uint4*) smem;                                // [prows_per_block]
    float* red = (float*) (smem + (size_t) a.prows_per_block * 16);

    const int tid = threadIdx.x;
    const int tx = tid % GC_TX, ty = tid / GC_TX;
    const GcMatrix& m = a.mat;
    const int col = blockIdx.x * GC_BN + tx * 4;
    const bool col_ok = col < m.N;
    const int prow_total = m.K >> 3;
    const int r0 = blockIdx.y * a.prows_per_block;
    const int nrows = min(prow_total, r0 + a.prows_per_block) - r0;
    const GcPlan plan = gc_plan(r0, nrows);
    uint4 wv[GC_MAXR];

    // fast paths keep every prologue load ahead of the weight stream, in straight-line code (<= 2 groups per thread)
    if constexpr (FAST == 2) {
        uint4 gv[NG], uv[NG];
#pragma unroll
        for (int i = 0; i < NG; ++i) {
            const int idx = tid + i * 256;
            const int k0 = (r0 + (idx < nrows ? idx : 0)) * 8;
            gv[i] = *(const uint4*) (a.g + k0);
            uv[i] = *(const uint4*) (a.u + k0);
        }
        gc_issue(m, plan, 0, col, col_ok, ty, wv);
#pragma unroll
        for (int i = 0; i < NG; ++i) {
            const int idx = tid + i * 256;
            if (idx < nrows) {
                const f16x8 g8 = __builtin_bit_cast(f16x8, gv[i]);
                const f16x8 u8 = __builtin_bit_cast(f16x8, uv[i]);
                f16x8 v;
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = silu_mul_f16(g8[j], u8[j]);
                xs[idx] = gc_permute(__builtin_bit_cast(uint4, v));
            }
        }
    } else if constexpr (FAST == 1) {
        // attention merge of 8 consecutive elements of one head per group
        float ms[NG][DEC_MAX_NSPLIT], ls[NG][DEC_MAX_NSPLIT];
        float2 po[NG][DEC_MAX_NSPLIT][4];
#pragma unroll
        for (int i = 0; i < NG; ++i) {
            const int idx = tid + i * 256;
            const int k0 = (r0 + (idx < nrows ? idx : 0)) * 8;
            const float* pp = a.partial + (size_t) (k0 >> 7) * a.nsplit * 130;
            const int d0 = k0 & 127;
            if (i * 256 < nrows) {                              // block-uniform
#pragma unroll
                for (int sp = 0; sp < DEC_MAX_NSPLIT; ++sp) {
                    const int cs = sp < a.nsplit ? sp : 0;
                    ms[i][sp] = pp[cs * 130 + 128];
                    ls[i][sp] = pp[cs * 130 + 129];
                    const float2* q2 = (const float2*) (pp + cs * 130 + d0);     // 130-float rows: 8-byte aligned
#pragma unroll
                    for (int j = 0; j < 4; ++j) po[i][sp][j] = q2[j];
                }
            }
        }
        gc_issue(m, plan, 0, col, col_ok, ty, wv);
#pragma unroll
        for (int i = 0; i < NG; ++i) {
            const int idx = tid + i * 256;
            if (i * 256 < nrows) {
                float M = -INFINITY;
#pragma unroll
                for (int sp = 0; sp < DEC_MAX_NSPLIT; ++sp) if (sp < a.nsplit) M = fmaxf(M, ms[i][sp]);
                float l = 0.f;
                float o[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int sp = 0; sp < DEC_MAX_NSPLIT; ++sp) {
                    const float w = (sp < a.nsplit && ms[i][sp] > -INFINITY) ? __expf(ms[i][sp] - M) : 0.f;
                    l = fmaf(ls[i][sp], w, l);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        o[2 * j] = fmaf(po[i][sp][j].x, w, o[2 * j]);
                        o[2 * j + 1] = fmaf(po[i][sp][j].y, w, o[2 * j + 1]);
                    }
                }
                const float inv = 1.0f / l;
                f16x8 v;
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = (f16) (o[j] * inv);
                if (idx < nrows) xs[idx] = gc_permute(__builtin_bit_cast(uint4, v));
            }
        }
    } else {
        // general path (act-order gather, plain vectors, long K ranges): activation first, then the weights
        for (int idx = tid; idx < nrows; idx += 256) {
            const int k0 = (r0 + idx) * 8;
            f16x8 v;
            if (!m.x_map && a.mode == 0) {
                v = *(const f16x8*) (a.vec + k0);
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int e = m.x_map ? (int) m.x_map[k0 + j] : k0 + j;
                    v[j] = a.mode == 0 ? a.vec[e] : a.mode == 1 ? attn_merge_elem(a.partial, a.nsplit, e) : silu_mul_f16(a.g[e], a.u[e]);
                }
            }
            xs[idx] = gc_permute(__builtin_bit_cast(uint4, v));
        }
        gc_issue(m, plan, 0, col, col_ok, ty, wv);
    }
    __syncthreads();

    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    gc_consume(m, plan, 0, col, col_ok, ty, wv, xs, acc);
    for (int pass = 1; pass < plan.npass; ++pass) {
        gc_issue(m, plan, pass, col, col_ok, ty, wv);
        gc_consume(m, plan, pass, col, col_ok, ty, wv, xs, acc);
    }
    const float v = gc_block_reduce(acc, red, tid);
    if (tid < GC_BN) {
        const int n = blockIdx.x * GC_BN + tid;
        if (n < m.N) a.slabs[(size_t) blockIdx.y * m.N + n] = v;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// K6: final residual + RMSNorm + fp16 lm_head GEMV (one wave per vocabulary row, 128-bit loads) -> fp32 logits
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void dec_head_kernel(const f16* __restrict__ hid, const float* __restrict__ slabs,
                                                       int nslab, const f16* __restrict__ norm_w, float eps, int h,
                                                       const f16* __restrict__ lm_head, int vocab,
                                                       float* __restrict__ logits, int rows_per_block,
                                                       int32_t* __restrict__ pos_dev, int advance)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    f16* xlin = (f16*) smem;
    float* red4 = (float*) (smem + (size_t) h * 2);
    const int tid = threadIdx.x;
    const int nvec = h >> 3;
    float ss = 0.f;
    for (int i = tid; i < nvec; i += 256) {
        f16x8 v = *(const f16x8*) (hid + i * 8);
        float f[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] = 0.f;
        for (int s = 0; s < nslab; ++s) {
            const float4 p0 = *(const float4*) (slabs + (size_t) s * h + i * 8);
            const float4 p1 = *(const float4*) (slabs + (size_t) s * h + i * 8 + 4);
            f[0] += p0.x; f[1] += p0.y; f[2] += p0.z; f[3] += p0.w;
            f[4] += p1.x; f[5] += p1.y; f[6] += p1.z; f[7] += p1.w;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) { v[j] = (f16) (f[j] + (float) v[j]); const float t = (float) v[j]; ss = fmaf(t, t, ss); }
        *(f16x8*) (xlin + i * 8) = v;
    }
    const float total = block_sum_256(ss, red4, tid);
    const f16 rm = (f16) (1.0f / sqrtf(total * (1.0f / (float) h) + eps));
    for (int i = tid; i < nvec; i += 256) {
        const f16x8 v = *(const f16x8*) (xlin + i * 8);
        const f16x8 w = *(const f16x8*) (norm_w + i * 8);
        f16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) { const f16 t = v[j] * rm; o[j] = t * w[j]; }
        *(f16x8*) (xlin + i * 8) = o;
    }
    __syncthreads();
    const int wave = tid >> 6, lane = tid & 63;
    const int row0 = blockIdx.x * rows_per_block;
    for (int r = wave; r < rows_per_block; r += 4) {
        const int row = row0 + r;
        if (row >= vocab) break;
        const f16* wr = lm_head + (size_t) row * h;
        float acc = 0.f;
        for (int i = lane; i < nvec; i += 64) {
            const uint4 wv = nt_load16(wr + i * 8);
            const uint4 xv = *(const uint4*) (xlin + i * 8);
            acc = __builtin_amdgcn_fdot2(gc_h2(wv.x), gc_h2(xv.x), acc, false);
            acc = __builtin_amdgcn_fdot2(gc_h2(wv.y), gc_h2(xv.y), acc, false);
            acc = __builtin_amdgcn_fdot2(gc_h2(wv.z), gc_h2(xv.z), acc, false);
            acc = __builtin_amdgcn_fdot2(gc_h2(wv.w), gc_h2(xv.w), acc, false);
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
        if (lane == 0) logits[row] = (float) (f16) acc;        // nn.Linear in fp16, then .float() (model.py:1077-1080)
    }
    if (advance && blockIdx.x == 0 && tid == 0) *pos_dev += 1;
}

// ---------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------
struct DecLayer {
    Q4Matrix *q, *k, *v, *o, *gate, *up, *down;
    const f16 *in_norm, *post_norm;
    f16 *kc, *vc;
    bool set;
};

struct Decoder {
    uint32_t magic;
    int device, L, h, inter, heads, kv_heads, hd, vocab, max_seq;
    float eps;
    const f16 *embed, *final_norm, *lm_head, *sin, *cos;
    std::vector<DecLayer> layers;
    f16 *hidA, *hidB, *qbuf, *kbuf, *vbuf, *gbuf, *ubuf;
    float *slab_o, *slab_d, *partial;
    int splitk_o, splitk_d, prows_o, prows_d, nsplit;
    void* block;                  // one hipMalloc
};
#define DEC_MAGIC 0x44454331u

static GcMatrix gc_view(const Q4Matrix* m)
{
    GcMatrix g;
    g.qweight = m->qweight; g.qzeros = m->qzeros; g.scales = m->scales; g.x_map = m->x_map;
    g.K = m->height; g.N = m->width; g.groupsize = m->groupsize;
    return g;
}

static void pick_splitk(int K, int N, int* splitk, int* prows)
{
    const int prow_total = K / 8;
    const int tiles = (N + GC_BN - 1) / GC_BN;
    int sk = (512 + tiles - 1) / tiles;
    if (sk > DEC_MAXS) sk = DEC_MAXS;
    while (sk > 1 && prow_total / sk < 64) --sk;
    int pr = (prow_total + sk - 1) / sk;
    pr = (pr + 3) & ~3;
    *prows = pr;
    *splitk = (prow_total + pr - 1) / pr;
}

extern "C" int exl_decoder_create(int device, int n_layers, int hidden, int inter, int heads, int kv_heads, int head_dim,
                                  int vocab, int max_seq_len, float eps, const void* embed, const void* final_norm,
                                  const void* lm_head, const void* sin, const void* cos, void** out)
{
    EXL_REQUIRE(out, EXL_E_INVALID, "decoder_create: out is null");
    *out = nullptr;
    EXL_REQUIRE(device >= 0 && device < EXL_MAX_DEVICES, EXL_E_INVALID, "decoder_create: invalid device %d", device);
    EXL_REQUIRE(head_dim == 128, EXL_E_UNSUPPORTED, "decoder: head_dim must be 128 (got %d)", head_dim);
    EXL_REQUIRE(hidden % 8 == 0 && hidden == heads * head_dim && heads % kv_heads == 0, EXL_E_UNSUPPORTED, "decoder: bad head geometry");
    EXL_REQUIRE(embed && final_norm && lm_head && sin && cos, EXL_E_INVALID, "decoder_create: null pointer");
    Decoder* d = new Decoder();
    d->magic = DEC_MAGIC;
    d->device = device; d->L = n_layers; d->h = hidden; d->inter = inter; d->heads = heads; d->kv_heads = kv_heads;
    d->hd = head_dim; d->vocab = vocab; d->max_seq = max_seq_len; d->eps = eps;
    d->embed = (const f16*) embed; d->final_norm = (const f16*) final_norm; d->lm_head = (const f16*) lm_head;
    d->sin = (const f16*) sin; d->cos = (const f16*) cos;
    d->layers.resize(n_layers);
    for (auto& l : d->layers) l.set = false;
    pick_splitk(hidden, hidden, &d->splitk_o, &d->prows_o);
    pick_splitk(inter, hidden, &d->splitk_d, &d->prows_d);
    int ns = 256 / heads;
    if (ns < 1) ns = 1;
    if (ns > DEC_MAX_NSPLIT) ns = DEC_MAX_NSPLIT;
    while ((max_seq_len + ns - 1) / ns + 16 > DEC_ATT_MAX_KEYS) ++ns;
    d->nsplit = ns;
    const int kvd = kv_heads * head_dim;
    size_t bytes = 0;
    auto carve = [&](size_t n) { const size_t off = bytes; bytes += (n + 255) & ~(size_t) 255; return off; };
    const size_t o_hidA = carve((size_t) hidden * 2), o_hidB = carve((size_t) hidden * 2), o_q = carve((size_t) hidden * 2);
    const size_t o_k = carve((size_t) kvd * 2), o_v = carve((size_t) kvd * 2);
    const size_t o_g = carve((size_t) inter * 2), o_u = carve((size_t) inter * 2);
    const size_t o_so = carve((size_t) d->splitk_o * hidden * 4), o_sd = carve((size_t) d->splitk_d * hidden * 4);
    const size_t o_p = carve((size_t) heads * ns * 130 * 4);
    int prev = 0;
    hipError_t e = hipGetDevice(&prev);
    if (e == hipSuccess) e = hipSetDevice(device);
    if (e == hipSuccess) e = hipMalloc(&d->block, bytes);
    (void) hipSetDevice(prev);
    if (e != hipSuccess) { delete d; EXL_FAIL((int) e, "decoder_create: %s", hipGetErrorString(e)); }
    unsigned char* b = (unsigned char*) d->block;
    d->hidA = (f16*) (b + o_hidA); d->hidB = (f16*) (b + o_hidB); d->qbuf = (f16*) (b + o_q);
    d->kbuf = (f16*) (b + o_k); d->vbuf = (f16*) (b + o_v); d->gbuf = (f16*) (b + o_g); d->ubuf = (f16*) (b + o_u);
    d->slab_o = (float*) (b + o_so); d->slab_d = (float*) (b + o_sd); d->partial = (float*) (b + o_p);
    *out = d;
    return 0;
}

static Decoder* dec_from(void* p)
{
    Decoder* d = (Decoder*) p;
    return (d && d->magic == DEC_MAGIC) ? d : nullptr;
}

extern "C" int exl_decoder_set_layer(void* dec, int index, void* q, void* k, void* v, void* o, void* gate, void* up,
                                     void* down, const void* in_norm, const void* post_norm, void* key_cache,
                                     void* value_cache)
{
    Decoder* d = dec_from(dec);
    EXL_REQUIRE(d, EXL_E_INVALID, "decoder_set_layer: invalid decoder");
    EXL_REQUIRE(index >= 0 && index < d->L, EXL_E_INVALID, "decoder_set_layer: layer %d out of range", index);
    DecLayer& l = d->layers[index];
    l.q = q4_from_handle(q); l.k = q4_from_handle(k); l.v = q4_from_handle(v); l.o = q4_from_handle(o);
    l.gate = q4_from_handle(gate); l.up = q4_from_handle(up); l.down = q4_from_handle(down);
    EXL_REQUIRE(l.q && l.k && l.v && l.o && l.gate && l.up && l.down, EXL_E_INVALID, "decoder_set_layer: invalid q4 handle");
    const int kvd = d->kv_heads * d->hd;
    EXL_REQUIRE(l.q->height == d->h && l.q->width == d->h && l.k->height == d->h && l.k->width == kvd &&
                l.v->height == d->h && l.v->width == kvd && l.o->height == d->h && l.o->width == d->h &&
                l.gate->height == d->h && l.gate->width == d->inter && l.up->height == d->h && l.up->width == d->inter &&
                l.down->height == d->inter && l.down->width == d->h, EXL_E_INVALID, "decoder_set_layer: matrix shapes do not match the model");
    for (Q4Matrix* m : {l.q, l.k, l.v, l.o, l.gate, l.up, l.down}) {
        EXL_REQUIRE(m->device == d->device, EXL_E_INVALID, "decoder_set_layer: matrix lives on another device");
        EXL_REQUIRE(m->width % 4 == 0 && m->groupsize % 8 == 0, EXL_E_UNSUPPORTED, "decoder_set_layer: unsupported matrix geometry");
    }
    EXL_REQUIRE(in_norm && post_norm && key_cache && value_cache, EXL_E_INVALID, "decoder_set_layer: null pointer");
    l.in_norm = (const f16*) in_norm; l.post_norm = (const f16*) post_norm;
    l.kc = (f16*) key_cache; l.vc = (f16*) value_cache;
    l.set = true;
    return 0;
}

extern "C" int exl_decoder_free(void* dec)
{
    Decoder* d = dec_from(dec);
    EXL_REQUIRE(d, EXL_E_INVALID, "decoder_free: invalid decoder");
    int prev = 0;
    if (hipGetDevice(&prev) == hipSuccess) { (void) hipSetDevice(d->device); (void) hipFree(d->block); (void) hipSetDevice(prev); }
    d->magic = 0;
    delete d;
    return 0;
}

static int launch_norm_gemv(const Decoder* d, const f16* hid_in, const int64_t* tok, const float* slabs, int nslab,
                            f16* hid_out, const f16* norm_w, int nmat, Q4Matrix* const* mats, f16* const* outs, hipStream_t s)
{
    ANormArgs a;
    a.hid_in = hid_in; a.tok = tok; a.slabs = slabs; a.nslab = nslab; a.hid_out = hid_out; a.norm_w = norm_w;
    a.eps = d->eps; a.h = d->h; a.nmat = nmat;
    int tiles = 0;
    for (int i = 0; i < DEC_MAX_MATS; ++i) {
        if (i < nmat) {
            a.mat[i] = gc_view(mats[i]);
            a.out[i] = outs[i];
            tiles += (mats[i]->width + GC_BN - 1) / GC_BN;
        } else {
            a.mat[i] = a.mat[0];
            a.out[i] = nullptr;
        }
        a.tile_end[i] = tiles;
    }
    const size_t smem = (size_t) d->h * 4 + (4 * GC_BN + 8) * sizeof(float);
    const int nv = (d->h / 8 + 255) / 256;
    if (nv <= 2)      hipLaunchKernelGGL(dec_norm_gemv_kernel<2>, dim3(tiles), dim3(256), smem, s, a);
    else if (nv <= 3) hipLaunchKernelGGL(dec_norm_gemv_kernel<3>, dim3(tiles), dim3(256), smem, s, a);
    else              hipLaunchKernelGGL(dec_norm_gemv_kernel<4>, dim3(tiles), dim3(256), smem, s, a);
    EXL_LAUNCH_CHECK();
    return 0;
}

static int launch_vec_gemv(int mode, const f16* vec, const float* partial, int nsplit, const f16* g, const f16* u,
                           const Q4Matrix* m, float* slabs, int splitk, int prows, hipStream_t s)
{
    BVecArgs a;
    a.mode = mode; a.vec = vec; a.partial = partial; a.nsplit = nsplit; a.g = g; a.u = u;
    a.mat = gc_view(m); a.slabs = slabs; a.prows_per_block = prows;
    const size_t smem = (size_t) prows * 16 + 4 * GC_BN * sizeof(float);
    dim3 grid((m->width + GC_BN - 1) / GC_BN, splitk);
    const bool fast = !m->x_map && prows <= 512 && (mode == 2 || (mode == 1 && nsplit <= DEC_MAX_NSPLIT));
    if (!fast)                      hipLaunchKernelGGL((dec_vec_gemv_kernel<0, 1>), grid, dim3(256), smem, s, a);
    else if (mode == 1 && prows <= 256) hipLaunchKernelGGL((dec_vec_gemv_kernel<1, 1>), grid, dim3(256), smem, s, a);
    else if (mode == 1)             hipLaunchKernelGGL((dec_vec_gemv_kernel<1, 2>), grid, dim3(256), smem, s, a);
    else if (prows <= 256)          hipLaunchKernelGGL((dec_vec_gemv_kernel<2, 1>), grid, dim3(256), smem, s, a);
    else                            hipLaunchKernelGGL((dec_vec_gemv_kernel<2, 2>), grid, dim3(256), smem, s, a);
    EXL_LAUNCH_CHECK();
    return 0;
}

extern "C" int exl_decoder_step(void* dec, const int64_t* token_dev, int32_t* pos_dev, float* logits_out, int advance,
                                void* stream)
{
    Decoder* d = dec_from(dec);
    EXL_REQUIRE(d, EXL_E_INVALID, "decoder_step: invalid decoder");
    EXL_REQUIRE(token_dev && pos_dev && logits_out, EXL_E_INVALID, "decoder_step: null pointer");
    for (const DecLayer& l : d->layers) EXL_REQUIRE(l.set, EXL_E_INVALID, "decoder_step: a layer was not set");
    hipStream_t s = (hipStream_t) stream;
    int prev = 0;
    EXL_HIP(hipGetDevice(&prev));
    if (prev != d->device) EXL_HIP(hipSetDevice(d->device));
    const float scale = 1.0f / sqrtf((float) d->hd);
    int rc = 0;
    for (int i = 0; i < d->L && rc == 0; ++i) {
        const DecLayer& l = d->layers[i];
        Q4Matrix* qkv[3] = {l.q, l.k, l.v};
        f16* qkv_out[3] = {d->qbuf, d->kbuf, d->vbuf};
        if (i == 0) rc = launch_norm_gemv(d, d->embed, token_dev, nullptr, 0, d->hidA, l.in_norm, 3, qkv, qkv_out, s);
        else        rc = launch_norm_gemv(d, d->hidB, nullptr, d->slab_d, d->splitk_d, d->hidA, l.in_norm, 3, qkv, qkv_out, s);
        if (rc) break;
        hipLaunchKernelGGL(dec_attn_kernel, dim3(d->nsplit, d->heads), dim3(256), 0, s, d->qbuf, d->kbuf, d->vbuf, l.kc, l.vc,
                           d->sin, d->cos, d->partial, pos_dev, d->heads, d->kv_heads, d->max_seq, d->nsplit, scale);
        { hipError_t e = hipGetLastError(); if (e != hipSuccess) { exl_set_error("decoder attn launch: %s", hipGetErrorString(e)); rc = (int) e; break; } }
        rc = launch_vec_gemv(1, nullptr, d->partial, d->nsplit, nullptr, nullptr, l.o, d->slab_o, d->splitk_o, d->prows_o, s);
        if (rc) break;
        Q4Matrix* gu[2] = {l.gate, l.up};
        f16* gu_out[2] = {d->gbuf, d->ubuf};
        rc = launch_norm_gemv(d, d->hidA, nullptr, d->slab_o, d->splitk_o, d->hidB, l.post_norm, 2, gu, gu_out, s);
        if (rc) break;
        rc = launch_vec_gemv(2, nullptr, nullptr, 0, d->gbuf, d->ubuf, l.down, d->slab_d, d->splitk_d, d->prows_d, s);
    }
    if (rc == 0) {
        const int rows_per_block = 32;
        const int blocks = (d->vocab + rows_per_block - 1) / rows_per_block;
        const size_t smem = (size_t) d->h * 2 + 8 * sizeof(float);
        hipLaunchKernelGGL(dec_head_kernel, dim3(blocks), dim3(256), smem, s, d->hidB, d->slab_d, d->splitk_d, d->final_norm,
                           d->eps, d->h, d->lm_head, d->vocab, logits_out, rows_per_block, pos_dev, advance);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) { exl_set_error("decoder head launch: %s", hipGetErrorString(e)); rc = (int) e; }
    }
    if (prev != d->device) (void) hipSetDevice(prev);
    return rc;
}
