// Device-side core of the int4 decode GEMV (one activation row), shared by the fused decode kernels.
//
// Work decomposition (256 threads): lane tx = tid % 8 owns 4 adjacent output columns (one uint4 of packed
// words per packed row = 128-bit loads, 8 lanes = one full 128-byte line); ty = tid / 8 in [0, 32) owns a
// contiguous run of packed rows.  Every thread issues ALL its weight loads for a pass before touching x, so the
// loads are in flight while the block is still building its activation vector in LDS (gemv_issue -> prologue ->
// __syncthreads -> gemv_consume).  Nibbles are expanded two at a time with the fp16 magic-number trick, the
// zero point is removed exactly in fp16, products accumulate in fp32 via v_dot2_f32_f16.
#pragma once
#include "common.h"

#define GC_MAGIC 0x64006400u
#define GC_TX 8
#define GC_TY 32
#define GC_BN 32
#define GC_MAXR 16          // packed rows per thread per pass

__device__ __forceinline__ f16x2 gc_h2(uint32_t v) { return __builtin_bit_cast(f16x2, v); }

// (h0..h7) -> (h0,h4),(h1,h5),(h2,h6),(h3,h7)
__device__ __forceinline__ uint4 gc_permute(uint4 d)
{
    uint4 o;
    o.x = (d.x & 0xFFFFu) | (d.z << 16);
    o.y = (d.x >> 16) | (d.z & 0xFFFF0000u);
    o.z = (d.y & 0xFFFFu) | (d.w << 16);
    o.w = (d.y >> 16) | (d.w & 0xFFFF0000u);
    return o;
}

__device__ __forceinline__ float gc_dot8(uint32_t w, const uint4& x4, f16x2 zc0, f16x2 zc1, float acc)
{
    const f16x2 sixteenth = {(f16) 0.0625f, (f16) 0.0625f};
    const uint32_t w8 = w >> 8;
    const f16x2 d0 = gc_h2((w & 0x000F000Fu) | GC_MAGIC) + zc0;
    const f16x2 d1 = gc_h2((w & 0x00F000F0u) | GC_MAGIC) * sixteenth + zc1;
    const f16x2 d2 = gc_h2((w8 & 0x000F000Fu) | GC_MAGIC) + zc0;
    const f16x2 d3 = gc_h2((w8 & 0x00F000F0u) | GC_MAGIC) * sixteenth + zc1;
    acc = __builtin_amdgcn_fdot2(d0, gc_h2(x4.x), acc, false);
    acc = __builtin_amdgcn_fdot2(d1, gc_h2(x4.y), acc, false);
    acc = __builtin_amdgcn_fdot2(d2, gc_h2(x4.z), acc, false);
    acc = __builtin_amdgcn_fdot2(d3, gc_h2(x4.w), acc, false);
    return acc;
}

struct GcMatrix {                   // device-visible view of a Q4Matrix
    const uint32_t* qweight;
    const uint32_t* qzeros;
    const f16* scales;
    const uint32_t* x_map;
    int K, N, groupsize;
};

// Rows [r0, r0 + nrows) of the block are cut into `npass` passes of GC_TY * rpt rows; thread ty owns rows
// [pass * GC_TY * rpt + ty * rpt, +rpt) of the block range.
struct GcPlan { int r0, nrows, rpt, npass; };

__device__ __forceinline__ GcPlan gc_plan(int r0, int nrows, int maxr = GC_MAXR)
{
    GcPlan p;
    p.r0 = r0;
    p.nrows = nrows;
    p.npass = (nrows + GC_TY * maxr - 1) / (GC_TY * maxr);
    p.rpt = (nrows + GC_TY * p.npass - 1) / (GC_TY * p.npass);
    return p;
}

template <int MAXR>
__device__ __forceinline__ void gc_issue(const GcMatrix& m, const GcPlan& p, int pass, int col, bool col_ok, int ty,
                                         uint4 (&wv)[MAXR])
{
    const uint4* wcol = (const uint4*) m.qweight + (col >> 2);
    const int n4 = m.N >> 2;
    const int c0 = (pass * GC_TY + ty) * p.rpt;
#pragma unroll
    for (int i = 0; i < MAXR; ++i) {
        const int rr = c0 + i;
        wv[i] = (col_ok && i < p.rpt && rr < p.nrows) ? nt_load16(wcol + (size_t) (p.r0 + rr) * n4) : make_uint4(0, 0, 0, 0);
    }
}

// xs: LDS, permuted 8-half groups, index = packed row relative to the block's r0
template <int MAXR>
__device__ __forceinline__ void gc_consume(const GcMatrix& m, const GcPlan& p, int pass, int col, bool col_ok, int ty,
                                           const uint4 (&wv)[MAXR], const uint4* xs, float (&acc)[4])
{
    const int c0 = (pass * GC_TY + ty) * p.rpt;
    if (c0 >= p.nrows) return;
    const int gprows = m.groupsize >> 3;
    const int rbase = p.r0 + c0;
    int g = rbase / gprows;
    int until = gprows - (rbase - g * gprows);
    f16x2 zc0[4], zc1[4];
    float sc[4], part[4];
    auto load_group = [&](int grp) {
        uint32_t zw = 0;
        f16x4 s4 = {(f16) 0.f, (f16) 0.f, (f16) 0.f, (f16) 0.f};
        if (col_ok) {
            zw = m.qzeros[(size_t) grp * (m.N >> 3) + (col >> 3)];
            s4 = *(const f16x4*) (m.scales + (size_t) grp * m.N + col);
        }
        const int sh = (col & 7) * 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int z = (int) ((zw >> (sh + 4 * j)) & 0xFu) + 1;
            const f16 a = (f16) (float) (-(1024 + z));
            const f16 b = (f16) (float) (-(64 + z));
            zc0[j] = (f16x2){a, a};
            zc1[j] = (f16x2){b, b};
            sc[j] = (float) s4[j];
            part[j] = 0.f;
        }
    };
    auto flush_group = [&]() {
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = fmaf(sc[j], part[j], acc[j]);
    };
    load_group(g);
#pragma unroll
    for (int i = 0; i < MAXR; ++i) {
        const int rr = c0 + i;
        if (i < p.rpt && rr < p.nrows) {
            if (until == 0) { flush_group(); ++g; load_group(g); until = gprows; }
            --until;
            const uint4 x4 = xs[rr];
            part[0] = gc_dot8(wv[i].x, x4, zc0[0], zc1[0], part[0]);
            part[1] = gc_dot8(wv[i].y, x4, zc0[1], zc1[1], part[1]);
            part[2] = gc_dot8(wv[i].z, x4, zc0[2], zc1[2], part[2]);
            part[3] = gc_dot8(wv[i].w, x4, zc0[3], zc1[3], part[3]);
        }
    }
    flush_group();
}

// Sum the 32 k-slices of a block: shuffles across the 8 slices of a wave, then LDS across the 4 waves.
// red: LDS float[4 * GC_BN]; returns the block total for column c (valid for tid < GC_BN) after the barrier.
__device__ __forceinline__ float gc_block_reduce(float (&acc)[4], float* red, int tid)
{
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float v = acc[j];
#pragma unroll
        for (int off = GC_TX; off < 64; off <<= 1) v += __shfl_xor(v, off, 64);
        acc[j] = v;
    }
    const int wave = tid >> 6, lane = tid & 63;
    if (lane < GC_TX) {
#pragma unroll
        for (int j = 0; j < 4; ++j) red[wave * GC_BN + lane * 4 + j] = acc[j];
    }
    __syncthreads();
    float v = 0.f;
    if (tid < GC_BN) v = red[tid] + red[GC_BN + tid] + red[2 * GC_BN + tid] + red[3 * GC_BN + tid];
    return v;
}

// Build the permuted (and x_map-gathered) LDS image of rows [r0, r0+nrows) from a linear fp16 vector in LDS.
__device__ __forceinline__ void gc_stage_from_lds(const f16* xlin, const uint32_t* x_map, int r0, int nrows, uint4* xs,
                                                  int tid)
{
    for (int idx = tid; idx < nrows; idx += 256) {
        const int k = (r0 + idx) * 8;
        uint4 v;
        if (x_map) {
            const uint4 m0 = *(const uint4*) (x_map + k);
            const uint4 m1 = *(const uint4*) (x_map + k + 4);
            f16x8 g;
            g[0] = xlin[m0.x]; g[1] = xlin[m0.y]; g[2] = xlin[m0.z]; g[3] = xlin[m0.w];
            g[4] = xlin[m1.x]; g[5] = xlin[m1.y]; g[6] = xlin[m1.z]; g[7] = xlin[m1.w];
            v = __builtin_bit_cast(uint4, g);
        } else {
            v = *(const uint4*) (xlin + k);
        }
        xs[idx] = gc_permute(v);
    }
}
