// Device-side core of the int4 decode GEMV on the RE-TILED weight layout (see q4_matrix.hip: retile_kernel).
//
// Layout in HBM ("column-tile major"):  word(n, r) of the GPTQ matrix [R = K/8 packed rows][N columns] lives at
//     qw[((n >> 3) * R + r) * 8 + (n & 7)]
// i.e. the 8-column tile t = n / 8 is ONE contiguous run of R * 32 bytes.  A wave64 streams a tile with 16-byte
// loads, lane l -> (row = l >> 1, column quad = l & 1): one wave instruction = 32 consecutive packed rows = 1 KiB of
// perfectly contiguous memory, and a block streams full-K tiles, so there is no split-K, no partial-sum slab and no
// cross-block reduction anywhere on the decode path.
//
// Per (tile, group) dequantisation constants sit in an LDS table built once per block:
//     entry (32 bytes) = { zc0 pair x 4 columns (f16x2 each), scale x 4 columns (f32) }      per column quad
// with zc0 = -(1024 + z + 1): (0x6400 | q) + zc0 == q - (z + 1) exactly in fp16 (magic-number nibble expansion).
// Products accumulate in fp32 (v_dot2_f32_f16) per packed row and are scaled per row: acc += scale * part.
#pragma once
#include "common.h"

#define GT_MAGIC 0x64006400u

__device__ __forceinline__ f16x2 gt_h2(uint32_t v) { return __builtin_bit_cast(f16x2, v); }

// (h0..h7) -> (h0,h4),(h1,h5),(h2,h6),(h3,h7): the order in which nibble pairs fall out of a GPTQ word
__device__ __forceinline__ uint4 gt_permute(uint4 d)
{
    uint4 o;
    o.x = (d.x & 0xFFFFu) | (d.z << 16);
    o.y = (d.x >> 16) | (d.z & 0xFFFF0000u);
    o.z = (d.y & 0xFFFFu) | (d.w << 16);
    o.w = (d.y >> 16) | (d.w & 0xFFFF0000u);
    return o;
}

// 8 weights of one word against 8 activations (permuted order), fp32 accumulate
__device__ __forceinline__ float gt_dot8(uint32_t w, const uint4& x4, f16x2 zc0, f16x2 zc1, float acc)
{
    const f16x2 sixteenth = {(f16) 0.0625f, (f16) 0.0625f};
    const uint32_t w8 = w >> 8;
    const f16x2 d0 = gt_h2((w & 0x000F000Fu) | GT_MAGIC) + zc0;
    const f16x2 d1 = gt_h2((w & 0x00F000F0u) | GT_MAGIC) * sixteenth + zc1;
    const f16x2 d2 = gt_h2((w8 & 0x000F000Fu) | GT_MAGIC) + zc0;
    const f16x2 d3 = gt_h2((w8 & 0x00F000F0u) | GT_MAGIC) * sixteenth + zc1;
    acc = __builtin_amdgcn_fdot2(d0, gt_h2(x4.x), acc, false);
    acc = __builtin_amdgcn_fdot2(d1, gt_h2(x4.y), acc, false);
    acc = __builtin_amdgcn_fdot2(d2, gt_h2(x4.z), acc, false);
    acc = __builtin_amdgcn_fdot2(d3, gt_h2(x4.w), acc, false);
    return acc;
}

struct GtMatrix {                   // device-visible view of a Q4Matrix (re-tiled qweight)
    const uint4* qw;                // [N/8][R][2] uint4
    const uint32_t* qzeros;         // [G][N/8]           (original GPTQ layout)
    const f16* scales;              // [G][N]             (original GPTQ layout)
    const uint32_t* x_map;          // [K] or NULL
    int K, N, R;                    // R = K / 8
    int gprows;                     // packed rows per group = groupsize / 8
    int gshift;                     // log2(gprows) or -1 when gprows is not a power of two
    int G;
};

__device__ __forceinline__ int gt_group_of(const GtMatrix& m, int r) { return m.gshift >= 0 ? (r >> m.gshift) : (r / m.gprows); }

struct GtEntry { uint4 z; float4 s; };          // 32 bytes: zc0 pairs of 4 columns, scales of 4 columns

// Build the constant table of tile `t` into tab[G][2].  Called by `nthreads` threads with consecutive `tid`.
__device__ __forceinline__ void gt_build_table(const GtMatrix& m, int t, GtEntry* tab, int tid, int nthreads)
{
    for (int g = tid; g < m.G; g += nthreads) {
        const uint32_t zw = m.qzeros[(size_t) g * (m.N >> 3) + t];
        const uint4 sraw = *(const uint4*) (m.scales + (size_t) g * m.N + t * 8);
        const f16x8 s8 = __builtin_bit_cast(f16x8, sraw);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            GtEntry e;
            uint32_t zz[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int z = (int) ((zw >> (4 * (q * 4 + j))) & 0xFu) + 1;
                const f16 a = (f16) (float) (-(1024 + z));
                const f16x2 p = {a, a};
                zz[j] = __builtin_bit_cast(uint32_t, p);
            }
            e.z = make_uint4(zz[0], zz[1], zz[2], zz[3]);
            e.s = make_float4((float) s8[q * 4 + 0], (float) s8[q * 4 + 1], (float) s8[q * 4 + 2], (float) s8[q * 4 + 3]);
            tab[g * 2 + q] = e;
        }
    }
}

// One wave streams rows [row0, row1) of tile `t` (row0 a multiple of 32) in passes of 32 * U packed rows.
// gt_issue puts the U 16-byte loads of one pass in flight (rows past row1 are clamped to a valid address and later
// weighted by a zero scale); gt_consume multiplies them with MROWS activation rows.
//   xs : LDS, [MROWS][xs_stride] uint4, permuted 8-half groups indexed by packed row
//   tab: LDS constant table of this tile
// acc[m][j]: column (t*8 + (lane&1)*4 + j) of activation row m, partial over this lane's packed rows.
template <int U>
__device__ __forceinline__ void gt_issue(const GtMatrix& m, int t, int r0, int row0, int row1, int lane, uint4 (&wv)[U])
{
    const uint4* base = m.qw + ((size_t) t * m.R) * 2 + (lane & 1);
    const int lr = lane >> 1;
#pragma unroll
    for (int i = 0; i < U; ++i) {
        const int r = r0 + i * 32 + lr;
        const int rc = r < row1 ? r : row0;
        wv[i] = nt_load16(base + (size_t) rc * 2);
    }
}

template <int U, int MROWS>
__device__ __forceinline__ void gt_consume(const GtMatrix& m, int r0, int row0, int row1, const uint4* xs, int xs_stride,
                                           const GtEntry* tab, int lane, const uint4 (&wv)[U], float (&acc)[MROWS][4])
{
    const int q = lane & 1;
    const int lr = lane >> 1;
    const f16x2 c960 = {(f16) 960.f, (f16) 960.f};
#pragma unroll
    for (int i = 0; i < U; ++i) {
        const int r = r0 + i * 32 + lr;
        if (r0 + i * 32 < row1) {                                       // wave-uniform
            const bool ok = r < row1;
            const int rc = ok ? r : row0;
            const GtEntry e = tab[gt_group_of(m, rc) * 2 + q];
            const f16x2 z0[4] = {gt_h2(e.z.x), gt_h2(e.z.y), gt_h2(e.z.z), gt_h2(e.z.w)};
            const float sc[4] = {ok ? e.s.x : 0.f, ok ? e.s.y : 0.f, ok ? e.s.z : 0.f, ok ? e.s.w : 0.f};
            const uint32_t ww[4] = {wv[i].x, wv[i].y, wv[i].z, wv[i].w};
#pragma unroll
            for (int mm = 0; mm < MROWS; ++mm) {
                const uint4 x4 = xs[mm * xs_stride + rc];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float part = gt_dot8(ww[j], x4, z0[j], z0[j] + c960, 0.f);
                    acc[mm][j] = fmaf(sc[j], part, acc[mm][j]);
                }
            }
        }
    }
}

// All passes of a wave's row range; the loads of the FIRST pass are already in flight in wvA (issued ahead of the
// block's prologue).  Double-buffered: pass p+1 is issued before pass p is consumed.
template <int U, int MROWS>
__device__ __forceinline__ void gt_finish(const GtMatrix& m, int t, int row0, int row1, const uint4* xs, int xs_stride,
                                          const GtEntry* tab, int lane, uint4 (&wvA)[U], float (&acc)[MROWS][4])
{
    constexpr int STEP = 32 * U;
    if (row0 + STEP >= row1) {                                          // the common case: one pass
        gt_consume<U, MROWS>(m, row0, row0, row1, xs, xs_stride, tab, lane, wvA, acc);
        return;
    }
    uint4 wvB[U];
    int r0 = row0;
    while (true) {
        const int r1 = r0 + STEP;
        if (r1 < row1) gt_issue<U>(m, t, r1, row0, row1, lane, wvB);
        gt_consume<U, MROWS>(m, r0, row0, row1, xs, xs_stride, tab, lane, wvA, acc);
        if (r1 >= row1) break;
        const int r2 = r1 + STEP;
        if (r2 < row1) gt_issue<U>(m, t, r2, row0, row1, lane, wvA);
        gt_consume<U, MROWS>(m, r1, row0, row1, xs, xs_stride, tab, lane, wvB, acc);
        if (r2 >= row1) break;
        r0 = r2;
    }
}

// Sum acc over the 32 row-lanes of a wave; afterwards lanes 0 and 1 hold the totals of column quads 0 and 1.
template <int MROWS>
__device__ __forceinline__ void gt_wave_reduce(float (&acc)[MROWS][4])
{
#pragma unroll
    for (int mm = 0; mm < MROWS; ++mm)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float v = acc[mm][j];
#pragma unroll
            for (int off = 2; off < 64; off <<= 1) v += __shfl_xor(v, off, 64);
            acc[mm][j] = v;
        }
}

// Build the permuted (and x_map-gathered) LDS image of a full activation row from a linear fp16 vector in LDS.
__device__ __forceinline__ void gt_stage_from_lds(const f16* xlin, const uint32_t* x_map, int R, uint4* xs, int tid,
                                                  int nthreads)
{
    for (int idx = tid; idx < R; idx += nthreads) {
        const int k = idx * 8;
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
        xs[idx] = gt_permute(v);
    }
}
