// Device-side core of the int4 decode GEMV on the T16 weight layout (q4_matrix.hip: retile_t16_kernel).
//
// T16 layout ("16-column tiles, 4-row pieces").  With R = K/8 packed rows and RB = R/16 row-blocks, the GPTQ word
// (packed row r, column n) lives in the 16-byte piece
//     piece(n, r) = (t * RB + rb) * 64 + rsub * 16 + col        t = n / 16, col = n % 16, rb = r / 16, rsub = (r % 16) / 4
// as dword j = r % 4, with the 8 nibbles of the word INTERLEAVED: weight k of the row sits at nibble (k >> 1) + 4 * (k & 1),
// so that the two-at-a-time magic-number expansion (mask 0x000F000F picks nibbles 0 and 4, ...) yields the weights in
// natural k order (k0,k1),(k2,k3),... and neither the weights nor the activations need a shuffle at run time.  So one piece = 4 consecutive packed rows (32 k) of ONE column, one wave64 load instruction
// (lane = rsub * 16 + col) = 16 packed rows x 16 columns = 1 KiB of contiguous memory, and a 16-column tile is one
// contiguous run of K * 8 bytes that a block streams front to back with full K: no split-K, no partial-sum slabs, no
// cross-block reduction on the decode path.
//
// The lane layout of a piece is exactly the B operand of v_mfma_f32_16x16x32_f16 (lane = (k-group, column), 8 k per
// lane): the matrix core is used as the wave-wide dot-product AND reduction engine.  Per 1 KiB instruction a lane
// expands its 4 words to fp16 (magic-number nibble expansion, exact zero-point subtraction) and issues 4 MFMAs against
// the activation k-groups read from LDS, fp32 accumulate; the group scale is applied to the fp32 sum of the row-block
// (groupsize % 128 == 0) or folded into the weights as ONE fp16 multiply, the reference's reconstruct bits,
// q4_matrix.cu:207 (groupsize 32 / 64).  No shuffles, no atomics.
//
// A operand: lane (m = l & 15, kg = l >> 4) supplies activation row m, k-group kg.  With one activation row every m
// reads the same LDS address (broadcast) and every D row is the same dot product; with up to 16 rows (op-level
// q4_matmul for M < 8) each m reads its own row and D rows 4*rsub + j land in acc[j].
#pragma once
#include "common.h"

#define T16_MAGIC 0x64006400u
#define T16_EH 9                    // entry slots per lane: up to 36 row-blocks per wave (K <= 36864 with 8 waves)

__device__ __forceinline__ f16x2 t16_h2(uint32_t v) { return __builtin_bit_cast(f16x2, v); }

// The magic constant lives in a VGPR the compiler cannot fold: (w & mask) | magic then becomes ONE v_and_or_b32
// (gfx9-family VALU instructions take a single literal / scalar operand: with two constants hipcc emits v_and + v_or).
__device__ __forceinline__ uint32_t t16_magic()
{
    uint32_t m;
    asm volatile("v_mov_b32 %0, 0x64006400" : "=v"(m));
    return m;
}

// 8 weights of one (nibble-interleaved) word as EXACT fp16 integers (q - z), natural k order: 1 shift, 4 and_or, 4 packed ops
__device__ __forceinline__ f16x8 t16_dequant_exact(uint32_t w, uint32_t magic, f16x2 zc0, f16x2 zc1)
{
    const f16x2 sixteenth = {(f16) 0.0625f, (f16) 0.0625f};
    const uint32_t w8 = w >> 8;
    const f16x2 d0 = t16_h2((w & 0x000F000Fu) | magic) + zc0;
    const f16x2 d1 = t16_h2((w & 0x00F000F0u) | magic) * sixteenth + zc1;
    const f16x2 d2 = t16_h2((w8 & 0x000F000Fu) | magic) + zc0;
    const f16x2 d3 = t16_h2((w8 & 0x00F000F0u) | magic) * sixteenth + zc1;
    const uint4 u = make_uint4(__builtin_bit_cast(uint32_t, d0), __builtin_bit_cast(uint32_t, d1),
                               __builtin_bit_cast(uint32_t, d2), __builtin_bit_cast(uint32_t, d3));
    return __builtin_bit_cast(f16x8, u);
}

// 8 weights of one (nibble-interleaved) word as fp16, natural k order, each h( h(q - z) * s )
__device__ __forceinline__ f16x8 t16_dequant(uint32_t w, f16x2 zc0, f16x2 zc1, f16x2 s2)
{
    const f16x2 sixteenth = {(f16) 0.0625f, (f16) 0.0625f};
    const uint32_t w8 = w >> 8;
    const f16x2 d0 = (t16_h2((w & 0x000F000Fu) | T16_MAGIC) + zc0) * s2;
    const f16x2 d1 = (t16_h2((w & 0x00F000F0u) | T16_MAGIC) * sixteenth + zc1) * s2;
    const f16x2 d2 = (t16_h2((w8 & 0x000F000Fu) | T16_MAGIC) + zc0) * s2;
    const f16x2 d3 = (t16_h2((w8 & 0x00F000F0u) | T16_MAGIC) * sixteenth + zc1) * s2;
    const uint4 u = make_uint4(__builtin_bit_cast(uint32_t, d0), __builtin_bit_cast(uint32_t, d1),
                               __builtin_bit_cast(uint32_t, d2), __builtin_bit_cast(uint32_t, d3));
    return __builtin_bit_cast(f16x8, u);
}

// One 1 KiB row-block against the activation: 4 MFMAs.
//   G16 (groupsize % 128 == 0): the whole row-block lies in one group -> weights enter the MFMA as exact integers
//       (q - z) and the fp32 sum of the row-block is scaled once: acc += s * sum   (fp32 dot per group, fp32 scale --
//       the accumulation the oracle's q4_matmul_gemv_f32 states);
//   otherwise (groupsize 32 / 64: the 4 k-groups of the MFMA belong to different GPTQ groups) the scale is folded into
//       the weights, h(h(q - z) * s) -- the reference's reconstruct bits -- and the MFMA accumulates straight into acc.
template <bool G16>
__device__ __forceinline__ void t16_rowblock(const uint4& w, uint32_t e, uint32_t magic, const uint4* xr, f32x4& acc)
{
    const f16x2 c960 = {(f16) 960.f, (f16) 960.f};
    const f16 sc = __builtin_bit_cast(f16, (uint16_t) (e & 0xFFFFu));
    const f16 za = __builtin_bit_cast(f16, (uint16_t) (e >> 16));    // -(1024 + z), stored as fp16 bits by t16_load_entry
    const f16x2 zc0 = {za, za};
    const f16x2 zc1 = zc0 + c960;
    const uint4 x0 = xr[0], x1 = xr[1], x2 = xr[2], x3 = xr[3];
    if constexpr (G16) {
        f32x4 c = {0.f, 0.f, 0.f, 0.f};
        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, x0), t16_dequant_exact(w.x, magic, zc0, zc1), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, x1), t16_dequant_exact(w.y, magic, zc0, zc1), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, x2), t16_dequant_exact(w.z, magic, zc0, zc1), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, x3), t16_dequant_exact(w.w, magic, zc0, zc1), c, 0, 0, 0);
        const float sf = (float) sc;
        acc[0] = fmaf(sf, c[0], acc[0]); acc[1] = fmaf(sf, c[1], acc[1]);
        acc[2] = fmaf(sf, c[2], acc[2]); acc[3] = fmaf(sf, c[3], acc[3]);
    } else {
        const f16x2 s2 = {sc, sc};
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, x0), t16_dequant(w.x, zc0, zc1, s2), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, x1), t16_dequant(w.y, zc0, zc1, s2), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, x2), t16_dequant(w.z, zc0, zc1, s2), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, x3), t16_dequant(w.w, zc0, zc1, s2), acc, 0, 0, 0);
    }
}

// Group sizes 32 / 64 with ONE activation row (the decode executor): the four k-groups of a row-block belong to different GPTQ
// groups, so their sums must stay apart until the scale is applied.  Instead of folding the scale into every weight (13 VALU
// per 8 weights), the A operand carries the activation in FOUR MASKED rows: row 4 g holds the 8 x 4 values of k-group g and
// zeros elsewhere (the caller points every other lane at a zero pad).  Then D[4 g][col] -- element 0 of the lane (col,
// k-group g), the lane that loaded the weights and the scale of exactly that group -- is the EXACT-integer sum
// sum_{k in group g} x_k (q_k - z): one fp32 multiply-add with the lane's own scale finishes it, as in the group-size-128 path
// (9 VALU per 8 weights).  acc[0] of the four lanes of a column are partial sums over their k-groups: the caller adds them once
// per tile (two shuffles).
__device__ __forceinline__ void t16_rowblock_groups(const uint4& w, uint32_t e, uint32_t magic, const uint4* xr, f32x4& acc)
{
    const f16x2 c960 = {(f16) 960.f, (f16) 960.f};
    const f16 sc = __builtin_bit_cast(f16, (uint16_t) (e & 0xFFFFu));
    const f16 za = __builtin_bit_cast(f16, (uint16_t) (e >> 16));
    const f16x2 zc0 = {za, za};
    const f16x2 zc1 = zc0 + c960;
    const uint4 x0 = xr[0], x1 = xr[1], x2 = xr[2], x3 = xr[3];
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, x0), t16_dequant_exact(w.x, magic, zc0, zc1), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, x1), t16_dequant_exact(w.y, magic, zc0, zc1), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, x2), t16_dequant_exact(w.z, magic, zc0, zc1), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, x3), t16_dequant_exact(w.w, magic, zc0, zc1), c, 0, 0, 0);
    acc[0] = fmaf((float) sc, c[0], acc[0]);
}

struct T16Matrix {                  // device-visible view of a Q4Matrix in T16 layout
    const uint4* qw;                // [N/16][RB][64] pieces
    const uint32_t* qzeros;         // [G][N/8]           (GPTQ layout)
    const f16* scales;              // [G][N]             (GPTQ layout)
    const uint32_t* x_map;          // [K] or NULL (act-order)
    int K, N, R, RB;
    int gprows;                     // packed rows per group = groupsize / 8 (a multiple of 4)
    int gshift;                     // log2(gprows) or -1
    int G;
};

__device__ __forceinline__ int t16_group_of_row(const T16Matrix& m, int r) { return m.gshift >= 0 ? (r >> m.gshift) : (r / m.gprows); }

// (scale bits) | fp16 bits of -(1024 + z) << 16 for (group g, column n), z = stored nibble + 1 (matrix.cuh:55-59).
// 1024 + z (z = 1..16) is 0x6400 + z in fp16 (ulp 1 in [1024, 2048)), so the negated constant is 0xE400 + z: the
// zero-point correction of the magic-number expansion costs no conversion instruction at the point of use.  An all-zero
// entry (row-blocks past a wave's range) still means scale 0 -> contributes nothing.
__device__ __forceinline__ uint32_t t16_load_entry(const T16Matrix& m, int g, int n)
{
    const uint32_t zw = m.qzeros[(size_t) g * (m.N >> 3) + (n >> 3)];
    const uint16_t sb = ((const uint16_t*) m.scales)[(size_t) g * m.N + n];
    return (uint32_t) sb | ((0xE401u + ((zw >> ((n & 7) * 4)) & 0xFu)) << 16);
}

// Per-wave streaming state.  U = 16-byte loads in flight per lane (one pass = U row-blocks), NP = max passes.
// G16 = true : groupsize is a multiple of 128 (every row-block lies in one group): entries are loaded once per
//              4 row-blocks per lane and handed around with ds_bpermute;
// G16 = false: groupsize 32 / 64: each lane loads the entry of its own 4 rows with every piece.
template <int U, int NP, bool G16>
struct T16Wave {
    static_assert(U * NP <= 4 * T16_EH, "too many row-blocks per wave");
    const uint4* base;
    int rb0, rb1, rbsafe, col, rsub, n;
    uint32_t ent[G16 ? T16_EH : 2 * U];     // !G16: per-piece entries, double-buffered with the pieces
    uint4 wv[2][U];                          // double buffer: pass p lives in wv[p & 1]

    __device__ __forceinline__ void init(const T16Matrix& m, int t, int lane, int rb_begin, int rb_end)
    {
        col = lane & 15; rsub = lane >> 4;
        rb0 = rb_begin; rb1 = rb_end;
        rbsafe = min(rb_begin, m.RB - 1);                             // waves past the end of K still form valid addresses
        n = t * 16 + col;
        base = m.qw + (size_t) t * m.RB * 64 + lane;
    }
    // issue the small loads of the wave's scale/zero entries (G16) -- call BEFORE the first issue()
    __device__ __forceinline__ void load_entries(const T16Matrix& m)
    {
        if constexpr (G16) {
#pragma unroll
            for (int h = 0; h < T16_EH; ++h) {
                if (h * 4 < U * NP) {
                    const int rb = min(rb0 + 4 * h + rsub, m.RB - 1);
                    const uint32_t e = t16_load_entry(m, t16_group_of_row(m, rb * 16), n);
                    ent[h] = (rb0 + 4 * h + rsub < rb1) ? e : 0u;       // rows past the range: scale 0 -> contribute nothing
                }
            }
        }
    }
    __device__ __forceinline__ void issue(const T16Matrix& m, int pass)
    {
#pragma unroll
        for (int i = 0; i < U; ++i) {
            const int rb = rb0 + pass * U + i;
            const int rbc = rb < rb1 ? rb : rbsafe;                    // clamped: always a valid address
            if constexpr (!G16) {
                const uint32_t e = t16_load_entry(m, t16_group_of_row(m, rbc * 16 + rsub * 4), n);
                ent[(pass & 1) * U + i] = rb < rb1 ? e : 0u;
            }
            wv[pass & 1][i] = nt_load16(base + (size_t) rbc * 64);
        }
    }
    // xrow: LDS activation image (8 halves per packed row, natural order) of the activation row this lane feeds to the
    // A operand (row m = lane & 15; with a single activation row every lane passes the same pointer).
    __device__ __forceinline__ void consume(int pass, const uint4* xrow, f32x4& c)
    {
        const uint32_t magic = t16_magic();
#pragma unroll
        for (int i = 0; i < U; ++i) {
            const int li = pass * U + i;
            const int rb = rb0 + li;
            const int rbc = rb < rb1 ? rb : rbsafe;                     // branch-free: out-of-range row-blocks carry scale 0
            uint32_t e;
            if constexpr (G16) e = (uint32_t) __shfl((int) ent[li >> 2], ((li & 3) << 4) | col, 64);
            else e = ent[(pass & 1) * U + i];
            t16_rowblock<G16>(wv[pass & 1][i], e, magic, xrow + rbc * 16 + rsub * 4, c);
        }
    }
    // all passes; the loads of pass 0 must already be in flight (issued ahead of the block's prologue).  Software
    // pipeline: pass p + 1 is issued into the other buffer before pass p is consumed.
    __device__ __forceinline__ void run(const T16Matrix& m, const uint4* xrow, f32x4& c)
    {
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            if (p + 1 < NP && rb0 + (p + 1) * U < rb1) issue(m, p + 1);     // wave-uniform
            if (rb0 + p * U < rb1) consume(p, xrow, c);
        }
    }
};

// Build the (x_map-gathered) LDS image of an activation row from a linear fp16 copy in LDS.
__device__ __forceinline__ void t16_stage_from_lds(const f16* xlin, const uint32_t* x_map, int R, uint4* xs, int tid,
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
        xs[idx] = v;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Explicit-buffer form of the same streaming steps, for the persistent decode kernel (decode_fused.hip), which keeps
// the loads of the NEXT work unit in flight while the current one is consumed.
// ---------------------------------------------------------------------------------------------------------------
struct T16Unit {                    // one wave's share of one 16-column tile
    const uint4* base;              // tile base + lane
    int rb0, rb1, rbsafe;           // row-block range [rb0, rb1); rbsafe: a valid row-block for clamped addresses
    int n;                          // this lane's column
};

__device__ __forceinline__ T16Unit t16_unit(const T16Matrix& m, int t, int lane, int rb_begin, int rb_end)
{
    T16Unit u;
    u.base = m.qw + (size_t) t * m.RB * 64 + lane;
    u.rb0 = rb_begin; u.rb1 = rb_end < rb_begin ? rb_begin : rb_end;
    u.rbsafe = min(rb_begin, m.RB - 1);
    u.n = t * 16 + (lane & 15);
    return u;
}

// G16 entries of a unit: slot h holds the entry of row-block rb0 + 4h + rsub
template <int NSLOT>
__device__ __forceinline__ void t16_unit_entries(const T16Matrix& m, const T16Unit& u, int rsub, uint32_t (&ent)[NSLOT])
{
#pragma unroll
    for (int h = 0; h < NSLOT; ++h) {
        const int rb = min(u.rb0 + 4 * h + rsub, m.RB - 1);
        const uint32_t e = t16_load_entry(m, t16_group_of_row(m, rb * 16), u.n);
        ent[h] = (u.rb0 + 4 * h + rsub < u.rb1) ? e : 0u;               // rows past the range: scale 0 -> contribute nothing
    }
}

template <int U, bool G16>
__device__ __forceinline__ void t16_unit_issue(const T16Matrix& m, const T16Unit& u, int pass, int rsub, uint4 (&wv)[U],
                                               uint32_t (&entp)[U])
{
#pragma unroll
    for (int i = 0; i < U; ++i) {
        const int rb = u.rb0 + pass * U + i;
        const int rbc = rb < u.rb1 ? rb : u.rbsafe;
        if constexpr (!G16) {
            const uint32_t e = t16_load_entry(m, t16_group_of_row(m, rbc * 16 + rsub * 4), u.n);
            entp[i] = rb < u.rb1 ? e : 0u;
        }
        wv[i] = nt_load16(u.base + (size_t) rbc * 64);
    }
}

template <int U, bool G16, int NSLOT>
__device__ __forceinline__ void t16_unit_consume(const T16Unit& u, int pass, int lane, const uint4 (&wv)[U],
                                                 const uint32_t (&ent)[NSLOT], const uint32_t (&entp)[U], const uint4* xrow,
                                                 f32x4& c)
{
    const uint32_t magic = t16_magic();
    const int col = lane & 15, rsub = lane >> 4;
#pragma unroll
    for (int i = 0; i < U; ++i) {
        const int li = pass * U + i;
        const int rb = u.rb0 + li;
        const int rbc = rb < u.rb1 ? rb : u.rbsafe;                     // branch-free: out-of-range row-blocks carry scale 0
        uint32_t e;
        if constexpr (G16) e = (uint32_t) __shfl((int) ent[(li >> 2) < NSLOT ? (li >> 2) : 0], ((li & 3) << 4) | col, 64);
        else e = entp[i];
        t16_rowblock<G16>(wv[i], e, magic, xrow + rbc * 16 + rsub * 4, c);
    }
}
