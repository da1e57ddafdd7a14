// Short-prompt GEMM (up to a few hundred activation rows): out[M,N] (+)= x[M,K] @ dequant(W[K,N]) on the T16 weight layout.
//
// Replaces, for short prompts (BASELINE configs[0]: the 128-token prompt), the reference's reconstruct-to-fp16 + cuBLAS
// path, /root/reference/exllama_ext/cuda_func/q4_matmul.cu:301-344, which at this height writes and re-reads a K x N fp16
// temporary per matmul.
//
// The 128 x 128 tile kernel (q4_gemm.hip: q4_gemm_t16m_kernel) gives 32 .. 86 blocks for a 7B layer at this height: a
// quarter of the 256 CUs work, each walking 64 .. 172 barrier-separated K steps (38 - 100 us per matmul whatever the
// height).  This kernel keeps the DECODE kernel's shape instead (q4_gemv.hip, gemv_t16.h): a block owns 32 activation rows x
// CT 16-column weight tiles (CT = 4, 2, 1: the widest that still gives ~256 blocks) and its 8 waves split K, so there is no
// barrier in the K loop.  Per 128-k row-block a wave
//   * loads its CT 1 KiB weight pieces and expands them ONCE to MFMA B operands (the reference's reconstruct bits,
//     h(h(q - z) * s), 13 VALU per 8 weights),
//   * copies its (32 or 64) x 128 activation slab with coalesced 1 KiB LDS-DMA loads into a wave-private, XOR-swizzled LDS
//     slab (conflict-free) and reads it back in the A operand layout -- lane (row r, k-group kg) holds 8 consecutive k of row r; read
//     straight from the row-major activations those 16 lanes touch 16 cache lines per instruction: measured 59 cycles per
//     1 KiB load instead of 16 -- and
//   * issues 2 x CT x 4 v_mfma_f32_16x16x32_f16, fp32 accumulate.
// The 8 K-slices are summed through LDS in a fixed order: bit-reproducible, no atomics.  All weight loads of (up to) four
// row-blocks go out together, ahead of the activation copies: vmcnt retires in order, so a wait for the (L2-resident)
// activations of the next step would otherwise also wait for any HBM weight load issued before it.
//
// How it got here (scripts/bench_gemm.cpp, eager launches, 7B q_proj at 128 rows; profiles/r02_short_prompt_gemm.txt): tile
// kernel 38 us -> one 16-column tile per block, 128 rows, A operands straight from global memory 33 us (the cache-line
// scatter above) -> the same from a pre-tiled activation copy 14 + 3 us (every block pulls the whole 1 MB activation
// through its L1: 18 TB/s of L2 traffic) -> 32 x 64 / 64 x 32 blocks with the LDS turn 11.7 us, of which 4.5 us is the
// empty launch in this eager loop and the four phases (weight latency, activation copies, arithmetic, reduction) cost
// 1.5 - 2 us each, one after the other: with 4 steps per wave and one block per CU nothing overlaps them.
#include "gemv_t16.h"
#include <type_traits>

#define GS_WAVES 8
#define GS_WD 4                       // row-blocks of weights requested at once

T16Matrix t16_view(const Q4Matrix* m);

// t16_dequant (gemv_t16.h) with the magic constant in a VGPR: (w & mask) | magic is ONE v_and_or_b32
__device__ __forceinline__ f16x8 gs_dequant(uint32_t w, uint32_t magic, f16x2 zc0, f16x2 zc1, f16x2 s2)
{
    const f16x2 sixteenth = {(f16) 0.0625f, (f16) 0.0625f};
    const uint32_t w8 = w >> 8;
    const f16x2 d0 = (t16_h2((w & 0x000F000Fu) | magic) + zc0) * s2;
    const f16x2 d1 = (t16_h2((w & 0x00F000F0u) | magic) * sixteenth + zc1) * s2;
    const f16x2 d2 = (t16_h2((w8 & 0x000F000Fu) | magic) + zc0) * s2;
    const f16x2 d3 = (t16_h2((w8 & 0x00F000F0u) | magic) * sixteenth + zc1) * s2;
    const uint4 u = make_uint4(__builtin_bit_cast(uint32_t, d0), __builtin_bit_cast(uint32_t, d1),
                               __builtin_bit_cast(uint32_t, d2), __builtin_bit_cast(uint32_t, d3));
    return __builtin_bit_cast(f16x8, u);
}

template <int MT, int CT>   // block = MT row tiles of 16 activation rows x CT 16-column weight tiles
__global__ __launch_bounds__(GS_WAVES * 64) void q4_gemm_t16s_kernel(const T16Matrix m, const f16* __restrict__ x,
                                                                      f16* __restrict__ out, int rows, int no_zero,
                                                                      int rb_per_wave, int nrg, int ncg)
{
    // slabs: [wave][16 MT rows][16 chunks of 16 B] = MT x 4 KiB per wave; after the K loop the same memory holds the partial sums
    constexpr int GS_ROWS = MT * 16;
    constexpr int RED_FLOATS = MT * CT * 256;
    constexpr int SLAB_BYTES = MT * 4096;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;

    // block -> (column group, row group): the row groups of one column group sit next to each other on ONE XCD, so a weight
    // tile comes from HBM once and from that XCD's L2 for the other row groups
    int b = blockIdx.x;
    if ((gridDim.x & 7) == 0) { const int per = gridDim.x >> 3; b = (b & 7) * per + (b >> 3); }
    const int cg = b / nrg, rg = b - cg * nrg;
    const int t0 = cg * CT;
    const int ntiles = m.N >> 4;
    const int r0 = rg * GS_ROWS;
    const int col = lane & 15, rsub = lane >> 4;
    const int rb0 = wave * rb_per_wave;
    const int rb1 = min(m.RB, rb0 + rb_per_wave);
    const int gsh = m.G == 1 ? 31 : m.gshift;                                 // group of a packed row: one shift (launcher: power-of-two groups)

    f32x4 acc[MT][CT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) acc[mt][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};

    if (rb0 < rb1) {                                                         // wave-uniform (short K: trailing waves idle)
        // activation slab loads: instruction i covers rows 4 i + (lane >> 4), 16-byte chunk (lane & 15) of the row-block
        uint32_t xoff[4 * MT];
#pragma unroll
        for (int i = 0; i < 4 * MT; ++i)                                      // (the DMA lands lane l in slot l: the swizzle is in the source address)
            xoff[i] = ((uint32_t) (min(r0 + 4 * i + rsub, rows - 1) - r0) * (uint32_t) m.K + (col ^ ((4 * i + rsub) & 15)) * 8) * 2u;
        const unsigned char* xb = (const unsigned char*) (x + (size_t) r0 * m.K);
        unsigned char* slab = smem + wave * SLAB_BYTES;
        // weights: piece of tile t, row-block rb, this lane
        uint32_t wb[CT];                                                       // byte offset of (tile, lane) in the weight array
        int ncol[CT];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            const int t = min(t0 + ct, ntiles - 1);                           // ragged last column group: a valid tile, result dropped
            wb[ct] = ((uint32_t) t * (uint32_t) m.RB * 64u + lane) * 16u;
            ncol[ct] = t * 16 + col;
        }
        u32x4 a[MT][4];
        uint4 w[GS_WD][CT];                                                    // the weights of GS_WD row-blocks: ALL requested before the first
        uint32_t z[GS_WD][CT];                                                 // is used (HBM latency is paid once per chunk, not once per step --
        uint32_t sb[GS_WD][CT];                                                // vmcnt retires in order: a wait for the L2-resident activations
                                                                               // of the next step would wait for any weight load issued before)
        // activation slab of row-block rb -> LDS by DMA (global_load_lds_dwordx4: 1 KiB per instruction, no VGPR round trip)
        auto load_g = [&](int rb) __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < 4 * MT; ++i)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*) (xb + (xoff[i] + (uint32_t) rb * 256u)),
                                                 (__attribute__((address_space(3))) unsigned char*) (slab + i * 1024), 16, 0, 0);
        };
        auto load_chunk = [&](int c0) __attribute__((always_inline)) {
#pragma unroll
            for (int u = 0; u < GS_WD; ++u) {
                const int rb = min(c0 + u, rb1 - 1);                           // past the wave's range: a valid address, never used
                const int grp = (rb * 16 + rsub * 4) >> gsh;
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) {
                    w[u][ct] = *(const uint4*) ((const unsigned char*) m.qw + (wb[ct] + (uint32_t) rb * 1024u));
                    z[u][ct] = m.qzeros[(size_t) grp * (m.N >> 3) + (ncol[ct] >> 3)];
                    sb[u][ct] = ((const uint16_t*) m.scales)[(size_t) grp * m.N + ncol[ct]];
                }
            }
        };
        // everything requested so far has arrived (the compiler does not count LDS-DMA: explicit wait; the "+v" operands tell its
        // own counter that the weight registers are complete, so it adds no conservative wait later, behind a fresh DMA)
        auto wait_all = [&]() __attribute__((always_inline)) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int u = 0; u < GS_WD; ++u)
#pragma unroll
                for (int ct = 0; ct < CT; ++ct)
                    asm volatile("" : "+v"(w[u][ct].x), "+v"(w[u][ct].y), "+v"(w[u][ct].z), "+v"(w[u][ct].w), "+v"(z[u][ct]), "+v"(sb[u][ct]));
        };
        // slab (chunk c of row r in 16-byte slot c ^ (r & 15)) -> A operands: MFMA j of row tile mt, lane (row r = lane & 15,
        // k-group kg = lane >> 4) takes chunk 4 kg + j of row 16 mt + r.  The slab is private to the wave: no barrier.
        auto read_a = [&]() __attribute__((always_inline)) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int j = 0; j < 4; ++j) a[mt][j] = *(const u32x4*) (slab + (mt * 16 + col) * 256 + (((rsub * 4 + j) ^ col) << 4));
        };
        const f16x2 c960 = {(f16) 960.f, (f16) 960.f};
        const uint32_t magic = t16_magic();
        auto compute = [&](int u) __attribute__((always_inline)) {
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                const f16 sc = __builtin_bit_cast(f16, (uint16_t) sb[u][ct]);
                const f16 za = __builtin_bit_cast(f16, (uint16_t) (0xE401u + ((z[u][ct] >> ((ncol[ct] & 7) * 4)) & 0xFu)));   // -(1024 + z), gemv_t16.h
                const f16x2 zc0 = {za, za};
                const f16x2 zc1 = zc0 + c960;
                const f16x2 s2 = {sc, sc};
                const f16x8 b0 = gs_dequant(w[u][ct].x, magic, zc0, zc1, s2);
                const f16x8 b1 = gs_dequant(w[u][ct].y, magic, zc0, zc1, s2);
                const f16x8 b2 = gs_dequant(w[u][ct].z, magic, zc0, zc1, s2);
                const f16x8 b3 = gs_dequant(w[u][ct].w, magic, zc0, zc1, s2);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    acc[mt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a[mt][0]), b0, acc[mt][ct], 0, 0, 0);
                    acc[mt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a[mt][1]), b1, acc[mt][ct], 0, 0, 0);
                    acc[mt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a[mt][2]), b2, acc[mt][ct], 0, 0, 0);
                    acc[mt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a[mt][3]), b3, acc[mt][ct], 0, 0, 0);
                }
            }
        };

        load_chunk(rb0);
        load_g(rb0);
        wait_all();
        read_a();
        for (int c0 = rb0; c0 < rb1; c0 += GS_WD) {
#pragma unroll
            for (int u = 0; u < GS_WD; ++u) {
                const int rb = c0 + u;
                if (rb < rb1) {                                               // wave-uniform
                    const bool more = rb + 1 < rb1;
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // this step's operands have left the slab
                    if (more) load_g(rb + 1);                                 // the next step's activations go out before this step's arithmetic
                    __builtin_amdgcn_sched_barrier(0);
                    compute(u);
                    __builtin_amdgcn_sched_barrier(0);
                    if (u == GS_WD - 1 && more) load_chunk(rb + 1);           // the next chunk's weights: after the last use of the registers
                    if (more) { wait_all(); read_a(); }
                }
            }
        }
    }

    // D rows 4 * rsub + j of tile (mt, ct), column col -> red[wave][mt * CT + ct][row][col]
    __syncthreads();                                                          // every wave is done with its slab
    float* red = (float*) smem;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int j = 0; j < 4; ++j) red[wave * RED_FLOATS + (mt * CT + ct) * 256 + (rsub * 4 + j) * 16 + col] = acc[mt][ct][j];
    __syncthreads();
    constexpr int BW = CT * 16;                                               // block width in columns
    for (int idx = tid; idx < GS_ROWS * BW; idx += GS_WAVES * 64) {
        const int r = idx / BW, c = idx - r * BW;
        const int row = r0 + r, n = t0 * 16 + c;
        if (row < rows && n < m.N) {
            const int ri = ((r >> 4) * CT + (c >> 4)) * 256 + (r & 15) * 16 + (c & 15);
            float v = 0.f;
#pragma unroll
            for (int wv = 0; wv < GS_WAVES; ++wv) v += red[wv * RED_FLOATS + ri];
            f16* o = out + (size_t) row * m.N + n;
            if (no_zero) v += (float) *o;
            *o = (f16) v;
        }
    }
}

// x: row-major activations, already gathered through x_map for act-order weights (the caller's column_remap).
// Returns 1 for what this kernel does not cover (the caller falls through to the tile kernel).
template <int MT, int CT>
static int launch_t16s_cfg(const T16Matrix& m, const f16* x, int rows, f16* out, int no_zero, hipStream_t s)
{
    const int rbw = (m.RB + GS_WAVES - 1) / GS_WAVES;
    const int ntiles = m.N / 16;
    const int nrg = (rows + MT * 16 - 1) / (MT * 16);
    const int ncg = (ntiles + CT - 1) / CT;
    const size_t slab = (size_t) MT * 4096, red = (size_t) MT * CT * 1024;
    const size_t smem = GS_WAVES * (slab > red ? slab : red);
    auto kfn = q4_gemm_t16s_kernel<MT, CT>;
    static bool big[EXL_MAX_DEVICES] = {};
    if (smem > 64 * 1024) EXL_TRY(exl_lds_opt_in((const void*) kfn, big));
    hipLaunchKernelGGL(kfn, dim3((unsigned) (nrg * ncg)), dim3(GS_WAVES * 64), smem, s, m, x, out, rows, no_zero, rbw, nrg, ncg);
    EXL_LAUNCH_CHECK();
    return 0;
}

int launch_gemm_t16s(const Q4Matrix* w, const f16* x, int rows, f16* out, int no_zero, hipStream_t s)
{
    const T16Matrix m = t16_view(w);
    EXL_REQUIRE(m.N % 16 == 0 && m.K % 128 == 0, EXL_E_UNSUPPORTED, "q4 gemm (short prompt): T16 layout needs N %% 16 == 0 and K %% 128 == 0");
    if (m.G > 1 && m.gshift < 0) return 1;                                    // odd group sizes: the tile kernel
    if ((uint64_t) 64 * (uint64_t) m.K >= (1ull << 31) || (uint64_t) m.K * (uint64_t) m.N >= (1ull << 32)) return 1;   // 32-bit offsets
    // Block shape: the dequantisation (13 VALU per 8 weights) is repeated by every row group, the activation slab by every
    // column group: 64 rows x 32 columns while that still gives ~a block per CU, then 32 x 64, 32 x 32, 32 x 16.
    const int ntiles = m.N / 16;
    auto blocks = [&](int mt, int ct) { return (long) ((rows + mt * 16 - 1) / (mt * 16)) * ((ntiles + ct - 1) / ct); };
    int cfg = 0;                                                              // 0: (4,2)  1: (2,4)  2: (2,2)  3: (2,1)
    if (blocks(4, 2) >= 224 && rows > 32) cfg = 0;
    else if (blocks(2, 4) >= 224) cfg = 1;
    else if (blocks(2, 2) >= 224) cfg = 2;
    else cfg = 3;
    static const char* force = getenv("EXL_GEMM_SKINNY_CFG");                 // measurement aid
    if (force && force[0] >= '0' && force[0] <= '3') cfg = force[0] - '0';
    switch (cfg) {
        case 0:  return launch_t16s_cfg<4, 2>(m, x, rows, out, no_zero, s);
        case 1:  return launch_t16s_cfg<2, 4>(m, x, rows, out, no_zero, s);
        case 2:  return launch_t16s_cfg<2, 2>(m, x, rows, out, no_zero, s);
        default: return launch_t16s_cfg<2, 1>(m, x, rows, out, no_zero, s);
    }
}
