// Short-prompt GEMMs on FRAGMENT-ORDER activations (up to a few hundred rows: BASELINE configs[0], the 128-token prompt, and every
// chat turn): out[M, N] (+)= x[M, K] @ dequant(W[K, N]) for one, two (gate / up + SiLU * mul) or three (q / k / v) T16 matrices per
// launch.  Replaces, at this height, the reference's reconstruct-to-fp16 + cuBLAS path (q4_matmul.cu:301-344; model.py:532-552 runs
// it as seven matmuls + norm / rope / SiLU launches per layer).
//
// Why another kernel.  q4_gemm_t16s (q4_gemm_skinny.hip) keeps the decode kernel's shape -- a block owns 64 rows x 32 columns, its 8
// waves split K, no barrier in the K loop -- but every wave turns its 64 x 128 activation slab through a private 16 KiB LDS slab per
// step (row-major activations cannot be read in the MFMA A layout without touching 16 cache lines per instruction).  One slab per
// wave is all the LDS holds, so the copy of step s + 1 starts when step s has been read out and the step is a serial chain
// copy -> wait -> read -> MFMA: 2.5 us per step where the arithmetic is 0.35 (r06e: 11.9 us for a 4096 x 4096 matrix at 128 rows,
// 20.9 us on average over a layer's seven).  Here the PRODUCER of every activation tensor writes it in the order the MFMA wants:
//     xf[(mt * (K / 32) + 4 rb + j) * 64 + lane]  (16 bytes)  =  x[16 mt + (lane & 15)][128 rb + 32 (lane >> 4) + 8 j .. + 8]
// (row-block rb, MFMA j of the row-block: the k a T16 weight piece feeds to that MFMA, gemv_t16.h)
// (the RMSNorm kernel below for q / k / v and gate / up, the gate / up epilogue for down_proj, a re-tile pass behind the attention
// kernel for o_proj), so an A operand is ONE coalesced 1 KiB load straight into registers: no LDS in the K loop, the loads of step
// s + 1 are issued while step s multiplies (hand-counted vmcnt, as in decode_ring.hip), and what bounds a block is the 64 B / clk
// at which a CU's L1 fills: rows x K x 2 bytes per block.
//
// Block = MT (4) row tiles x CT column tiles, 8 waves splitting K (fixed order LDS reduction: bit-reproducible, no atomics); weights
// are dequantised ONCE per block and step to the reference's reconstruct bits h(h(q - z) * s) (q4_matrix.cu:207), 13 VALU per 8.
#include "decode_ring.h"

#include <mutex>

#define GR_WAVES 8

struct GrMat { const unsigned char* qw; const uint32_t* qz; const uint16_t* sc; int N, RB, gsh; };
struct GrArgs {
    GrMat m[3];
    int tile_end[3];              // cumulative 16-column tiles over the matrices of the launch (EPI 1: tiles of m[0]; m[1] in lock-step)
    const unsigned char* xf;      // fragment-order activations, [mtiles][K / 32][64] x 16 bytes (rows past `rows` are zero)
    int rows, K;
    f16* out[3];                  // EPI 0: row-major outputs (leading dimension m[i].N)
    int no_zero;                  // EPI 0: out += (the residual add of o_proj / down_proj)
    unsigned char* out_frag;      // EPI 1: silu(gate) * up in fragment order for a consumer with K = m[0].N
    int rb_per_wave, nrg, ncg;
    int stagger;                  // blocks start their walk over K at different row-blocks (see q4_gemm_t16g_kernel)
    int ks;                       // q4_gemm_t16g, EPI 0: K is cut over ks blocks per output tile; the last one to arrive adds the slices up
    float* kws;                   //   fp32 slices [tile][ks][items][8]
    int* kcnt;                    //   arrivals per tile: zero before the launch, zero behind it
    float* rowsq;                 // EPI 0, one matrix: rowsq[row * rowsq_stride + slot] = sum of the squares of the row's FINAL fp16 values in the
    int rowsq_stride;             // columns of slot (a column group / column-wave): the RMSNorm behind this launch adds the slots up (to_frag_kernel)
};

#ifdef EXL_T16G_PROBE
// Phase stamps of q4_gemm_t16g's pipelined step (scripts/probe_t16g.py; NOT in the product build): s_memtime of wave 0 and wave NCW (the
// first wave of each K-group) of blocks 0 and gridDim.x / 2, at the marks below.
__device__ unsigned long long g_t16g_probe[4][64];
#define GG_STAMP(i) do { if (pslot >= 0 && (i) < 64) g_t16g_probe[pslot][(i)] = __builtin_amdgcn_s_memtime(); } while (0)
extern "C" int exl_debug_t16g_probe(unsigned long long* out)
{
    return (int) hipMemcpyFromSymbol(out, HIP_SYMBOL(g_t16g_probe), sizeof(unsigned long long) * 4 * 64);
}
#else
#define GG_STAMP(i) do { } while (0)
#endif

namespace {
__device__ __forceinline__ void gr_ld16(u32x4& d, uint32_t voff, const void* sbase)   // uniform base + lane offset (L2-resident or re-read: no nt)
{
    asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2" : "=v"(d) : "v"(voff), "s"(rg_uniform(sbase)) : "memory");
}
// the wait of a step: the six registers it releases are operands (nothing that reads them can be placed above it)
template <int N> __device__ __forceinline__ void gg_wait(u32x4& a, u32x4& b, uint32_t& c, uint32_t& d, uint32_t& e, uint32_t& f)
{
    asm volatile("s_waitcnt vmcnt(%6)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f) : "n"(N) : "memory");
}
__device__ __forceinline__ f16x8 gr_dequant(uint32_t w, uint32_t magic, f16x2 zc0, f16x2 zc1, f16x2 s2)
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
}  // namespace

// EPI 0: plain / accumulating row-major store; EPI 1: tiles ct < CT / 2 are gate tiles, ct >= CT / 2 the same tiles of up: silu(g) * u
// is stored in fragment order.
template <int MT, int CT, int EPI>
__global__ __launch_bounds__(GR_WAVES * 64) void q4_gemm_t16r_kernel(const GrArgs a)
{
    constexpr int ROWS = MT * 16;
    constexpr int RED = MT * CT * 256;                               // floats per wave
    constexpr int NW_LOADS = 3 * CT;                                 // weight piece + zero word + scale per tile and step
    constexpr int NA_LOADS = 4;                                      // per row tile and step
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int col = lane & 15, kg = lane >> 4;
    const uint32_t lane16 = (uint32_t) lane * 16u;

    // block -> (column group, row group): the row groups of one column group sit next to each other on ONE XCD (the weight tile comes
    // from HBM once, from that XCD's L2 for the other row groups)
    int b = blockIdx.x;
    if ((gridDim.x & 7) == 0) { const int per = gridDim.x >> 3; b = (b & 7) * per + (b >> 3); }
    const int nrg = a.nrg;
    const int cg = b / nrg, rg = b - cg * nrg;
    const int K32 = a.K >> 5;

    // the CT tiles of this block: (matrix, tile, first column)
    const unsigned char* wq[CT]; const unsigned char* zq[CT]; const unsigned char* sq[CT];
    uint32_t zoff[CT], soff[CT];                                     // lane offsets of the zero word / scale of (this lane's k-group, column)
    int gsh[CT], n8[CT], nn[CT], mi_[CT], n0_[CT], shz[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        int mi = 0, tile;
        if constexpr (EPI == 1) { mi = ct >= CT / 2 ? 1 : 0; tile = cg * (CT / 2) + (ct % (CT / 2)); }
        else {
            tile = cg * CT + ct;
            if (tile >= a.tile_end[0]) { mi = 1; if (tile >= a.tile_end[1]) mi = 2; }
            tile -= mi == 0 ? 0 : a.tile_end[mi - 1];
        }
        const GrMat m = mi == 0 ? a.m[0] : mi == 1 ? a.m[1] : a.m[2];
        const int ntiles = m.N >> 4;
        const int t = tile < ntiles ? tile : ntiles - 1;              // ragged last column group: a valid tile, result dropped
        wq[ct] = m.qw + (size_t) (uint32_t) t * (uint32_t) m.RB * 1024u;
        const int n = t * 16 + col;
        const uint32_t gl = (uint32_t) (kg * 4) >> (uint32_t) m.gsh;  // group of the lane's 32 k inside the row-block (group sizes 32 / 64)
        zq[ct] = (const unsigned char*) m.qz; sq[ct] = (const unsigned char*) m.sc;
        zoff[ct] = (gl * (uint32_t) (m.N >> 3) + ((uint32_t) n >> 3)) * 4u;
        soff[ct] = (gl * (uint32_t) m.N + (uint32_t) n) * 2u;
        gsh[ct] = m.gsh; n8[ct] = m.N >> 3; nn[ct] = m.N; mi_[ct] = mi; n0_[ct] = tile < ntiles ? tile * 16 : -1;
        shz[ct] = (n & 7) * 4;
    }
    const int RB = a.m[0].RB;
    // (which K slice a wave takes rotates from block to block as well: with the start offset below, the blocks of an XCD touch a given
    // activation line at 8 x rb_per_wave different times)
    const int li = a.stagger ? (int) (blockIdx.x >> 3) : 0;
    const int slice = (wave + li / max(a.rb_per_wave, 1)) & (GR_WAVES - 1);
    const int rb0 = slice * a.rb_per_wave;
    const int rb1 = min(RB, rb0 + a.rb_per_wave);
    const int nsteps = rb1 - rb0;                                     // (wave-uniform; <= 0: this wave idles)

    f32x4 acc[MT][CT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) acc[mt][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};

    if (nsteps > 0) {
        u32x4 av[MT][4], wr[CT];
        uint32_t zr[CT], sr[CT];
        const unsigned char* xb = a.xf + (size_t) (uint32_t) (rg * MT) * (uint32_t) K32 * 1024u;
        auto issue_w = [&](int rb) {                                 // NW_LOADS loads
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                gr_ld16(wr[ct], lane16, wq[ct] + (size_t) (uint32_t) rb * 1024u);
                const uint32_t g0 = (uint32_t) (rb * 16) >> (uint32_t) gsh[ct];
                rg_ld4s(zr[ct], zoff[ct], zq[ct] + (size_t) g0 * (uint32_t) n8[ct] * 4u);
                rg_ld2s(sr[ct], soff[ct], sq[ct] + (size_t) g0 * (uint32_t) nn[ct] * 2u);
            }
        };
        auto issue_a = [&](int rb, auto mc) {                        // NA_LOADS loads: the four k-blocks of the row-block, row tile mt
            constexpr int mt = decltype(mc)::value;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                gr_ld16(av[mt][j], lane16, xb + ((size_t) (uint32_t) mt * (uint32_t) K32 + (uint32_t) (rb * 4 + j)) * 1024u);
        };
        const f16x2 c960 = {(f16) 960.f, (f16) 960.f};
        const uint32_t magic = t16_magic();
        // one step: LAST = nothing behind it (no requests go out; the waits count down)
        auto step = [&](auto last_tag, int rbn) {                     // rbn: the row-block to request while this one multiplies
            constexpr bool LAST = decltype(last_tag)::value;
            rg_wait<NA_LOADS * MT>();                                 // the weights of this step (the A loads behind them may still fly)
            f16x8 bq[CT][4];
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                rg_tie(wr[ct]); rg_tie(zr[ct]); rg_tie(sr[ct]);
                const f16 sc = __builtin_bit_cast(f16, (uint16_t) sr[ct]);
                const f16 za = __builtin_bit_cast(f16, (uint16_t) (0xE401u + ((zr[ct] >> (uint32_t) shz[ct]) & 0xFu)));   // -(1024 + z), gemv_t16.h
                const f16x2 zc0 = {za, za};
                const f16x2 zc1 = zc0 + c960;
                const f16x2 s2 = {sc, sc};
                bq[ct][0] = gr_dequant(wr[ct][0], magic, zc0, zc1, s2);
                bq[ct][1] = gr_dequant(wr[ct][1], magic, zc0, zc1, s2);
                bq[ct][2] = gr_dequant(wr[ct][2], magic, zc0, zc1, s2);
                bq[ct][3] = gr_dequant(wr[ct][3], magic, zc0, zc1, s2);
            }
            if constexpr (!LAST) issue_w(rbn);
            static_for<0, MT>([&](auto mc) {
                constexpr int mt = decltype(mc)::value;
                // row tile mt of this step: behind it in the queue are the later row tiles of this step, the next step's weights and
                // the next step's earlier row tiles -- the same number at every mt
                if constexpr (!LAST) rg_wait<NA_LOADS * (MT - 1) + NW_LOADS>(av[mt][0]);
                else rg_wait<NA_LOADS * (MT - 1 - mt)>(av[mt][0]);
                rg_tie(av[mt][1]); rg_tie(av[mt][2]); rg_tie(av[mt][3]);
#pragma unroll
                for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[mt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, av[mt][j]), bq[ct][j], acc[mt][ct], 0, 0, 0);
                if constexpr (!LAST) issue_a(rbn, mc);
            });
        };
        // the walk over the wave's row-blocks starts at a different one in every block of an XCD (q4_gemm_t16g_kernel: why)
        const int off = a.stagger ? (int) ((unsigned) (blockIdx.x >> 3) % (unsigned) nsteps) : 0;
        auto at = [&](int i) { int t = i + off; t = t >= nsteps ? t - nsteps : t; return rb0 + t; };
        issue_w(at(0));
        static_for<0, MT>([&](auto mc) { issue_a(at(0), mc); });
        int i = 0;
        for (; i + 1 < nsteps; ++i) step(std::false_type{}, at(i + 1));
        step(std::true_type{}, 0);
    }

    // ---- K-slice reduction through LDS, fixed order -----------------------------------------------------------------------------------
    float* red = (float*) smem;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int j = 0; j < 4; ++j) red[wave * RED + (mt * CT + ct) * 256 + (kg * 4 + j) * 16 + col] = acc[mt][ct][j];
    __syncthreads();
    constexpr int OCT = EPI == 1 ? CT / 2 : CT;                       // output tiles per block
    constexpr int ITEMS = ROWS * OCT * 2;                             // (row, tile, half): 8 consecutive columns each
    const int r0 = rg * ROWS;
    for (int it = tid; it < ITEMS; it += GR_WAVES * 64) {
        const int hf = it & 1, ct = (it >> 1) % OCT, r = it / (2 * OCT);
        const int ri = ((r >> 4) * CT + ct) * 256 + (r & 15) * 16 + hf * 8;
        float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int wv = 0; wv < GR_WAVES; ++wv) {
            const float4 p0 = *(const float4*) (red + wv * RED + ri), p1 = *(const float4*) (red + wv * RED + ri + 4);
            v[0] += p0.x; v[1] += p0.y; v[2] += p0.z; v[3] += p0.w; v[4] += p1.x; v[5] += p1.y; v[6] += p1.z; v[7] += p1.w;
        }
        const int row = r0 + r;
        // (the tile description lives in per-lane registers indexed by a compile-time ct: select with a short chain)
        int n0 = -1, mi = 0, ldn = 0;
#pragma unroll
        for (int q = 0; q < OCT; ++q) if (q == ct) { n0 = n0_[q]; mi = mi_[q]; ldn = nn[q]; }
        if constexpr (EPI == 1) {
            if (n0 < 0) continue;
            float u[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            const int ru = ((r >> 4) * CT + ct + CT / 2) * 256 + (r & 15) * 16 + hf * 8;
#pragma unroll
            for (int wv = 0; wv < GR_WAVES; ++wv) {
                const float4 p0 = *(const float4*) (red + wv * RED + ru), p1 = *(const float4*) (red + wv * RED + ru + 4);
                u[0] += p0.x; u[1] += p0.y; u[2] += p0.z; u[3] += p0.w; u[4] += p1.x; u[5] += p1.y; u[6] += p1.z; u[7] += p1.w;
            }
            f16x8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = silu_mul_f16((f16) v[j], (f16) u[j]);
            const int n = n0 + hf * 8;                                // the consumer's k index of the first of the 8 values
            const int c = n >> 3;                                     // chunk of the consumer's K (to_frag_kernel: the same placement)
            const size_t piece = ((size_t) (row >> 4) * (size_t) (ldn >> 5) + (size_t) ((c >> 4) * 4 + (c & 3))) * 64 + (size_t) (((c >> 2) & 3) * 16 + (row & 15));
            *(uint4*) (a.out_frag + piece * 16) = __builtin_bit_cast(uint4, o);      // (padding rows: silu(0) * 0 = 0, stays zero)
        } else {
            float ss = 0.f;
            if (n0 >= 0 && row < a.rows) {
                f16* o = a.out[mi] + (size_t) row * ldn + n0 + hf * 8;
                f16x8 ov;
                if (a.no_zero) {
                    const f16x8 old = *(const f16x8*) o;
#pragma unroll
                    for (int j = 0; j < 8; ++j) ov[j] = (f16) (v[j] + (float) old[j]);
                } else {
#pragma unroll
                    for (int j = 0; j < 8; ++j) ov[j] = (f16) v[j];
                }
                *(f16x8*) o = ov;
#pragma unroll
                for (int j = 0; j < 8; ++j) { const float f = (float) ov[j]; ss = fmaf(f, f, ss); }
            }
            if (a.rowsq) {                                            // (uniform; the 2 OCT items of a row are neighbouring lanes of one wave)
#pragma unroll
                for (int m = 1; m < 2 * OCT; m <<= 1) ss += __shfl_xor(ss, m);
                if ((it & (2 * OCT - 1)) == 0 && row < a.rows) a.rowsq[(size_t) row * a.rowsq_stride + cg] = ss;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// q4_gemm_t16g: the same product with the activations SHARED through LDS (wide matrices: q / k / v, gate / up).
//
// What bounds q4_gemm_t16r (r06f, 7B layer at 128 rows: gate / up 42 us, q / k / v 29 us): every wave of a block pulls the activation
// slab of ITS K slice from L2 into registers -- rows x K x 2 bytes per block, 0.44 GB per gate / up launch, 1.24 GB per layer
// against 0.1 GB of weights -- and a CU takes that in at ~46 GB/s.  Here a block is NKW K-groups x NCW column-waves: the NCW waves of
// a K-group copy one slab (MT row tiles x 128 k = MT x 4 KiB, fragment order: 1 KiB LDS-DMA pieces, no registers) into a two-slot
// ring and ALL read it from LDS (256 B / clk) for their own two column tiles, so a block fetches rows x K x 2 bytes for NCW x 32 (EPI 1:
// NCW x 16 of gate AND up) columns instead of for CT x 16, and with MT = 8 every weight is expanded once for 128 rows.
// Step s of K-group kw = row-block s NKW + kw; per step and wave PPW pieces + 6 register loads, requested one step ahead;
// `s_waitcnt vmcnt(PPW + 6)`, barrier (the slab of step s is whole), LDS reads + MFMAs, barrier (its slot may be refilled).
// K-groups are summed through LDS in a fixed order (the ring's memory, after the loop); no atomics.
template <int MT, int NCW, int EPI, int PF>
__global__ __launch_bounds__(GR_WAVES * 64) void q4_gemm_t16g_kernel(const GrArgs a)
{
    constexpr int CT = 2;
    constexpr int NKW = GR_WAVES / NCW;
    constexpr int ROWS = MT * 16;
    constexpr int SLAB = MT * 4096;                                  // bytes: one step's activations of one K-group
    constexpr int PPW = MT * 4 / NCW;                                // 1 KiB pieces a wave copies per step
    constexpr int NW_LOADS = 3 * CT;
    constexpr int LOADS = PPW + NW_LOADS;                            // vector-memory instructions per wave and step
    constexpr int RED = MT * CT * 256;                               // floats per wave
    static_assert(MT * 4 % NCW == 0 && GR_WAVES % NCW == 0, "pieces / waves divide");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kw = wave / NCW, cw = wave - kw * NCW;
    const int col = lane & 15, kg = lane >> 4;
    const uint32_t lane16 = (uint32_t) lane * 16u;

    int b = blockIdx.x;
    if ((gridDim.x & 7) == 0) { const int per = gridDim.x >> 3; b = (b & 7) * per + (b >> 3); }   // (q4_gemm_t16r: row groups of a column group on one XCD)
    const int nrg = a.nrg;
    const int ks = a.ks > 1 ? a.ks : 1;                               // blocks per output tile (K ranges); the row groups of a range are neighbours
    const int cg = b / (nrg * ks), rem = b - cg * nrg * ks;
    const int kz = rem / nrg, rg = rem - kz * nrg;
    const int K32 = a.K >> 5;

    const unsigned char* wq[CT]; const unsigned char* zq[CT]; const unsigned char* sq[CT];
    uint32_t zoff[CT], soff[CT];
    int gsh[CT], n8[CT], nn[CT], mi_[CT], n0_[CT], shz[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        int mi = 0, tile;
        if constexpr (EPI == 1) { mi = ct; tile = cg * NCW + cw; }
        else {
            tile = (cg * NCW + cw) * CT + ct;
            if (tile >= a.tile_end[2]) tile = 1 << 28;                // behind the last matrix (ragged last column group)
            else {
                if (tile >= a.tile_end[0]) { mi = 1; if (tile >= a.tile_end[1]) mi = 2; }
                tile -= mi == 0 ? 0 : a.tile_end[mi - 1];
            }
        }
        const GrMat m = mi == 0 ? a.m[0] : mi == 1 ? a.m[1] : a.m[2];
        const int ntiles = m.N >> 4;
        const int t = tile < ntiles ? tile : ntiles - 1;              // ragged last column group: a valid tile, result dropped
        wq[ct] = m.qw + (size_t) (uint32_t) t * (uint32_t) m.RB * 1024u;
        const int n = t * 16 + col;
        const uint32_t gl = (uint32_t) (kg * 4) >> (uint32_t) m.gsh;
        zq[ct] = (const unsigned char*) m.qz; sq[ct] = (const unsigned char*) m.sc;
        zoff[ct] = (gl * (uint32_t) (m.N >> 3) + ((uint32_t) n >> 3)) * 4u;
        soff[ct] = (gl * (uint32_t) m.N + (uint32_t) n) * 2u;
        gsh[ct] = m.gsh; n8[ct] = m.N >> 3; nn[ct] = m.N; mi_[ct] = mi; n0_[ct] = tile < ntiles ? tile * 16 : -1;
        shz[ct] = (n & 7) * 4;
    }
    const int rb_lo = (int) ((long) kz * a.m[0].RB / ks);             // this block's row-blocks: [rb_lo, RB)
    const int RB = (int) ((long) (kz + 1) * a.m[0].RB / ks);
    const int nsteps = (RB - rb_lo + NKW - 1) / NKW;                  // the same for every wave (barriers); a K-group's last step may be empty
    // Where the walk over K starts (a.stagger, an experiment kept as a switch: blocks that run side by side on one XCD read the SAME
    // activation lines at the same moment; started one step apart they would not.  The L2 turned out to serve them either way.)
    const int off = a.stagger ? (int) ((unsigned) (blockIdx.x >> 3) % (unsigned) nsteps) : 0;

    f32x4 acc[MT][CT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) acc[mt][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const uint32_t ring = rg_lds_addr(smem) + (uint32_t) (kw * 2 * SLAB);
    const unsigned char* ringp = smem + kw * 2 * SLAB;
    const unsigned char* xb = a.xf + (size_t) (uint32_t) (rg * MT) * (uint32_t) K32 * 1024u + lane16;
    u32x4 wr[2][CT];
    uint32_t zr[2][CT], sr[2][CT];
    const f16x2 c960 = {(f16) 960.f, (f16) 960.f};
    const uint32_t magic = t16_magic();
    auto rb_of = [&](int s) {                                        // row-block of step s of this K-group (>= RB: an empty step)
        int sp = s + off;                                             // rotated step (the surplus step behind the last one stays behind)
        sp = s >= nsteps ? nsteps : sp >= nsteps ? sp - nsteps : sp;
        return rb_lo + sp * NKW + kw;
    };
    auto issue_piece = [&](int rb, int set, int i) {                  // piece i of this wave's share of the slab: (row tile p / 4, MFMA p % 4)
        const int p = cw * PPW + i;
        rg_dma16(ring + (uint32_t) (set * SLAB + p * 1024), xb + ((size_t) (uint32_t) (p >> 2) * (uint32_t) K32 + (uint32_t) (rb * 4 + (p & 3))) * 1024u);
    };
    if constexpr (PF == 1) {
    // ---- pipelined: everything that is not an MFMA is issued between the MFMAs of a row tile --------------------------------------------
    // r06h (the plain loop below): a wave spends 27 % of its cycles issuing, 42 % stalled on issue, 31 % parked at waits -- the phases of a
    // step (request, expand the weights, read LDS, multiply) run one after the other and the two waves of a SIMD are in the same phase.
    // Here a step is MT groups, one per row tile: each carries its share of the requests, of the expansion of the NEXT step's weights
    // (into the other half of bq) and the LDS reads of the next row tile, next to the 8 MFMAs of its own.
    // State when step s begins (p = s & 1): ring[p] = slab s, whole; bq[p] = the weights of step s, expanded; raw[p ^ 1] = the weights
    // of step s + 1, landed; raw[p] = those of step s + 2, on their way.  During step s: the pieces of slab s + 1 are requested in the
    // FIRST half of the groups (L2 hits: they have the rest of the step to land), raw[p ^ 1] is expanded into bq[p ^ 1], and when its
    // last word has been read the weights of step s + 3 are requested into it -- LAST: they come from HBM and have all of the next step
    // to arrive.  The wait at the end of the step is `vmcnt(6)`: the six youngest requests, exactly those, stay in flight; everything
    // older (slab s + 1, the weights of step s + 2) has landed -- then ONE barrier.  (r06i, with `vmcnt(0)`: 35 % of a wave's cycles
    // parked.)  An empty step (a K-group's surplus) expands its weights with scale 0: the MFMAs add zeros, no branch.
    f16x8 bq[2][CT][4];
    auto issue_wt = [&](int rb, auto rc, int ct) {
        constexpr int r = decltype(rc)::value;
        gr_ld16(wr[r][ct], lane16, wq[ct] + (size_t) (uint32_t) rb * 1024u);
        const uint32_t g0 = (uint32_t) (rb * 16) >> (uint32_t) gsh[ct];
        rg_ld4s(zr[r][ct], zoff[ct], zq[ct] + (size_t) g0 * (uint32_t) n8[ct] * 4u);
        rg_ld2s(sr[r][ct], soff[ct], sq[ct] + (size_t) g0 * (uint32_t) nn[ct] * 2u);
    };
    auto drain = [&]() {
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(wr[0][0]), "+v"(wr[0][1]), "+v"(zr[0][0]), "+v"(zr[0][1]), "+v"(sr[0][0]), "+v"(sr[0][1]),
                     "+v"(wr[1][0]), "+v"(wr[1][1]), "+v"(zr[1][0]), "+v"(zr[1][1]), "+v"(sr[1][0]), "+v"(sr[1][1]) :: "memory");
    };
    constexpr int HALF = MT >= 2 ? MT / 2 : 1;                        // groups that carry slab requests
    constexpr int PPG = (PPW + HALF - 1) / HALF;                      // pieces per such group
    constexpr int WPG = (CT * 4 + MT - 1) / MT;                       // weight words expanded per group
#ifdef EXL_T16G_PROBE
    const int pslot = (lane == 0 && (wave == 0 || wave == NCW) && (blockIdx.x == 0 || blockIdx.x == gridDim.x / 2))
                      ? (blockIdx.x == 0 ? 0 : 2) + (wave == 0 ? 0 : 1) : -1;
#endif
    GG_STAMP(0);
    auto step = [&](auto set_tag, int s) {
        constexpr int p = decltype(set_tag)::value;                   // s & 1
        constexpr int pn = p ^ 1;
        if (s < 10) GG_STAMP(2 + 6 * s);                              // step start
        int rbn = rb_of(s + 1), rbw = rb_of(s + 3);
        const bool live_n = rbn < RB;
        rbn = rbn < RB ? rbn : RB - 1;                                // (empty steps request the last row-block again: same counts)
        rbw = rbw < RB ? rbw : RB - 1;
        const unsigned char* slab = ringp + p * SLAB + lane16;
        f16x8 av[2][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) av[0][j] = *(const f16x8*) (slab + j * 1024);
        f16x2 zc0[CT], zc1[CT], sc2[CT];                              // per tile: -(1024 + z), that + 960, the scale (0 for an empty step)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            const f16 sc = live_n ? __builtin_bit_cast(f16, (uint16_t) sr[pn][ct]) : (f16) 0.f;
            const f16 za = __builtin_bit_cast(f16, (uint16_t) (0xE401u + ((zr[pn][ct] >> (uint32_t) shz[ct]) & 0xFu)));   // gemv_t16.h
            zc0[ct] = (f16x2){za, za};
            zc1[ct] = zc0[ct] + c960;
            sc2[ct] = (f16x2){sc, sc};
        }
        static_for<0, MT>([&](auto gc) {
            constexpr int g = decltype(gc)::value;
            if constexpr (g < HALF) {
#pragma unroll
                for (int i = g * PPG; i < (g + 1) * PPG && i < PPW; ++i) issue_piece(rbn, pn, i);
            }
            if constexpr (g + 1 < MT) {
#pragma unroll
                for (int j = 0; j < 4; ++j) av[(g + 1) & 1][j] = *(const f16x8*) (slab + ((g + 1) * 4 + j) * 1024);
            }
#pragma unroll
            for (int w = g * WPG; w < (g + 1) * WPG && w < CT * 4; ++w)
                bq[pn][w >> 2][w & 3] = gr_dequant(wr[pn][w >> 2][w & 3], magic, zc0[w >> 2], zc1[w >> 2], sc2[w >> 2]);
            if constexpr (g == MT - 1) {                              // raw[pn] has been read to its last word: the youngest requests of the step
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) issue_wt(rbw, std::integral_constant<int, pn>{}, ct);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int ct = 0; ct < CT; ++ct)
                    acc[g][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av[g & 1][j], bq[p][ct][j], acc[g][ct], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (g == 0) { if (s < 10) GG_STAMP(3 + 6 * s); }      // first group issued
            if constexpr (g == HALF - 1) { if (s < 10) GG_STAMP(4 + 6 * s); } // slab requests out
        });
        if (s < 10) GG_STAMP(5 + 6 * s);                              // all groups issued
        gg_wait<NW_LOADS>(wr[p][0], wr[p][1], zr[p][0], zr[p][1], sr[p][0], sr[p][1]);   // (releases the weights of step s + 2)
        if (s < 10) GG_STAMP(6 + 6 * s);                              // requests landed
        rg_barrier();
        if (s < 10) GG_STAMP(7 + 6 * s);                              // barrier passed
    };
    {   // prologue: slab 0, the weights of steps 0 and 1; expand step 0; request step 2
        int r0 = rb_of(0), r1 = rb_of(1), r2 = rb_of(2);
        const bool live0 = r0 < RB;
        r0 = r0 < RB ? r0 : RB - 1;
        r1 = r1 < RB ? r1 : RB - 1;
        r2 = r2 < RB ? r2 : RB - 1;
#pragma unroll
        for (int i = 0; i < PPW; ++i) issue_piece(r0, 0, i);
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            issue_wt(r0, std::integral_constant<int, 0>{}, ct);
            issue_wt(r1, std::integral_constant<int, 1>{}, ct);
        }
        drain();
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            const f16 sc = live0 ? __builtin_bit_cast(f16, (uint16_t) sr[0][ct]) : (f16) 0.f;
            const f16 za = __builtin_bit_cast(f16, (uint16_t) (0xE401u + ((zr[0][ct] >> (uint32_t) shz[ct]) & 0xFu)));
            const f16x2 zc0 = {za, za}, zc1 = zc0 + c960, s2 = {sc, sc};
#pragma unroll
            for (int j = 0; j < 4; ++j) bq[0][ct][j] = gr_dequant(wr[0][ct][j], magic, zc0, zc1, s2);
        }
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) issue_wt(r2, std::integral_constant<int, 0>{}, ct);
        rg_barrier();
        GG_STAMP(1);                                                  // prologue done
    }
    for (int s = 0; s < nsteps; s += 2) {                             // (an odd walk ends with one surplus step: scale 0)
        step(std::integral_constant<int, 0>{}, s);
        step(std::integral_constant<int, 1>{}, s + 1);
    }
    drain();                                                          // the last weight requests (registers only: the last slab landed before the last barrier)
    } else {
    auto issue = [&](int s, auto set_tag) {                           // LOADS instructions: the slab pieces of this wave, then its weights
        constexpr int set = decltype(set_tag)::value;
        int sp = s + off;                                             // rotated step (the surplus step behind the last one stays behind)
        sp = s >= nsteps ? nsteps : sp >= nsteps ? sp - nsteps : sp;
        int rb = rb_lo + sp * NKW + kw;
        rb = rb < RB ? rb : RB - 1;                                   // (an empty last step copies the last row-block again: same counts, product skipped)
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            const int p = cw * PPW + i;                               // piece (row tile p / 4, MFMA p % 4) of the slab
            rg_dma16(ring + (uint32_t) (set * SLAB + p * 1024), xb + ((size_t) (uint32_t) (p >> 2) * (uint32_t) K32 + (uint32_t) (rb * 4 + (p & 3))) * 1024u);
        }
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            gr_ld16(wr[set][ct], lane16, wq[ct] + (size_t) (uint32_t) rb * 1024u);
            const uint32_t g0 = (uint32_t) (rb * 16) >> (uint32_t) gsh[ct];
            rg_ld4s(zr[set][ct], zoff[ct], zq[ct] + (size_t) g0 * (uint32_t) n8[ct] * 4u);
            rg_ld2s(sr[set][ct], soff[ct], sq[ct] + (size_t) g0 * (uint32_t) nn[ct] * 2u);
        }
    };
    const f16x2 c960 = {(f16) 960.f, (f16) 960.f};
    const uint32_t magic = t16_magic();
    // (one step body, no peeled last step: a value that is still in flight must not cross a merge of two differently allocated paths --
    // hipcc copies registers there, ahead of the wait, scripts/isa_lint.py -- so the step behind the last one requests the last
    // row-block again and the loop runs an even number of steps; `s * NKW + kw < RB` skips the products of the surplus)
    auto step = [&](auto set_tag, int s) {
        constexpr int set = decltype(set_tag)::value;
        issue(s + 1, std::integral_constant<int, set ^ 1>{});
        gg_wait<LOADS>(wr[set][0], wr[set][1], zr[set][0], zr[set][1], sr[set][0], sr[set][1]);
        rg_barrier();                                                 // every wave's pieces of slab s have landed
        int sp = s + off;
        sp = s >= nsteps ? nsteps : sp >= nsteps ? sp - nsteps : sp;
        if (rb_lo + sp * NKW + kw < RB) {
            f16x8 bq[CT][4];
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                const f16 sc = __builtin_bit_cast(f16, (uint16_t) sr[set][ct]);
                const f16 za = __builtin_bit_cast(f16, (uint16_t) (0xE401u + ((zr[set][ct] >> (uint32_t) shz[ct]) & 0xFu)));   // -(1024 + z), gemv_t16.h
                const f16x2 zc0 = {za, za};
                const f16x2 zc1 = zc0 + c960;
                const f16x2 s2 = {sc, sc};
#pragma unroll
                for (int j = 0; j < 4; ++j) bq[ct][j] = gr_dequant(wr[set][ct][j], magic, zc0, zc1, s2);
            }
            const unsigned char* slab = ringp + set * SLAB + lane16;
            if constexpr (PF == 0) {                                  // LDS reads placed by the compiler (two in flight)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    f16x8 av[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) av[j] = *(const f16x8*) (slab + (mt * 4 + j) * 1024);
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int ct = 0; ct < CT; ++ct)
                            acc[mt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av[j], bq[ct][j], acc[mt][ct], 0, 0, 0);
                }
            } else {                                                  // the fragments of row tile mt + 1 are requested before the MFMAs of row tile mt
                f16x8 av[2][4];
#pragma unroll
                for (int j = 0; j < 4; ++j) av[0][j] = *(const f16x8*) (slab + j * 1024);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    if (mt + 1 < MT) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) av[(mt + 1) & 1][j] = *(const f16x8*) (slab + ((mt + 1) * 4 + j) * 1024);
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int ct = 0; ct < CT; ++ct)
                            acc[mt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av[mt & 1][j], bq[ct][j], acc[mt][ct], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        rg_barrier();                                                 // slab s has been read by all: its slot is the target of step s + 2
    };
    issue(0, std::integral_constant<int, 0>{});
    for (int s = 0; s < nsteps; s += 2) {
        step(std::integral_constant<int, 0>{}, s);
        step(std::integral_constant<int, 1>{}, s + 1);
    }
    gg_wait<0>(wr[0][0], wr[0][1], zr[0][0], zr[0][1], sr[0][0], sr[0][1]);      // the surplus request (registers AND LDS: the ring is reused below)
    rg_barrier();

    }
    // ---- K-group reduction through LDS (the ring's memory: every slab has been read, nothing is in flight), fixed order -----------------
    float* red = (float*) smem;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int j = 0; j < 4; ++j) red[wave * RED + (mt * CT + ct) * 256 + (kg * 4 + j) * 16 + col] = acc[mt][ct][j];
    __syncthreads();
    constexpr int OCT = EPI == 1 ? 1 : CT;                            // output tiles per wave
    constexpr int ITEMS = ROWS * 2;                                   // (row, half) of each of THIS wave's column tiles, dealt over its NKW waves
    constexpr int NIT = (ITEMS + NKW * 64 - 1) / (NKW * 64);
    const int r0 = rg * ROWS;
    auto gather = [&](int ri, float* v) {                             // the K-groups' sums of one item, fixed order
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = 0.f;
#pragma unroll
        for (int q = 0; q < NKW; ++q) {
            const float* pr = red + (q * NCW + cw) * RED + ri;
            const float4 p0 = *(const float4*) pr, p1 = *(const float4*) (pr + 4);
            v[0] += p0.x; v[1] += p0.y; v[2] += p0.z; v[3] += p0.w; v[4] += p1.x; v[5] += p1.y; v[6] += p1.z; v[7] += p1.w;
        }
    };
    if constexpr (EPI == 1) {
#pragma unroll
        for (int n = 0; n < NIT; ++n) {
            const int it = kw * 64 + lane + n * NKW * 64;
            if (it >= ITEMS) continue;                                // (wave-uniform: ITEMS is a multiple of 64)
            const int hf = it & 1, r = it >> 1;
            const int ri = (r >> 4) * CT * 256 + (r & 15) * 16 + hf * 8;
            const int row = r0 + r;
            const int n0 = n0_[0], ldn = nn[0];
            if (n0 < 0) continue;
            float v[8], u[8];
            gather(ri, v);
            gather(ri + 256, u);                                      // the up tile sits behind the gate tile
            f16x8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = silu_mul_f16((f16) v[j], (f16) u[j]);
            const int nc = n0 + hf * 8;
            const int c = nc >> 3;                                    // chunk of the consumer's K (to_frag_kernel: the same placement)
            const size_t piece = ((size_t) (row >> 4) * (size_t) (ldn >> 5) + (size_t) ((c >> 4) * 4 + (c & 3))) * 64 + (size_t) (((c >> 2) & 3) * 16 + (row & 15));
            *(uint4*) (a.out_frag + piece * 16) = __builtin_bit_cast(uint4, o);
        }
    } else {
        float vv[OCT][NIT][8];
        static_for<0, OCT>([&](auto ctc) {
            constexpr int ct = decltype(ctc)::value;
#pragma unroll
            for (int n = 0; n < NIT; ++n) {
                const int it = kw * 64 + lane + n * NKW * 64;
                if (it >= ITEMS) continue;
                const int hf = it & 1, r = it >> 1;
                gather(((r >> 4) * CT + ct) * 256 + (r & 15) * 16 + hf * 8, vv[ct][n]);
            }
        });
        if (ks > 1) {
            // K cut over ks blocks: every block leaves its slice of the tile in a.kws; the one that arrives LAST (a counter per tile) adds
            // the slices up in the order of their K ranges -- whoever it is: bit-reproducible -- and finishes the tile.  Visibility across
            // CUs / XCDs: 16-byte sc1 (write-through) stores, drained, then the arrival (agent-scope atomic); the last arrival reads with sc1
            // loads (MI355X_MICROARCH.md, correctness boundaries: "16 B sc1 stores AND sc1 loads").
            volatile int* s_last = (volatile int*) smem;             // (the reduction buffer is free behind the barrier below; static LDS next to
                                                                      // 128 KiB of dynamic would need its own opt-in arithmetic)
            constexpr int PER = NCW * OCT * ITEMS;                    // items of a tile
            const size_t tile = (size_t) rg * (size_t) a.ncg + (size_t) cg;
            float* mine = a.kws + ((tile * ks + kz) * PER) * 8;
            static_for<0, OCT>([&](auto ctc) {
                constexpr int ct = decltype(ctc)::value;
#pragma unroll
                for (int n = 0; n < NIT; ++n) {
                    const int it = kw * 64 + lane + n * NKW * 64;
                    if (it >= ITEMS) continue;
                    float* d = mine + ((size_t) ((cw * OCT + ct) * ITEMS + it)) * 8;
                    const f32x4 lo = {vv[ct][n][0], vv[ct][n][1], vv[ct][n][2], vv[ct][n][3]}, hi = {vv[ct][n][4], vv[ct][n][5], vv[ct][n][6], vv[ct][n][7]};
                    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\tglobal_store_dwordx4 %0, %2, off offset:16 sc1" :: "v"(d), "v"(lo), "v"(hi) : "memory");
                }
            });
            // (write-through stores, drained, instead of a release fence: `__threadfence()` writes the whole L2 back -- measured with it,
            // r06l: 85 us per launch at 8 ranges where K whole takes 18)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) {
                const int old = __hip_atomic_fetch_add(a.kcnt + tile, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                *s_last = old == ks - 1;
                if (old == ks - 1) a.kcnt[tile] = 0;                  // (every block of the tile has arrived: ready for the next launch)
            }
            __syncthreads();
            if (!*s_last) return;
            const float* all = a.kws + (tile * ks * PER) * 8;
            static_for<0, OCT>([&](auto ctc) {
                constexpr int ct = decltype(ctc)::value;
#pragma unroll
                for (int n = 0; n < NIT; ++n) {
                    const int it = kw * 64 + lane + n * NKW * 64;
                    if (it >= ITEMS) continue;
#pragma unroll
                    for (int j = 0; j < 8; ++j) vv[ct][n][j] = 0.f;
                    for (int z0 = 0; z0 < ks; z0 += 4) {              // four slices per round trip (sc1: past this CU's and this XCD's caches)
                        const float* d[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u)
                            d[u] = all + ((size_t) (z0 + u < ks ? z0 + u : ks - 1) * PER + (size_t) ((cw * OCT + ct) * ITEMS + it)) * 8;
                        f32x4 q[8];
                        asm volatile("global_load_dwordx4 %0, %8, off sc1\n\tglobal_load_dwordx4 %1, %8, off offset:16 sc1\n\t"
                                     "global_load_dwordx4 %2, %9, off sc1\n\tglobal_load_dwordx4 %3, %9, off offset:16 sc1\n\t"
                                     "global_load_dwordx4 %4, %10, off sc1\n\tglobal_load_dwordx4 %5, %10, off offset:16 sc1\n\t"
                                     "global_load_dwordx4 %6, %11, off sc1\n\tglobal_load_dwordx4 %7, %11, off offset:16 sc1\n\ts_waitcnt vmcnt(0)"
                                     : "=&v"(q[0]), "=&v"(q[1]), "=&v"(q[2]), "=&v"(q[3]), "=&v"(q[4]), "=&v"(q[5]), "=&v"(q[6]), "=&v"(q[7])
                                     : "v"(d[0]), "v"(d[1]), "v"(d[2]), "v"(d[3]) : "memory");
#pragma unroll
                        for (int u = 0; u < 4; ++u)
                            if (z0 + u < ks) {                        // (in the order of the K ranges)
#pragma unroll
                                for (int j = 0; j < 4; ++j) { vv[ct][n][j] += q[2 * u][j]; vv[ct][n][4 + j] += q[2 * u + 1][j]; }
                            }
                    }
                }
            });
        }
        float ssr[NIT];                                               // squares of what this thread stores, per item
#pragma unroll
        for (int n = 0; n < NIT; ++n) ssr[n] = 0.f;
        static_for<0, OCT>([&](auto ctc) {
            constexpr int ct = decltype(ctc)::value;
#pragma unroll
            for (int n = 0; n < NIT; ++n) {
                const int it = kw * 64 + lane + n * NKW * 64;
                if (it >= ITEMS) continue;
                const int hf = it & 1, row = r0 + (it >> 1);
                const int n0 = n0_[ct], mi = mi_[ct], ldn = nn[ct];
                if (n0 < 0 || row >= a.rows) continue;
                f16* ob = mi == 0 ? a.out[0] : mi == 1 ? a.out[1] : a.out[2];
                f16* o = ob + (size_t) row * ldn + n0 + hf * 8;
                f16x8 ov;
                if (a.no_zero) {
                    const f16x8 old = *(const f16x8*) o;
#pragma unroll
                    for (int j = 0; j < 8; ++j) ov[j] = (f16) (vv[ct][n][j] + (float) old[j]);
                } else {
#pragma unroll
                    for (int j = 0; j < 8; ++j) ov[j] = (f16) vv[ct][n][j];
                }
                *(f16x8*) o = ov;
#pragma unroll
                for (int j = 0; j < 8; ++j) { const float f = (float) ov[j]; ssr[n] = fmaf(f, f, ssr[n]); }
            }
        });
        if (a.rowsq) {                                                // (uniform) slot = this column-wave: the two halves of a row are neighbouring lanes
#pragma unroll
            for (int n = 0; n < NIT; ++n) {
                const int it = kw * 64 + lane + n * NKW * 64;
                if (it >= ITEMS) continue;
                const float ss = ssr[n] + __shfl_xor(ssr[n], 1);
                const int row = r0 + (it >> 1);
                if ((it & 1) == 0 && row < a.rows) a.rowsq[(size_t) row * a.rowsq_stride + cg * NCW + cw] = ss;
            }
        }
    }
}

// ---- producers of fragment-order activations --------------------------------------------------------------------------------------
// xf = fragment order of [ RMSNorm(x) * w  (norm_w != NULL)  |  x ], optionally gathered through an act-order map (column k of the
// result is column x_map[k] of the input, reference column_remap.cu:7-36).  A wave = one row-block of 128 k of one row tile at a
// time: lane (kg, r) reads the 64 contiguous bytes x[16 mt + r][128 rb + 32 kg .. + 32] and stores them as its 16 bytes of the four
// pieces (mt, 4 rb + j) -- every wave-store is one contiguous 1 KiB, no LDS turn, no barrier in the loop.
// Where the squares come from decides the grid.  rowsq != NULL: the GEMM that wrote x left, per row, `nslots` partial sums of squares
// (its epilogue held the final fp16 values: GrArgs::rowsq) -- a block adds them up for its 16 rows and any number of blocks can share
// a row tile: grid (row tiles, K slices), 4 waves, a few KiB each, spread over the chip.  rowsq == NULL with a norm (the first layer of
// a pass, or x came from somewhere else): ONE block of 8 waves per row tile reads whole rows -- all reads requested before the first is
// used (up to TF_HOLD row-blocks per wave stay in registers: K <= 8192; beyond that the second pass reads again) -- 6-14 us for a
// 1 MiB x, because a CU takes in lines that miss its L2 at ~28 GB/s (r06h: 65 % misses, x was written on other XCDs).
#define TF_WAVES 8
#define TF_HOLD 8
__global__ __launch_bounds__(TF_WAVES * 64) void to_frag_kernel(const f16* __restrict__ x, const f16* __restrict__ norm_w, float eps,
                                                                const uint32_t* __restrict__ x_map, unsigned char* __restrict__ xf, int rows, int K,
                                                                const float* __restrict__ rowsq, int nslots)
{
    __shared__ float part[TF_WAVES * 4][16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nw = blockDim.x >> 6;                                   // waves of this block (8: whole rows; 4: a K slice)
    const int r = lane & 15, kg = lane >> 4;
    const int mt = blockIdx.x, row = mt * 16 + r;
    const bool live = row < rows;
    const f16* xr = x + (size_t) (live ? row : 0) * K;
    const int nrb_all = K >> 7, K32 = K >> 5;
    const int per = (nrb_all + gridDim.y - 1) / gridDim.y;            // row-blocks of this block's K slice
    const int rb_lo = blockIdx.y * per, nrb = min(nrb_all, rb_lo + per) - rb_lo;
    const bool hold = nrb <= nw * TF_HOLD;
    f16x8 v[TF_HOLD][4];
    auto fetch = [&](int rb, f16x8* d) {                              // the lane's 32 values of row-block rb (gathered through the map)
        const int k0 = rb * 128 + kg * 32;
        if (!live) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int e = 0; e < 8; ++e) d[j][e] = (f16) 0.f;
        } else if (x_map) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint4 m0 = *(const uint4*) (x_map + k0 + j * 8), m1 = *(const uint4*) (x_map + k0 + j * 8 + 4);
                const uint32_t mm[8] = {m0.x, m0.y, m0.z, m0.w, m1.x, m1.y, m1.z, m1.w};
#pragma unroll
                for (int e = 0; e < 8; ++e) d[j][e] = xr[mm[e]];
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) d[j] = *(const f16x8*) (xr + k0 + j * 8);
        }
    };
    float rm = 1.f;
    if (hold) {
#pragma unroll
        for (int i = 0; i < TF_HOLD; ++i) { const int q = wave + i * nw; if (q < nrb) fetch(rb_lo + q, v[i]); }
    }
    if (norm_w) {
        float ss = 0.f;
        if (rowsq) {                                                  // the producer's partial sums: slot s of the row to thread s mod (4 nw)
            if (live) {                                               // (eight requests at a time, then added in slot order: a loop of dependent
                const float* pr = rowsq + (size_t) row * nslots;      // load-and-add took 7-10 us for 256 slots, r06m)
                const int st = nw * 4;
                for (int sl = wave * 4 + kg; sl < nslots; sl += 8 * st) {
                    float t[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) t[u] = sl + u * st < nslots ? pr[sl + u * st] : 0.f;
#pragma unroll
                    for (int u = 0; u < 8; ++u) ss += t[u];
                }
            }
        } else if (hold) {
#pragma unroll
            for (int i = 0; i < TF_HOLD; ++i)
                if (wave + i * nw < nrb)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int e = 0; e < 8; ++e) { const float f = (float) v[i][j][e]; ss = fmaf(f, f, ss); }
        } else {
            for (int q = wave; q < nrb; q += nw) {
                f16x8 t[4];
                fetch(rb_lo + q, t);
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int e = 0; e < 8; ++e) { const float f = (float) t[j][e]; ss = fmaf(f, f, ss); }
            }
        }
        part[wave * 4 + kg][r] = ss;
        __syncthreads();
        float sum = 0.f;
        for (int q = 0; q < nw * 4; ++q) sum += part[q][r];           // (every thread of row r adds the same numbers in the same order)
        rm = 1.0f / sqrtf(sum * (1.0f / (float) K) + eps);
    }
    const f16 rmh = (f16) rm;
    auto put = [&](int rb, f16x8* d) {
        if (norm_w && live) {
            const int k0 = rb * 128 + kg * 32;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (x_map) {
                    const uint4 m0 = *(const uint4*) (x_map + k0 + j * 8), m1 = *(const uint4*) (x_map + k0 + j * 8 + 4);
                    const uint32_t mm[8] = {m0.x, m0.y, m0.z, m0.w, m1.x, m1.y, m1.z, m1.w};
#pragma unroll
                    for (int e = 0; e < 8; ++e) { const f16 q = d[j][e] * rmh; d[j][e] = q * norm_w[mm[e]]; }
                } else {
                    const f16x8 w = *(const f16x8*) (norm_w + k0 + j * 8);
#pragma unroll
                    for (int e = 0; e < 8; ++e) { const f16 q = d[j][e] * rmh; d[j][e] = q * w[e]; }   // the rounding points of rms_norm.cu:21-213 (elementwise.hip)
                }
            }
        }
        // piece (mt, 4 rb + j), lane (kg, r) <- the lane's chunk j (gemv_t16.h: a lane's dword j holds packed row 4 kg + j of the
        // row-block, i.e. k 32 kg + 8 j .. + 8)
#pragma unroll
        for (int j = 0; j < 4; ++j)
            *(uint4*) (xf + (((size_t) mt * K32 + (size_t) (rb * 4 + j)) * 64 + lane) * 16) = __builtin_bit_cast(uint4, d[j]);
    };
    if (hold) {
#pragma unroll
        for (int i = 0; i < TF_HOLD; ++i) { const int q = wave + i * nw; if (q < nrb) put(rb_lo + q, v[i]); }
    } else {
        for (int q = wave; q < nrb; q += nw) {
            f16x8 t[4];
            fetch(rb_lo + q, t);
            put(rb_lo + q, t);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------------------------
// (buffers hold whole row groups -- 64 rows, from 65 rows on 128: a block reads MT = 4 / 8 row tiles whatever `rows` is; the producers
// write the padding as zeros; nothing a padding row holds reaches a live row)
static int frag_rows(int rows) { return rows <= 64 ? 64 : (rows + 127) / 128 * 128; }
size_t frag_bytes(int rows, int K) { return (size_t) frag_rows(rows) * (size_t) K * 2; }

int launch_to_frag(const f16* x, const f16* norm_w, float eps, const uint32_t* x_map, void* xf, int rows, int K, hipStream_t s, const float* rowsq,
                   int nslots)
{
    EXL_REQUIRE(K % 128 == 0 && rows > 0, EXL_E_UNSUPPORTED, "fragment order: K (%d) must be a multiple of 128", K);
    const int mt = frag_rows(rows) / 16;                             // (whole row groups: the padding tiles are written as zeros)
    const bool whole_rows = norm_w && !(rowsq && nslots > 0);        // the squares have to be summed here: a block reads whole rows
    int ks = 1, waves = TF_WAVES;
    if (!whole_rows) {                                               // K slices of four row-blocks (one per wave): the copy spread over the chip
        waves = 4;
        ks = (K / 128 + 3) / 4;
        while (ks > 1 && (long) mt * ks > 1024) ks = (ks + 1) / 2;
    }
    hipLaunchKernelGGL(to_frag_kernel, dim3(mt, ks), dim3(waves * 64), 0, s, x, norm_w, eps, x_map, (unsigned char*) xf, rows, K,
                       whole_rows ? nullptr : rowsq, whole_rows ? 0 : nslots);
    EXL_LAUNCH_CHECK();
    return 0;
}

size_t gemm_frag_ksplit_floats(int rows, int N) { return (size_t) 8 * (size_t) frag_rows(rows) * (size_t) ((N + 127) / 128 * 128); }

static int gr_gshift(const Q4Matrix* w)
{
    if (w->groups <= 1) return 31;
    const int gp = w->groupsize / 8;                                 // packed rows per group
    int sh = 0;
    while ((1 << sh) < gp) ++sh;
    return (1 << sh) == gp ? sh : -1;
}

static thread_local bool g_gr_dry = false;                          // gemm_t16r_covers: run the launcher's tests, launch nothing (per thread:
                                                                     // another host thread's real launch must not see it)

template <int MT, int CT, int EPI>
static int gr_go(GrArgs& a, int rows, int col_groups, hipStream_t s)
{
    if (g_gr_dry) return 0;
    a.nrg = (rows + MT * 16 - 1) / (MT * 16);
    a.ncg = col_groups;
    a.rb_per_wave = (a.m[0].RB + GR_WAVES - 1) / GR_WAVES;
    const size_t smem = (size_t) GR_WAVES * MT * CT * 1024;
    auto kfn = q4_gemm_t16r_kernel<MT, CT, EPI>;
    static bool big[EXL_MAX_DEVICES] = {};
    if (smem > 64 * 1024) EXL_TRY(exl_lds_opt_in((const void*) kfn, big));
    hipLaunchKernelGGL(kfn, dim3((unsigned) (a.nrg * a.ncg)), dim3(GR_WAVES * 64), smem, s, a);
    EXL_LAUNCH_CHECK();
    return 0;
}

static int t16g_pipelined()                                          // A/B: the pipelined step (default) / the plain step (EXL_GEMM_T16G_PF=0)
{
    static const int pf = getenv("EXL_GEMM_T16G_PF") ? atoi(getenv("EXL_GEMM_T16G_PF")) : 1;
    return pf;
}

template <int MT, int NCW, int EPI>
static int gg_go(GrArgs& a, int rows, int units, hipStream_t s)
{
    if (g_gr_dry) return 0;
    constexpr int NKW = GR_WAVES / NCW;
    a.nrg = (rows + MT * 16 - 1) / (MT * 16);
    a.ncg = (units + NCW - 1) / NCW;
    const int ks = a.ks > 1 ? a.ks : 1;
    const size_t smem = (size_t) NKW * MT * 8192;                     // the ring (two slabs per K-group) >= the reduction (MT * 16 KiB)
    const int pf = ks > 1 ? 1 : t16g_pipelined();                     // (K cut over blocks: the pipelined kernel only)
    auto kfn = pf ? q4_gemm_t16g_kernel<MT, NCW, EPI, 1> : q4_gemm_t16g_kernel<MT, NCW, EPI, 0>;
    static bool big[2][EXL_MAX_DEVICES] = {};
    if (smem > 64 * 1024) EXL_TRY(exl_lds_opt_in((const void*) kfn, big[pf ? 1 : 0]));
    hipLaunchKernelGGL(kfn, dim3((unsigned) (a.nrg * a.ncg * ks)), dim3(GR_WAVES * 64), smem, s, a);
    EXL_LAUNCH_CHECK();
    return 0;
}

// arrival counters of the K-cut launches: one int per output tile, zero between launches (the last arrival resets its own)
#define GG_KCNT 16384
static int ksplit_counters(int device, int** out)
{
    static std::mutex lock;
    static int* cnt[EXL_MAX_DEVICES] = {};
    std::lock_guard<std::mutex> hold(lock);
    EXL_REQUIRE(device >= 0 && device < EXL_MAX_DEVICES, EXL_E_INVALID, "q4 gemm: device %d out of range", device);
    if (!cnt[device]) {
        int* p = nullptr;
        EXL_HIP(hipMalloc((void**) &p, GG_KCNT * sizeof(int)));
        EXL_HIP(hipMemset(p, 0, GG_KCNT * sizeof(int)));
        EXL_HIP(hipDeviceSynchronize());
        cnt[device] = p;
    }
    *out = cnt[device];
    return 0;
}

// The matrices of one launch: T16 layout, same K, power-of-two group size >= 32, no map left to apply (the producer of xf gathered).
// Returns 1 for what the kernel does not cover.  outs: row-major outputs (EPI 0) / out_frag: fragment-order silu(gate) * up (dual).
int launch_gemm_t16r(int nmat, const Q4Matrix* const* w, const void* xf, int rows, f16* const* outs, int no_zero, int dual, void* out_frag,
                     hipStream_t s, int force, float* rowsq, int* rowsq_slots, float* kws, size_t kws_floats)
{
    if (rowsq_slots) *rowsq_slots = 0;
    if (nmat != 1 || dual) rowsq = nullptr;                           // (one slot = the columns of one block / column-wave of ONE matrix)
    if (nmat < 1 || nmat > 3 || rows < 1 || (dual && nmat != 2)) return 1;
    GrArgs a = {};
    const int K = w[0]->height;
    if (K % 128 != 0) return 1;
    if ((uint64_t) frag_bytes(rows, K) >= (1ull << 32)) return 1;
    int tiles = 0;
    for (int i = 0; i < 3; ++i) {
        const Q4Matrix* m = w[i < nmat ? i : 0];
        if (i < nmat) {
            if (m->layout != EXL_LAYOUT_T16 || m->height != K || m->width % 16 != 0) return 1;
            if ((uint64_t) K * (uint64_t) m->width >= (1ull << 32)) return 1;
            if (gr_gshift(m) < 0 || (m->groups > 1 && m->groupsize % 32 != 0)) return 1;
            if (dual && i == 1 && (m->width != w[0]->width || m->groupsize != w[0]->groupsize)) return 1;
            if (!dual || i == 0) tiles += m->width / 16;
        }
        a.m[i].qw = (const unsigned char*) m->qweight; a.m[i].qz = m->qzeros; a.m[i].sc = (const uint16_t*) m->scales;
        a.m[i].N = m->width; a.m[i].RB = K / 128; a.m[i].gsh = gr_gshift(m);
        a.tile_end[i] = tiles;
        a.out[i] = (outs && i < nmat) ? outs[i] : nullptr;
    }
    a.xf = (const unsigned char*) xf; a.rows = rows; a.K = K; a.no_zero = no_zero; a.out_frag = (unsigned char*) out_frag;
    // (measured, r06h: no difference -- 156.9 vs 156.8 us per 7B layer at 128 rows, the L2 already serves the shared lines, PMC FETCH_SIZE
    // = 1.2 x the weights -- so off; EXL_GEMM_STAGGER=1 is the A/B switch)
    static const bool stagger = getenv("EXL_GEMM_STAGGER") != nullptr;
    a.stagger = stagger ? 1 : 0;
    a.rowsq = rowsq;
    auto slots = [&](int n) { a.rowsq_stride = n; if (rowsq_slots && rowsq) *rowsq_slots = n; };   // (set before the launch that reads a)
    // Wide launches: activations shared through LDS (q4_gemm_t16g), as many rows and column-waves per block as still leave
    // about two blocks for every three CUs (a block is 8 waves: it fills its CU's matrix pipes alone).
    static const bool no_g = getenv("EXL_GEMM_NO_T16G") != nullptr;
    static const int g_min = getenv("EXL_GEMM_T16G_MIN_BLOCKS") ? atoi(getenv("EXL_GEMM_T16G_MIN_BLOCKS")) : 160;
    // (force, the op-level entry point's kernel choice for tests and A/B runs: 0 as above, 1 the narrow kernel, 2 the wide kernel at any
    // width, 3 / 4 / 5 its <4, 4> / <4, 2> / <8, 4> block shape)
    // Row tiles per block: what the rows fill -- 1 / 2 / 4 / 8 tiles of 16 (a prompt of a few tokens, a chat turn, does a sixteenth of
    // the LDS reads and MFMAs of a 128-row block and runs at the rate the weights stream).
    // (force, the op-level entry point's kernel choice for tests and A/B runs: 0 as above, 1 the narrow kernel, 2 the wide kernel at any
    // width, 3 / 4 / 5 its <4, 4> / <4, 2> / <8, 4> block shape, 7 .. 10 <1, 4> / <1, 2> / <2, 4> / <2, 2>)
    const int mt_fit = rows <= 16 ? 1 : rows <= 32 ? 2 : rows <= 64 ? 4 : 8;
    // (EXL_GEMM_SMALL_NARROW=1, an A/B switch: prompts of up to 32 rows on the narrow kernel for EVERY launch -- the decode GEMV's shape)
    static const bool small_narrow = getenv("EXL_GEMM_SMALL_NARROW") != nullptr;
    if ((!no_g && force == 0 && !(small_narrow && mt_fit <= 2)) || (force >= 2 && force <= 5) || (force >= 7 && force <= 10)) {
        const int units = dual ? w[0]->width / 16 : (tiles + 1) / 2;  // what one wave owns: a gate + an up tile / two tiles
        const long need = force ? 0 : g_min;
        auto blocks = [&](int mt, int ncw) { return (long) ((rows + mt * 16 - 1) / (mt * 16)) * ((units + ncw - 1) / ncw); };
        int mt = 0, ncw = 0;
        if (force <= 2) {
            if (blocks(mt_fit, 4) >= need) { mt = mt_fit; ncw = 4; }
            else if (mt_fit == 8 && blocks(4, 4) >= need) { mt = 4; ncw = 4; }
            else if (blocks(mt_fit == 8 ? 4 : mt_fit, 2) >= need) { mt = mt_fit == 8 ? 4 : mt_fit; ncw = 2; }
        } else {
            static const int shape[11][2] = {{0, 0}, {0, 0}, {0, 0}, {4, 4}, {4, 2}, {8, 4}, {0, 0}, {1, 4}, {1, 2}, {2, 4}, {2, 2}};
            mt = shape[force][0]; ncw = shape[force][1];
            if (mt == 8 && rows <= 64) mt = 0;                        // (the buffers of <= 64 rows hold 64)
        }
        if (mt) {
            if (!dual) slots((units + ncw - 1) / ncw * ncw);
#define GG_CASE(M, N) case M * 10 + N: return dual ? gg_go<M, N, 1>(a, rows, units, s) : gg_go<M, N, 0>(a, rows, units, s)
            switch (mt * 10 + ncw) {
                GG_CASE(8, 4); GG_CASE(4, 4); GG_CASE(4, 2); GG_CASE(2, 4); GG_CASE(2, 2); GG_CASE(1, 4); GG_CASE(1, 2);
            }
#undef GG_CASE
        }
        if (force >= 2) return 1;                                     // the forced shape does not take this launch (6: the K-cut form, below)
    }
    // Narrow launches (o_proj, down_proj: N = hidden).  With K whole, a block per 64 x 32 outputs is what fills the chip, and every such
    // block pulls 64 rows x K of activations through its CU (r06j, 7B at 128 rows: 11.4 / 28.2 us).  The alternative -- K cut over `ks`
    // blocks per tile, the wide kernel's 128 x 128 tiles, the block that arrives last at a tile adding the fp32 slices up
    // (q4_gemm_t16g_kernel: ks) -- is built, bit-exact run to run and parity-tested (kernel choice 6 of exl_q4_matmul_frag), and LOSES:
    // 23.7 / 42.7 us at 4 / 8 ranges (profiles/r06_short_prompt.txt: a block of 4-5 steps is mostly start-up, and the hand-over costs
    // 64 KiB of write-through stores per block plus a last block that reads ks x 64 KiB back).  Off unless EXL_GEMM_KSPLIT=n (n ranges;
    // -1: the launcher's choice) or the caller forces it.
    static const int ks_env = getenv("EXL_GEMM_KSPLIT") ? atoi(getenv("EXL_GEMM_KSPLIT")) : 0;
    if (!dual && nmat == 1 && kws && (force == 6 || (force == 0 && ks_env != 0 && !no_g))) {
        const int units = (tiles + 1) / 2, RBn = K / 128;
        const bool m8 = rows > 64;
        const int rg_n = m8 ? (rows + 127) / 128 : (rows + 63) / 64, cg_n = (units + 3) / 4;
        int ks = ks_env > 0 ? ks_env : (256 + rg_n * cg_n / 2) / (rg_n * cg_n);
        ks = ks > 8 ? 8 : ks;
        if (ks_env <= 0) ks = ks > RBn / 8 ? RBn / 8 : ks;           // (four steps per K-group: below that a block is start-up and hand-over)
        ks = ks > RBn / 4 ? RBn / 4 : ks;                            // (never fewer than two)
        const size_t need = (size_t) rg_n * cg_n * ks * (m8 ? 128 : 64) * 128;
        if (ks >= 2 && need <= kws_floats && rg_n * cg_n <= GG_KCNT) {
            a.ks = ks;
            a.kws = kws;
            if (!g_gr_dry) EXL_TRY(ksplit_counters(w[0]->device, &a.kcnt));
            slots(cg_n * 4);
            return m8 ? gg_go<8, 4, 0>(a, rows, units, s) : gg_go<4, 4, 0>(a, rows, units, s);
        }
    }
    if (force == 6) return 1;                                         // (the K-cut form was asked for and does not take this launch)
    // K whole: column tiles per block as wide as still gives about a block per CU -- the activation traffic of the launch is
    // (tiles / CT) x rows x K x 2 bytes through the CUs' L1s, the weight expansion is repeated by every row group.
    const int nrg = (rows + 63) / 64;
    if (dual) {
        if (mt_fit <= 2) return mt_fit == 1 ? gr_go<1, 2, 1>(a, rows, w[0]->width / 16, s) : gr_go<2, 2, 1>(a, rows, w[0]->width / 16, s);
        if (w[0]->width % 32 != 0) return 1;                          // two gate + two up tiles per block
        return gr_go<4, 4, 1>(a, rows, w[0]->width / 32, s);
    }
    if (mt_fit <= 2) {                                               // a few rows: one or two row tiles per block, the decode GEMV's shape
        bool by2 = true;
        for (int i = 0; i < nmat; ++i) by2 = by2 && (w[i]->width % 32 == 0);
        if (by2 && tiles / 2 >= 200) { slots(tiles / 2); return mt_fit == 1 ? gr_go<1, 2, 0>(a, rows, tiles / 2, s) : gr_go<2, 2, 0>(a, rows, tiles / 2, s); }
        slots(tiles);
        return mt_fit == 1 ? gr_go<1, 1, 0>(a, rows, tiles, s) : gr_go<2, 1, 0>(a, rows, tiles, s);
    }
    bool by4 = true;                                                 // a block's tiles never straddle two matrices
    for (int i = 0; i < nmat; ++i) by4 = by4 && (w[i]->width % 64 == 0);
    if (by4 && (long) nrg * (tiles / 4) >= 224) { slots(tiles / 4); return gr_go<4, 4, 0>(a, rows, tiles / 4, s); }
    for (int i = 0; i < nmat; ++i) if (w[i]->width % 32 != 0) return 1;
    slots(tiles / 2);
    return gr_go<4, 2, 0>(a, rows, tiles / 2, s);
}

// Would launch_gemm_t16r take this launch?  (The layer entry point asks for all four of its GEMMs BEFORE it enqueues anything.)
bool gemm_t16r_covers(int nmat, const Q4Matrix* const* w, int rows, int dual)
{
    g_gr_dry = true;
    const int r = launch_gemm_t16r(nmat, w, nullptr, rows, nullptr, 0, dual, nullptr, nullptr, 0, nullptr, nullptr, nullptr, 0);
    g_gr_dry = false;
    return r == 0;
}
