// Prefill path: out[M,N] (+)= x[M,K] @ dequant(W[K,N]) for large M -- int4 dequant fused into an MFMA GEMM.
//
// Replaces /root/reference/exllama_ext/cuda_func/q4_matmul.cu:301-344 (q4_matmul_recons_cuda), which
// materialises the whole fp16 weight (q4_matrix.cu:170-224 reconstruct_kernel, K*N*2 bytes written and read
// back per call) and then calls cublasHgemm.  Here nothing is materialised.  Kernels in this file:
//   q4_gemm_kernel<T16>      generic register-B kernel (any layout / shape the T16 tiling cannot express; A/B reference):
//                            the packed word is the per-lane B fragment of v_mfma_f32_32x32x16_f16, dequantised in
//                            registers by every wave; the activation tile goes through LDS in the nibble order;
//   q4_gemm_t16m_kernel      T16 layout, weights dequantised ONCE per block into an LDS tile, mid-step barrier schedule
//                            (256 x 128 tile: the fallback of the next one; 128 x 128 tile, optionally with K cut in two:
//                            257 .. 512 rows, see launch_q4_gemm);
//   q4_gemm_t16w_kernel<EPI> the default above 256 rows: 8 MFMA waves + 4 loader waves on a 256 x 128 tile; EPI 1 is the
//                            q/k/v projection with RoPE and the KV-cache write as its epilogue (> 512 rows);
//   q4_gemm_t16d2_kernel     gate and up projections of the MLP in one kernel with the SiLU epilogue, software-pipelined
//                            (> 512 rows);
//   (q4_gemm_skinny.hip)     <= 256 rows: the decode-shaped short-prompt kernel;
//   half_gemm_kernel         plain fp16 GEMM of the LoRA path (correctness first).
// Act-order weights: x is gathered through x_map into the borrowed temp_state buffer first (the reference's column_remap,
// q4_matmul.cu:320-325) -- ONCE for the matrices that share a map (q / k / v and gate / up of a GPTQ checkpoint: quantised against
// the same input, hence the same permutation), and by the RMSNorm kernel that produces the activations where there is one
// (prompt_prologue).  A 2-byte gather cannot ride in the 16-byte LDS-DMA pieces of the A tile.
#include "gemv_t16.h"
#include <stdlib.h>
#include <string.h>

#define MAGIC_1024 0x64006400u
#define BM 128
#define BN 128
#define BK 64

__device__ __forceinline__ f16x2 as_h2(uint32_t v) { return __builtin_bit_cast(f16x2, v); }

__device__ __forceinline__ uint4 permute_x8(uint4 d)
{
    uint4 o;
    o.x = (d.x & 0xFFFFu) | (d.z << 16);
    o.y = (d.x >> 16) | (d.z & 0xFFFF0000u);
    o.z = (d.y & 0xFFFFu) | (d.w << 16);
    o.w = (d.y >> 16) | (d.w & 0xFFFF0000u);
    return o;
}

// LDS byte offset of the 16-byte slot (row, c8) of a [BM][BK] fp16 tile (128-byte rows, 8 slots per row).
__device__ __forceinline__ int a_lds_off(int row, int c8) { return row * 128 + ((c8 ^ ((row >> 1) & 7)) << 4); }

// word -> 8 fp16 weights in slot order (q0,q4,q1,q5,q2,q6,q3,q7), each h( h(q - z) * s )
__device__ __forceinline__ f16x8 dequant_word(uint32_t w, f16x2 zc0, f16x2 zc1, f16x2 s2)
{
    const f16x2 sixteenth = {(f16) 0.0625f, (f16) 0.0625f};
    const uint32_t w8 = w >> 8;
    const f16x2 d0 = (as_h2((w & 0x000F000Fu) | MAGIC_1024) + zc0) * s2;
    const f16x2 d1 = (as_h2((w & 0x00F000F0u) | MAGIC_1024) * sixteenth + zc1) * s2;
    const f16x2 d2 = (as_h2((w8 & 0x000F000Fu) | MAGIC_1024) + zc0) * s2;
    const f16x2 d3 = (as_h2((w8 & 0x00F000F0u) | MAGIC_1024) * sixteenth + zc1) * s2;
    f16x8 r;
    r[0] = d0[0]; r[1] = d0[1]; r[2] = d1[0]; r[3] = d1[1];
    r[4] = d2[0]; r[5] = d2[1]; r[6] = d3[0]; r[7] = d3[1];
    return r;
}

template <bool T16>      // weight layout: GPTQ [K/8][N] words, or T16 pieces (gemv_t16.h)
__global__ __launch_bounds__(256) void q4_gemm_kernel(const f16* __restrict__ x, const uint32_t* __restrict__ qweight,
                                                      const uint32_t* __restrict__ qzeros,
                                                      const f16* __restrict__ scales, f16* __restrict__ out, int M,
                                                      int K, int N, int gshift, int groupsize, int no_zero, int mtiles,
                                                      int ntiles)
{
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * BM * BK * 2];

    // ---- XCD-aware tile mapping: block b runs on XCD b % 8 (observed dispatch; speed only) ------------
    const int b = blockIdx.x;
    const int xcd = b & 7;
    const int idx = b >> 3;
    const int nl = idx / mtiles;
    const int mt = idx - nl * mtiles;
    const int nt = nl * 8 + xcd;
    if (nt >= ntiles) return;
    const int m0 = mt * BM;
    const int n0 = nt * BN;

    const int tid = threadIdx.x;
    const int wave = tid >> 6;
    const int lane = tid & 63;
    const int wm = wave >> 1;
    const int wn = wave & 1;
    const int g = lane >> 5;                                 // k half of the 16-wide MFMA step
    const int c = lane & 31;

    // B addressing: this lane owns columns n, n+1 (one uint2 per packed row)
    const int n = n0 + wn * 64 + 2 * c;
    const bool n_ok = n < N;
    const uint32_t* wptr = qweight + n;
    const int zsh = (n & 7) * 4;

    // A staging: 4 x 16 bytes per thread per tile
    int a_row[4], a_c8[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int id = tid + i * 256;
        a_row[i] = id >> 3;
        a_c8[i] = id & 7;
    }

    f32x16 acc[2][2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][j][r] = 0.f;

    const int nk = (K + BK - 1) / BK;

    auto load_a = [&](int k0, uint4 (&ar)[4]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = m0 + a_row[i];
            const int k = k0 + a_c8[i] * 8;
            ar[i] = (row < M && k < K) ? *(const uint4*) (x + (size_t) row * K + k) : make_uint4(0, 0, 0, 0);
        }
    };
    auto store_a = [&](int buf, const uint4 (&ar)[4]) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            *(uint4*) (lds + buf * (BM * BK * 2) + a_lds_off(a_row[i], a_c8[i])) = T16 ? ar[i] : permute_x8(ar[i]);   // T16 words are nibble-interleaved: natural k order
    };
    // B operand of MFMA step kk, lane half g = the 8 k of ONE packed row.  GPTQ layout: row 2*kk + g of the K tile (one
    // uint2 = columns n, n+1).  T16 layout: the lane's two pieces hold rows 4*g .. 4*g+3 of the K tile for columns n and
    // n+1 (32 contiguous bytes), step kk uses dword kk -- the A fragment is read with the matching k-order below.
    const uint4* t16p = (const uint4*) qweight + ((size_t) (n >> 4) * (K >> 7)) * 64 + (n & 15);
    auto load_b = [&](int k0, uint2 (&br)[4]) {
        if constexpr (T16) {
            const int rb = k0 >> 7, rsub = ((k0 >> 6) & 1) * 2 + g;
            uint4 p0 = make_uint4(0, 0, 0, 0), p1 = p0;
            if (n_ok) { const uint4* p = t16p + (size_t) rb * 64 + rsub * 16; p0 = p[0]; p1 = p[1]; }
            br[0] = make_uint2(p0.x, p1.x); br[1] = make_uint2(p0.y, p1.y);
            br[2] = make_uint2(p0.z, p1.z); br[3] = make_uint2(p0.w, p1.w);
        } else {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const int prow = (k0 >> 3) + 2 * kk + g;
                br[kk] = (n_ok && prow * 8 < K) ? *(const uint2*) (wptr + (size_t) prow * N) : make_uint2(0, 0);
            }
        }
    };

    uint4 areg[4];
    uint2 bcur[4], bnext[4];
    load_a(0, areg);
    load_b(0, bcur);
    store_a(0, areg);
    __syncthreads();

    int cur_group = -1;
    f16x2 zc0[2], zc1[2], s2[2];

    for (int it = 0; it < nk; ++it) {
        const int k0 = it * BK;
        const bool more = it + 1 < nk;
        if (more) {
            load_a(k0 + BK, areg);
            load_b(k0 + BK, bnext);
        }
        const unsigned char* abuf = lds + (it & 1) * (BM * BK * 2);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int k = T16 ? k0 + 32 * g + 8 * kk : k0 + kk * 16;    // first k this lane multiplies in this step
            if (k0 + kk * 16 < K) {                            // block-uniform
                const int grp = gshift >= 0 ? (k >> gshift) : (k / groupsize);
                if (grp != cur_group) {                        // uniform unless groupsize == 32 in the T16 layout
                    cur_group = grp;
                    uint32_t zw = 0;
                    f16x2 sv = {(f16) 0.f, (f16) 0.f};
                    if (n_ok) {
                        zw = qzeros[(size_t) grp * (N >> 3) + (n >> 3)] >> zsh;
                        sv = *(const f16x2*) (scales + (size_t) grp * N + n);
                    }
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const int z = (int) ((zw >> (4 * j)) & 0xFu) + 1;
                        const f16 a = (f16) (float) (-(1024 + z));
                        const f16 bb = (f16) (float) (-(64 + z));
                        zc0[j] = (f16x2){a, a};
                        zc1[j] = (f16x2){bb, bb};
                        s2[j] = (f16x2){sv[j], sv[j]};
                    }
                }
                f16x8 bf[2];
                bf[0] = dequant_word(bcur[kk].x, zc0[0], zc1[0], s2[0]);
                bf[1] = dequant_word(bcur[kk].y, zc0[1], zc1[1], s2[1]);
                f16x8 af[2];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const int row = wm * 64 + t * 32 + c;
                    af[t] = *(const f16x8*) (abuf + a_lds_off(row, T16 ? 4 * g + kk : kk * 2 + g));
                }
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[t][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[t], bf[j], acc[t][j], 0, 0, 0);
            }
        }
        if (more) {
            store_a((it + 1) & 1, areg);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) bcur[kk] = bnext[kk];
        }
        __syncthreads();
    }

    // ---- epilogue: C[row = (r&3) + 8*(r>>2) + 4*g][col = c] per 32x32 tile --------------------------------
    if (!n_ok) return;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = m0 + wm * 64 + t * 32 + (r & 3) + 8 * (r >> 2) + 4 * g;
            if (row < M) {
                f16* op = out + (size_t) row * N + n;
                float v0 = acc[t][0][r], v1 = acc[t][1][r];
                if (no_zero) {
                    const f16x2 prev = *(const f16x2*) op;
                    v0 += (float) prev[0];
                    v1 += (float) prev[1];
                }
                *(f16x2*) op = (f16x2){(f16) v0, (f16) v1};
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// T16 layout, LDS-staged B: the prefill kernels of the product path (what all of them share).
//
//   * block tile BM(M) x 128(N), K step 64; BM = 256 with 8 MFMA waves (4 x 2) for large M, 128 with 4 waves (2 x 2) otherwise;
//     each wave 64 x 64 = 4 x 4 tiles of v_mfma_f32_16x16x32_f16, fp32 accumulate (64 accumulator registers);
//   * activations: global -> LDS by LDS-DMA (global_load_lds_dwordx4, no VGPR round trip), 128-byte rows XOR-swizzled
//     through the per-lane SOURCE address (the LDS image of a DMA is lane-linear), so the fragment reads
//     (ds_read_b128, 16 rows x 4 k-chunks per wave) are bank-conflict free (SQ_LDS_BANK_CONFLICT = 0 measured);
//   * weights: each thread loads one T16 piece (or half of one at 512 threads) per K step -- 32 lanes read 512
//     contiguous bytes --, expands it ONCE per block to fp16 values h(h(q - z) * s), bit-identical to the reference's
//     reconstruct (q4_matrix.cu:207), already in natural k order (the T16 words are nibble-interleaved), and writes
//     them to the swizzled [n][k] LDS tile.  The dequantisation cost is paid once per BM rows of M instead of once per
//     wave as in the register-B kernel above; at BM = 256 it is 1/4 of the VALU work of the 128-row tile per MFMA;
//   * activations run two K steps ahead through a 3-slot LDS ring, the packed weights of step t+2 are in flight in
//     registers while step t feeds the MFMAs and step t+1 is dequantised; hipcc cannot count LDS-DMA next to ordinary
//     loads (it drains with vmcnt(0)), so the register loads are inline asm with hand-counted s_waitcnt, barriers are raw
//     s_barrier, LDS stores go through inline asm; one barrier per K step;
//   * MFMA roles are swapped (A operand = weights, B operand = activations) so that a lane ends up with 4 CONSECUTIVE
//     output columns of one row: 8-byte stores instead of 2-byte ones;
//   * blocks of one XCD walk the m-tiles of the same weight column tile first (weights are pulled from HBM once per
//     XCD L2, not once per m-tile).
// ---------------------------------------------------------------------------------------------------------------
#define GT_BN 128
#define GT_BK 64
#define GT_BTILE_BYTES (128 * 64 * 2)

// byte offset of the 16-byte chunk c (8 halves) of row r in a swizzled [rows][64] fp16 tile
__device__ __forceinline__ int gt_off(int r, int c) { return r * 128 + ((c ^ (r & 7)) << 4); }

// Stall attribution for scripts/probe_gemm.hip (compiled only there): cycles a wave spends in the VMEM wait, the dequant +
// LDS store, and the barrier of each K step; [block][wave][4] = {total, wait, store, barrier}.
#ifdef EXL_GEMM_PROBE
__device__ unsigned long long g_gemm_probe[1024 * 8 * 4];
#define GP_CLK(v) const unsigned long long v = __builtin_readcyclecounter()
#define GP_ACC(dst, a, b) dst += (b) - (a)
#else
#define GP_CLK(v) do { } while (0)
#define GP_ACC(dst, a, b) do { } while (0)
#endif

// ---------------------------------------------------------------------------------------------------------------
// Mid-step barrier schedule (the fallback of the wave-specialised kernel; 128-row tile variant for 257 .. 512 rows).  Stall attribution of its
// predecessor, which loaded, dequantised and multiplied tile by tile (scripts/probe_gemm.hip, profiles/r01_gemm_ablation.txt):
// of ~2100 cycles per K step only ~1000 were MFMA issue; after every barrier both waves of a
// SIMD wait for their first fragments, and the dequant + LDS store (300 cycles) and the barrier (150-500) run with an
// idle matrix pipe.  Here a K step is cut into two halves of 16 MFMAs and the barrier sits between the SECOND half of
// tile t-1 and the FIRST half of tile t:
//
//     barrier(t-1) | read Q <- frags(t, k 0..31) | 16 MFMAs on P = frags(t-1, k 32..63), DMA A(t+2) + loads B(t+2) between them
//                  | read P <- frags(t, k 32..63)| 16 MFMAs on Q, wait batch t+1 + dequant/store B(t+1) between them | barrier(t)
//
// so every MFMA group runs on fragments requested 16 MFMAs earlier and all VMEM / VALU / LDS-store work is issued in the
// shadow of MFMAs.  __builtin_amdgcn_sched_barrier(0) pins the groups in source order.  The first half-step runs on
// zero fragments (P starts as 0), the last two steps re-fetch the last tile instead of branching (straight-line loop:
// every hand-counted wait stays unconditional).
// ---------------------------------------------------------------------------------------------------------------
template <int WAVES_M, int WAVES_N, int TM, int TN, int KS>      // KS: 1, or 2 = the K range is cut in two, fp32 slices go to `ws`
__global__ __launch_bounds__(WAVES_M * WAVES_N * 64) void q4_gemm_t16m_kernel(const f16* __restrict__ x, const uint4* __restrict__ qw,
                                                               const uint32_t* __restrict__ qzeros,
                                                               const f16* __restrict__ scales, f16* __restrict__ out, int M,
                                                               int K, int N, int gshift, int groupsize, int no_zero, int mtiles,
                                                               int ntiles, float* __restrict__ ws)
{
    static_assert(WAVES_N * TN == 8, "block is 128 columns wide");
    static_assert(TM == 4 && TN == 4, "the MFMA groups below are written for 4 x 4 tiles per wave");
    constexpr int TBM = WAVES_M * TM * 16;
    constexpr int NW = WAVES_M * WAVES_N;
    constexpr int NTH = NW * 64;
    constexpr int APW = (TBM / 8) / NW;                               // 1 KiB activation pieces per wave per K step
    static_assert(APW == 4, "4 activation pieces per wave per K step");
    constexpr int A_BYTES = TBM * 128;
    constexpr int PW = 4 * 256 / NTH;                                 // words of a piece per thread (4 or 2)
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];       // [3][A] then [2][B]
    unsigned char* const ldsB = lds + 3 * A_BYTES;

    const int b = blockIdx.x;
    const int xcd = b & 7;
    const int idx = b >> 3;
    const int nl = idx / (mtiles * KS);
    const int rem = idx - nl * (mtiles * KS);
    const int kz = KS == 1 ? 0 : rem / mtiles;                        // which part of K (all row tiles of a part are neighbours: shared weights)
    const int mt = KS == 1 ? rem : rem - kz * mtiles;
    const int nt = nl * 8 + xcd;
    if (nt >= ntiles) return;
    const int m0 = mt * TBM;
    const int n0 = nt * GT_BN;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int RB = K >> 7;
    const int nk = K / GT_BK / KS;                                    // K steps of this block: even (K % (128 KS) == 0)
    const int it0 = kz * nk;                                          // its first K tile

    uint32_t a_off[APW];                                               // element offsets from x (M * K < 2^32 checked on the host)
#pragma unroll
    for (int i = 0; i < APW; ++i) {
        const int c = wave * APW + i;
        const int row = c * 8 + (lane >> 3);
        const int slot = lane & 7;
        const int grow = min(m0 + row, M - 1);
        a_off[i] = (uint32_t) grow * (uint32_t) K + ((slot ^ (row & 7)) << 3);
    }
    auto stage_a = [&](int i, int slot3, int k0) {                      // one 1 KiB piece
        __attribute__((address_space(3))) unsigned char* dst =
            (__attribute__((address_space(3))) unsigned char*) (lds + (size_t) slot3 * A_BYTES + (wave * APW + i) * 1024);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*) (x + (size_t) (a_off[i] + (uint32_t) k0)), dst, 16, 0, 0);
    };

    const int pid = PW == 4 ? tid : tid >> 1;                          // 256 or 512 threads
    const int ph = PW == 4 ? 0 : tid & 1;
    const int b_tile = pid >> 5;
    const int b_rs = (pid >> 4) & 1;
    const int b_col = pid & 15;
    const int b_nloc = b_tile * 16 + b_col;
    const int b_n = min(n0 + b_nloc, N - 1);
    const uint32_t* b_src = (const uint32_t*) (qw + ((size_t) (b_n >> 4) * RB) * 64 + (b_n & 15)) + ph * PW;
    const int b_zsh = (b_n & 7) * 4;
    const uint32_t magic = t16_magic();
    const uint32_t ginv = (uint32_t) (((1ull << 32) + (uint32_t) groupsize - 1) / (uint32_t) groupsize);
    const uint32_t b_lds = (uint32_t) (uintptr_t) (__attribute__((address_space(3))) unsigned char*) ldsB;
    uint32_t b_dst[PW];                                                // LDS byte offsets of this thread's chunks inside a B tile
#pragma unroll
    for (int j = 0; j < PW; ++j) b_dst[j] = b_lds + (uint32_t) gt_off(b_nloc, b_rs * 4 + ph * PW + j);

    struct BRegs { u32x4 w4; u32x2 w2; uint32_t zw, sc; };
    auto issue_w = [&](int it, BRegs& r) {
        const int rb = it >> 1, rsub = (it & 1) * 2 + b_rs;
        const uint32_t* p = b_src + ((size_t) rb * 64 + rsub * 16) * 4;
        if constexpr (PW == 4) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(r.w4) : "v"(p) : "memory");
        else                   asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(r.w2) : "v"(p) : "memory");
    };
    auto issue_zs = [&](int it, BRegs& r) {
        const uint32_t k = (uint32_t) (it * GT_BK + b_rs * 32);
        const uint32_t grp = __umulhi(k, ginv);                             // = k / groupsize (k < 2^16), no branch in the loop
        const uint32_t* zp = qzeros + (size_t) grp * (N >> 3) + (b_n >> 3);
        const f16* sp = scales + (size_t) grp * N + b_n;
        asm volatile("global_load_dword %0, %1, off" : "=v"(r.zw) : "v"(zp) : "memory");
        asm volatile("global_load_ushort %0, %1, off" : "=v"(r.sc) : "v"(sp) : "memory");
    };
#define GM_WAIT(NSTR, r)                                                                                              \
    do {                                                                                                              \
        if constexpr (PW == 4) asm volatile("s_waitcnt vmcnt(" NSTR ")" : "+v"(r.w4), "+v"(r.zw), "+v"(r.sc) :: "memory"); \
        else                   asm volatile("s_waitcnt vmcnt(" NSTR ")" : "+v"(r.w2), "+v"(r.zw), "+v"(r.sc) :: "memory"); \
    } while (0)
    // dequantise word j of the landed register set and store its 8 halves (one 16-byte chunk of the [n][k] tile)
    auto store_word = [&](int slot2, const BRegs& r, int j) {
        const int z = (int) ((r.zw >> b_zsh) & 0xFu) + 1;
        const f16 za = (f16) (float) (-(1024 + z));
        const f16 zb = (f16) (float) (-(64 + z));
        const f16 bsc = __builtin_bit_cast(f16, (uint16_t) (r.sc & 0xFFFFu));
        const f16x2 zc0 = {za, za}, zc1 = {zb, zb}, s2 = {bsc, bsc};
        const uint32_t word = PW == 4 ? r.w4[j] : r.w2[j & 1];
        const f16x8 d = t16_dequant_exact(word, magic, zc0, zc1);
        const uint4 u = __builtin_bit_cast(uint4, d);
        const u32x4 ov = {__builtin_bit_cast(uint32_t, as_h2(u.x) * s2), __builtin_bit_cast(uint32_t, as_h2(u.y) * s2),
                          __builtin_bit_cast(uint32_t, as_h2(u.z) * s2), __builtin_bit_cast(uint32_t, as_h2(u.w) * s2)};
        asm volatile("ds_write_b128 %0, %1" :: "v"(b_dst[j] + (uint32_t) (slot2 * GT_BTILE_BYTES)), "v"(ov) : "memory");
    };
    auto block_barrier = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");               // this wave's ds_writes are in LDS
        __builtin_amdgcn_s_barrier();
    };

    f32x4 acc[TN][TM];                                                   // [n-tile][m-tile]
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int fr = lane & 15, fk = lane >> 4;
    int fx_off[TM], fw_off[TN];                                           // fragment byte offsets for k-chunk (kk * 4 + fk) = fk; kk = 1 flips bit 6 of the swizzled chunk
#pragma unroll
    for (int i = 0; i < TM; ++i) fx_off[i] = gt_off((wm * TM + i) * 16 + fr, fk);
#pragma unroll
    for (int i = 0; i < TN; ++i) fw_off[i] = gt_off((wn * TN + i) * 16 + fr, fk);
    struct Frags { f16x8 x[TM], w[TN]; };
    auto read_frags = [&](int slot3, int slot2, int kk, Frags& f) {
        const unsigned char* at = lds + (size_t) slot3 * A_BYTES;
        const unsigned char* bt = ldsB + (size_t) slot2 * GT_BTILE_BYTES;
#pragma unroll
        for (int i = 0; i < TM; ++i) f.x[i] = *(const f16x8*) (at + (fx_off[i] ^ (kk << 6)));
#pragma unroll
        for (int i = 0; i < TN; ++i) f.w[i] = *(const f16x8*) (bt + (fw_off[i] ^ (kk << 6)));
    };
#define GM_MFMA(f, i) acc[(i) / TM][(i) % TM] = __builtin_amdgcn_mfma_f32_16x16x32_f16(f.w[(i) / TM], f.x[(i) % TM], acc[(i) / TM][(i) % TM], 0, 0, 0)
#define GM_MFMA2(f, i) do { GM_MFMA(f, i); GM_MFMA(f, (i) + 1); } while (0)
#define GM_MFMA4(f, i) do { GM_MFMA2(f, i); GM_MFMA2(f, (i) + 2); } while (0)
#define GM_SB() __builtin_amdgcn_sched_barrier(0)

    Frags P, Q;
#pragma unroll
    for (int i = 0; i < TM; ++i) P.x[i] = (f16x8){0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < TN; ++i) P.w[i] = (f16x8){0, 0, 0, 0, 0, 0, 0, 0};

    // ---- prologue: batches 0 and 1 in flight, B(0) dequantised --------------------------------------------------------
    BRegs rX, rY;
#pragma unroll
    for (int i = 0; i < APW; ++i) stage_a(i, 0, it0 * GT_BK);
    issue_w(it0, rX); issue_zs(it0, rX);
#pragma unroll
    for (int i = 0; i < APW; ++i) stage_a(i, 1, (it0 + 1) * GT_BK);      // nk >= 2 always
    issue_w(it0 + 1, rY); issue_zs(it0 + 1, rY);
    GM_WAIT("7", rX);                                                     // batch 0 landed (batch 1 may still fly)
#pragma unroll
    for (int j = 0; j < PW; ++j) store_word(0, rX, j);
    block_barrier();

#ifdef EXL_GEMM_PROBE
    unsigned long long p_wait = 0, p_store = 0, p_bar = 0;                // here: first half / second half / barrier
    const unsigned long long p_t0 = __builtin_readcyclecounter();
#endif
    int a_slot = 0;                                                       // LDS ring slot of tile t
    auto ring = [](int s, int d) { const int v = s + d; return v >= 3 ? v - 3 : v; };
    // one K step; rI receives the packed weights of tile t+2, rW holds tile t+1 (landing), bcur = LDS slot of B(t)
    auto step = [&](int t, BRegs& rI, BRegs& rW, int bcur) {
        const int tf = it0 + min(t + 2, nk - 1);                          // the last two steps re-fetch the last tile (never read)
        const int adma = ring(a_slot, 2);
        GP_CLK(c0);
        GM_MFMA2(P, 0);   stage_a(0, adma, tf * GT_BK);  GM_SB();
        read_frags(a_slot, bcur, 0, Q);                  GM_SB();          // behind the first MFMAs: hipcc waits with lgkmcnt(0)
        GM_MFMA2(P, 2);   stage_a(1, adma, tf * GT_BK);  GM_SB();
        GM_MFMA2(P, 4);   stage_a(2, adma, tf * GT_BK);  GM_SB();
        GM_MFMA2(P, 6);   stage_a(3, adma, tf * GT_BK);  GM_SB();
        GM_MFMA2(P, 8);   issue_w(tf, rI);               GM_SB();
        GM_MFMA4(P, 10);  issue_zs(tf, rI);              GM_SB();
        GM_MFMA2(P, 14);                                 GM_SB();
        GP_CLK(c1);
        GM_MFMA4(Q, 0);   GM_WAIT("7", rW);              GM_SB();
        read_frags(a_slot, bcur, 1, P);                  GM_SB();
        if constexpr (PW == 2) {
            GM_MFMA4(Q, 4);   store_word(bcur ^ 1, rW, 0);   GM_SB();
            GM_MFMA4(Q, 8);   store_word(bcur ^ 1, rW, 1);   GM_SB();
            GM_MFMA4(Q, 12);                                 GM_SB();
        } else {
            GM_MFMA2(Q, 4);   store_word(bcur ^ 1, rW, 0);   GM_SB();
            GM_MFMA2(Q, 6);   store_word(bcur ^ 1, rW, 1);   GM_SB();
            GM_MFMA2(Q, 8);   store_word(bcur ^ 1, rW, 2);   GM_SB();
            GM_MFMA2(Q, 10);  store_word(bcur ^ 1, rW, 3);   GM_SB();
            GM_MFMA4(Q, 12);                                 GM_SB();
        }
        GP_CLK(c2);
        block_barrier();
        GP_CLK(c3);
        GP_ACC(p_wait, c0, c1); GP_ACC(p_store, c1, c2); GP_ACC(p_bar, c2, c3);
        a_slot = ring(a_slot, 1);
    };
    for (int t = 0; t < nk; t += 2) {
        step(t, rX, rY, 0);
        step(t + 1, rY, rX, 1);
    }
    GM_MFMA4(P, 0); GM_MFMA4(P, 4); GM_MFMA4(P, 8); GM_MFMA4(P, 12);      // second half of the last tile
    // The two redundant fetches.  Their destination registers are operands of the wait on purpose: to the compiler the values are
    // dead after the loop, and without the tie it shuffled accumulators through the very registers a dwordx4 was still on its way
    // to (v_accvgpr_read v4..v7 / v_accvgpr_write between the loop and this wait): whole accumulator registers of a wave came out
    // as weight bits whenever the last fetch was slow (first launch after an idle period, two blocks per CU): the round-2 "400 rows
    // x 11008 columns" defect, profiles/HISTORY.md 9.5.
    GM_WAIT("0", rX); GM_WAIT("0", rY);
#ifdef EXL_GEMM_PROBE
    if (lane == 0 && b < 1024) {
        unsigned long long* pp = g_gemm_probe + ((size_t) b * 8 + wave) * 4;
        pp[0] = __builtin_readcyclecounter() - p_t0; pp[1] = p_wait; pp[2] = p_store; pp[3] = p_bar;
    }
#endif
#undef GM_WAIT
#undef GM_MFMA
#undef GM_MFMA2
#undef GM_MFMA4
#undef GM_SB

#pragma unroll
    for (int im = 0; im < TM; ++im) {
        const int row = m0 + (wm * TM + im) * 16 + fr;
        if (row < M) {
#pragma unroll
            for (int in = 0; in < TN; ++in) {
                const int n = n0 + (wn * TN + in) * 16 + fk * 4;
                if (n < N) {
                    if constexpr (KS > 1) {                               // fp32 slice of this K part; q4_gemm_splitk_reduce_kernel adds them up
                        *(f32x4*) (ws + ((size_t) kz * M + row) * N + n) = acc[in][im];
                        continue;
                    }
                    f16* op = out + (size_t) row * N + n;
                    float v[4] = {acc[in][im][0], acc[in][im][1], acc[in][im][2], acc[in][im][3]};
                    if (no_zero) {
                        const f16x4 prev = *(const f16x4*) op;
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] += (float) prev[j];
                    }
                    *(f16x4*) op = (f16x4){(f16) v[0], (f16) v[1], (f16) v[2], (f16) v[3]};
                }
            }
        }
    }
}

// out (+)= sum of the KS fp32 slices of the split-K form above (4 columns per thread)
__global__ __launch_bounds__(256) void q4_gemm_splitk_reduce_kernel(const float* __restrict__ ws, f16* __restrict__ out, size_t cells4,
                                                                    size_t slice, int ks, int no_zero)
{
    const size_t i = (size_t) blockIdx.x * 256 + threadIdx.x;
    if (i >= cells4) return;
    f32x4 v = *(const f32x4*) (ws + i * 4);
    for (int z = 1; z < ks; ++z) {
        const f32x4 u = *(const f32x4*) (ws + (size_t) z * slice + i * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] += u[j];
    }
    f16* op = out + i * 4;
    if (no_zero) {
        const f16x4 prev = *(const f16x4*) op;
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] += (float) prev[j];
    }
    *(f16x4*) op = (f16x4){(f16) v[0], (f16) v[1], (f16) v[2], (f16) v[3]};
}

// ---------------------------------------------------------------------------------------------------------------
// Wave-specialised version of the same 256 x 128 x 64 tile (the default for > 256 rows).  Stall attribution of the
// kernels above (scripts/probe_gemm.hip): the half step that carries the 4 LDS-DMA instructions + 3 loads of a wave takes
// 640 cycles for the older and 1030 for the younger wave of a SIMD instead of 256 -- the CU's vector-memory pipe takes
// ~16 cycles per wave instruction (32 KiB of activations per K step = 512 cycles) and a wave stalled on a VMEM issue
// cannot issue the MFMAs behind it.  So the roles are split:
//   * waves 0-7 (consumers, 64 x 64 each): nothing but fragment reads and MFMAs, software-pipelined over the barrier
//     (first half of tile t runs on fragments requested before the second half of tile t-1 was issued);
//   * waves 8-11 (producers, one per SIMD): per K step 8 LDS-DMA pieces of A(t+2), one 16-byte piece quarter of
//     B(t+2) + zero word + scale into registers, then dequantise B(t+1) (landed a step ago) into the [n][k] LDS tile.
// One s_barrier per K step over all 12 waves; same 3-slot A ring / 2-slot B ring, same hand-counted vmcnt.
// ---------------------------------------------------------------------------------------------------------------
#ifndef GW_ABL
#define GW_ABL 0                // scripts/probe_gemm.hip only: 1 = no activation DMA, 2 = no B-tile LDS stores, 8 = no fragment reads
#endif
#define GW_CONS 8
#define GW_PROD 4
#define GW_BATCH_STR "11"       // 3 loads + 8 DMA pieces
// EPI 1 = the attention front half of the prompt pass in ONE launch (exl_q4_qkv_rope_cache): the block's 128 columns are one
// head of q, k or v (three matrices sharing x and K, tiles numbered across them); the epilogue applies RoPE to q and k
// (the partner column d +- 64 sits in the other column wave: exchanged through LDS) and writes q to its buffer, k and v
// straight into the KV cache -- the reference runs 3 matmuls, 2 RoPE kernels and a cache scatter (model.py:431-445).
struct GemmTail { int b_split, parts; float* ws; int mr; };      // see plan_gemm_tail; mr: row tiles per XCD rectangle (gemm_tile_of)

// Logical block -> (row tile, column tile).  Block b runs on XCD b % 8 (observed dispatch order; speed only), so the blocks b, b + 8,
// b + 16 ... are what ONE XCD's 32 CUs hold at a time and its 4 MiB L2 serves.  They walk the column tiles the XCD owns (nt % 8 ==
// xcd) for `mr` row tiles at a time.  mr = mtiles (the default, the order of rounds 1-3): every row tile of 4 column tiles at a
// time -- a weight tile is shared by 8 blocks, but the whole activation matrix (16.8 MB at K = 4096) streams through each L2 for
// every four column tiles: PMC in round 4 saw 667 MB of fabric traffic per q/k/v launch against 93 MB of algorithmic bytes
// (profiles/r04_pmc_traffic.json).  mr = 2 (EXL_GEMM_TILE_ROWS=2): 2 row tiles x up to 16 column tiles, K slices of activations
// and packed weights of about equal size, the rectangle with the fewest bytes from outside the L2 per tile -- 283 MB for the same
// launch.  MEASURED AND NOT MADE THE DEFAULT: 13B act-order prompt 38.5 / 38.7 k tokens/s with the rectangles against 39.8 / 39.6 k
// with the old order (alternating runs on one box), 7B 78.5 vs 78.0 k: the traffic is not what bounds these kernels (the loader
// waves wait 111 of 1473 cycles per K step, profiles/HISTORY.md 9.3), and a weight tile shared by 2 blocks instead of 8 makes the
// latency-critical register loads of B miss the L2 more often than the deep activation DMA ring ever stalled.
__host__ __device__ __forceinline__ bool gemm_tile_of(int b, int mtiles, int ntiles, int mr, int* mt, int* nt)
{
    const int xcd = b & 7, idx = b >> 3;
    const int nper = (ntiles + 7) >> 3;
    const int grp = mr * nper;
    const int mg = idx / grp, r = idx - mg * grp;
    const int m0 = mg * mr;
    int gsz = mtiles - m0;
    if (gsz > mr) gsz = mr;
    if (gsz <= 0) return false;
    const int j = r / gsz;
    *mt = m0 + (r - j * gsz);
    *nt = j * 8 + xcd;
    return j < nper && *nt < ntiles;
}
struct GwQkv {
    const uint4* qw[3]; const uint32_t* qz[3]; const f16* sc[3];
    int n[3];                      // out_features of q, k, v
    int nt_end[3];                 // cumulative 128-column tile counts
    f16* q_out;                    // [rows][n[0]]
    const f16 *sin, *cos;          // [max_seq][128]
    f16 *kc, *vc;                  // [bsz][kv_heads][max_seq][128]
    int q_len, past_len, max_seq, kv_heads;
};
template <int EPI>
__global__ __launch_bounds__((GW_CONS + GW_PROD) * 64) void q4_gemm_t16w_kernel(const f16* __restrict__ x, const uint4* __restrict__ qw,
                                                               const uint32_t* __restrict__ qzeros,
                                                               const f16* __restrict__ scales, f16* __restrict__ out, int M,
                                                               int K, int N, int gshift, int no_zero, int mtiles, int ntiles,
                                                               const GwQkv e, const GemmTail tail)
{
    constexpr int TM = 4, TN = 4, WAVES_N = 2;
    constexpr int TBM = 256;
    constexpr int A_BYTES = TBM * 128;
    constexpr int APW = (TBM / 8) / GW_PROD;                          // 8 one-KiB activation pieces per producer wave per K step
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];       // [3][A] then [2][B]
    unsigned char* const ldsB = lds + 3 * A_BYTES;

    int b = blockIdx.x;
    int kz = 0, nparts = 1, tseq = 0;                                  // tail blocks: part kz of `parts` of the K range of logical block b_split + tseq
    if (EPI == 0 && b >= tail.b_split) {
        const int j = b - tail.b_split;
        tseq = j / tail.parts;
        kz = j - tseq * tail.parts;
        b = tail.b_split + tseq;
        nparts = tail.parts;
    }
    int mt, nt;
    if (!gemm_tile_of(b, mtiles, ntiles, tail.mr, &mt, &nt)) return;
    const int m0 = mt * TBM;
    int mi = 0;                                                        // EPI 1: which of q / k / v this block works on
    if constexpr (EPI == 1) {
        if (nt >= e.nt_end[1]) { mi = 2; nt -= e.nt_end[1]; }
        else if (nt >= e.nt_end[0]) { mi = 1; nt -= e.nt_end[0]; }
        qw = e.qw[mi]; qzeros = e.qz[mi]; scales = e.sc[mi]; N = e.n[mi];
    }
    const int n0 = nt * GT_BN;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nk = K / GT_BK / nparts;                                // K steps of this block: even (K % (128 parts) == 0), >= 2
    const int it0 = kz * nk;                                          // its first K tile
    auto ring = [](int s, int d) { const int v = s + d; return v >= 3 ? v - 3 : v; };

    if (wave >= GW_CONS) {
        // ================================================= producer =================================================
        // Every address is (uniform base, advanced by scalar adds per K step) + (fixed 32-bit byte offset per lane): the
        // loads use the saddr form and cost no vector instruction -- a single wave issues only ~1 instruction per 5-6
        // cycles, so the loader's instruction count per K step is what bounds it.
        const int pw = wave - GW_CONS;
        const int ptid = tid - GW_CONS * 64;                            // 0..255
        const int RB = K >> 7;
        uint32_t a_voff[APW];                                            // byte offsets from x (M * K * 2 < 2^32 checked on the host)
#pragma unroll
        for (int i = 0; i < APW; ++i) {
            const int c = pw * APW + i;
            const int row = c * 8 + (lane >> 3);
            const int slot = lane & 7;
            const int grow = min(m0 + row, M - 1);
            a_voff[i] = ((uint32_t) grow * (uint32_t) K + ((slot ^ (row & 7)) << 3)) * 2u;
        }
        const uint32_t lds_a0 = (uint32_t) (uintptr_t) (__attribute__((address_space(3))) unsigned char*) lds + (uint32_t) (pw * APW * 1024);
        // pieces 2 * pair and 2 * pair + 1 of this wave's 8: M0 carries the LDS address of an LDS-DMA
        auto stage_a2 = [&](int pair, int slot3, int k0) {
            const f16* xk = x + k0;                                      // uniform
            const uint32_t l = lds_a0 + (uint32_t) slot3 * A_BYTES + (uint32_t) pair * 2048;
            if (GW_ABL & 1) return;
            asm volatile("s_mov_b32 m0, %[l]\n\t"
                         "global_load_lds_dwordx4 %[v0], %[sb]\n\t"
                         "s_add_u32 m0, m0, 0x400\n\t"
                         "global_load_lds_dwordx4 %[v1], %[sb]"
                         :: [l] "s"(l), [sb] "s"(xk), [v0] "v"(a_voff[2 * pair]), [v1] "v"(a_voff[2 * pair + 1]) : "memory", "scc");
        };
        auto stage_a = [&](int slot3, int k0) {                         // all 8 pieces (prologue)
#pragma unroll
            for (int pr = 0; pr < APW / 2; ++pr) stage_a2(pr, slot3, k0);
        };
        const int b_tile = ptid >> 5;
        const int b_rs = (ptid >> 4) & 1;
        const int b_col = ptid & 15;
        const int b_nloc = b_tile * 16 + b_col;
        const int b_n = min(n0 + b_nloc, N - 1);
        // packed words of K step `it`: piece (n, rb = it / 2), sub-row (it & 1) * 2 + b_rs  ->  uniform it * 512 bytes + lane part
        const uint32_t w_voff = (uint32_t) (((size_t) (b_n >> 4) * RB * 64 + (b_n & 15)) * 16 + (size_t) b_rs * 256);
        // group of k = it * 64 + b_rs * 32 (groupsize a power of two >= 32): uniform (it * 64) >> gshift, + b_rs for groupsize 32
        const int zs_lane_grp = gshift == 5 ? b_rs : 0;
        const uint32_t z_voff = (uint32_t) ((zs_lane_grp * (N >> 3) + (b_n >> 3)) * 4);
        const uint32_t s_voff = (uint32_t) ((zs_lane_grp * N + b_n) * 2);
        const int b_zsh = (b_n & 7) * 4;
        const uint32_t magic = t16_magic();
        const uint32_t b_lds = (uint32_t) (uintptr_t) (__attribute__((address_space(3))) unsigned char*) ldsB;
        uint32_t b_dst[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) b_dst[j] = b_lds + (uint32_t) gt_off(b_nloc, b_rs * 4 + j);

        struct BRegs { u32x4 w4; uint32_t zw, sc; };
        auto issue_b = [&](int it, BRegs& r) {
            const unsigned char* wp = (const unsigned char*) qw + (size_t) it * 512;                       // uniform
            const int grp = (it * GT_BK) >> gshift;
            const uint32_t* zp = qzeros + (size_t) grp * (N >> 3);
            const f16* sp = scales + (size_t) grp * N;
            asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(r.w4) : "v"(w_voff), "s"(wp) : "memory");
            asm volatile("global_load_dword %0, %1, %2" : "=v"(r.zw) : "v"(z_voff), "s"(zp) : "memory");
            asm volatile("global_load_ushort %0, %1, %2" : "=v"(r.sc) : "v"(s_voff), "s"(sp) : "memory");
        };
        // one batch = 3 loads + 8 DMA pieces = 11 VMEM operations, in that order
#define GW_WAIT(NSTR, r) asm volatile("s_waitcnt vmcnt(" NSTR ")" : "+v"(r.w4), "+v"(r.zw), "+v"(r.sc) :: "memory")
        // -(1024 + z + 1) and -(64 + z + 1) as fp16 bit patterns: 0xE400 + n, 0xD400 + 16 n (exact integers below 2048 / 128)
        auto store_word = [&](int slot2, const BRegs& r, int j) {
            const uint32_t z1 = ((r.zw >> b_zsh) & 0xFu) + 1u;
            const uint32_t z2 = z1 | (z1 << 16);
            const f16x2 zc0 = as_h2(0xE400E400u + z2), zc1 = as_h2(0xD400D400u + (z2 << 4));
            const uint32_t sb = r.sc & 0xFFFFu;
            const f16x2 s2 = as_h2(sb | (sb << 16));
            const f16x8 d = t16_dequant_exact(r.w4[j], magic, zc0, zc1);
            const uint4 u = __builtin_bit_cast(uint4, d);
            const u32x4 ov = {__builtin_bit_cast(uint32_t, as_h2(u.x) * s2), __builtin_bit_cast(uint32_t, as_h2(u.y) * s2),
                              __builtin_bit_cast(uint32_t, as_h2(u.z) * s2), __builtin_bit_cast(uint32_t, as_h2(u.w) * s2)};
            if (GW_ABL & 2) asm volatile("; keep %0 %1" :: "v"(b_dst[j] + (uint32_t) (slot2 * GT_BTILE_BYTES)), "v"(ov));
            else asm volatile("ds_write_b128 %0, %1" :: "v"(b_dst[j] + (uint32_t) (slot2 * GT_BTILE_BYTES)), "v"(ov) : "memory");
        };
        auto store_b = [&](int slot2, const BRegs& r) {
#pragma unroll
            for (int j = 0; j < 4; ++j) store_word(slot2, r, j);
        };
        auto publish = [&]() {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           // this wave's ds_writes are in LDS
            __builtin_amdgcn_s_barrier();
        };
        BRegs rX, rY;
        issue_b(it0, rX);      stage_a(0, it0 * GT_BK);
        issue_b(it0 + 1, rY);  stage_a(1, (it0 + 1) * GT_BK);
        GW_WAIT("11", rX);                                               // batch 0 landed
        store_b(0, rX);
        publish();
        int a_slot = 0;
#ifdef EXL_GEMM_PROBE
        unsigned long long p_wait = 0, p_store = 0, p_bar = 0, p_iss = 0;
        const unsigned long long p_t0 = __builtin_readcyclecounter();
#endif
        auto half_step = [&](int adma, int tf, BRegs& rI, BRegs& rW, int bstore) {
            GP_CLK(q0);
            issue_b(tf, rI);
            GP_CLK(q1);
            GW_WAIT(GW_BATCH_STR, rW);                                   // B(t+1) in registers; its DMA pieces + the 3 new loads may still fly
            GP_CLK(q2);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                store_word(bstore, rW, j);
                stage_a2(j, adma, tf * GT_BK);
                __builtin_amdgcn_sched_barrier(0);
            }
            GP_CLK(q3);
            asm volatile("s_waitcnt vmcnt(" GW_BATCH_STR ")" ::: "memory");   // A(t+1) landed (B(t+2) + A(t+2) are the operations behind it)
            publish();
            GP_CLK(q4);
            GP_ACC(p_iss, q0, q1); GP_ACC(p_wait, q1, q2); GP_ACC(p_store, q2, q3); GP_ACC(p_bar, q3, q4);
        };
        for (int t = 0; t < nk; t += 2) {                                // straight line: every wait is unconditional
            const int t2 = it0 + min(t + 2, nk - 1), t3 = it0 + min(t + 3, nk - 1);   // the last two steps re-fetch the last tile (never read)
            // B(t+2) first, then the wait for B(t+1)'s registers, then the 8 DMA
            // pieces of A(t+2) BETWEEN the dequantised words: the CU's vector-memory pipe takes ~16 cycles per 1 KiB piece
            // whoever issues it, so DMA issue and the VALU work overlap instead of adding up
            half_step(ring(a_slot, 2), t2, rX, rY, 1);
            a_slot = ring(a_slot, 1);
            half_step(ring(a_slot, 2), t3, rY, rX, 0);
            a_slot = ring(a_slot, 1);
        }
        GW_WAIT("0", rX); GW_WAIT("0", rY);                              // the redundant fetches must not outlive the block's LDS (registers tied: see q4_gemm_t16m_kernel)
        if constexpr (EPI == 1) {
            if (mi < 2) { __builtin_amdgcn_s_barrier(); __builtin_amdgcn_s_barrier(); }   // the RoPE exchange of the MFMA waves reuses the tiles
        }
#ifdef EXL_GEMM_PROBE
        if (lane == 0 && b < 1024) {                                    // producers report in slots 4..7 of the block: {issue, wait, store, barrier} of the even steps
            unsigned long long* pp = g_gemm_probe + ((size_t) b * 8 + 4 + pw) * 4;
            pp[0] = p_iss; pp[1] = p_wait; pp[2] = p_store; pp[3] = p_bar;
        }
#endif
#undef GW_WAIT
        return;
    }

    // ===================================================== consumer =====================================================
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    f32x4 acc[TN][TM];                                                   // [n-tile][m-tile]
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int fr = lane & 15, fk = lane >> 4;
    // fragment byte offsets of tile 0, k-chunk fk; tile i is 16 rows = 2048 bytes further (same swizzle), the second
    // half of the K step (chunk 4 + fk) flips bit 6 of the swizzled offset
    const int fx0 = gt_off(wm * TM * 16 + fr, fk), fw0 = gt_off(wn * TN * 16 + fr, fk);
    struct Frags { f16x8 x[TM], w[TN]; };
    auto read_frags = [&](int slot3, int slot2, int kk, Frags& f) {
        const unsigned char* at = lds + (size_t) slot3 * A_BYTES + (fx0 ^ (kk << 6));
        const unsigned char* bt = ldsB + (size_t) slot2 * GT_BTILE_BYTES + (fw0 ^ (kk << 6));
#pragma unroll
        for (int i = 0; i < TM; ++i) f.x[i] = *(const f16x8*) (at + i * 2048);
#pragma unroll
        for (int i = 0; i < TN; ++i) f.w[i] = *(const f16x8*) (bt + i * 2048);
    };
#define GW_MFMA(f, i) acc[(i) / TM][(i) % TM] = __builtin_amdgcn_mfma_f32_16x16x32_f16(f.w[(i) / TM], f.x[(i) % TM], acc[(i) / TM][(i) % TM], 0, 0, 0)
#define GW_MFMA2(f, i) do { GW_MFMA(f, i); GW_MFMA(f, (i) + 1); } while (0)
#define GW_MFMA4(f, i) do { GW_MFMA2(f, i); GW_MFMA2(f, (i) + 2); } while (0)
#define GW_SB() __builtin_amdgcn_sched_barrier(0)
    Frags P, Q;
#pragma unroll
    for (int i = 0; i < TM; ++i) P.x[i] = (f16x8){0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < TN; ++i) P.w[i] = (f16x8){0, 0, 0, 0, 0, 0, 0, 0};
    __builtin_amdgcn_s_barrier();                                        // A(0), B(0) published
    int a_slot = 0;
#ifdef EXL_GEMM_PROBE
    unsigned long long p_wait = 0, p_store = 0, p_bar = 0;
    const unsigned long long p_t0 = __builtin_readcyclecounter();
    const unsigned long long p_r0 = __builtin_amdgcn_s_memrealtime();      // constant 100 MHz
#endif
    auto step = [&](int t, int bcur) {
        GP_CLK(c0);
        GW_MFMA4(P, 0);                       GW_SB();                    // second half of tile t-1 (zeros at t = 0)
        if ((GW_ABL & 4) && t > 1) { bcur = 0; }
        if (!(GW_ABL & 8) || t < 2)
        read_frags(a_slot, bcur, 0, Q);       GW_SB();
        GW_MFMA4(P, 4); GW_MFMA4(P, 8); GW_MFMA4(P, 12);  GW_SB();
        GW_MFMA4(Q, 0);                       GW_SB();
        if (!(GW_ABL & 8) || t < 2)
        read_frags(a_slot, bcur, 1, P);       GW_SB();
        GW_MFMA4(Q, 4); GW_MFMA4(Q, 8); GW_MFMA4(Q, 12);  GW_SB();
        GP_CLK(c1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");               // this wave's reads of tile t are done before the slots are recycled
        __builtin_amdgcn_s_barrier();
        GP_CLK(c2);
        GP_ACC(p_wait, c0, c1); GP_ACC(p_bar, c1, c2);
        a_slot = ring(a_slot, 1);
    };
    for (int t = 0; t < nk; t += 2) { step(t, 0); step(t + 1, 1); }
    GW_MFMA4(P, 0); GW_MFMA4(P, 4); GW_MFMA4(P, 8); GW_MFMA4(P, 12);      // second half of the last tile
#ifdef EXL_GEMM_PROBE
    if (lane == 0 && b < 1024 && wave < 4) {                            // consumers 0..3 report {total, mfma part, -, barrier}
        unsigned long long* pp = g_gemm_probe + ((size_t) b * 8 + wave) * 4;
        pp[0] = __builtin_readcyclecounter() - p_t0; pp[1] = p_wait; pp[2] = __builtin_amdgcn_s_memrealtime() - p_r0; pp[3] = p_bar;
    }
#endif
#undef GW_MFMA
#undef GW_MFMA2
#undef GW_MFMA4
#undef GW_SB

    if constexpr (EPI == 1) {
        // acc[in][im][j]: row m0 + (wm * 4 + im) * 16 + fr, column d = wn * 64 + in * 16 + fk * 4 + j of head n0 / 128
        const int head = n0 >> 7;
        uint2 val[TN][TM];                                             // fp16 results, 4 consecutive columns per (in, im)
#pragma unroll
        for (int in = 0; in < TN; ++in)
#pragma unroll
            for (int im = 0; im < TM; ++im) {
                const f16x4 h4 = {(f16) acc[in][im][0], (f16) acc[in][im][1], (f16) acc[in][im][2], (f16) acc[in][im][3]};
                val[in][im] = __builtin_bit_cast(uint2, h4);
            }
        uint2 oth[TN][TM];
        if (mi < 2) {                                                  // q, k: fetch the partner half (d +- 64) from the other column wave
            uint2* xch = (uint2*) lds;                                 // [8 waves][16 tiles][64 lanes]
            __builtin_amdgcn_s_barrier();                              // every wave is past its last fragment read, the loaders' DMAs have landed
#pragma unroll
            for (int in = 0; in < TN; ++in)
#pragma unroll
                for (int im = 0; im < TM; ++im) xch[(wave * 16 + in * 4 + im) * 64 + lane] = val[in][im];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
#pragma unroll
            for (int in = 0; in < TN; ++in)
#pragma unroll
                for (int im = 0; im < TM; ++im) oth[in][im] = xch[((wave ^ 1) * 16 + in * 4 + im) * 64 + lane];
        }
#pragma unroll
        for (int im = 0; im < TM; ++im) {
            const int row = m0 + (wm * TM + im) * 16 + fr;
            if (row >= M) continue;
            const int bb = row / e.q_len, pos = e.past_len + (row - bb * e.q_len);
#pragma unroll
            for (int in = 0; in < TN; ++in) {
                const int d = wn * 64 + in * 16 + fk * 4;
                f16x4 r = __builtin_bit_cast(f16x4, val[in][im]);
                if (mi < 2) {                                          // rope.cu:21-88: l' = fma(l, cos_l, h(r * h(-sin_l))), r' = fma(r, cos_r, h(l * sin_r))
                    const f16x4 o4 = __builtin_bit_cast(f16x4, oth[in][im]);
                    const f16x4 sn = *(const f16x4*) (e.sin + (size_t) pos * 128 + d);
                    const f16x4 cs = *(const f16x4*) (e.cos + (size_t) pos * 128 + d);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const f16 t = o4[j] * (wn == 0 ? (f16) (-sn[j]) : sn[j]);
                        r[j] = __builtin_fmaf16(r[j], cs[j], t);
                    }
                }
                f16* dst = mi == 0 ? e.q_out + (size_t) row * e.n[0] + head * 128 + d
                                   : (mi == 1 ? e.kc : e.vc) + (((size_t) bb * e.kv_heads + head) * e.max_seq + pos) * 128 + d;
                *(f16x4*) dst = r;
            }
        }
        return;
    }

    if (nparts > 1) {                                                    // fp32 slice of this part of K: q4_gemm_tail_reduce_kernel adds them up
        float* wt = tail.ws + (size_t) (tseq * nparts + kz) * (256 * 128);
#pragma unroll
        for (int im = 0; im < TM; ++im)
#pragma unroll
            for (int in = 0; in < TN; ++in)
                *(f32x4*) (wt + ((wm * TM + im) * 16 + fr) * 128 + (wn * TN + in) * 16 + fk * 4) = acc[in][im];
        return;
    }
#pragma unroll
    for (int im = 0; im < TM; ++im) {
        const int row = m0 + (wm * TM + im) * 16 + fr;
        if (row < M) {
#pragma unroll
            for (int in = 0; in < TN; ++in) {
                const int n = n0 + (wn * TN + in) * 16 + fk * 4;
                if (n < N) {
                    f16* op = out + (size_t) row * N + n;
                    float v[4] = {acc[in][im][0], acc[in][im][1], acc[in][im][2], acc[in][im][3]};
                    if (no_zero) {
                        const f16x4 prev = *(const f16x4*) op;
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] += (float) prev[j];
                    }
                    *(f16x4*) op = (f16x4){(f16) v[0], (f16) v[1], (f16) v[2], (f16) v[3]};
                }
            }
        }
    }
}

// Epilogue of the tail blocks (GemmTail): logical block b_split + blockIdx.x sums its `parts` fp32 slices in a fixed order and
// finishes the tile -- DUAL 0: out (+)= sum; DUAL 1: a slice holds the gate tile then the up tile, out = silu(gate) * up in the
// fp16 order of q4_gemm_t16d2_kernel's own epilogue; DUAL 2: the two products go to out and out2.
template <int DUAL>
__global__ __launch_bounds__(256) void q4_gemm_tail_reduce_kernel(f16* __restrict__ out, f16* __restrict__ out2, int M, int N, int no_zero,
                                                                  int mtiles, int ntiles, const GemmTail tail)
{
    const int b = tail.b_split + blockIdx.x;
    int mt, nt;
    if (!gemm_tile_of(b, mtiles, ntiles, tail.mr, &mt, &nt)) return;
    const int m0 = mt * 256, n0 = nt * GT_BN;
    constexpr int SLICE = (DUAL ? 2 : 1) * 256 * 128;
    const float* w0 = tail.ws + (size_t) blockIdx.x * tail.parts * SLICE;
    for (int c = threadIdx.x; c < 256 * 32; c += 256) {                  // 4-column chunks of the 256 x 128 tile
        const int r = c >> 5, c4 = (c & 31) * 4;
        const int row = m0 + r, n = n0 + c4;
        if (row >= M || n >= N) continue;
        f32x4 v = *(const f32x4*) (w0 + r * 128 + c4), u = {0.f, 0.f, 0.f, 0.f};
        if (DUAL) u = *(const f32x4*) (w0 + 256 * 128 + r * 128 + c4);
        for (int z = 1; z < tail.parts; ++z) {
            const f32x4 a = *(const f32x4*) (w0 + (size_t) z * SLICE + r * 128 + c4);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] += a[j];
            if (DUAL) {
                const f32x4 a2 = *(const f32x4*) (w0 + (size_t) z * SLICE + 256 * 128 + r * 128 + c4);
#pragma unroll
                for (int j = 0; j < 4; ++j) u[j] += a2[j];
            }
        }
        f16* op = out + (size_t) row * N + n;
        f16x4 o;
        if (DUAL == 2) {
            o = (f16x4){(f16) v[0], (f16) v[1], (f16) v[2], (f16) v[3]};
            *(f16x4*) (out2 + (size_t) row * N + n) = (f16x4){(f16) u[0], (f16) u[1], (f16) u[2], (f16) u[3]};
        } else if (DUAL) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {                                // exactly elementwise.hip: silu_mul_h on the fp16 values
                const f16 g = (f16) v[j], up = (f16) u[j];
                const f16 e = (f16) __expf((float) (f16) (-g));
                const f16 sm = (f16) 1.0f + e;
                const f16 rc = (f16) (1.0f / (float) sm);
                const f16 t = g * rc;
                o[j] = t * up;
            }
        } else {
            if (no_zero) {
                const f16x4 prev = *(const f16x4*) op;
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] += (float) prev[j];
            }
            o = (f16x4){(f16) v[0], (f16) v[1], (f16) v[2], (f16) v[3]};
        }
        *(f16x4*) op = o;
    }
}

// Tail of a launch whose tile count is not a multiple of the CU count (7B gate / up: 688 tiles = 2.69 rounds of 256 CUs; 13B
// o / down: 320 = 1.25): the tiles of the last, partly filled round are cut along K into `parts` blocks each, so the round's work
// is spread over the whole chip; their fp32 slices are summed by q4_gemm_tail_reduce_kernel.  Only when at least one FULL round
// precedes the tail (small problems keep their one-block-per-tile form and their bit patterns).  Deterministic: the split depends
// on (tiles, K) only, the slices are added in a fixed order -- but the K split makes the fp32 sums of the tail tiles differ from
// the unsplit kernel's by rounding.  Returns the grid size; t->parts == 1: no split.
static int plan_gemm_tail(int device, int mtiles, int ntiles, int K, int slices_per_tile, GemmTail* t, int* n_tail)
{
    static const bool off = getenv("EXL_GEMM_NO_TAIL_SPLIT") != nullptr;      // A/B switch
    const int padded = 8 * ((ntiles + 7) / 8) * mtiles;
    static const int mr_env = getenv("EXL_GEMM_TILE_ROWS") ? atoi(getenv("EXL_GEMM_TILE_ROWS")) : 0;   // 0 (default) = every row tile of 4 column tiles; 2 = the low-traffic rectangles (measured slower)
    t->b_split = padded; t->parts = 1; t->ws = nullptr;
    t->mr = mr_env > 0 && mr_env < mtiles ? mr_env : mtiles;
    *n_tail = 0;
    int ncu = 256;
    {
        static int cus[EXL_MAX_DEVICES] = {};
        if (device >= 0 && device < EXL_MAX_DEVICES) {
            if (!cus[device]) { hipDeviceProp_t p; if (hipGetDeviceProperties(&p, device) == hipSuccess) cus[device] = p.multiProcessorCount; }
            if (cus[device] > 0) ncu = cus[device];
        }
    }
    const int tiles = mtiles * ntiles;
    const int full = tiles / ncu * ncu, rem = tiles - full;
    if (off || full == 0 || rem == 0) return padded;
    // choose the split: parts in {2, 4, 8} with whole 128-row blocks of K per part.  Cost of the tail in units of one whole tile's
    // time (2 * 256 * 128 * K flops at the ~4.7 TFLOP/s a CU sustains in these kernels): rounds / parts, ~3 K steps of fill / drain
    // per part, and the fp32 slices' round trip through HBM at ~6 TB/s -- the term that decides: measured in round 4
    // (gpurun_out/r04a), 7B gate / up with 176 tail tiles cut in 4 wrote + read 360 MB and ran 360 us instead of 305; 13B o / down (64
    // tail tiles, 67 MB) gained.  slices_per_tile scales the tile's time and its traffic alike, so a pair of matrices and either
    // of them alone get the same split (same bits).
    const int RB = K >> 7;
    const double tile_us = 2.0 * 256 * 128 * K / 4.7e6;
    int best = 1; double best_cost = 1.0;
    for (int p = 2; p <= 8; p *= 2) {
        if (RB % p != 0 || RB / p < 2) continue;
        const double rounds = (double) ((rem * p + ncu - 1) / ncu);
        const double traffic_us = (double) rem * p * (256 * 128 * 4) * 2.0 / 6.0e6;
        const double cost = rounds / p * (1.0 + 3.0 * p / (2.0 * RB)) + traffic_us / tile_us;
        if (cost < best_cost - 0.1) { best = p; best_cost = cost; }
    }
    if (best == 1) return padded;
    // the logical block after which `full` valid tiles have been dispatched
    int valid = 0, bs = 0;
    for (; bs < padded && valid < full; ++bs) {
        int mt_, nt_;
        if (gemm_tile_of(bs, mtiles, ntiles, t->mr, &mt_, &nt_)) ++valid;
    }
    const int nl_tail = padded - bs;
    float* ws = nullptr;
    if (exl_gemm_workspace(device, (size_t) nl_tail * best * slices_per_tile * 256 * 128, &ws) != 0) return padded;   // no room: unsplit
    t->b_split = bs; t->parts = best; t->ws = ws;
    *n_tail = nl_tail;
    return bs + nl_tail * best;
}

bool q4_same_map(const Q4Matrix* a, const Q4Matrix* b)
{
    if (!a->x_map || !b->x_map) return !a->x_map && !b->x_map;
    if (a->height != b->height || a->xmap_hash != b->xmap_hash) return false;
    // equal hashes are a hint, not a proof: the entries themselves decide (host copies kept by make_q4; ~K * 4 bytes per call)
    return a->xmap_host.size() == b->xmap_host.size() &&
           memcmp(a->xmap_host.data(), b->xmap_host.data(), a->xmap_host.size() * sizeof(uint32_t)) == 0;
}

// Runs the prologue of a fused prompt-pass launch (common.h: PromptPrologue) and points *x at the kernel's input.
static int prompt_prologue(const PromptPrologue& pro, const Q4Matrix* w, const f16** x, int rows, hipStream_t s)
{
    if (!pro.norm_w && !w->x_map) return 0;
    EXL_REQUIRE(pro.tmp && pro.tmp_numel >= (size_t) rows * w->height, EXL_E_TOO_SMALL,
                "q4 gemm: temp_state buffer is too small for the normalised / gathered activations (%zu < %zu halves)",
                pro.tmp_numel, (size_t) rows * w->height);
    EXL_REQUIRE(*x != pro.tmp, EXL_E_INVALID, "q4 gemm: the activations may not live in the temp_state buffer");
    if (pro.norm_w) EXL_TRY(launch_rms_norm_gather(*x, pro.norm_w, pro.tmp, w->x_map, pro.eps, rows, w->height, s));
    else            EXL_TRY(launch_column_remap(*x, pro.tmp, rows, w->height, w->x_map, s));
    *x = pro.tmp;
    return 0;
}

static int launch_gemm_t16w(const Q4Matrix* w, const f16* xin, int rows, f16* out, int no_zero, int gshift, hipStream_t s)
{
    const int K = w->height, N = w->width;
    const int mtiles = (rows + 255) / 256;
    const int ntiles = (N + GT_BN - 1) / GT_BN;
    GemmTail tail;
    int n_tail = 0;
    const int grid = plan_gemm_tail(w->device, mtiles, ntiles, K, 1, &tail, &n_tail);
    const size_t smem = 3 * (size_t) 256 * 128 + 2 * GT_BTILE_BYTES;
    static bool big[EXL_MAX_DEVICES] = {};
    EXL_TRY(exl_lds_opt_in((const void*) q4_gemm_t16w_kernel<0>, big));
    GwQkv none = {};
    hipLaunchKernelGGL(q4_gemm_t16w_kernel<0>, dim3(grid), dim3((GW_CONS + GW_PROD) * 64), smem, s, xin, (const uint4*) w->qweight, w->qzeros,
                       w->scales, out, rows, K, N, gshift, no_zero, mtiles, ntiles, none, tail);
    EXL_LAUNCH_CHECK();
    if (n_tail) {
        hipLaunchKernelGGL(q4_gemm_tail_reduce_kernel<0>, dim3(n_tail), dim3(256), 0, s, out, (f16*) nullptr, rows, N, no_zero, mtiles, ntiles, tail);
        EXL_LAUNCH_CHECK();
    }
    return 0;
}

// q = rope(x @ Wq), cache_k <- rope(x @ Wk), cache_v <- x @ Wv in one launch.  Returns 1 (nothing launched) when the
// matrices / shapes are outside what the fused kernel covers; the caller then runs the separate ops.
int launch_q4_qkv_rope_cache(const Q4Matrix* wq, const Q4Matrix* wk, const Q4Matrix* wv, const f16* x, int rows, f16* q_out,
                             const f16* sin, const f16* cos, f16* kc, f16* vc, int q_len, int heads, int kv_heads, int head_dim,
                             int past_len, int max_seq, const PromptPrologue& pro, hipStream_t s)
{
    static const bool off = getenv("EXL_GEMM_NO_QKV_FUSION") != nullptr;      // A/B switch
    const Q4Matrix* m[3] = {wq, wk, wv};
    const int K = wq->height;
    if (off || rows <= 512 || head_dim != 128 || q_len <= 0 || rows % q_len != 0) return 1;
    if (wq->width != heads * 128 || wk->width != kv_heads * 128 || wv->width != kv_heads * 128) return 1;
    if ((uint64_t) rows * (uint64_t) K >= (1ull << 31)) return 1;
    int gshift = -1;
    for (int i = 0; i < 3; ++i) {
        if (m[i]->layout != EXL_LAYOUT_T16 || m[i]->height != K || m[i]->groupsize != wq->groupsize || !q4_same_map(m[i], wq)) return 1;
        if ((uint64_t) K * (uint64_t) m[i]->width >= (1ull << 32)) return 1;
    }
    if ((wq->groupsize & (wq->groupsize - 1)) == 0) { gshift = 0; while ((1 << gshift) < wq->groupsize) ++gshift; }
    if (gshift < 5) return 1;
    EXL_TRY(prompt_prologue(pro, wq, &x, rows, s));                           // RMSNorm and / or the (shared) act-order gather
    GwQkv e = {};
    int tiles = 0;
    for (int i = 0; i < 3; ++i) {
        e.qw[i] = (const uint4*) m[i]->qweight; e.qz[i] = m[i]->qzeros; e.sc[i] = m[i]->scales; e.n[i] = m[i]->width;
        tiles += m[i]->width / 128;
        e.nt_end[i] = tiles;
    }
    e.q_out = q_out; e.sin = sin; e.cos = cos; e.kc = kc; e.vc = vc;
    e.q_len = q_len; e.past_len = past_len; e.max_seq = max_seq; e.kv_heads = kv_heads;
    const int mtiles = (rows + 255) / 256;
    const int grid = 8 * ((tiles + 7) / 8) * mtiles;
    GemmTail qkv_order;
    int no_tail = 0;
    (void) plan_gemm_tail(-1, mtiles, tiles, 128, 1, &qkv_order, &no_tail);        // (K = 128: no part count divides it -> the tile order only, whole tiles)
    const size_t smem = 3 * (size_t) 256 * 128 + 2 * GT_BTILE_BYTES;
    static bool big[EXL_MAX_DEVICES] = {};
    EXL_TRY(exl_lds_opt_in((const void*) q4_gemm_t16w_kernel<1>, big));
    hipLaunchKernelGGL(q4_gemm_t16w_kernel<1>, dim3(grid), dim3((GW_CONS + GW_PROD) * 64), smem, s, x, (const uint4*) nullptr, (const uint32_t*) nullptr,
                       (const f16*) nullptr, (f16*) nullptr, rows, K, 0, gshift, 0, mtiles, tiles, e, qkv_order);
    EXL_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// Dual-weight kernel for the MLP up-projection of the prompt pass: out = silu(x @ Wgate) * (x @ Wup) (or both products) in ONE
// kernel.  A block computes the SAME 256 x 128 output tile of both matrices from one activation tile: the activation DMA and
// fragment reads are shared (24 instead of 32 LDS fragment reads per 64 MFMAs) and the gate / up products never travel through
// HBM.  It carries the software pipeline of q4_gemm_t16m_kernel.  SQ counters of its un-pipelined predecessor (removed in round
// 3; profiles/r02_pmc_dual_gemm.txt): matrix pipes busy 48 % of a launch, a wave 37 % of its life in s_waitcnt / s_barrier
// and only 4 % of it waiting for LDS -- both waves of a SIMD walk the same phases (fragment reads -> 64 MFMAs -> wait ->
// dequantise + store -> barrier) at the same time, so the matrix pipe idles through every wait, store and barrier.
// Here a K step is cut into 8 groups of 8 MFMAs (one weight fragment pair x 4 activation fragments x 2 matrices); the
// fragments of group g + 2 are requested before group g is issued, the activation DMA / weight loads of tile t + 2, the
// wait for tile t + 1 and its dequantisation + LDS stores sit BETWEEN the groups, and the LAST group of tile t is held
// back across the barrier: it runs while the first fragments of tile t + 1 are on their way.  Registers: 128 accumulators
// + 56 fragment registers (two activation sets, three weight pairs) -- the full P / Q double set of the single-matrix
// kernel (96) does not fit next to 128 accumulators.  3-slot activation ring + 2 x 2 weight tiles = 160 KiB of LDS.
// (Bit-identical to its un-pipelined predecessor q4_gemm_t16d_kernel, removed in round 3.)
// ---------------------------------------------------------------------------------------------------------------
template <bool SILU>
__global__ __launch_bounds__(512) void q4_gemm_t16d2_kernel(const f16* __restrict__ x, const uint4* __restrict__ qw1,
                                                            const uint32_t* __restrict__ qz1, const f16* __restrict__ sc1,
                                                            const uint4* __restrict__ qw2, const uint32_t* __restrict__ qz2,
                                                            const f16* __restrict__ sc2, f16* __restrict__ out1,
                                                            f16* __restrict__ out2, int M, int K, int N, int gshift,
                                                            int groupsize, int mtiles, int ntiles, const GemmTail tail)
{
    constexpr int TBM = 256;
    constexpr int A_BYTES = TBM * 128;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];       // [3][A] then [2][B1 | B2]
    unsigned char* const ldsB = lds + 3 * A_BYTES;

    int b = blockIdx.x;
    int kz = 0, nparts = 1, tseq = 0;                                  // tail blocks (plan_gemm_tail): part kz of the K range of logical block b_split + tseq
    if (b >= tail.b_split) {
        const int j = b - tail.b_split;
        tseq = j / tail.parts;
        kz = j - tseq * tail.parts;
        b = tail.b_split + tseq;
        nparts = tail.parts;
    }
    int mt, nt;
    if (!gemm_tile_of(b, mtiles, ntiles, tail.mr, &mt, &nt)) return;
    const int m0 = mt * TBM;
    const int n0 = nt * GT_BN;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int RB = K >> 7;
    const int nk = K / GT_BK / nparts;                                // K steps of this block: even, >= 2
    const int it0 = kz * nk;                                          // its first K tile

    // Every address is (uniform base, advanced per K step by scalar arithmetic) + (fixed 32-bit byte offset per lane): saddr-form
    // loads, no 64-bit address registers (the register file is full: 128 accumulators + 56 fragment registers).
    uint32_t a_voff[4];                                                // byte offsets from x (M * K * 2 < 2^32 checked on the host)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = wave * 4 + i;
        const int row = c * 8 + (lane >> 3);
        const int slot = lane & 7;
        const int grow = min(m0 + row, M - 1);
        a_voff[i] = ((uint32_t) grow * (uint32_t) K + ((slot ^ (row & 7)) << 3)) * 2u;
    }
    const uint32_t lds_a0 = (uint32_t) (uintptr_t) (__attribute__((address_space(3))) unsigned char*) lds + (uint32_t) (wave * 4 * 1024);
    auto stage_a2 = [&](int pair, int slot3, int k0) {                 // pieces 2 * pair and 2 * pair + 1; M0 carries the LDS address
        const f16* xk = x + k0;                                        // uniform
        const uint32_t l = lds_a0 + (uint32_t) slot3 * A_BYTES + (uint32_t) pair * 2048;
        asm volatile("s_mov_b32 m0, %[l]\n\t"
                     "global_load_lds_dwordx4 %[v0], %[sb]\n\t"
                     "s_add_u32 m0, m0, 0x400\n\t"
                     "global_load_lds_dwordx4 %[v1], %[sb]"
                     :: [l] "s"(l), [sb] "s"(xk), [v0] "v"(a_voff[2 * pair]), [v1] "v"(a_voff[2 * pair + 1]) : "memory", "scc");
    };

    // loader part: waves 0-3 fetch the packed weights of matrix 1, waves 4-7 those of matrix 2 -- one whole T16 piece quarter (4 words)
    // per thread and K step: 3 vector-memory instructions per wave and step instead of 6 (a wave that queues on the memory pipe
    // cannot issue the MFMAs behind it)
    const int pid = tid & 255, mat = wave >> 2;
    const int b_tile = pid >> 5;
    const int b_rs = (pid >> 4) & 1;
    const int b_col = pid & 15;
    const int b_nloc = b_tile * 16 + b_col;
    const int b_n = min(n0 + b_nloc, N - 1);
    // packed words of K step `it`: piece (n, rb = it / 2), sub-row (it & 1) * 2 + b_rs, words 2 ph, 2 ph + 1 -> uniform it * 512 bytes + lane part
    const uint32_t w_voff = (uint32_t) (((size_t) (b_n >> 4) * RB * 64 + (b_n & 15)) * 16 + (size_t) b_rs * 256);
    // group of k = it * 64 + b_rs * 32 (groupsize a power of two >= 32): uniform (it * 64) >> gshift, + b_rs for groupsize 32
    const int zs_lane_grp = gshift == 5 ? b_rs : 0;
    const uint32_t z_voff = (uint32_t) ((zs_lane_grp * (N >> 3) + (b_n >> 3)) * 4);
    const uint32_t s_voff = (uint32_t) ((zs_lane_grp * N + b_n) * 2);
    const int b_zsh = (b_n & 7) * 4;
    const uint32_t magic = t16_magic();
    const uint32_t b_lds = (uint32_t) (uintptr_t) (__attribute__((address_space(3))) unsigned char*) ldsB;
    uint32_t b_dst[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) b_dst[j] = b_lds + (uint32_t) (mat * GT_BTILE_BYTES) + (uint32_t) gt_off(b_nloc, b_rs * 4 + j);

    struct BRegs { u32x4 w4; uint32_t zw, sc; };
    const unsigned char* const qw_m = (const unsigned char*) (mat ? qw2 : qw1);       // wave-uniform
    const uint32_t* const qz_m = mat ? qz2 : qz1;
    const f16* const sc_m = mat ? sc2 : sc1;
    auto issue_w = [&](int it, BRegs& r) {
        const size_t wo = (size_t) it * 512;                                                              // uniform
        asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(r.w4) : "v"(w_voff), "s"(qw_m + wo) : "memory");
    };
    auto issue_zs = [&](int it, BRegs& r) {
        const int grp = (it * GT_BK) >> gshift;
        const size_t zo = (size_t) grp * (N >> 3), so = (size_t) grp * N;
        asm volatile("global_load_dword %0, %1, %2" : "=v"(r.zw) : "v"(z_voff), "s"(qz_m + zo) : "memory");
        asm volatile("global_load_ushort %0, %1, %2" : "=v"(r.sc) : "v"(s_voff), "s"(sc_m + so) : "memory");
    };
#define GD2_WAIT(NSTR, r) asm volatile("s_waitcnt vmcnt(" NSTR ")" : "+v"(r.w4), "+v"(r.zw), "+v"(r.sc) :: "memory")
    // dequantise word j of the landed register set and store its 8 halves (one 16-byte chunk of this wave's [n][k] tile)
    auto store_word = [&](int slot2, const BRegs& r, int j) {
        const int z = (int) ((r.zw >> b_zsh) & 0xFu) + 1;
        const f16 za = (f16) (float) (-(1024 + z));
        const f16 zb = (f16) (float) (-(64 + z));
        const f16 bsc = __builtin_bit_cast(f16, (uint16_t) (r.sc & 0xFFFFu));
        const f16x2 zc0 = {za, za}, zc1 = {zb, zb}, s2 = {bsc, bsc};
        const f16x8 d = t16_dequant_exact(r.w4[j], magic, zc0, zc1);
        const uint4 u = __builtin_bit_cast(uint4, d);
        const u32x4 ov = {__builtin_bit_cast(uint32_t, as_h2(u.x) * s2), __builtin_bit_cast(uint32_t, as_h2(u.y) * s2),
                          __builtin_bit_cast(uint32_t, as_h2(u.z) * s2), __builtin_bit_cast(uint32_t, as_h2(u.w) * s2)};
        asm volatile("ds_write_b128 %0, %1" :: "v"(b_dst[j] + (uint32_t) (slot2 * 2 * GT_BTILE_BYTES)), "v"(ov) : "memory");
    };
    auto block_barrier = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    };

    f32x4 acc1[4][4], acc2[4][4];                                        // [n-tile][m-tile]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) { acc1[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc2[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f}; }

    const int fr = lane & 15, fk = lane >> 4;
    // fragment byte offsets: row tiles are 16 rows = 2048 bytes apart and share r & 7, so ONE swizzled base per operand + immediates;
    // kk = 1 flips bit 6 of the swizzled chunk
    const int fx0 = gt_off(wm * 64 + fr, fk), fw0 = gt_off(wn * 64 + fr, fk);
    struct Pair { f16x8 w1, w2; };
    f16x8 X0[4], X1[4];
    Pair Q0, Q1, Q2, Q3, Q4;                                              // Q0-Q2 rotate through groups 0-5, Q3 / Q4 carry groups 6 / 7 across the barrier
    const f16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 4; ++i) { X0[i] = zero8; X1[i] = zero8; }
    Q3.w1 = zero8; Q3.w2 = zero8; Q4 = Q3; Q0 = Q3; Q1 = Q3; Q2 = Q3;
#define GD2_RDX(X, at, kk) do { _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) X[i_] = *(const f16x8*) ((at) + (fx0 ^ ((kk) << 6)) + i_ * 2048); } while (0)
#define GD2_RDW(P, b1, in, kk) do { P.w1 = *(const f16x8*) ((b1) + (fw0 ^ ((kk) << 6)) + (in) * 2048); P.w2 = *(const f16x8*) ((b1) + GT_BTILE_BYTES + (fw0 ^ ((kk) << 6)) + (in) * 2048); } while (0)
#define GD2_MF(P, X, in) do { _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) {                                         \
        acc1[in][i_] = __builtin_amdgcn_mfma_f32_16x16x32_f16(P.w1, X[i_], acc1[in][i_], 0, 0, 0);                       \
        acc2[in][i_] = __builtin_amdgcn_mfma_f32_16x16x32_f16(P.w2, X[i_], acc2[in][i_], 0, 0, 0); } } while (0)
#define GD2_SB() __builtin_amdgcn_sched_barrier(0)

    // ---- prologue: batches 0 and 1 in flight, B(0) dequantised --------------------------------------------------------
    BRegs rX, rY;
    stage_a2(0, 0, it0 * GT_BK); stage_a2(1, 0, it0 * GT_BK);
    issue_w(it0, rX); issue_zs(it0, rX);
    stage_a2(0, 1, (it0 + 1) * GT_BK); stage_a2(1, 1, (it0 + 1) * GT_BK);    // nk >= 2 always
    issue_w(it0 + 1, rY); issue_zs(it0 + 1, rY);
    GD2_WAIT("7", rX);                                                    // batch 0 landed (batch 1 may still fly)
#pragma unroll
    for (int j = 0; j < 4; ++j) store_word(0, rX, j);
    block_barrier();

    int a_slot = 0;
    auto ring = [](int s, int d) { const int v = s + d; return v >= 3 ? v - 3 : v; };
#ifdef EXL_GEMM_PROBE
    unsigned long long p_wait = 0, p_store = 0, p_bar = 0;                // here: first half (incl. the vmcnt wait) / second half / barrier
    const unsigned long long p_t0 = __builtin_readcyclecounter();
#endif
    // one K step; rI receives the packed weights of tile t + 2, rW holds tile t + 1 (landing), bcur = LDS slot of B(t)
    auto step = [&](int t, BRegs& rI, BRegs& rW, int bcur) {
        const int tf = it0 + min(t + 2, nk - 1);                          // the last two steps re-fetch the last tile (never read)
        const int adma = ring(a_slot, 2);
        const unsigned char* at = lds + (size_t) a_slot * A_BYTES;
        const unsigned char* b1 = ldsB + (size_t) (bcur * 2) * GT_BTILE_BYTES;
        GP_CLK(c0);
        // group g = (kk = g >> 2, in = g & 3); fragments are requested TWO groups ahead of their use (a ds_read_b128 under this load
        // takes ~250 cycles, a group of 8 MFMAs 128: with one group of lead a wave stalled ~120 cycles per group -- probe: 2055
        // cycles for its 64 MFMAs); groups 6 and 7 of the previous tile open the step while this tile's first fragments fly
        GD2_RDX(X0, at, 0); GD2_RDW(Q0, b1, 0, 0); GD2_RDW(Q1, b1, 1, 0);   GD2_SB();
        GD2_MF(Q3, X1, 2);  stage_a2(0, adma, tf * GT_BK);                  GD2_SB();   // group 6 of tile t - 1
        GD2_RDW(Q2, b1, 2, 0);                                              GD2_SB();
        GD2_MF(Q4, X1, 3);  stage_a2(1, adma, tf * GT_BK);                  GD2_SB();   // group 7 of tile t - 1
        GD2_RDX(X1, at, 1);                                                 GD2_SB();
        GD2_MF(Q0, X0, 0);  issue_w(tf, rI);                                GD2_SB();
        GD2_RDW(Q0, b1, 3, 0);                                              GD2_SB();
        GD2_MF(Q1, X0, 1);  issue_zs(tf, rI);                               GD2_SB();
        GD2_RDW(Q1, b1, 0, 1);                                              GD2_SB();
        GD2_MF(Q2, X0, 2);  GD2_WAIT("7", rW);                              GD2_SB();
        GP_CLK(c1);
        GD2_RDW(Q2, b1, 1, 1);                                              GD2_SB();
        GD2_MF(Q0, X0, 3);  store_word(bcur ^ 1, rW, 0);                    GD2_SB();
        GD2_RDW(Q3, b1, 2, 1);                                              GD2_SB();
        GD2_MF(Q1, X1, 0);  store_word(bcur ^ 1, rW, 1);                    GD2_SB();
        GD2_RDW(Q4, b1, 3, 1);                                              GD2_SB();
        GD2_MF(Q2, X1, 1);  store_word(bcur ^ 1, rW, 2); store_word(bcur ^ 1, rW, 3);   GD2_SB();
        GP_CLK(c2);
        block_barrier();
        GP_CLK(c3);
        GP_ACC(p_wait, c0, c1); GP_ACC(p_store, c1, c2); GP_ACC(p_bar, c2, c3);
        a_slot = ring(a_slot, 1);
    };
    for (int t = 0; t < nk; t += 2) {
        step(t, rX, rY, 0);
        step(t + 1, rY, rX, 1);
    }
    GD2_MF(Q3, X1, 2); GD2_MF(Q4, X1, 3);                                 // the held groups of the last tile
    GD2_WAIT("0", rX); GD2_WAIT("0", rY);                                 // the two redundant fetches (registers tied: see q4_gemm_t16m_kernel)
#ifdef EXL_GEMM_PROBE
    if (lane == 0 && b < 1024) {
        unsigned long long* pp = g_gemm_probe + ((size_t) b * 8 + wave) * 4;
        pp[0] = __builtin_readcyclecounter() - p_t0; pp[1] = p_wait; pp[2] = p_store; pp[3] = p_bar;
    }
#endif
#undef GD2_WAIT
#undef GD2_RDX
#undef GD2_RDW
#undef GD2_MF
#undef GD2_SB

    // ---- epilogue ------------------------------------------------------------------------------------------------------
    if (nparts > 1) {                                                    // fp32 slices (gate tile, then up tile) of this part of K: q4_gemm_tail_reduce_kernel<1>
        float* wt = tail.ws + (size_t) (tseq * nparts + kz) * (2 * 256 * 128);
#pragma unroll
        for (int im = 0; im < 4; ++im)
#pragma unroll
            for (int in = 0; in < 4; ++in) {
                float* p = wt + ((wm * 4 + im) * 16 + fr) * 128 + (wn * 4 + in) * 16 + fk * 4;
                *(f32x4*) p = acc1[in][im];
                *(f32x4*) (p + 256 * 128) = acc2[in][im];
            }
        return;
    }
#pragma unroll
    for (int im = 0; im < 4; ++im) {
        const int row = m0 + (wm * 4 + im) * 16 + fr;
        if (row < M) {
#pragma unroll
            for (int in = 0; in < 4; ++in) {
                const int n = n0 + (wn * 4 + in) * 16 + fk * 4;
                if (n < N) {
                    const size_t o = (size_t) row * N + n;
                    f16x4 g, u;
#pragma unroll
                    for (int j = 0; j < 4; ++j) { g[j] = (f16) acc1[in][im][j]; u[j] = (f16) acc2[in][im][j]; }
                    if constexpr (SILU) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) {                 // exactly elementwise.hip: silu_mul_h on the fp16 values
                            const f16 e = (f16) __expf((float) (f16) (-g[j]));
                            const f16 sm = (f16) 1.0f + e;
                            const f16 rc = (f16) (1.0f / (float) sm);
                            const f16 v = g[j] * rc;
                            g[j] = v * u[j];
                        }
                        *(f16x4*) (out1 + o) = g;
                    } else {
                        *(f16x4*) (out1 + o) = g;
                        *(f16x4*) (out2 + o) = u;
                    }
                }
            }
        }
    }
}

// out1 = silu(x @ W1) * (x @ W2) (silu != 0) or out1 = x @ W1, out2 = x @ W2.  Returns 1 when the pair is not eligible
// for the dual kernel (the caller then runs the two products separately), 0 on success, otherwise an error.
int launch_q4_gemm_dual(const Q4Matrix* w1, const Q4Matrix* w2, const f16* x, int rows, f16* out1, f16* out2, int silu,
                        const PromptPrologue& pro, hipStream_t s)
{
    static const bool off = getenv("EXL_GEMM_NO_DUAL") != nullptr;
    if (off || rows <= 512 || w1->layout != EXL_LAYOUT_T16 || w2->layout != EXL_LAYOUT_T16 || !q4_same_map(w1, w2) ||
        w1->height != w2->height || w1->width != w2->width || w1->groupsize != w2->groupsize || w1->device != w2->device ||
        (size_t) rows * w1->height >= ((size_t) 1 << 32))
        return 1;
    const int K = w1->height, N = w1->width;
    int gshift = -1;
    if ((w1->groupsize & (w1->groupsize - 1)) == 0) { gshift = 0; while ((1 << gshift) < w1->groupsize) ++gshift; }
    const int mtiles = (rows + 255) / 256;
    const int ntiles = (N + GT_BN - 1) / GT_BN;
    GemmTail tail = {1 << 30, 1, nullptr, mtiles};
    int n_tail = 0;
    const int grid = plan_gemm_tail(w1->device, mtiles, ntiles, K, 2, &tail, &n_tail);                 // the same split as either product alone: same bits
#define GD_ARGS x, (const uint4*) w1->qweight, w1->qzeros, w1->scales, (const uint4*) w2->qweight, w2->qzeros, w2->scales, out1, out2, \
                rows, K, N, gshift, w1->groupsize, mtiles, ntiles, tail
    // power-of-two groups (one shift), 32-bit byte offsets into the weight / scale / activation arrays; anything else runs as the
    // separate products
    const bool pipelined_ok = gshift >= 5 && (uint64_t) K * (uint64_t) N < (1ull << 32) && (uint64_t) rows * (uint64_t) K < (1ull << 31);
    if (!pipelined_ok) return 1;
    EXL_TRY(prompt_prologue(pro, w1, &x, rows, s));                           // RMSNorm and / or the (shared) act-order gather
    {
        const size_t smem = 3 * (size_t) 256 * 128 + 4 * GT_BTILE_BYTES;              // 160 KiB: the whole LDS of a CU
        static bool big_silu[EXL_MAX_DEVICES] = {}, big_pair[EXL_MAX_DEVICES] = {};
        EXL_TRY(exl_lds_opt_in((const void*) q4_gemm_t16d2_kernel<true>, big_silu));
        EXL_TRY(exl_lds_opt_in((const void*) q4_gemm_t16d2_kernel<false>, big_pair));
        if (silu) hipLaunchKernelGGL(q4_gemm_t16d2_kernel<true>, dim3(grid), dim3(512), smem, s, GD_ARGS);
        else      hipLaunchKernelGGL(q4_gemm_t16d2_kernel<false>, dim3(grid), dim3(512), smem, s, GD_ARGS);
    }
#undef GD_ARGS
    EXL_LAUNCH_CHECK();
    if (n_tail) {
        if (silu) hipLaunchKernelGGL(q4_gemm_tail_reduce_kernel<1>, dim3(n_tail), dim3(256), 0, s, out1, (f16*) nullptr, rows, N, 0, mtiles, ntiles, tail);
        else      hipLaunchKernelGGL(q4_gemm_tail_reduce_kernel<2>, dim3(n_tail), dim3(256), 0, s, out1, out2, rows, N, 0, mtiles, ntiles, tail);
        EXL_LAUNCH_CHECK();
    }
    return 0;
}

template <int WAVES_M, int WAVES_N, int TM, int TN, int KS = 1>
static int launch_gemm_t16m(const Q4Matrix* w, const f16* xin, int rows, f16* out, int no_zero, int gshift, hipStream_t s)
{
    constexpr int TBM = WAVES_M * TM * 16;
    const int K = w->height, N = w->width;
    const int mtiles = (rows + TBM - 1) / TBM;
    const int ntiles = (N + GT_BN - 1) / GT_BN;
    const int grid = 8 * ((ntiles + 7) / 8) * mtiles * KS;
    const size_t smem = 3 * (size_t) TBM * 128 + 2 * GT_BTILE_BYTES;
    auto kfn = q4_gemm_t16m_kernel<WAVES_M, WAVES_N, TM, TN, KS>;
    static bool big[EXL_MAX_DEVICES] = {};
    if (smem > 64 * 1024) EXL_TRY(exl_lds_opt_in((const void*) kfn, big));
    float* ws = nullptr;
    if constexpr (KS > 1) { if (exl_gemm_workspace(w->device, (size_t) KS * rows * N, &ws) != 0) EXL_FAIL(EXL_E_TOO_SMALL, "q4 GEMM: no room for the split-K workspace (%zu floats)", (size_t) KS * rows * N); }
    hipLaunchKernelGGL(kfn, dim3(grid), dim3(WAVES_M * WAVES_N * 64), smem, s, xin, (const uint4*) w->qweight, w->qzeros,
                       w->scales, out, rows, K, N, gshift, w->groupsize, no_zero, mtiles, ntiles, ws);
    EXL_LAUNCH_CHECK();
    if constexpr (KS > 1) {
        const size_t cells4 = (size_t) rows * N / 4;
        hipLaunchKernelGGL(q4_gemm_splitk_reduce_kernel, dim3((unsigned) ((cells4 + 255) / 256)), dim3(256), 0, s, ws, out, cells4,
                           (size_t) rows * N, KS, no_zero);
        EXL_LAUNCH_CHECK();
    }
    return 0;
}

int launch_gemm_t16s(const Q4Matrix* w, const f16* x, int rows, f16* out, int no_zero, hipStream_t s);

int launch_q4_gemm(const Q4Matrix* w, const f16* x, int rows, f16* out, int no_zero, f16* remap_tmp,
                   size_t remap_tmp_numel, hipStream_t s)
{
    if (rows <= 0) return 0;
    const int K = w->height, N = w->width;
    EXL_REQUIRE(N % 2 == 0 && K % 16 == 0, EXL_E_UNSUPPORTED, "q4 gemm: need N %% 2 == 0 and K %% 16 == 0 (K=%d N=%d)", K, N);
    EXL_REQUIRE(w->groupsize % 16 == 0, EXL_E_UNSUPPORTED, "q4 gemm: groupsize (%d) must be a multiple of 16", w->groupsize);
    const f16* xin = x;
    if (w->x_map) {
        EXL_REQUIRE(remap_tmp && remap_tmp_numel >= (size_t) rows * K, EXL_E_TOO_SMALL,
                    "q4 gemm: temp_state buffer is too small for the act-order gather (%zu < %zu halves)",
                    remap_tmp_numel, (size_t) rows * K);
        EXL_TRY(launch_column_remap(x, remap_tmp, rows, K, w->x_map, s));
        xin = remap_tmp;
    }
    // short prompts: the decode-shaped kernel, 32 rows x (16 .. 64) columns per block, waves split K (q4_gemm_skinny.hip)
    static const int skinny_max = getenv("EXL_GEMM_SKINNY_MAX") ? atoi(getenv("EXL_GEMM_SKINNY_MAX")) : 256;
    if (w->layout == EXL_LAYOUT_T16 && rows <= skinny_max) {
        const int r = launch_gemm_t16s(w, xin, rows, out, no_zero, s);
        if (r != 1) return r;
    }
    int gshift = -1;
    if ((w->groupsize & (w->groupsize - 1)) == 0) { gshift = 0; while ((1 << gshift) < w->groupsize) ++gshift; }
    const int mtiles = (rows + BM - 1) / BM;
    const int ntiles = (N + BN - 1) / BN;
    const int grid = 8 * ((ntiles + 7) / 8) * mtiles;
    if (w->layout == EXL_LAYOUT_T16 && (uint64_t) rows * (uint64_t) K < (1ull << 31)) {    // 32-bit activation byte offsets
        // 257 .. 512 rows: the 128 x 128 tile (about 10 % faster than the 256-row kernels at 300 - 384 rows).  Round 2 found it
        // corrupting accumulators at 400 rows x 11008 columns on cold launches (in-flight "redundant" fetches landing in registers the
        // compiler had reused; fixed by tying the final wait to those registers, profiles/HISTORY.md 9.5) and routed around it; round 3
        // validated the fix with the cold-launch stress test (tests/test_cold_launch_gpu.py: every hand-counted kernel as the first
        // GEMM of a fresh process on poisoned memory) and scripts/isa_lint.py.  EXL_GEMM_NO_TILE128=1 restores the detour.
        static const bool tile128 = getenv("EXL_GEMM_NO_TILE128") == nullptr;
        const int big_rows = tile128 ? 512 : 256;
        const bool spec = rows > big_rows && gshift >= 5 && (uint64_t) K * (uint64_t) N < (1ull << 32);   // loader waves: power-of-two groups, 32-bit weight offsets
        if (spec) return launch_gemm_t16w(w, xin, rows, out, no_zero, gshift, s);                 // 256 x 128, 8 MFMA waves + 4 loader waves
        if (rows > big_rows) return launch_gemm_t16m<4, 2, 4, 4>(w, xin, rows, out, no_zero, gshift, s);  // 256 x 128, 8 waves
        // K cut in two for 257 .. 512 rows: 2 x the blocks at half the length (the 128-row tile alone leaves 344 blocks of a 7B
        // down projection on 256 CUs), fp32 slices in the workspace + a reduce kernel.  Round 3, 7B layer at 300 / 384 / 512 rows:
        // 0.397 / 0.405 / 0.423 ms without, 0.318 / 0.327 / 0.355 ms with (profiles/r03_tile128_validation.txt); validated by the
        // GEMM op tests and the cold-launch case t16m128k.  EXL_GEMM_NO_SPLITK=1 is the A/B switch.
        static const bool splitk = getenv("EXL_GEMM_NO_SPLITK") == nullptr;
        if (splitk && rows > 256 && K % 256 == 0 && N % 4 == 0) {
            // the slices live in the per-device workspace, grown on demand: when it cannot be had (a device packed to the weights),
            // the unsplit kernel below computes the same product (its error message stays in exl_last_error for the curious)
            float* probe = nullptr;
            if (exl_gemm_workspace(w->device, (size_t) 2 * rows * N, &probe) == 0)
                return launch_gemm_t16m<2, 2, 4, 4, 2>(w, xin, rows, out, no_zero, gshift, s);
        }
        return launch_gemm_t16m<2, 2, 4, 4>(w, xin, rows, out, no_zero, gshift, s);                // 128 x 128, 4 waves
    }
    if (w->layout == EXL_LAYOUT_T16)
        hipLaunchKernelGGL(q4_gemm_kernel<true>, dim3(grid), dim3(256), 0, s, xin, w->qweight, w->qzeros, w->scales, out, rows, K,
                           N, gshift, w->groupsize, no_zero, mtiles, ntiles);
    else
        hipLaunchKernelGGL(q4_gemm_kernel<false>, dim3(grid), dim3(256), 0, s, xin, w->qweight, w->qzeros, w->scales, out, rows, K,
                           N, gshift, w->groupsize, no_zero, mtiles, ntiles);
    EXL_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------------
// Plain fp16 GEMM out (+)= x[M,K] @ w[K,N] for the LoRA side path (reference: half_matmul.cu).  Small and
// simple on purpose (64x64 tile, MFMA 32x32x16, fp32 accumulate): this op is not on the no-LoRA hot path.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void half_gemm_kernel(const f16* __restrict__ x, const f16* __restrict__ w,
                                                        f16* __restrict__ out, int M, int K, int N, int no_zero)
{
    __shared__ f16 As[64][16 + 8];      // [m][k]
    __shared__ f16 Bs[64][16 + 8];      // [n][k]  (transposed on the way in)
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int wm = wave >> 1, wn = wave & 1;
    const int g = lane >> 5, c = lane & 31;
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;

    for (int k0 = 0; k0 < K; k0 += 16) {
        for (int i = tid; i < 64 * 16; i += 256) {
            const int r = i >> 4, kk = i & 15;
            const int row = m0 + r, k = k0 + kk;
            As[r][kk] = (row < M && k < K) ? x[(size_t) row * K + k] : (f16) 0.f;
            const int kb = i >> 6, nn = i & 63;            // coalesced along n
            const int col = n0 + nn, k2 = k0 + kb;
            Bs[nn][kb] = (col < N && k2 < K) ? w[(size_t) k2 * N + col] : (f16) 0.f;
        }
        __syncthreads();
        const f16x8 af = *(const f16x8*) &As[wm * 32 + c][g * 8];
        const f16x8 bf = *(const f16x8*) &Bs[wn * 32 + c][g * 8];
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, bf, acc, 0, 0, 0);
        __syncthreads();
    }
    const int col = n0 + wn * 32 + c;
    if (col >= N) return;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * g;
        if (row < M) {
            float v = acc[r];
            if (no_zero) v += (float) out[(size_t) row * N + col];
            out[(size_t) row * N + col] = (f16) v;
        }
    }
}

// Same 64 x 64 tile, K step 64 with 16-byte global loads (K % 8 == 0, N % 8 == 0, 16-byte aligned pointers): the LoRA
// down-projection x[M, hidden] @ A[hidden, r] walks a long K with a handful of blocks, so the K loop is what costs; the
// scalar kernel above (16 k per step, 2-byte loads) stays for odd shapes.  B is transposed on its way into LDS.
__global__ __launch_bounds__(256) void half_gemm64_kernel(const f16* __restrict__ x, const f16* __restrict__ w,
                                                          f16* __restrict__ out, int M, int K, int N, int no_zero)
{
    __shared__ __attribute__((aligned(16))) f16 As[64][64 + 8];      // [m][k]
    __shared__ __attribute__((aligned(16))) f16 Bs[64][64 + 8];      // [n][k]
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int wm = wave >> 1, wn = wave & 1;
    const int g = lane >> 5, c = lane & 31;
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const f16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};

    // this thread's two 16-byte chunks of each tile: A rows ar, ar + 32 at k-chunk ac; B k-rows bk, bk + 32 at n-chunk bc
    const int ar = tid >> 3, ac = tid & 7;
    const int bk = tid >> 3, bc = tid & 7;
    f16x8 ra[2], rb[2];
    auto load = [&](int k0) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int row = m0 + ar + 32 * h, k = k0 + ac * 8;
            ra[h] = (row < M && k < K) ? *(const f16x8*) (x + (size_t) row * K + k) : zero8;
            const int kr = k0 + bk + 32 * h, col = n0 + bc * 8;
            rb[h] = (kr < K && col < N) ? *(const f16x8*) (w + (size_t) kr * N + col) : zero8;
        }
    };
    load(0);
    for (int k0 = 0; k0 < K; k0 += 64) {
        __syncthreads();                                   // the previous step's fragment reads are done
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            *(f16x8*) &As[ar + 32 * h][ac * 8] = ra[h];
#pragma unroll
            for (int j = 0; j < 8; ++j) Bs[bc * 8 + j][bk + 32 * h] = rb[h][j];
        }
        __syncthreads();
        if (k0 + 64 < K) load(k0 + 64);                    // next tile in flight during the MFMAs
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const f16x8 af = *(const f16x8*) &As[wm * 32 + c][kk * 16 + g * 8];
            const f16x8 bf = *(const f16x8*) &Bs[wn * 32 + c][kk * 16 + g * 8];
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, bf, acc, 0, 0, 0);
        }
    }
    const int col = n0 + wn * 32 + c;
    if (col >= N) return;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * g;
        if (row < M) {
            float v = acc[r];
            if (no_zero) v += (float) out[(size_t) row * N + col];
            out[(size_t) row * N + col] = (f16) v;
        }
    }
}

// LoRA down-projection of a few rows: out[M <= 8, N <= 64] (+)= x[M, K] @ w[K, N] with a LONG K (hidden / intermediate size) and a
// handful of output columns (the adapter rank).  The 64 x 64 tile kernels above walk that K in ONE block (64-170 us per call:
// 7 such products per layer made decoding with an adapter 11 x slower than without, profiles/r04_lora.json); here K is cut over
// up to 64 blocks, each lane owns 8 consecutive columns of one k row (16-byte loads, rows of w are contiguous), the lanes of a
// wave that share the column chunk are summed by shuffles, the waves through LDS, and the blocks' fp32 partial sums land in a
// workspace that half_skinny_finish_kernel adds up in a fixed order (deterministic; one rounding to fp16, like the tile kernels).
#define HS_MAXM 8
__global__ __launch_bounds__(256) void half_skinny_partial_kernel(const f16* __restrict__ x, const f16* __restrict__ w, float* __restrict__ part,
                                                                  int M, int K, int N, int kslice)
{
    __shared__ float red[4][8][HS_MAXM][8];                          // [wave][column chunk][row][column in chunk]
    const int nc = (N + 7) >> 3;                                      // column chunks of 8 (<= 8)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane % nc, kl = lane / nc;                          // lanes of a wave: (k row in the pass, column chunk); 64 / nc rows per wave and pass
    const int rows_per_wave = 64 / nc;                                // (nc in {1, 2, 4, 8}: the launcher pads N to that)
    const int k0 = blockIdx.x * kslice, k1 = min(K, k0 + kslice);
    float acc[HS_MAXM][8];
#pragma unroll
    for (int m = 0; m < HS_MAXM; ++m)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[m][j] = 0.f;
    for (int k = k0 + wave * rows_per_wave + kl; k < k1; k += 4 * rows_per_wave) {
        f16x8 wv = {0, 0, 0, 0, 0, 0, 0, 0};
        if (c * 8 + 8 <= N) wv = *(const f16x8*) (w + (size_t) k * N + c * 8);
        else for (int j = 0; j < 8 && c * 8 + j < N; ++j) wv[j] = w[(size_t) k * N + c * 8 + j];
#pragma unroll
        for (int m = 0; m < HS_MAXM; ++m) {
            if (m >= M) break;
            const float xv = (float) x[(size_t) m * K + k];
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[m][j] = fmaf(xv, (float) wv[j], acc[m][j]);
        }
    }
#pragma unroll
    for (int m = 0; m < HS_MAXM; ++m) {
        if (m >= M) break;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float v = acc[m][j];
            for (int off = nc; off < 64; off <<= 1) v += __shfl_xor(v, off, 64);    // lanes with the same column chunk
            if (kl == 0) red[wave][c][m][j] = v;
        }
    }
    __syncthreads();
    for (int i = tid; i < M * nc * 8; i += 256) {
        const int m = i / (nc * 8), cc = (i / 8) % nc, j = i & 7;
        const float v = red[0][cc][m][j] + red[1][cc][m][j] + red[2][cc][m][j] + red[3][cc][m][j];
        if (cc * 8 + j < N) part[((size_t) blockIdx.x * M + m) * N + cc * 8 + j] = v;
    }
}

__global__ __launch_bounds__(256) void half_skinny_finish_kernel(const float* __restrict__ part, f16* __restrict__ out, int cells, int nparts, int no_zero)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= cells) return;
    float v = 0.f;
    for (int p = 0; p < nparts; ++p) v += part[(size_t) p * cells + i];
    if (no_zero) v += (float) out[i];
    out[i] = (f16) v;
}

// LoRA down-projection of a PROMPT: out[M, N <= 64] (+)= x[M, K] @ w[K, N] -- many rows, a long K, a handful of columns.  The 64 x 64
// tile kernels give that shape M / 64 blocks (32 for a 2048-token prompt on 256 CUs) each walking the whole K: 70 us per call, seven
// calls per layer -- the prompt pass with an adapter ran at half the rate without (profiles/r04_lora.json).  Here a block is ONE
// wave owning 16 rows and 1 / KS of K (grid M / 16 x KS: >= 1024 waves): per 32-k step the x fragment is a 16-byte load per lane
// (the MFMA's B operand: lane = (row, k-group)), the [32][N] slab of w -- contiguous in memory -- goes through LDS and comes back
// as the A operand (lane = (column, k-group): 8 strided 2-byte reads), one v_mfma_f32_16x16x32_f16 per 16 columns; the next step's
// loads are requested before the current step's MFMAs.  fp32 partial tiles per K part, summed by half_skinny_finish_kernel.
template <int NT>
__global__ __launch_bounds__(64) void half_tall_partial_kernel(const f16* __restrict__ x, const f16* __restrict__ w, float* __restrict__ part,
                                                               int M, int K, int N, int kslice)
{
    __shared__ __attribute__((aligned(16))) f16 slab[32 * NT * 16];
    const int lane = threadIdx.x;
    const int m0 = blockIdx.x * 16, ks = blockIdx.y;
    const int k0 = ks * kslice, k1 = min(K, k0 + kslice);               // kslice % 32 == 0, K % 32 == 0
    const int fr = lane & 15, kg = lane >> 4;
    const f16* xp = x + (size_t) min(m0 + fr, M - 1) * K + kg * 8;
    f32x4 acc[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    constexpr int SLAB16 = 32 * NT * 16 / 8;                             // 16-byte pieces of a slab (64 NT)
    f16x8 xf = *(const f16x8*) (xp + k0);
    uint4 sl[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i) sl[i] = *(const uint4*) (w + (size_t) k0 * N + (size_t) (i * 64 + lane) * 8);
    for (int k = k0; k < k1; k += 32) {
#pragma unroll
        for (int i = 0; i < NT; ++i) *(uint4*) (slab + (size_t) (i * 64 + lane) * 8) = sl[i];
        const f16x8 xc = xf;
        if (k + 32 < k1) {                                               // next step in flight during this step's LDS reads and MFMAs
            xf = *(const f16x8*) (xp + k + 32);
#pragma unroll
            for (int i = 0; i < NT; ++i) sl[i] = *(const uint4*) (w + (size_t) (k + 32) * N + (size_t) (i * 64 + lane) * 8);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                // this wave's slab is in LDS (one wave per block: no barrier)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            f16x8 wf;
#pragma unroll
            for (int j = 0; j < 8; ++j) wf[j] = slab[(kg * 8 + j) * (NT * 16) + nt * 16 + fr];
            acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf, xc, acc[nt], 0, 0, 0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                // the reads are done before the next step overwrites the slab
    }
    if (m0 + fr < M) {
        float* pp = part + ((size_t) ks * M + m0 + fr) * N;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) *(f32x4*) (pp + nt * 16 + kg * 4) = acc[nt];       // lane: row fr, columns nt * 16 + 4 kg .. + 3
    }
}

int launch_half_gemm(const f16* x, const f16* w, f16* out, int M, int K, int N, int no_zero, hipStream_t s)
{
    if (M <= 0 || N <= 0) return 0;
    if (M > 64 && N <= 64 && N % 16 == 0 && K % 32 == 0 && K >= 1024 && (((uintptr_t) w | (uintptr_t) x) & 15) == 0 && (uint64_t) M * K < (1ull << 31)) {
        int dev = 0;
        EXL_HIP(hipGetDevice(&dev));
        int ks = 8;
        while (ks > 1 && (K / ks) % 32 != 0) ks >>= 1;
        const int kslice = K / ks;
        float* ws = nullptr;
        if (kslice % 32 == 0 && exl_gemm_workspace(dev, (size_t) ks * M * N, &ws) == 0) {
            const dim3 grid((M + 15) / 16, ks);
            switch (N / 16) {
            case 1: hipLaunchKernelGGL(half_tall_partial_kernel<1>, grid, dim3(64), 0, s, x, w, ws, M, K, N, kslice); break;
            case 2: hipLaunchKernelGGL(half_tall_partial_kernel<2>, grid, dim3(64), 0, s, x, w, ws, M, K, N, kslice); break;
            case 3: hipLaunchKernelGGL(half_tall_partial_kernel<3>, grid, dim3(64), 0, s, x, w, ws, M, K, N, kslice); break;
            default: hipLaunchKernelGGL(half_tall_partial_kernel<4>, grid, dim3(64), 0, s, x, w, ws, M, K, N, kslice); break;
            }
            EXL_LAUNCH_CHECK();
            hipLaunchKernelGGL(half_skinny_finish_kernel, dim3((M * N + 255) / 256), dim3(256), 0, s, ws, out, M * N, ks, no_zero);
            EXL_LAUNCH_CHECK();
            return 0;
        }
    }
    if (M <= HS_MAXM && N <= 64 && K >= 1024 && N % 8 == 0 && (((uintptr_t) w) & 15) == 0) {
        int nc = N / 8;
        if (nc == 3) nc = 4; else if (nc > 4) nc = 8;                 // a power of two chunks per k row (the lanes of a wave split evenly)
        // (the kernel derives its chunk count from N: pass the padded width through the launch geometry only when N itself is 8, 16, 32 or 64)
        if (nc * 8 == N) {
            int dev = 0;
            EXL_HIP(hipGetDevice(&dev));
            int nparts = (K + 255) / 256;
            if (nparts > 64) nparts = 64;
            const int kslice = ((K + nparts - 1) / nparts + 7) & ~7;
            nparts = (K + kslice - 1) / kslice;
            float* ws = nullptr;
            if (exl_gemm_workspace(dev, (size_t) nparts * M * N, &ws) == 0) {
                hipLaunchKernelGGL(half_skinny_partial_kernel, dim3(nparts), dim3(256), 0, s, x, w, ws, M, K, N, kslice);
                EXL_LAUNCH_CHECK();
                hipLaunchKernelGGL(half_skinny_finish_kernel, dim3((M * N + 255) / 256), dim3(256), 0, s, ws, out, M * N, nparts, no_zero);
                EXL_LAUNCH_CHECK();
                return 0;
            }
        }
    }
    dim3 grid((N + 63) / 64, (M + 63) / 64);
    const bool vec = K % 8 == 0 && N % 8 == 0 && (((uintptr_t) x | (uintptr_t) w) & 15) == 0;
    if (vec) hipLaunchKernelGGL(half_gemm64_kernel, grid, dim3(256), 0, s, x, w, out, M, K, N, no_zero);
    else     hipLaunchKernelGGL(half_gemm_kernel, grid, dim3(256), 0, s, x, w, out, M, K, N, no_zero);
    EXL_LAUNCH_CHECK();
    return 0;
}
