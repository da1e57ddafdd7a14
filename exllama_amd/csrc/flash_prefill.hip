// MFMA flash attention for the prompt pass (long q_len), head_dim = 128, fp16 in / fp32 softmax+accumulate.
// Replaces F.scaled_dot_product_attention / the matmul-softmax-matmul chain at
// /root/reference/model.py:463-495 (and the repeat_kv copies of model.py:310-319: GQA is an index here).
// Two kernels: flash_prefill_kernel (4 waves, register staging; prompts of one query block) and flash_prefill8_kernel (8 waves, LDS-DMA
// staging, transposing V reads; everything longer) -- launch_flash_prefill at the end of the file picks.  The arithmetic is the same.
//
// Structure (one block = 128 query rows of one head = 4 waves x 32 rows; KV tiles of 64 keys):
//   S^T = K Q^T   "swapped" product: MFMA A = K tile rows straight from LDS (d contiguous), B = Q held in
//                 registers for the whole kernel.  Each lane then owns ONE query column with 16 keys per
//                 32x32 result in its registers, so the row max / row sum are in-lane (+ one exchange with
//                 lane^32) and the running statistics m, l are plain per-lane scalars.
//   O^T = V^T P^T  P^T in its MFMA result layout IS the B operand of the second product (keys along the
//                 register index) once the key order inside each 16-key block is taken as
//                 key(g,e) = (e&3) + 8*(e>>2) + 4*g; V^T is written to LDS in exactly that key order, so
//                 no cross-lane traffic is needed for P.  O^T keeps the query in the lane: rescaling by
//                 alpha and the final 1/l are in-lane as well.
//   V is transposed once per block on its way into LDS (4 keys x 8 d per thread -> 8 x ds_write_b64),
//   both LDS tiles are XOR-swizzled for ds_read_b128.
// Causal: query i (global row q0+i) sees keys j <= past_len + q0 + i; tiles beyond the diagonal are skipped.
#include "common.h"
#include <stdlib.h>

#define FA_BQ 128
#define FA_BKV 64
#define FA_HD 128

// Phase attribution (scripts/bench_flash.hip builds this file with -DEXL_FLASH_PROBE): cycles of wave 0 of every block, summed over its
// key tiles: {total, wait for the tile in flight + barrier, LDS store, barrier, issue of the next tile's loads, compute, tiles}
#ifdef EXL_FLASH_PROBE
#define FA_PROBE_BLOCKS 512
__device__ unsigned long long g_flash_probe[FA_PROBE_BLOCKS * 8];
__device__ unsigned long long g_flash_probe8[FA_PROBE_BLOCKS * 2 * 8];       // the 8-wave kernel: [block][set][phase]
__device__ __forceinline__ unsigned long long fa_clk() { unsigned long long t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory"); return t; }
#define FP8(v) const unsigned long long v = fa_clk()
#define FP_CLK(v) const unsigned long long v = __builtin_readcyclecounter()
#define FP_ACC(i, a, b) fp_acc[i] += (b) - (a)
#else
#define FP_CLK(v) do { } while (0)
#define FP_ACC(i, a, b) do { } while (0)
#define FP8(v) do { } while (0)
#endif

__device__ __forceinline__ uint32_t pack_lo(uint32_t a, uint32_t b) { return (a & 0xFFFFu) | (b << 16); }
__device__ __forceinline__ uint32_t pack_hi(uint32_t a, uint32_t b) { return (a >> 16) | (b & 0xFFFF0000u); }

// K tile [64 keys][128 d]: 256-byte rows, 16 slots of 16 B, slot ^= key & 15
__device__ __forceinline__ int k_lds_off(int key, int slot) { return key * 256 + ((slot ^ (key & 15)) << 4); }
// V^T tile [128 d][64 key positions]: 128-byte rows, 8 slots of 16 B, slot ^= f(d)
__device__ __forceinline__ int vt_lds_off(int d, int slot) { return d * 128 + ((slot ^ (((d >> 1) ^ (d >> 4)) & 7)) << 4); }

// Per-lane LDS offsets of the MFMA operand fragments, computed ONCE (the loop body is VALU-issue bound: every address
// instruction saved per tile counts).  K: k_lds_off(32 * kt + c, 2 * kb + g) = kofs[kb] + kt * 8192 (the swizzle only sees
// c & 15).  V^T: vt_lds_off(32 * dt + c, 2 * kb2 + g) = vofs[kb2 ^ dt] + dt * 4096: with d = 32 * dt + c the swizzle term
// ((d >> 1) ^ (d >> 4)) & 7 is ((c >> 1) ^ (c >> 4)) & 7 xor 2 * dt, and xor-ing 2 * dt into slot 2 * kb2 + g permutes kb2.
struct FaFragOffsets { int kofs[8]; int vofs[4]; };
__device__ __forceinline__ FaFragOffsets fa_frag_offsets(int c, int g)
{
    FaFragOffsets o;
#pragma unroll
    for (int kb = 0; kb < 8; ++kb) o.kofs[kb] = k_lds_off(c, 2 * kb + g);
#pragma unroll
    for (int t = 0; t < 4; ++t) o.vofs[t] = vt_lds_off(c, 2 * t + g);
    return o;
}

__global__ __launch_bounds__(256, 2) void flash_prefill_kernel(const f16* __restrict__ q, const f16* __restrict__ kc,
                                                            const f16* __restrict__ vc, f16* __restrict__ out,
                                                            int q_len, int heads, int kv_heads, int max_seq,
                                                            int past_len, float c1 /* scale * log2(e) */, int frag)
{
    __shared__ __attribute__((aligned(16))) unsigned char lds[FA_BKV * FA_HD * 2 * 2];
    unsigned char* k_lds = lds;
    unsigned char* vt_lds = lds + FA_BKV * FA_HD * 2;

    // 1-D grid.  Causal work grows with the query block (block qb walks qb + 1 key tiles of 128), and the blocks that
    // end up co-resident on a CU are `half` apart in launch order (two 256-thread blocks per CU when the grid is 2 x CUs):
    // the second half of the grid walks the query blocks in REVERSE, so a CU gets qb and nqb-1-qb -- equal work per CU.
    const int nqb = (q_len + FA_BQ - 1) / FA_BQ;
    const int total = gridDim.x;
    const int L = blockIdx.x;
    const int qi = L % nqb;
    const int hb = L / nqb;
    const int qb = (2 * L >= total) ? nqb - 1 - qi : qi;
    const int h = hb % heads;
    const int b = hb / heads;
    const int kvh = h / (heads / kv_heads);
    const int tid = threadIdx.x;
    const int wave = tid >> 6;
    const int lane = tid & 63;
    const int g = lane >> 5;
    const int c = lane & 31;
    const FaFragOffsets fo = fa_frag_offsets(c, g);

    const int q0 = qb * FA_BQ;
    const int qrow = q0 + wave * 32 + c;                       // this lane's query (row inside q_len)
    const int kv_len = past_len + q_len;
    const int last_q = min(q0 + FA_BQ, q_len) - 1;
    const int ntiles = (min(kv_len, past_len + last_q + 1) + FA_BKV - 1) / FA_BKV;
    const int wave_last_key = past_len + min(q0 + wave * 32 + 31, q_len - 1);   // last key any row of this wave sees

    const f16* kbase = kc + ((size_t) b * kv_heads + kvh) * max_seq * FA_HD;
    const f16* vbase = vc + ((size_t) b * kv_heads + kvh) * max_seq * FA_HD;

    // ---- Q fragments (B operand of S^T): Q[qrow][16*kb + 8*g .. +8] -------------------------------------
    f16x8 qf[8];
    {
        const f16* qp = q + (((size_t) b * q_len + qrow) * heads + h) * FA_HD + 8 * g;
#pragma unroll
        for (int kb = 0; kb < 8; ++kb) {
            if (qrow < q_len) qf[kb] = *(const f16x8*) (qp + 16 * kb);
            else {
#pragma unroll
                for (int e = 0; e < 8; ++e) qf[kb][e] = (f16) 0.f;
            }
        }
    }

    f32x16 acc_o[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc_o[dt][r] = 0.f;
    float m_run = -INFINITY;
    float l_run = 0.f;

    // staging roles
    const int s_d8 = tid & 15;            // 16-byte column of a K / V row
    const int s_kq = tid >> 4;            // 0..15
    uint4 kreg[4], vreg[4];
    auto load_tile = [&](int tile) {
        const int kv0 = tile * FA_BKV;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int kk = kv0 + s_kq + 16 * i;                // K: rows s_kq, +16, +32, +48
            kreg[i] = kk < kv_len ? *(const uint4*) (kbase + (size_t) kk * FA_HD + s_d8 * 8) : make_uint4(0, 0, 0, 0);
            const int kvv = kv0 + 4 * s_kq + i;                // V: rows 4*s_kq .. +3
            vreg[i] = kvv < kv_len ? *(const uint4*) (vbase + (size_t) kvv * FA_HD + s_d8 * 8) : make_uint4(0, 0, 0, 0);
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) *(uint4*) (k_lds + k_lds_off(s_kq + 16 * i, s_d8)) = kreg[i];
        // V: 4 keys x 8 d -> 8 x (4 keys of one d).  Position of key 4*s_kq + i inside the tile:
        //   16-key block kb2 = s_kq >> 2, pos16 = 8*(s_kq & 1) + 4*((s_kq >> 1) & 1) + i
        const int kb2 = s_kq >> 2;
        const int pos = 16 * kb2 + 8 * (s_kq & 1) + 4 * ((s_kq >> 1) & 1);      // multiple of 4
        const int slot = pos >> 3;
        const int inslot = (pos & 7) * 2;                                         // 0 or 8 bytes
        const uint32_t r0[4] = {vreg[0].x, vreg[0].y, vreg[0].z, vreg[0].w};
        const uint32_t r1[4] = {vreg[1].x, vreg[1].y, vreg[1].z, vreg[1].w};
        const uint32_t r2[4] = {vreg[2].x, vreg[2].y, vreg[2].z, vreg[2].w};
        const uint32_t r3[4] = {vreg[3].x, vreg[3].y, vreg[3].z, vreg[3].w};
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int d_even = s_d8 * 8 + 2 * p;
            uint2 lo, hi;
            lo.x = pack_lo(r0[p], r1[p]); lo.y = pack_lo(r2[p], r3[p]);
            hi.x = pack_hi(r0[p], r1[p]); hi.y = pack_hi(r2[p], r3[p]);
            *(uint2*) (vt_lds + vt_lds_off(d_even, slot) + inslot) = lo;
            *(uint2*) (vt_lds + vt_lds_off(d_even + 1, slot) + inslot) = hi;
        }
    };

    load_tile(0);
#ifdef EXL_FLASH_PROBE
    unsigned long long fp_acc[6] = {0, 0, 0, 0, 0, 0};
    const unsigned long long fp_t0 = __builtin_readcyclecounter();
#endif
    for (int tile = 0; tile < ntiles; ++tile) {
        FP_CLK(c0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                       // previous tile's readers are done
        FP_CLK(c1);
        store_tile();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        FP_CLK(c2);
        __syncthreads();
        FP_CLK(c3);
        if (tile + 1 < ntiles) load_tile(tile + 1);
        FP_CLK(c4);
        FP_ACC(1, c0, c1); FP_ACC(2, c1, c2); FP_ACC(3, c2, c3); FP_ACC(4, c3, c4);

        const int kv0 = tile * FA_BKV;
        if (kv0 > wave_last_key) continue;                     // wave-uniform: everything masked for this wave

        // ---- S^T = K Q^T : two 32-key chains ------------------------------------------------------------
        f32x16 s[2];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[kt][r] = 0.f;
#pragma unroll
            for (int kb = 0; kb < 8; ++kb) {
                const f16x8 kf = *(const f16x8*) (k_lds + fo.kofs[kb] + kt * 8192);
                s[kt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[kb], s[kt], 0, 0, 0);
            }
        }
        // ---- causal mask + online softmax (this lane: query qrow, keys kv0 + kt*32 + (r&3)+8*(r>>2)+4*g) ----
        const int limit = past_len + qrow;                     // last visible key
        float mx = -INFINITY;
        if (kv0 + FA_BKV - 1 <= past_len + q0 + wave * 32) {    // wave-uniform: the whole tile is visible to every row of this wave
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[kt][r]);
        } else {
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = kv0 + kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * g;
                    const float v = key <= limit ? s[kt][r] : -INFINITY;
                    s[kt][r] = v;
                    mx = fmaxf(mx, v);
                }
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        // Lazy reference maximum: softmax is invariant to the reference point as long as l and O share it, so the running
        // maximum only moves when a row's maximum grew by more than 2^8 in the exp2 domain (p <= 256 stays exact-range in
        // fp16, O and l accumulate in fp32).  After the first tiles almost no tile rescales O (64 multiplies per lane).
        const float m_new = (mx - m_run) * c1 > 8.0f ? mx : m_run;     // m_run = -inf on the first tile: always taken
        const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c1);
        const float mc = m_new * c1;
        float psum = 0.f;
        f16x8 pf[4];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float p = __builtin_amdgcn_exp2f(fmaf(s[kt][r], c1, -mc));   // v_exp_f32: exponent <= 8, a flushed denormal is 0
                psum += p;
                pf[kt * 2 + (r >> 3)][r & 7] = (f16) p;
            }
        l_run = fmaf(l_run, alpha, psum);
        m_run = m_new;
        if (!__all(alpha == 1.0f)) {                           // wave-uniform: no row's running maximum moved -> nothing to rescale
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc_o[dt][r] *= alpha;
        }

        // ---- O^T += V^T P^T ---------------------------------------------------------------------------------
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
            for (int kb2 = 0; kb2 < 4; ++kb2) {
                const f16x8 vf = *(const f16x8*) (vt_lds + fo.vofs[kb2 ^ dt] + dt * 4096);
                acc_o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf[kb2], acc_o[dt], 0, 0, 0);
            }
    }

#ifdef EXL_FLASH_PROBE
    if (tid == 0 && blockIdx.x < FA_PROBE_BLOCKS) {
        unsigned long long* pp = g_flash_probe + (size_t) blockIdx.x * 8;
        pp[0] = __builtin_readcyclecounter() - fp_t0;
        for (int i = 1; i < 5; ++i) pp[i] = fp_acc[i];
        pp[5] = pp[0] - fp_acc[1] - fp_acc[2] - fp_acc[3] - fp_acc[4];
        pp[6] = (unsigned long long) ntiles;
    }
#endif
    // ---- epilogue: O[qrow][d] = O^T / l,  d = dt*32 + (r&3) + 8*(r>>2) + 4*g ------------------------------
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    if (qrow >= q_len) return;
    const float inv = 1.0f / l_tot;
    f16* op = out + (((size_t) b * q_len + qrow) * heads + h) * FA_HD;
    // frag: the output in the FRAGMENT ORDER the short-prompt o_proj GEMM reads (q4_gemm_frag.hip; K = heads * 128: head h is row-block h,
    // d = 32 dt + 8 rq + 4 g + e is k-group dt, MFMA rq): the 8 bytes below land at halves 4 g .. of piece (row / 16, 4 h + rq), lane
    // 16 dt + row % 16 -- the re-tile launch between attention and o_proj (5-6 us in the chain of a layer) is not needed
    const size_t R = (size_t) b * q_len + qrow;
    f16* fp = out + ((((R >> 4) * (size_t) (heads * 4) + (size_t) (4 * h)) * 64 + (R & 15)) * 8 + 4 * g);
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
            f16x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (f16) (acc_o[dt][rq * 4 + e] * inv);
            if (frag) *(f16x4*) (fp + (size_t) (rq * 64 + dt * 16) * 8) = o;
            else *(f16x4*) (op + dt * 32 + 8 * rq + 4 * g) = o;
        }
}

// ---------------------------------------------------------------------------------------------------------------
// The 8-wave form (round 4; the default).  What the phase probe of the kernel above showed at S = 2048 (scripts/bench_flash.hip
// -DEXL_FLASH_PROBE): the launch lasts as long as its LONGEST block (32 key tiles for the last query block), which runs alone on its
// SIMDs for most of that time -- one wave per SIMD, 4037 cycles per key tile of which 162 + 555 + 88 + 666 are the staging of the
// tile (wait for the registers in flight, transpose + LDS stores, barrier, issue of the next tile's 8 loads per thread) and 2565
// the chain S -> softmax -> P V of a single wave.  Here
//   * K and V tiles travel by LDS-DMA (global_load_lds_dwordx4), a pair of tiles ahead, into a 2-deep ring: no staging registers,
//     no transposition pass, no LDS stores, one barrier per PAIR of tiles;
//   * V stays key-major in LDS, as [32 keys][16 d] sub-tiles (1 KiB = one DMA instruction), and the V^T operand of O^T += V^T P^T
//     is read with the transposing ds_read_b64_tr_b16: lane c of a 16-lane group, element j receives element c % 4 of the 4-key run
//     addressed by lane 4 j + c / 4 (scripts/probe_tr16.hip) -- 4 consecutive keys of one d, which is the key order
//     key(g, e) = (e & 3) + 8 (e >> 2) + 4 g of P^T's registers with two reads per operand;
//   * a block is 8 waves: waves w and w + 4 own the SAME 32 query rows and take the even / the odd key tiles, so every SIMD holds
//     two independent S -> softmax -> P V chains; the two partial results (O, reference maximum, sum) are merged through LDS at the
//     end;
//   * one block per CU, dispatched heaviest query block first.
// ---------------------------------------------------------------------------------------------------------------
typedef __fp16 fa_h4 __attribute__((__vector_size__(8)));
#ifndef FA8_ABL
#define FA8_ABL 0               // scripts/bench_flash.hip only (wrong results, timing): 1 no DMA inside the loop, 2 no V^T reads, 4 no softmax arithmetic, 8 no K reads
#endif
#define FA8_TILE_BYTES (FA_BKV * FA_HD * 2)                 // 16 KiB
#define FA8_PAIR_BYTES (4 * FA8_TILE_BYTES)                 // K even, V even, K odd, V odd
#define FA8_LDS_BYTES (2 * FA8_PAIR_BYTES)                  // 128 KiB

__global__ __launch_bounds__(512) void flash_prefill8_kernel(const f16* __restrict__ q, const f16* __restrict__ kc,
                                                             const f16* __restrict__ vc, f16* __restrict__ out,
                                                             int q_len, int heads, int kv_heads, int max_seq,
                                                             int past_len, float c1 /* scale * log2(e) */, int frag)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds8[];
    const uint32_t lds_base = (uint32_t) (uintptr_t) (__attribute__((address_space(3))) unsigned char*) lds8;

    const int nqb = (q_len + FA_BQ - 1) / FA_BQ;
    const int per_q = gridDim.x / nqb;                          // heads * bsz
    const int L = blockIdx.x;
    const int hb = L % per_q;                                   // block L runs on XCD L % 8: the query blocks of a head share an L2
    const int qb = nqb - 1 - L / per_q;                         // heaviest first
    const int h = hb % heads;
    const int b = hb / heads;
    const int kvh = h / (heads / kv_heads);
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int set = wave >> 2;                                  // 0: even key tiles, 1: odd key tiles
    const int wq = wave & 3;                                    // 32 query rows of the block
    const int g = lane >> 5;
    const int c = lane & 31;

    const int q0 = qb * FA_BQ;
    const int qrow = q0 + wq * 32 + c;
    const int kv_len = past_len + q_len;
    const int last_q = min(q0 + FA_BQ, q_len) - 1;
    const int ntiles = (min(kv_len, past_len + last_q + 1) + FA_BKV - 1) / FA_BKV;
    const int niter = (ntiles + 1) >> 1;

    const f16* kbase = kc + ((size_t) b * kv_heads + kvh) * max_seq * FA_HD;
    const f16* vbase = vc + ((size_t) b * kv_heads + kvh) * max_seq * FA_HD;

    // ---- the DMA pieces of this wave: 4 of the 16 K pieces (4 key rows each) and 4 of the 16 V sub-tiles of ITS set's tile ------
    // K piece j: rows wq * 16 + j * 4 + (lane >> 4), LDS slot lane & 15 of the row holds the row's 16-byte slot (lane & 15) ^ (row & 15)
    // V sub-tile st = wq * 4 + j = kq * 4 + cb ([16 keys][32 d], 64-byte rows): key kq * 16 + (lane >> 2), d = cb * 32 + (lane & 3) * 8 .. + 8
    // Rows past kv_len re-read the last valid row (finite values under a zero weight; the rows behind kv_len may hold anything).
    int k_row[4], v_row[4];
    uint32_t k_col[4], v_col[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = wq * 16 + j * 4 + (lane >> 4);
        k_row[j] = r;
        k_col[j] = (uint32_t) (((lane & 15) ^ (r & 15)) * 16);
        const int st = wq * 4 + j;
        v_row[j] = (st >> 2) * 16 + (lane >> 2);
        v_col[j] = (uint32_t) ((st & 3) * 64 + (lane & 3) * 16);
    }
    auto issue_piece = [&](int tile, int slot, int j) {          // j = 0..3: K pieces, 4..7: V sub-tiles.  Straight-line: no branch per piece
        const int kv0 = tile * FA_BKV;
        const bool is_v = j >= 4;
        const int jj = j & 3;
        const uint32_t dst = lds_base + (uint32_t) (slot * FA8_PAIR_BYTES + set * 2 * FA8_TILE_BYTES + wq * 4096 + (is_v ? FA8_TILE_BYTES : 0) + jj * 1024);
        const int row = is_v ? v_row[jj] : k_row[jj];
        const uint32_t col = is_v ? v_col[jj] : k_col[jj];
        const uint32_t off = (uint32_t) min(kv0 + row, kv_len - 1) * 256u + col;
        asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2" :: "s"(dst), "v"(off), "s"(is_v ? vbase : kbase));
    };
    auto issue_tile = [&](int tile, int slot) {
#pragma unroll
        for (int j = 0; j < 8; ++j) issue_piece(tile, slot, j);
    };

    // ---- Q fragments (B operand of S^T): Q[qrow][16*kb + 8*g .. +8] -------------------------------------
    f16x8 qf[8];
    {
        const f16* qp = q + (((size_t) b * q_len + qrow) * heads + h) * FA_HD + 8 * g;
#pragma unroll
        for (int kb = 0; kb < 8; ++kb) {
            if (qrow < q_len) qf[kb] = *(const f16x8*) (qp + 16 * kb);
            else {
#pragma unroll
                for (int e = 0; e < 8; ++e) qf[kb][e] = (f16) 0.f;
            }
        }
    }
    int kofs[8];                                                 // K fragment: k_lds_off(32 * kt + c, 2 * kb + g) = kofs[kb] + kt * 8192
#pragma unroll
    for (int kb = 0; kb < 8; ++kb) kofs[kb] = k_lds_off(c, 2 * kb + g);
    // V^T fragment of (dt, kb2), half hf: keys kb2 * 16 + 4 g + 8 hf + (i >> 2) of the group's lane i, d = dt * 32 + 16 ((lane >> 4) & 1) + 4 (lane & 3):
    // ONE per-lane byte offset + the immediate kb2 * 4096 + dt * 1024 + hf * 512.  The two 16-lane groups of a 32-lane LDS pass read the
    // two 32-byte halves of the same four 64-byte rows: 256 contiguous bytes, every bank once.
    const int vt_lane = (4 * (lane >> 5) + ((lane & 15) >> 2)) * 64 + ((lane >> 4) & 1) * 32 + (lane & 3) * 8;

    f32x16 acc_o[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc_o[dt][r] = 0.f;
    float m_run = -1e30f;                                       // finite: a fully masked tile leaves exp2(m_run - m_run) = 1, not NaN
    float l_run = 0.f;

    // tile number of the set's j-th tile; behind the set's last tile the block's last one (a finite source for a slot that is read fully
    // masked or not at all: no branch in the instruction stream)
    auto tile_of = [&](int j) { const int t = 2 * j + set; return t < ntiles ? t : ntiles - 1; };

    // ---- S^T = K Q^T of the tile in K slot `kslot`: two 32-key chains.  Hand-issued reads, hand-counted waits (LDS returns in order; no
    // scalar load is in flight inside the loops): left to itself hipcc re-used one register quad and waited lgkmcnt(0) in front of every MFMA.
    // The 8 K fragments of a chain are requested together, the second chain's while the first one multiplies; `dma(kb)` issues DMA piece kb
    // between the MFMAs of the second chain (plain asm volatile, no "memory" clobber: a clobber pins every LDS read behind it).
    auto s_phase = [&](int kslot, f32x16 (&s)[2], auto&& dma) {
        f16x8 ka[8], kb_[8];
        const uint32_t ka0 = lds_base + (uint32_t) (kslot * FA8_PAIR_BYTES + set * 2 * FA8_TILE_BYTES);
#pragma unroll
        for (int kb = 0; kb < 8; ++kb) {
            if (FA8_ABL & 8) ka[kb] = qf[(kb + 1) & 7];
            else asm volatile("ds_read_b128 %0, %1" : "=v"(ka[kb]) : "v"(ka0 + (uint32_t) kofs[kb]));
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[0][r] = 0.f; s[1][r] = 0.f; }
#pragma unroll
        for (int kb = 0; kb < 8; ++kb) {
            if (!(FA8_ABL & 8)) asm volatile("s_waitcnt lgkmcnt(7)" : "+v"(ka[kb]));
            s[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ka[kb], qf[kb], s[0], 0, 0, 0);
            if (FA8_ABL & 8) kb_[kb] = qf[(kb + 2) & 7];
            else asm volatile("ds_read_b128 %0, %1 offset:8192" : "=v"(kb_[kb]) : "v"(ka0 + (uint32_t) kofs[kb]));
        }
#define FA8_STEPB(N, kb) if (!(FA8_ABL & 8)) asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(kb_[kb])); \
                         s[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kb_[kb], qf[kb], s[1], 0, 0, 0); if (!(FA8_ABL & 1)) dma(kb)
        FA8_STEPB(7, 0); FA8_STEPB(6, 1); FA8_STEPB(5, 2); FA8_STEPB(4, 3); FA8_STEPB(3, 4); FA8_STEPB(2, 5); FA8_STEPB(1, 6); FA8_STEPB(0, 7);
#undef FA8_STEPB
        // The counted waits above hold as long as NOTHING ELSE enters the LDS queue between the first read and the last wait: the V^T reads
        // of the softmax / P V part below depend on nothing computed here, so neither the compiler (memory clobber) nor the machine
        // scheduler (sched_barrier) may lift them into this stretch.
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    };

    // ---- causal mask + online softmax of S (this lane: query qrow, keys kv0 + kt*32 + (r&3)+8*(r>>2)+4*g), then O^T += V^T P^T with the V
    // tile in V slot `vslot`: V^T operands by transposing reads of the key-major tile.  (No "nothing to do for this wave" short cut: a tile
    // behind the wave's last visible key is computed fully masked -- p = 0 exactly, the reference point stays put -- ~3 % of the wave-tiles.
    // With the short cut hipcc kept the 64 accumulators in two places and copied them twice per tile: 64-96 v_mov_b64 beside 32 MFMAs.)
    auto softmax_pv = [&](f32x16 (&s)[2], int tile, int vslot) {
        const int kv0 = tile * FA_BKV;
        const int limit = past_len + qrow;
        float mx = -INFINITY;
        if (FA8_ABL & 4) {
            mx = s[0][0];
        } else
        if (kv0 + FA_BKV - 1 <= past_len + q0 + wq * 32) {     // wave-uniform: the whole tile is visible to every row of this wave
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[kt][r]);
        } else {
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = kv0 + kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * g;
                    const float v = key <= limit ? s[kt][r] : -INFINITY;
                    s[kt][r] = v;
                    mx = fmaxf(mx, v);
                }
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = (mx - m_run) * c1 > 8.0f ? mx : m_run;     // lazy reference maximum, as in the kernel above
        const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c1);
        const float mc = m_new * c1;
        float psum = 0.f;
        f16x8 pf[4];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if (FA8_ABL & 4) { pf[kt * 2 + (r >> 3)][r & 7] = (f16) s[kt][r]; continue; }
                const float p = __builtin_amdgcn_exp2f(fmaf(s[kt][r], c1, -mc));
                psum += p;
                pf[kt * 2 + (r >> 3)][r & 7] = (f16) p;
            }
        l_run = fmaf(l_run, alpha, psum);
        m_run = m_new;
        if (!__all(alpha == 1.0f)) {
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) acc_o[dt] = acc_o[dt] * alpha;
        }
        __attribute__((address_space(3))) unsigned char* v_lds =
            (__attribute__((address_space(3))) unsigned char*) lds8 + vslot * FA8_PAIR_BYTES + set * 2 * FA8_TILE_BYTES + FA8_TILE_BYTES + vt_lane;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
            for (int kb2 = 0; kb2 < 4; ++kb2) {
                const int imm = kb2 * 4096 + dt * 1024;
                if (FA8_ABL & 2) { acc_o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(qf[(dt + kb2) & 7], pf[kb2], acc_o[dt], 0, 0, 0); continue; }
                const fa_h4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fa_h4*) (v_lds + imm));
                const fa_h4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fa_h4*) (v_lds + imm + 512));
                const uint2 lo2 = __builtin_bit_cast(uint2, lo), hi2 = __builtin_bit_cast(uint2, hi);
                const f16x8 vf = __builtin_bit_cast(f16x8, make_uint4(lo2.x, lo2.y, hi2.x, hi2.y));
                acc_o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf[kb2], acc_o[dt], 0, 0, 0);
            }
    };
    auto step_sync = [&]() {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // this wave's DMA pieces have landed
        __syncthreads();                                        // ... everybody's have, and the slots about to be refilled have been read
    };

    // One step = one pair of key tiles (the even set's and the odd set's): wait for the pair's DMA, barrier, S of the own tile -- with the
    // DMA of the set's next tile (into the other slot, read one step ago) between its MFMAs --, softmax, P V.
    // Measured on the way and not kept (scripts/bench_flash.hip, S = 2048, 32 heads; this form: 51.9 us): P V of the tile before issued in
    // one basic block with this tile's exponentials (the matrix and the vector pipe fed by ONE wave): 55.3 us; the two sets a third of a
    // step apart (one instruction stream, the odd set at the barrier between softmax / P V and S): 55.4 us; the odd set started late by
    // s_sleep: 52.7-59 us; the DMA pieces issued in the vector stretch behind S instead of between its MFMAs: 52.4 against 53.1 us, within
    // what the validated form is worth keeping for.  Phase probe of the second (cycles per step, 2 wave-tiles per SIMD): 3950 = 2 x ~1000 in S, 2 x ~1800 in softmax
    // + P V, the rest at the barrier; the matrix pipe is busy 2048 of them.  At 56 us per launch those 17 x 3950 cycles mean a shader clock
    // of ~1.3-1.4 GHz: like the GEMMs of the prompt pass this kernel runs into the power limit, and what is left is energy per tile
    // rather than an idle pipe.
    issue_tile(tile_of(0), 0);
    for (int j = 0; j < niter; ++j) {
        step_sync();
        f32x16 s[2];
        s_phase(j & 1, s, [&](int kb) { issue_piece(tile_of(j + 1), (j + 1) & 1, kb); });
        softmax_pv(s, 2 * j + set, j & 1);
    }

    // ---- the odd-tile waves hand (O, m, l) to their even-tile partners through the ring's memory -----------------------------
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    float* mo = (float*) lds8 + (size_t) wq * (66 * 64);
    if (set == 1) {
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) mo[(dt * 16 + r) * 64 + lane] = acc_o[dt][r];
        mo[64 * 64 + lane] = m_run;
        mo[65 * 64 + lane] = l_run;
    }
    __syncthreads();
    if (set == 1) return;
    {
        const float mb = mo[64 * 64 + lane], lb = mo[65 * 64 + lane];
        const float m = fmaxf(m_run, mb);
        const float fa = __builtin_amdgcn_exp2f((m_run - m) * c1), fb = __builtin_amdgcn_exp2f((mb - m) * c1);   // mb = -1e30 (no visible odd tile): 0
        l_run = l_run * fa + lb * fb;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc_o[dt][r] = acc_o[dt][r] * fa + mo[(dt * 16 + r) * 64 + lane] * fb;
    }

    // ---- epilogue: O[qrow][d] = O^T / l,  d = dt*32 + (r&3) + 8*(r>>2) + 4*g ------------------------------
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    if (qrow >= q_len) return;
    const float inv = 1.0f / l_tot;
    f16* op = out + (((size_t) b * q_len + qrow) * heads + h) * FA_HD;
    // frag: the output in the FRAGMENT ORDER the short-prompt o_proj GEMM reads (q4_gemm_frag.hip; K = heads * 128: head h is row-block h,
    // d = 32 dt + 8 rq + 4 g + e is k-group dt, MFMA rq): the 8 bytes below land at halves 4 g .. of piece (row / 16, 4 h + rq), lane
    // 16 dt + row % 16 -- the re-tile launch between attention and o_proj (5-6 us in the chain of a layer) is not needed
    const size_t R = (size_t) b * q_len + qrow;
    f16* fp = out + ((((R >> 4) * (size_t) (heads * 4) + (size_t) (4 * h)) * 64 + (R & 15)) * 8 + 4 * g);
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
            f16x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (f16) (acc_o[dt][rq * 4 + e] * inv);
            if (frag) *(f16x4*) (fp + (size_t) (rq * 64 + dt * 16) * 8) = o;
            else *(f16x4*) (op + dt * 32 + 8 * rq + 4 * g) = o;
        }
}

int launch_flash_prefill(const f16* q, const f16* kc, const f16* vc, f16* out, int bsz, int q_len, int heads,
                         int kv_heads, int hd, int max_seq, int past_len, hipStream_t s, int frag)
{
    EXL_REQUIRE(hd == FA_HD, EXL_E_UNSUPPORTED, "flash prefill: head_dim must be 128 (got %d)", hd);
    const float c1 = (1.0f / sqrtf((float) hd)) * 1.4426950408889634f;
    dim3 grid(((q_len + FA_BQ - 1) / FA_BQ) * heads * bsz);
    // (Round 1, 4-wave kernel: double-buffered K / V^T tiles with one barrier per tile measured 69.5 us against 66.6 -- the store of tile
    // t + 1 sits in front of tile t's MFMAs; an 8-wave block with REGISTER staging, whose wave pairs split every tile's keys, 79 us: every
    // wave paid the staging of both tiles.  The 8-wave kernel above stages by DMA.)
    // One query block of at most 4 key tiles (a 128-token prompt): the 4-wave kernel (7.2 against 8.5 us at 32 heads); everything longer --
    // 14.5 against 17 us at 512 tokens, 52.8 against 65 at 2048, 31 against 51 for a 64-token chunk behind 1984 cached tokens (same box,
    // alternating, profiles/r04_flash_ab.txt) -- the 8-wave kernel.  EXL_FLASH_4WAVE=1 keeps the 4-wave kernel everywhere (A/B).
    static const bool four_waves_env = getenv("EXL_FLASH_4WAVE") != nullptr;
    const bool four_waves = four_waves_env || (q_len <= FA_BQ && past_len + q_len <= 4 * FA_BKV);
    if (four_waves) {
        hipLaunchKernelGGL(flash_prefill_kernel, grid, dim3(256), 0, s, q, kc, vc, out, q_len, heads, kv_heads, max_seq,
                           past_len, c1, frag);
        EXL_LAUNCH_CHECK();
        return 0;
    }
    static bool big[EXL_MAX_DEVICES] = {};
    EXL_TRY(exl_lds_opt_in((const void*) flash_prefill8_kernel, big));
    hipLaunchKernelGGL(flash_prefill8_kernel, grid, dim3(512), FA8_LDS_BYTES, s, q, kc, vc, out, q_len, heads, kv_heads, max_seq,
                       past_len, c1, frag);
    EXL_LAUNCH_CHECK();
    return 0;
}
