// MFMA flash attention for the prompt pass (long q_len), head_dim = 128, fp16 in / fp32 softmax+accumulate.
// Replaces F.scaled_dot_product_attention / the matmul-softmax-matmul chain at
// /root/reference/model.py:463-495 (and the repeat_kv copies of model.py:310-319: GQA is an index here).
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
                                                            int past_len, float c1 /* scale * log2(e) */)
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
    for (int tile = 0; tile < ntiles; ++tile) {
        __syncthreads();                                       // previous tile's readers are done
        store_tile();
        __syncthreads();
        if (tile + 1 < ntiles) load_tile(tile + 1);

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

    // ---- epilogue: O[qrow][d] = O^T / l,  d = dt*32 + (r&3) + 8*(r>>2) + 4*g ------------------------------
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    if (qrow >= q_len) return;
    const float inv = 1.0f / l_tot;
    f16* op = out + (((size_t) b * q_len + qrow) * heads + h) * FA_HD;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
            f16x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (f16) (acc_o[dt][rq * 4 + e] * inv);
            *(f16x4*) (op + dt * 32 + 8 * rq + 4 * g) = o;
        }
}

int launch_flash_prefill(const f16* q, const f16* kc, const f16* vc, f16* out, int bsz, int q_len, int heads,
                         int kv_heads, int hd, int max_seq, int past_len, hipStream_t s)
{
    EXL_REQUIRE(hd == FA_HD, EXL_E_UNSUPPORTED, "flash prefill: head_dim must be 128 (got %d)", hd);
    const float c1 = (1.0f / sqrtf((float) hd)) * 1.4426950408889634f;
    dim3 grid(((q_len + FA_BQ - 1) / FA_BQ) * heads * bsz);
    // Measured and dropped (round 1): double-buffered K / V^T tiles with one barrier per tile: 69.5 us against 66.6 (the
    // store of tile t+1 sits in front of tile t's MFMAs); an 8-wave block whose wave pairs split every tile's keys and merge at the end (one
    // block per CU, heaviest-first dispatch) is balanced by construction but 79 us against 67 us for this kernel at
    // S = 2048: the loop is bound by instruction issue (PMC: ~310 VALU instructions per 32 MFMAs, MFMA pipe 21 % busy),
    // not by the causal imbalance.
    hipLaunchKernelGGL(flash_prefill_kernel, grid, dim3(256), 0, s, q, kc, vc, out, q_len, heads, kv_heads, max_seq,
                       past_len, c1);
    EXL_LAUNCH_CHECK();
    return 0;
}
