// Native single-token decode executor: one C call (capturable into one hipGraph) runs a whole Llama token step
// as 5 kernels per layer + 1 head kernel, instead of the ~20 launches per layer of the op-by-op path.
//
// It is the MI355X answer to the reference's decode loop (/root/reference/model.py:1053-1058 driving
// q4_attn -> ATen attention -> q4_attn_2 -> q4_mlp, i.e. q4_attn.cu:74-228 + q4_mlp.cu:100-199 + model.py:376-409):
// same arithmetic contract per op (fp16 values at the same points: normed x, q/k/v,
// attention output, gate/up, activation, residual stream), different decomposition:
//
//   K1 qkv      RMSNorm(hid) [layer 0: the embedding row] staged in LDS; q, k, v projections as ONE launch over the
//               three matrices -> fp16 q/k/v
//   K2 attn     RoPE(q), RoPE(k_new) in registers, k_new/v_new appended to the cache, split-KV attention over the
//               cache -> per (head, split): its own fp16 attention output + fp32 (max, sum); one split: the output itself
//   K3 o_proj   log-sum-exp merge of the splits while the activation image is built (a kernel boundary costs more than
//               the merge: the stand-alone K2b kernel serves the models whose attention output is wider than one block stages, hidden > 4096), then
//               hid += attn_out @ Wo   (residual added in the epilogue, fp16 residual stream updated in place)
//   K4 gate_up  RMSNorm(hid); one block computes the SAME 16 columns of gate and up -> act = silu(gate) * up (fp16)
//   K5 down     hid += act @ Wdown
//   K6 head     final RMSNorm; fp16 lm_head GEMV -> fp32 logits; advances the device-side position
//
// Every GEMV block streams one contiguous 16-column weight tile over the FULL K range (T16 layout, gemv_t16.h), so
// there is no split-K, no partial-sum slab, no atomics: results are bit-reproducible run to run.  The position is
// read from device memory, so one captured graph serves every context length.
#include <mutex>
#include "decode_args.h"

#include <vector>
#include <stdlib.h>

// One wave's share of one 16-column tile.  Everything here is wave-uniform (SGPRs): a weight load is
// (uniform base + uniform row-block offset) + lane * 16 bytes -- the saddr form, one VALU instruction for all of them.
struct DecUnit {
    const uint4* base;              // tile base (piece 0 of row-block 0)
    int rb0, rb1, rbsafe;           // row-block range [rb0, rb1); rbsafe: a valid row-block for clamped addresses
    int n0;                         // first column of the tile
};
__device__ __forceinline__ DecUnit dec_unit(const T16Matrix& m, int t, int rb_begin, int rb_end)
{
    DecUnit u;
    u.base = m.qw + (size_t) (uint32_t) t * (uint32_t) m.RB * 64u;
    u.rb0 = rb_begin; u.rb1 = rb_end < rb_begin ? rb_begin : rb_end;
    u.rbsafe = min(rb_begin, m.RB - 1);
    u.n0 = t * 16;
    return u;
}
// G16 entries of a unit: slot h holds the entry of row-block rb0 + 4h + rsub
template <int NSLOT>
__device__ __forceinline__ void dec_unit_entries(const T16Matrix& m, const DecUnit& u, int lane, uint32_t (&ent)[NSLOT])
{
    const int rsub = lane >> 4, n = u.n0 + (lane & 15);
#pragma unroll
    for (int h = 0; h < NSLOT; ++h) {
        const int rb = min(u.rb0 + 4 * h + rsub, m.RB - 1);
        const uint32_t e = t16_load_entry(m, t16_group_of_row(m, rb * 16), n);
        ent[h] = (u.rb0 + 4 * h + rsub < u.rb1) ? e : 0u;               // rows past the range: scale 0 -> contribute nothing
    }
}
template <int U, bool G16>
__device__ __forceinline__ void dec_unit_issue(const T16Matrix& m, const DecUnit& u, int pass, uint32_t lane, uint4 (&wv)[U],
                                               uint32_t (&entp)[U])
{
    // Branch-free on purpose: a conditional load makes the compiler drain the whole queue (s_waitcnt vmcnt(0)) at every join and
    // the pass-ahead pipelining is gone.  Slots past the wave's range re-read a valid row-block with scale 0 -- so the (U, NP)
    // configuration is chosen per shape to have (almost) no such slots (launch_dec_gemv_cfg).
#pragma unroll
    for (int i = 0; i < U; ++i) {
        const int rb = u.rb0 + pass * U + i;
        const int rbc = rb < u.rb1 ? rb : u.rbsafe;                     // uniform
        if constexpr (!G16) {
            const uint32_t e = t16_load_entry(m, t16_group_of_row(m, rbc * 16 + (int) (lane >> 4) * 4), u.n0 + (int) (lane & 15));
            entp[i] = rb < u.rb1 ? e : 0u;
        }
        wv[i] = nt_load16(u.base + (size_t) (uint32_t) rbc * 64u + lane);
    }
}
template <int U, bool G16, int NSLOT>
__device__ __forceinline__ void dec_unit_consume(const DecUnit& u, int pass, int lane, const uint4 (&wv)[U],
                                                 const uint32_t (&ent)[NSLOT], const uint32_t (&entp)[U], const uint4* xrow,
                                                 f32x4& c, const uint4* zpad)
{
    const uint32_t magic = t16_magic();
    const int col = lane & 15, rsub = lane >> 4;
    const bool live = col == 4 * rsub;                                  // group sizes 32 / 64: the lane that carries A row 4 * k-group (gemv_t16.h)
#pragma unroll
    for (int i = 0; i < U; ++i) {
        const int li = pass * U + i;
        const int rb = u.rb0 + li;
        const int rbc = rb < u.rb1 ? rb : u.rbsafe;                     // branch-free: out-of-range row-blocks carry scale 0
        uint32_t e;
        if constexpr (G16) e = (uint32_t) __shfl((int) ent[(li >> 2) < NSLOT ? (li >> 2) : 0], ((li & 3) << 4) | col, 64);
        else e = entp[i];
        if constexpr (G16) t16_rowblock<true>(wv[i], e, magic, xrow + rbc * 16 + rsub * 4, c);
        else t16_rowblock_groups(wv[i], e, magic, live ? xrow + rbc * 16 + rsub * 4 : zpad, c);
    }
}
// The (gathered) LDS image of the activation from its linear copy in LDS: packed row idx takes x[map[8 idx .. 8 idx + 7]]; the map
// travels as 16-bit indices (one 16-byte load per packed row: half the L2 traffic of the 32-bit x_map, which every block reads).
__device__ __forceinline__ void dec_stage_from_lds(const f16* xlin, const uint16_t* map, int R, uint4* xs, int tid, int nthreads)
{
    for (int idx = tid; idx < R; idx += nthreads) {
        const uint4 m = *(const uint4*) (map + idx * 8);
        f16x8 g;
        g[0] = xlin[m.x & 0xFFFFu]; g[1] = xlin[m.x >> 16]; g[2] = xlin[m.y & 0xFFFFu]; g[3] = xlin[m.y >> 16];
        g[4] = xlin[m.z & 0xFFFFu]; g[5] = xlin[m.z >> 16]; g[6] = xlin[m.w & 0xFFFFu]; g[7] = xlin[m.w >> 16];
        xs[idx] = __builtin_bit_cast(uint4, g);
    }
}
// PNORM: 0 plain vector, 1 RMSNorm(residual stream).  EMODE: 0 store fp16, 1 residual add, 2 silu(gate) * up pair.
// NV = 8-half vectors of the activation per thread (K <= 4096 * NV).
//
// PERSISTENT: the grid is one block per CU (not one per tile).  A block builds its activation image in LDS ONCE, then
// walks its share of the 16-column tiles; the 8 waves split the K range of each tile, and every wave keeps the loads of
// its NEXT step (next pass of this tile, or first pass of the next tile, plus that tile's scale/zero entries) in flight
// while it dequantises and multiplies the current one -- so after the start-up the weight stream never stops and the
// only per-tile synchronisation is one barrier for the 8-way partial-sum reduction (double-buffered by tile parity).
// Load order matters: the small L2-resident prologue loads (x, norm weight) are issued BEFORE the first weight loads
// (loads return in order).
// Phase attribution (probe builds only, scripts/probe_attn.sh): cycles since block start at 6 points, per kernel class.
#ifdef EXL_ATTN_PROBE
__device__ unsigned long long g_stream_probe[8 * 512 * 12];         // [class = PNORM * 2 + (EMODE == 2 ? 1 : EMODE)][block][point]
#define SP_CLK(i) sp_t[i] = __builtin_readcyclecounter()
#else
#define SP_CLK(i) do { } while (0)
#endif
// Register budget: two blocks per CU (4 waves per SIMD) need <= 128 VGPRs.  Most streams meet that on their own; the ones named
// here are a few registers above it and are capped (a handful of spills outside the inner loop; measured: 13B gate/up 22.8 ->
// 19.4 us, 65B 46.5 -> 43.5).  Capping the 6- and 7-deep streams costs 100+ bytes of scratch per lane and loses (65B down_proj
// 26.8 -> 34.6 us): those run one block per CU and are only chosen where the grid has no more blocks than CUs.
#define DEC_MIN_WAVES(U, NP, G16, PNORM, EMODE) ((((EMODE) == 2 && (G16) && (U) <= 5) || ((PNORM) == 0 && (U) == 4)) ? 4 : 2)
template <int U, int NP, bool G16, int PNORM, int EMODE, int NV>
__global__ __launch_bounds__(DEC_THREADS, DEC_MIN_WAVES(U, NP, G16, PNORM, EMODE)) void dec_stream_kernel(const DecGemvArgs a)
{
#ifdef EXL_ATTN_PROBE
    unsigned long long sp_t[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    const unsigned long long sp_t0 = __builtin_readcyclecounter();
#endif
    constexpr int NSLOT = G16 ? (U * NP + 3) / 4 : 1;
    constexpr int WPT = EMODE == 2 ? DEC_WAVES / 2 : DEC_WAVES;      // waves per tile
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // ---- 0. every kernel argument the start-up path needs, requested as ONE batch of scalar loads -----------------
    T16Matrix M0 = a.mat[0], M1 = a.mat[1], M2 = a.mat[2];
    dec_pin(M0); dec_pin(M1); dec_pin(M2);
    int K = M0.K, R = M0.R;
    const f16* a_vec = dec_pin_ptr(a.vec); const f16* a_norm_w = dec_pin_ptr(a.norm_w); const int64_t* a_tok = dec_pin_ptr(a.tok);
    int te0 = a.tile_end[0], te1 = a.tile_end[1], te2 = a.tile_end[2], a_nmat = a.nmat;
    int a_rbw = a.rb_per_wave, a_images = a.xs_images, nb = a.nblocks, units_lo = a.units_lo, units_rem = a.units_rem;
#ifdef EXL_DEC_ABLATE_BUILD                                           // measurement builds only (-DEXL_DEC_ABLATE_BUILD): run-time ablation levels
    int abl = a.ablate;
    DEC_PIN_S(abl);
#else
    constexpr int abl = 0;
#endif
    const uint16_t* g0 = dec_pin_ptr(a.map16[0]); const uint16_t* g1 = dec_pin_ptr(a.map16[1]); const uint16_t* g2 = dec_pin_ptr(a.map16[2]);
    DEC_PIN_S(K); DEC_PIN_S(R);
    DEC_PIN_S(te0); DEC_PIN_S(te1); DEC_PIN_S(te2); DEC_PIN_S(a_nmat); DEC_PIN_S(a_rbw); DEC_PIN_S(a_images);
    DEC_PIN_S(nb); DEC_PIN_S(units_lo); DEC_PIN_S(units_rem);
    SP_CLK(5);                                                       // kernel arguments have arrived
    const int RB = M0.RB;
    uint4* xs = (uint4*) smem;                                       // [xs_images][R]
    float* red = (float*) (smem + (size_t) a_images * R * 16);       // [2][DEC_WAVES][16] + [DEC_WAVES]
    constexpr int RED_FLOATS = 2 * DEC_WAVES * 16 + DEC_WAVES + 16;   // + 64 zero bytes: the A rows of the lanes that carry nothing (group sizes 32 / 64)
    const uint4* zpad = (const uint4*) (red + 2 * DEC_WAVES * 16 + DEC_WAVES);
    if (threadIdx.x < 16) red[2 * DEC_WAVES * 16 + DEC_WAVES + threadIdx.x] = 0.f;   // (read only after the image barrier)
    f16* xlin = (f16*) (smem + (size_t) a_images * R * 16 + RED_FLOATS * sizeof(float));   // [K], act-order only

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rsub = lane >> 4;
    const int nunits = EMODE == 2 ? te0 : (a_nmat == 1 ? te0 : a_nmat == 2 ? te1 : te2);
    const int b = blockIdx.x;
    const int n_my = units_lo + (b < units_rem ? 1 : 0);             // = ceil((nunits - b) / nb), divided on the host
    const bool remap = (nunits & 7) == 0 && (nb & 7) == 0;           // XCD x (= b % 8) walks one contiguous eighth of the tiles
    const int per = nunits >> 3;
    const int rb_lo = (wave % WPT) * a_rbw;
    const int rb_hi = min(RB, rb_lo + a_rbw);

    // unit i of this block -> (matrix, tile) of this wave
    auto describe = [&](int i, int& mi, int& tile) {
        const int v = b + i * nb;
        const int g = remap ? (v & 7) * per + (v >> 3) : v;
        mi = 0; tile = g;
        if constexpr (EMODE == 2) {
            mi = wave / WPT;                                         // waves 0-3: gate tile g, waves 4-7: up tile g
        } else {
            if (a_nmat > 1 && g >= te0) { mi = 1; tile = g - te0; }
            if (a_nmat > 2 && g >= te1) { mi = 2; tile = g - te1; }
        }
    };

    // ---- 1. prologue loads ------------------------------------------------------------------------------------
    const int nvec = K >> 3;
    const f16* src = a_vec;
    if constexpr (PNORM == 1) { if (a_tok) src = a_vec + (size_t) (*a_tok) * K; }
    uint4 xraw[NV], wraw[NV];
    constexpr int MS = PNORM == 3 ? DEC_MAX_NSPLIT : 1;
    uint4 praw[MS];                                                  // PNORM 3: this thread's 8 dims of every split's output (one vector at a time)
    float2 pml = make_float2(-INFINITY, 0.f);                        //          (max, sum) of split lane & 15 of this thread's head
    auto load_splits = [&](int i) {                                  // 16 consecutive 8-dim vectors = one head
        const int idx = tid + i * DEC_THREADS;
        const int ci = idx < nvec ? idx : 0;
        const int hd = ci >> 4, sp = lane & 15;
        pml = sp < a.att_nsplit ? *(const float2*) (a.att_ml + ((size_t) hd * a.att_nsplit + sp) * 2) : make_float2(-INFINITY, 0.f);
        // one per-lane base + uniform offsets (splits beyond nsplit re-read the last one: their coefficient is 0)
        const uint4* base = (const uint4*) (src + (size_t) hd * a.att_nsplit * 128 + (ci & 15) * 8);
#pragma unroll
        for (int sp2 = 0; sp2 < MS; ++sp2) praw[sp2] = base[min(sp2, a.att_nsplit - 1) * 16];
    };
    // PNORM 0 without a gather (o_proj after the merge kernel, down_proj): the image is a plain copy of the vector -> LDS-DMA
    // (global_load_lds_dwordx4: 1 KiB per wave instruction straight into LDS, no VGPR round trip).  The K = intermediate-size
    // kernels held 2 x NV x 4 registers for this copy, which pushed them over 128 VGPRs = one block per CU, i.e. a second
    // round for the 320-512 tile o_proj / down_proj of the 13B-65B models.
    // (PNORM 0 never gathers: its producers store the vector in the consumer's row order -- launch_dec_gemv checks.)
    constexpr bool dma_image = PNORM == 0;
    if constexpr (PNORM == 3) {
        load_splits(0);
    } else if constexpr (PNORM == 0) {
        {
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int idx0 = wave * 64 + i * DEC_THREADS;        // first packed row of this wave's 1 KiB piece (uniform)
                if (idx0 < nvec) {
                    const int idx = idx0 + lane;
                    const int ci = idx < nvec ? idx : 0;             // lanes past the end copy row 0 into the reduction scratch: harmless
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*) (src + ci * 8),
                                                     (__attribute__((address_space(3))) unsigned char*) (xs + idx0), 16, 0, 0);
                }
            }
        }
    } else {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int idx = tid + i * DEC_THREADS;
            const int ci = idx < nvec ? idx : 0;
            xraw[i] = make_uint4(0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u); wraw[i] = xraw[i];
            if (abl >= 4) continue;                                  // measurement only: no dependent activation load
            xraw[i] = *(const uint4*) (src + ci * 8);
            if constexpr (PNORM == 1) wraw[i] = *(const uint4*) (a_norm_w + ci * 8);
        }
    }
    SP_CLK(6);                                                       // activation / split loads issued
    // ---- 2. first unit's weight stream ----------------------------------------------------------------------
    uint4 wv0[U], wv1[U];
    uint32_t ep0[U], ep1[U];
    uint32_t entA[NSLOT], entB[NSLOT];
    DecUnit uA, uB;
    T16Matrix mA, mB;
    int miA = 0, miB = 0, tileA = 0, tileB = 0;
    float resA = 0.f, resB = 0.f;                                    // EMODE 1: residual value of this thread's column
    describe(0, miA, tileA);
    mA = dec_pick(M0, M1, M2, miA); mB = mA;
    uA = dec_unit(mA, tileA, rb_lo, rb_hi);
    SP_CLK(7);                                                       // first unit described
    // The first weight batch is issued before the activation has landed (waiting for it first measured slower on 7B in round 2:
    // 599 / 712 vs 619 / 731 tokens/s worst / best case; the switch is gone).
    dec_unit_issue<U, G16>(mA, uA, 0, lane, wv0, ep0);               // addresses: scalar arithmetic + one VALU
    SP_CLK(8);                                                       // first weight batch issued (entries / residual loads follow)
    if constexpr (G16) { if (abl < 3) dec_unit_entries<NSLOT>(mA, uA, lane, entA); }
    if constexpr (EMODE == 1) { if (tid < 16) resA = (float) a.res_in[tileA * 16 + tid]; }

    SP_CLK(0);                                                       // prologue loads + first weight batch issued
    // ---- 3. activation image (once per block) ---------------------------------------------------------------
    f16x8 xv[NV];
    if constexpr (PNORM == 1) {
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int idx = tid + i * DEC_THREADS;
            xv[i] = __builtin_bit_cast(f16x8, xraw[i]);
            if (idx < nvec) {
                if (a_tok && a.hid_copy && b == 0) *(f16x8*) (a.hid_copy + idx * 8) = xv[i];
#pragma unroll
                for (int j = 0; j < 8; ++j) { const float f = (float) xv[i][j]; ss = fmaf(f, f, ss); }
            }
        }
        ss = dec_wave_sum(ss);
        if (lane == 0) red[2 * DEC_WAVES * 16 + wave] = ss;
        __syncthreads();
        float total = 0.f;
#pragma unroll
        for (int i = 0; i < DEC_WAVES; ++i) total += red[2 * DEC_WAVES * 16 + i];
        const f16 rm = (f16) (1.0f / sqrtf(total * (1.0f / (float) K) + a.eps));
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const f16x8 nw = __builtin_bit_cast(f16x8, wraw[i]);
#pragma unroll
            for (int j = 0; j < 8; ++j) { const f16 t = xv[i][j] * rm; xv[i][j] = t * nw[j]; }
        }
    } else if constexpr (PNORM == 3) {
        // log-sum-exp merge of the attention splits (what the stand-alone dec_attn_merge_kernel does), per head inside its
        // 16-lane group: lane s holds (m_s, l_s); coefficient of split s = l_s e^(m_s - M) / sum of those.  One 8-dim vector
        // at a time: 16 x 16 bytes of split outputs per thread are the register budget.  With two vectors per thread (hidden >
        // 4096) the loop stays ROLLED and each merged vector goes straight to LDS: unrolled, hipcc keeps both vectors' 32 split
        // loads live and spills under the 128-register cap of two blocks per CU.
        auto merge_one = [&]() {
            float M = pml.x;
#pragma unroll
            for (int off = 1; off < 16; off <<= 1) M = fmaxf(M, __shfl_xor(M, off, 64));
            const float lw = pml.x > -INFINITY ? pml.y * __expf(pml.x - M) : 0.f;
            float L = lw;
#pragma unroll
            for (int off = 1; off < 16; off <<= 1) L += __shfl_xor(L, off, 64);
            const float coef = lw / L;
            float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int sp2 = 0; sp2 < MS; ++sp2) {
                const float cf = __shfl(coef, sp2, 16);
                const f16x8 o8 = __builtin_bit_cast(f16x8, praw[sp2]);
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] = fmaf((float) o8[j], cf, acc[j]);
            }
            f16x8 r;
#pragma unroll
            for (int j = 0; j < 8; ++j) r[j] = (f16) acc[j];
            return r;
        };
        if constexpr (NV == 1) {
            xv[0] = merge_one();
        } else {
#pragma unroll 1
            for (int i = 0; i < NV; ++i) {
                if (i) load_splits(i);                               // (vector 0 was requested in the prologue)
                const f16x8 r = merge_one();
                const int idx = tid + i * DEC_THREADS;
                if (idx < nvec) {
                    if (g0 != nullptr) *(f16x8*) (xlin + idx * 8) = r;
                    else               xs[idx] = __builtin_bit_cast(uint4, r);
                }
            }
        }
    }
    const bool gather = PNORM != 0 && g0 != nullptr;                 // all matrices of a launch agree (checked on the host)
    if constexpr (!(PNORM == 3 && NV > 1) && !dma_image) {
        {
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int idx = tid + i * DEC_THREADS;
                if (idx < nvec) {
                    if (gather) *(f16x8*) (xlin + idx * 8) = xv[i];
                    else        xs[idx] = __builtin_bit_cast(uint4, xv[i]);
                }
            }
        }
    }
    __syncthreads();
    if (gather) {                                                    // act-order: one image per matrix (each has its own x_map)
        if constexpr (EMODE == 2) {                                  // gate image built by waves 0-3, up image by waves 4-7
            const int mi = wave / WPT;
            dec_stage_from_lds(xlin, mi ? g1 : g0, R, xs + (size_t) mi * R, tid % (WPT * 64), WPT * 64);
        } else {
            for (int k = 0; k < a_images; ++k)
                dec_stage_from_lds(xlin, k == 0 ? g0 : k == 1 ? g1 : g2, R, xs + (size_t) k * R, tid, DEC_THREADS);
        }
        __syncthreads();
    }

    SP_CLK(1);                                                       // activation image staged
    // ---- 4. walk the tiles ----------------------------------------------------------------------------------
#define DEC_BUF(k) (((k) & 1) ? wv1 : wv0)
#define DEC_EP(k)  (((k) & 1) ? ep1 : ep0)
#define DEC_UNIT_BODY(P, uC, mC, miC, tileC, entC, resC, uN, mN, miN, tileN, entN, resN)                                    \
    {                                                                                                                       \
        const int i = 2 * j + P;                                                                                            \
        if (i >= n_my) break;                                                                                               \
        const bool have_next = i + 1 < n_my;                                                                                \
        if (have_next) { describe(i + 1, miN, tileN); mN = dec_pick(M0, M1, M2, miN); uN = dec_unit(mN, tileN, rb_lo, rb_hi); } \
        const uint4* xrow = xs + (a_images > 1 ? (size_t) miC * R : 0);                                                     \
        f32x4 c = {0.f, 0.f, 0.f, 0.f};                                                                                     \
        _Pragma("unroll") for (int p = 0; p < NP; ++p) {                                                                    \
            if (p + 1 < NP) {                                                                                               \
                if (uC.rb0 + (p + 1) * U < uC.rb1) dec_unit_issue<U, G16>(mC, uC, p + 1, lane, DEC_BUF(P * NP + p + 1), DEC_EP(P * NP + p + 1)); \
            } else if (have_next) {                                                                                         \
                if constexpr (G16) { if (abl < 3) dec_unit_entries<NSLOT>(mN, uN, lane, entN); }                            \
                if constexpr (EMODE == 1) { if (tid < 16) resN = (float) a.res_in[tileN * 16 + tid]; }                      \
                dec_unit_issue<U, G16>(mN, uN, 0, lane, DEC_BUF((P ^ 1) * NP), DEC_EP((P ^ 1) * NP));                       \
            }                                                                                                               \
            if (abl) { _Pragma("unroll") for (int q = 0; q < U; ++q) c[0] += __builtin_bit_cast(float, DEC_BUF(P * NP + p)[q].x ^ DEC_BUF(P * NP + p)[q].w); } \
            else if (uC.rb0 + p * U < uC.rb1) dec_unit_consume<U, G16, NSLOT>(uC, p, lane, DEC_BUF(P * NP + p), entC, DEC_EP(P * NP + p), xrow, c, zpad); \
        }                                                                                                                   \
        if (i == 0) SP_CLK(2);                                                                                              \
        float* rp = red + P * DEC_WAVES * 16;                                                                               \
        if constexpr (!G16) { c[0] += __shfl_xor(c[0], 16, 64); c[0] += __shfl_xor(c[0], 32, 64); }   /* the four k-groups of a column */ \
        if (lane < 16) rp[wave * 16 + lane] = c[0];                                                                         \
        __syncthreads();                                                                                                    \
        if (i == 0) SP_CLK(3);                                                                                              \
        if (tid < 16) {                                                                                                     \
            const int n = tileC * 16 + tid;                                                                                 \
            if constexpr (EMODE == 2) {                                                                                     \
                float g = 0.f, u = 0.f;                                                                                     \
                _Pragma("unroll") for (int k = 0; k < WPT; ++k) { g += rp[k * 16 + tid]; u += rp[(WPT + k) * 16 + tid]; }   \
                a.out[0][a.out_perm ? (int) a.out_perm[n] : n] = silu_mul_f16((f16) g, (f16) u);                           \
            } else {                                                                                                        \
                float v = 0.f;                                                                                              \
                _Pragma("unroll") for (int k = 0; k < DEC_WAVES; ++k) v += rp[k * 16 + tid];                                \
                if constexpr (EMODE == 0) a.out[miC][n] = (f16) v;                                                          \
                else a.hid_io[n] = (f16) (v + resC);                                                                        \
            }                                                                                                               \
        }                                                                                                                   \
    }
    for (int j = 0;; ++j) {
        DEC_UNIT_BODY(0, uA, mA, miA, tileA, entA, resA, uB, mB, miB, tileB, entB, resB)
        DEC_UNIT_BODY(1, uB, mB, miB, tileB, entB, resB, uA, mA, miA, tileA, entA, resA)
    }
    SP_CLK(4);                                                       // all units done
#ifdef EXL_ATTN_PROBE
    if (tid == 0 && b < 512) {
        unsigned long long* dst = g_stream_probe + ((size_t) (PNORM * 2 + (EMODE == 2 ? 1 : EMODE)) * 512 + b) * 12;
#pragma unroll
        for (int q = 0; q < 5; ++q) dst[q] = sp_t[q] - sp_t0;
        dst[6] = sp_t[5] - sp_t0;
        dst[5] = (unsigned long long) n_my;
        dst[7] = 1;
        dst[8] = sp_t[6] - sp_t0; dst[9] = sp_t[7] - sp_t0; dst[10] = sp_t[8] - sp_t0; dst[11] = 0;
    }
#endif
#undef DEC_UNIT_BODY
#undef DEC_BUF
#undef DEC_EP
}

// ---------------------------------------------------------------------------------------------------------------
// K2: RoPE + cache append + split-KV attention (head_dim 128, 16 lanes per key row, 16 key rows per step)
//
// One block = (head, split).  The first DEC_ATT_UN * 16 keys of the split -- the whole split up to context 2560 with 16
// splits -- are requested for K AND V before anything else happens, so the kernel pays ONE memory latency, not two
// (scores -> softmax -> PV with the V rows already in registers).  Longer splits loop over further chunks.
// ---------------------------------------------------------------------------------------------------------------
#define DEC_ATT_CHUNK 160        // keys per pass of a block
#define DEC_ATT_AHEAD 5          // rows per thread in flight during the score phase
// Phase attribution (probe builds only, scripts/probe_attn.sh): cycles summed over blocks at 7 points of the kernel.
#ifdef EXL_ATTN_PROBE
__device__ unsigned long long g_attn_probe[512 * 8];                // [block][point]; read out by exl_debug_attn_probe
#define AP_CLK(i) ap_t[i] = __builtin_readcyclecounter()
#else
#define AP_CLK(i) do { } while (0)
#endif
// SHORT (the one-split bucket, context <= 160): rows past the context are skipped with block-uniform branches; in the long
// buckets every split is full and the same branches only break up the load / score interleave (measured: 9.5 -> 11.7 us).
// NW = waves per block (EXL_DEC_ATTN_WAVES): 4 -> 16 key rows per step, 10 rows per thread and chunk, 5 in flight;
//                                            8 -> 32 key rows per step,  5 rows per thread and chunk, all 5 K rows in flight
// (twice the waves per CU at the same bytes in flight per thread).
// The one-split bucket (context <= 160; the benchmark's "best case" runs here): round 2's kernel with, since round 4, the reductions of the
// kernel below (DPP row sums instead of 100 ds_bpermute round trips per thread, barriers that do not drain the V loads).  The chunked form
// below measured 0.3 us slower per launch here and 35 us per token slower inside the graph (same box, A/B of the two libraries:
// 806 -> 773 tokens/s at context 4), where there is nothing to speculate on and one split to balance.
template <int NW>
__global__ __launch_bounds__(NW * 64) void dec_attn_short_kernel(const f16* __restrict__ q, const f16* __restrict__ k_new,
                                                       const f16* __restrict__ v_new, f16* __restrict__ kc,
                                                       f16* __restrict__ vc, const f16* __restrict__ sin,
                                                       const f16* __restrict__ cos, float* __restrict__ partial,
                                                       const int32_t* __restrict__ pos_dev, int heads, int kv_heads,
                                                       int max_seq, int nsplit, float scale, f16* __restrict__ direct_out,
                                                       const uint16_t* __restrict__ out_perm)
{
    constexpr bool SHORT = true;
    constexpr int HD = 128, LPK = 16, KPI = NW * 4, UN = DEC_ATT_CHUNK / 16, NT = NW * 64;   // 10 rows per thread and pass: 160 keys with 4 waves, 320 with 8
    __shared__ float sc[DEC_ATT_MAX_KEYS];
    __shared__ float red[NW][HD + 1];
    __shared__ float stat[2 * NW];

    // 1-D grid; block id -> (head, split) such that every split of head h runs on XCD h % 8 (block b runs on XCD b % 8,
    // observed, speed only): the merge block of head h (XCD h % 8 as well) then finds the partials in its own L2.
    int h, split;
    if ((heads & 7) == 0) {
        const int r = blockIdx.x & 7, j = blockIdx.x >> 3;
        h = r + 8 * (j / nsplit);
        split = j % nsplit;
    } else {
        h = blockIdx.x / nsplit;
        split = blockIdx.x % nsplit;
    }
    const int tid = threadIdx.x;
    const int d8 = tid & 15, ks = tid >> 4;
#ifdef EXL_ATTN_PROBE
    unsigned long long ap_t[7];
    const unsigned long long ap_t0 = __builtin_readcyclecounter();
#endif
    const int kvh = h / (heads / kv_heads);
    f16* kbase = kc + (size_t) kvh * max_seq * HD + d8 * 8;
    f16* vbase = vc + (size_t) kvh * max_seq * HD + d8 * 8;
    const int past = *pos_dev;
    const int vis = past + 1;
    int L = (vis + nsplit - 1) / nsplit;
    L = (L + 15) & ~15;
    const int s0 = min(vis, split * L), s1 = min(vis, s0 + L);
    const int nkeys = s1 - s0;

    AP_CLK(0);                                                       // position known
    // ---- every load of the first chunk up front: small ones first, then K rows, then V rows -------------------------
    // (Round 4, measured and not kept: q, the new k / v and the first K rows of every thread requested BEFORE the position is read --
    // with one split their addresses do not depend on it -- made no difference to the replay rate at context 4, scripts/gpu_calls/r04r.sh.)
    const f16x8 sn = *(const f16x8*) (sin + (size_t) past * HD + d8 * 8);
    const f16x8 cs = *(const f16x8*) (cos + (size_t) past * HD + d8 * 8);
    const f16x8 qraw = *(const f16x8*) (q + (size_t) h * HD + d8 * 8);
    const f16x8 kraw = *(const f16x8*) (k_new + (size_t) kvh * HD + d8 * 8);
    const f16x8 vn = *(const f16x8*) (v_new + (size_t) kvh * HD + d8 * 8);
    // K and V rows of the first chunk (the whole split up to 160 keys) go through registers with a ROLLING prefetch of
    // DEC_ATT_AHEAD rows per thread: a wave that issues all 20 row loads at once sits in the issue queue until most of the
    // data is back (the CU's memory queue is full) and only then starts on the scores; with a bounded number in flight the
    // score of row u is computed while rows u + AHEAD .. stream in, and the V rows arrive during scores and softmax.
    f16x8 kv0[UN], vv0[UN];
    auto load_k0 = [&](int u) {
        const int j = u * KPI + ks;
        kv0[u] = (f16x8){0, 0, 0, 0, 0, 0, 0, 0};
        if (j < nkeys) kv0[u] = *(const f16x8*) (kbase + (size_t) (s0 + j) * HD);
    };
    auto load_v0 = [&](int u) {
        const int j = u * KPI + ks;
        vv0[u] = (f16x8){0, 0, 0, 0, 0, 0, 0, 0};
        if (j < nkeys) vv0[u] = *(const f16x8*) (vbase + (size_t) (s0 + j) * HD);
    };
#pragma unroll
    for (int u = 0; u < DEC_ATT_AHEAD; ++u) load_k0(u);
    __builtin_amdgcn_sched_barrier(0);
    AP_CLK(1);                                                       // loads issued
    // RoPE on q and on the new key: element d pairs with d +- 64, i.e. lane d8 with lane d8 ^ 8
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
    const f16x8 qr = rope8(qraw);
    const f16x8 kr = rope8(kraw);
    if (split == 0 && (h % (heads / kv_heads)) == 0 && ks == 0) {
        *(f16x8*) (kbase + (size_t) past * HD) = kr;
        *(f16x8*) (vbase + (size_t) past * HD) = vn;
    }
    float qf[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) qf[j] = (float) qr[j] * scale;

    AP_CLK(2);                                                       // q / k / sin / cos landed, RoPE done
    // ---- scores ---------------------------------------------------------------------------------------------------
    float mx = -INFINITY;
    auto score_one = [&](int j0, int u, const f16x8& kv) {
        const int j = j0 + u * KPI + ks;
        const f16x8 kk = (s0 + j == past) ? kr : kv;                   // the new key never comes from memory
        float dot = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) dot = fmaf(qf[e], (float) kk[e], dot);
        dot = dec_row_sum(dot);                                        // 16 lanes of a key row: DPP, no LDS-pipe round trips
        if (j < nkeys) {
            if (d8 == 0) sc[j] = dot;
            mx = fmaxf(mx, dot);
        }
    };
    auto score_chunk = [&](int j0, const f16x8 (&kv)[UN]) {
#pragma unroll
        for (int u = 0; u < UN; ++u) score_one(j0, u, kv[u]);
    };
#pragma unroll
    for (int u = 0; u < UN; ++u) {                                     // first chunk: row u + AHEAD of the K-then-V list is requested, row u scored
        const int t = u + DEC_ATT_AHEAD;
        if (t < UN) load_k0(t);
        else if (t - UN < UN) load_v0(t - UN);
        if (!SHORT || u * KPI < nkeys) score_one(0, u, kv0[u]);
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int u = DEC_ATT_AHEAD; u < UN; ++u) load_v0(u);               // the rest of V streams in during the softmax
    __builtin_amdgcn_sched_barrier(0);
    for (int j0 = KPI * UN; j0 < nkeys; j0 += KPI * UN) {
        f16x8 kv[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int j = j0 + u * KPI + ks;
            kv[u] = *(const f16x8*) (kbase + (size_t) (s0 + (j < nkeys ? j : 0)) * HD);
        }
        score_chunk(j0, kv);
    }
    AP_CLK(3);                                                       // K rows landed, scores done
    mx = dec_row_max(mx);
    mx = fmaxf(fmaxf(dec_lane(mx, 0), dec_lane(mx, 16)), fmaxf(dec_lane(mx, 32), dec_lane(mx, 48)));
    if ((tid & 63) == 0) stat[tid >> 6] = mx;
    dec_lds_barrier();                                               // LDS hand-off only: the V rows stay in flight across it (__syncthreads drains vmcnt)
    mx = stat[0];
#pragma unroll
    for (int w = 1; w < NW; ++w) mx = fmaxf(mx, stat[w]);
    float lsum = 0.f;
    for (int j = tid; j < nkeys; j += NT) {
        const float p = __expf(sc[j] - mx);
        sc[j] = p;
        lsum += p;
    }
    lsum = dec_wave_sum(lsum);
    if ((tid & 63) == 0) stat[NW + (tid >> 6)] = lsum;
    dec_lds_barrier();
    lsum = stat[NW];
#pragma unroll
    for (int w = 1; w < NW; ++w) lsum += stat[NW + w];

    AP_CLK(4);                                                       // softmax done
    // ---- P V ------------------------------------------------------------------------------------------------------
    float o[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = 0.f;
    auto pv_chunk = [&](int j0, const f16x8 (&vv)[UN]) {
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int j = j0 + u * KPI + ks;
            if (SHORT && j0 + u * KPI >= nkeys) continue;
            const float p = j < nkeys ? sc[j] : 0.f;
            const f16x8 v8 = (s0 + j == past) ? vn : vv[u];
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = fmaf(p, (float) v8[e], o[e]);
        }
    };
    pv_chunk(0, vv0);
    for (int j0 = KPI * UN; j0 < nkeys; j0 += KPI * UN) {
        f16x8 vv[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int j = j0 + u * KPI + ks;
            vv[u] = *(const f16x8*) (vbase + (size_t) (s0 + (j < nkeys ? j : 0)) * HD);
        }
        pv_chunk(j0, vv);
    }
    AP_CLK(5);                                                       // V rows landed, P V done
    // the 4 key rows of a wave first (v_permlane16_swap / v_permlane32_swap), then one row per wave through LDS
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = dec_rows_sum(o[e]);
    if ((tid & 63) < 16) {
#pragma unroll
        for (int e = 0; e < 8; ++e) red[tid >> 6][d8 * 8 + e] = o[e];
    }
    __syncthreads();
    // Several splits: each writes its OWN attention output (normalised by its own sum: a convex combination of V rows,
    // safe in fp16) plus (max, sum); the o_proj kernel combines them while it builds its activation image.
    f16* po = (f16*) partial + ((size_t) h * nsplit + split) * HD;
    float* pml = partial + (size_t) heads * nsplit * (HD / 2) + ((size_t) h * nsplit + split) * 2;
    if (tid < HD) {
        float v = 0.f;
#pragma unroll
        for (int s = 0; s < NW; ++s) v += red[s][tid];
        const f16 r = (f16) (nkeys > 0 ? v / lsum : 0.f);
        if (direct_out) direct_out[out_perm ? (int) out_perm[h * HD + tid] : h * HD + tid] = r;   // a single split: this IS the attention output
                                                                        // (stored where an act-order o_proj reads it linearly)
        else po[tid] = r;
    }
    if (tid == 0 && !direct_out) {
        pml[0] = nkeys > 0 ? mx : -INFINITY;
        pml[1] = nkeys > 0 ? lsum : 0.f;
    }
    AP_CLK(6);                                                       // partials written
#ifdef EXL_ATTN_PROBE
    if (tid == 0 && blockIdx.x < 512) {
#pragma unroll
        for (int i = 0; i < 7; ++i) g_attn_probe[blockIdx.x * 8 + i] = ap_t[i] - ap_t0;
        g_attn_probe[blockIdx.x * 8 + 7] = 1;
    }
#endif
}

template <bool SHORT, int NW>
__global__ __launch_bounds__(NW * 64) void dec_attn_kernel(const f16* __restrict__ q, const f16* __restrict__ k_new,
                                                       const f16* __restrict__ v_new, f16* __restrict__ kc,
                                                       f16* __restrict__ vc, const f16* __restrict__ sin,
                                                       const f16* __restrict__ cos, float* __restrict__ partial,
                                                       const int32_t* __restrict__ pos_dev, int heads, int kv_heads,
                                                       int max_seq, int nsplit, float scale, f16* __restrict__ direct_out,
                                                       const uint16_t* __restrict__ out_perm)
{
    constexpr int HD = 128, LPK = 16, KPI = NW * 4, UN = DEC_ATT_CHUNK / 16, NT = NW * 64;   // 10 rows per thread and pass: 160 keys with 4 waves, 320 with 8
    __shared__ float sc[DEC_ATT_MAX_KEYS];
    __shared__ float red[NW][HD];
    __shared__ float stat[2 * NW];

    // 1-D grid; block id -> (head, split) such that every split of head h runs on XCD h % 8 (block b runs on XCD b % 8,
    // observed, speed only): the merge block of head h (XCD h % 8 as well) then finds the partials in its own L2.
    int h, split;
    if ((heads & 7) == 0) {
        const int r = blockIdx.x & 7, j = blockIdx.x >> 3;
        h = r + 8 * (j / nsplit);
        split = j % nsplit;
    } else {
        h = blockIdx.x / nsplit;
        split = blockIdx.x % nsplit;
    }
    const int tid = threadIdx.x;
    const int d8 = tid & 15, ks = tid >> 4;
#ifdef EXL_ATTN_PROBE
    unsigned long long ap_t[7];
    const unsigned long long ap_t0 = __builtin_readcyclecounter();
#endif
    // Keys are dealt to the splits in CHUNKS of KPI (16 / 32) consecutive rows, round robin: chunk c belongs to split c % nsplit.
    // The addresses of a block's first rows therefore do not depend on the position -- only how many of its chunks are visible
    // does -- and the first DEC_ATT_AHEAD K rows are requested before the position has arrived (round 3 phase stamps of the
    // contiguous form: first K request 4,300 cycles into a 17,800-cycle block, behind the kernel arguments and then the position,
    // two dependent round trips).  Rows beyond the context are in-bounds reads of the cache (clamped to max_seq - 1) and masked.
    // Every split still gets the same number of chunks (+- 1) at every position.
    const int past_raw = *pos_dev;
    const int kvh = h / (heads / kv_heads);
    f16* kbase = kc + (size_t) kvh * max_seq * HD + d8 * 8;
    f16* vbase = vc + (size_t) kvh * max_seq * HD + d8 * 8;
    auto key_of = [&](int u) { return (u * nsplit + split) * KPI + ks; };       // global key index of this thread's row u

    AP_CLK(0);
    const f16x8 qraw = *(const f16x8*) (q + (size_t) h * HD + d8 * 8);
    const f16x8 kraw = *(const f16x8*) (k_new + (size_t) kvh * HD + d8 * 8);
    const f16x8 vn = *(const f16x8*) (v_new + (size_t) kvh * HD + d8 * 8);
    f16x8 kv0[UN], vv0[UN];
    if constexpr (!SHORT) {
#pragma unroll
        for (int u = 0; u < DEC_ATT_AHEAD; ++u)                        // speculative: before the position is known
            kv0[u] = *(const f16x8*) (kbase + (size_t) min(key_of(u), max_seq - 1) * HD);
        __builtin_amdgcn_sched_barrier(0);
    }
    const int past = past_raw;
    const int vis = past + 1;
    const int nchunk = vis > split * KPI ? (vis - split * KPI + nsplit * KPI - 1) / (nsplit * KPI) : 0;   // chunks of this split with a visible key
    const int nkeys = nchunk * KPI;                                    // local key slots in use (the last chunk may be partly masked)
    const f16x8 sn = *(const f16x8*) (sin + (size_t) past * HD + d8 * 8);
    const f16x8 cs = *(const f16x8*) (cos + (size_t) past * HD + d8 * 8);
    auto load_k0 = [&](int u) {
        kv0[u] = (f16x8){0, 0, 0, 0, 0, 0, 0, 0};
        if (key_of(u) < vis) kv0[u] = *(const f16x8*) (kbase + (size_t) key_of(u) * HD);
    };
    auto load_v0 = [&](int u) {
        vv0[u] = (f16x8){0, 0, 0, 0, 0, 0, 0, 0};
        if (key_of(u) < vis) vv0[u] = *(const f16x8*) (vbase + (size_t) key_of(u) * HD);
    };
    if constexpr (SHORT) {                                           // one split of <= 160 keys: only the rows that exist (a context of 4 has one)
#pragma unroll
        for (int u = 0; u < DEC_ATT_AHEAD; ++u) load_k0(u);
        __builtin_amdgcn_sched_barrier(0);
    }
    AP_CLK(1);                                                       // first K rows requested, position known
    // RoPE on q and on the new key: element d pairs with d +- 64, i.e. lane d8 with lane d8 ^ 8
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
    const f16x8 qr = rope8(qraw);
    const f16x8 kr = rope8(kraw);
    if (split == 0 && (h % (heads / kv_heads)) == 0 && ks == 0) {
        *(f16x8*) (kbase + (size_t) past * HD) = kr;
        *(f16x8*) (vbase + (size_t) past * HD) = vn;
    }
    float qf[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) qf[j] = (float) qr[j] * scale;

    AP_CLK(2);                                                       // q / k / sin / cos landed, RoPE done
    // ---- scores ---------------------------------------------------------------------------------------------------
    float mx = -INFINITY;
    auto score_one = [&](int uabs, const f16x8& kv) {                  // uabs: row index of this thread counted over all passes
        const int g = key_of(uabs);
        const f16x8 kk = (g == past) ? kr : kv;                        // the new key never comes from memory
        float dot = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) dot = fmaf(qf[e], (float) kk[e], dot);
#ifdef EXL_ATTN_VAR_SHFL
#pragma unroll
        for (int off = 1; off < LPK; off <<= 1) dot += __shfl_xor(dot, off, 64);
#else
        dot = dec_row_sum(dot);                                        // 16 lanes of a key row: DPP, no LDS-pipe round trips
#endif
        if (g < vis) {
            if (d8 == 0) sc[uabs * KPI + ks] = dot;
            mx = fmaxf(mx, dot);
        }
    };
#pragma unroll
    for (int u = 0; u < UN; ++u) {                                     // first pass: row u + AHEAD of the K-then-V list is requested, row u scored
        const int t = u + DEC_ATT_AHEAD;
        if (t < UN) load_k0(t);
        else if (t - UN < UN) load_v0(t - UN);
        if (!SHORT || u < nchunk) score_one(u, kv0[u]);
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int u = DEC_ATT_AHEAD; u < UN; ++u) load_v0(u);               // the rest of V streams in during the softmax
    __builtin_amdgcn_sched_barrier(0);
    for (int u0 = UN; u0 < nchunk; u0 += UN) {                         // splits longer than one pass (UN chunks)
        f16x8 kv[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) kv[u] = *(const f16x8*) (kbase + (size_t) min(key_of(u0 + u), max_seq - 1) * HD);
#pragma unroll
        for (int u = 0; u < UN; ++u) score_one(u0 + u, kv[u]);
    }
    AP_CLK(3);                                                       // K rows landed, scores done
    mx = dec_row_max(mx);
    mx = fmaxf(fmaxf(dec_lane(mx, 0), dec_lane(mx, 16)), fmaxf(dec_lane(mx, 32), dec_lane(mx, 48)));
    if ((tid & 63) == 0) stat[tid >> 6] = mx;
#ifdef EXL_ATTN_VAR_SYNC
    __syncthreads();
#else
    dec_lds_barrier();                                               // (the V rows stay in flight across these barriers)
#endif
    mx = stat[0];
#pragma unroll
    for (int w = 1; w < NW; ++w) mx = fmaxf(mx, stat[w]);
    float lsum = 0.f;
    for (int j = tid; j < nkeys; j += NT) {                            // local slot j = chunk (j / KPI) of this split, row j % KPI
        const bool live = ((j / KPI) * nsplit + split) * KPI + (j % KPI) < vis;
        const float p = live ? __expf(sc[j] - mx) : 0.f;
        sc[j] = p;
        lsum += p;
    }
    lsum = dec_wave_sum(lsum);
    if ((tid & 63) == 0) stat[NW + (tid >> 6)] = lsum;
#ifdef EXL_ATTN_VAR_SYNC
    __syncthreads();
#else
    dec_lds_barrier();
#endif
    lsum = stat[NW];
#pragma unroll
    for (int w = 1; w < NW; ++w) lsum += stat[NW + w];

    AP_CLK(4);                                                       // softmax done
    // ---- P V ------------------------------------------------------------------------------------------------------
    float o[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = 0.f;
    auto pv_one = [&](int uabs, const f16x8& vv) {
        const int g = key_of(uabs);
        const float p = g < vis ? sc[uabs * KPI + ks] : 0.f;
        const f16x8 v8 = (g == past) ? vn : vv;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = fmaf(p, (float) v8[e], o[e]);
    };
#pragma unroll
    for (int u = 0; u < UN; ++u) {
        if (SHORT && u >= nchunk) continue;
        pv_one(u, vv0[u]);
    }
    for (int u0 = UN; u0 < nchunk; u0 += UN) {
        f16x8 vv[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) vv[u] = *(const f16x8*) (vbase + (size_t) min(key_of(u0 + u), max_seq - 1) * HD);
#pragma unroll
        for (int u = 0; u < UN; ++u) pv_one(u0 + u, vv[u]);
    }
    AP_CLK(5);                                                       // V rows landed, P V done
    // the 4 key rows of a wave first (two cross-row exchanges per value), then one row per wave through LDS
#ifdef EXL_ATTN_VAR_SHFL
#pragma unroll
    for (int e = 0; e < 8; ++e) { o[e] += __shfl_xor(o[e], 16, 64); o[e] += __shfl_xor(o[e], 32, 64); }
#else
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = dec_rows_sum(o[e]);
#endif
    if ((tid & 63) < 16) {
#pragma unroll
        for (int e = 0; e < 8; ++e) red[tid >> 6][d8 * 8 + e] = o[e];
    }
    __syncthreads();
    // Several splits: each writes its OWN attention output (normalised by its own sum: a convex combination of V rows,
    // safe in fp16) plus (max, sum); the o_proj kernel combines them while it builds its activation image.
    f16* po = (f16*) partial + ((size_t) h * nsplit + split) * HD;
    float* pml = partial + (size_t) heads * nsplit * (HD / 2) + ((size_t) h * nsplit + split) * 2;
    if (tid < HD) {
        float v = 0.f;
#pragma unroll
        for (int s = 0; s < NW; ++s) v += red[s][tid];
        const f16 r = (f16) (nkeys > 0 ? v / lsum : 0.f);
        if (direct_out) direct_out[out_perm ? (int) out_perm[h * HD + tid] : h * HD + tid] = r;   // a single split: this IS the attention output
                                                                        // (stored where an act-order o_proj reads it linearly)
        else po[tid] = r;
    }
    if (tid == 0 && !direct_out) {
        pml[0] = nkeys > 0 ? mx : -INFINITY;
        pml[1] = nkeys > 0 ? lsum : 0.f;
    }
    AP_CLK(6);                                                       // partials written
#ifdef EXL_ATTN_PROBE
    if (tid == 0 && blockIdx.x < 512) {
#pragma unroll
        for (int i = 0; i < 7; ++i) g_attn_probe[blockIdx.x * 8 + i] = ap_t[i] - ap_t0;
        g_attn_probe[blockIdx.x * 8 + 7] = 1;
    }
#endif
}
#ifdef EXL_ATTN_PROBE
extern "C" int exl_debug_stream_probe(int cls, unsigned long long* out12)    // sums over blocks; out12[7] = block count, out12[5] = units
{
    static unsigned long long h[512 * 12];
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_stream_probe), sizeof(h), (size_t) cls * sizeof(h)) != hipSuccess) return -1;
    for (int i = 0; i < 12; ++i) out12[i] = 0;
    for (int b = 0; b < 512; ++b)
        for (int i = 0; i < 12; ++i) out12[i] += h[b * 12 + i];
    return 0;
}
extern "C" int exl_debug_attn_probe(unsigned long long* out8)         // sums over the blocks of the LAST launch; out8[7] = block count
{
    static unsigned long long h[512 * 8];
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_attn_probe), sizeof(h)) != hipSuccess) return -1;
    for (int i = 0; i < 8; ++i) out8[i] = 0;
    for (int b = 0; b < 512; ++b)
        for (int i = 0; i < 8; ++i) out8[i] += h[b * 8 + i];
    return 0;
}
#endif

// ---------------------------------------------------------------------------------------------------------------
// K2b: merge the split-KV partials of one head -> fp16 attention output (the value the reference's ATen attention
// rounds to fp16 before o_proj, model.py:407-409)
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(128) void dec_attn_merge_kernel(const float* __restrict__ partial, f16* __restrict__ out, int nsplit, int heads,
                                                             const uint16_t* __restrict__ out_perm)
{
    const int h = blockIdx.x, d = threadIdx.x;
    const f16* po = (const f16*) partial + (size_t) h * nsplit * 128;
    const float2* pml = (const float2*) (partial + (size_t) heads * nsplit * 64) + (size_t) h * nsplit;
    // every load up front, no data-dependent control flow: one memory round trip
    float2 ml[DEC_MAX_NSPLIT];
    f16 os[DEC_MAX_NSPLIT];
#pragma unroll
    for (int s = 0; s < DEC_MAX_NSPLIT; ++s) {
        const int cs = s < nsplit ? s : 0;
        ml[s] = pml[cs];
        os[s] = po[cs * 128 + d];
    }
    float M = -INFINITY;
#pragma unroll
    for (int s = 0; s < DEC_MAX_NSPLIT; ++s) if (s < nsplit) M = fmaxf(M, ml[s].x);
    float L = 0.f, o = 0.f;
#pragma unroll
    for (int s = 0; s < DEC_MAX_NSPLIT; ++s) {
        const float w = (s < nsplit && ml[s].x > -INFINITY) ? ml[s].y * __expf(ml[s].x - M) : 0.f;
        L += w;
        o = fmaf((float) os[s], w, o);
    }
    out[out_perm ? (int) out_perm[h * 128 + d] : h * 128 + d] = (f16) (o / L);
}

// act-order maps of one matrix as 16-bit indices: map16 = x_map (gather: x'[c] = x[x_map[c]]), inv16 = its inverse (a producer
// that stores column n at inv16[n] hands the consumer an already gathered vector)
__global__ void dec_map16_kernel(const uint32_t* __restrict__ x_map, uint16_t* __restrict__ map16, uint16_t* __restrict__ inv16, int K)
{
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= K) return;
    const uint32_t src = x_map[c];
    map16[c] = (uint16_t) src;
    inv16[src] = (uint16_t) c;
}

__global__ void dec_ident16_kernel(uint16_t* __restrict__ m, int K)
{
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c < K) m[c] = (uint16_t) c;
}

// ---------------------------------------------------------------------------------------------------------------
// K6: final RMSNorm + fp16 lm_head GEMV (one wave per vocabulary row, 128-bit loads) -> fp32 logits
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void dec_head_kernel(const f16* __restrict__ hid, const f16* __restrict__ norm_w, float eps, int h,
                                                       const f16* __restrict__ lm_head, int vocab,
                                                       float* __restrict__ logits, int rows_per_block,
                                                       int32_t* __restrict__ pos_dev, int advance, float2* __restrict__ blk_best)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    f16* xlin = (f16*) smem;
    float* red4 = (float*) (smem + (size_t) h * 2);
    const int tid = threadIdx.x;
    const int nvec = h >> 3;
    float ss = 0.f;
    for (int i = tid; i < nvec; i += 256) {
        const f16x8 v = *(const f16x8*) (hid + i * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) { const float t = (float) v[j]; ss = fmaf(t, t, ss); }
        *(f16x8*) (xlin + i * 8) = v;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) ss += __shfl_xor(ss, off, 64);
    if ((tid & 63) == 0) red4[tid >> 6] = ss;
    __syncthreads();
    const float total = red4[0] + red4[1] + red4[2] + red4[3];
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
    float best = -INFINITY;                                          // this wave's largest logit and its row (greedy generation)
    int best_row = 0x7fffffff;
    for (int r = wave; r < rows_per_block; r += 4) {
        const int row = row0 + r;
        if (row >= vocab) break;
        const f16* wr = lm_head + (size_t) row * h;
        float acc = 0.f;
        for (int i = lane; i < nvec; i += 64) {
            const uint4 wv = nt_load16(wr + i * 8);
            const uint4 xv = *(const uint4*) (xlin + i * 8);
            acc = __builtin_amdgcn_fdot2(t16_h2(wv.x), t16_h2(xv.x), acc, false);
            acc = __builtin_amdgcn_fdot2(t16_h2(wv.y), t16_h2(xv.y), acc, false);
            acc = __builtin_amdgcn_fdot2(t16_h2(wv.z), t16_h2(xv.z), acc, false);
            acc = __builtin_amdgcn_fdot2(t16_h2(wv.w), t16_h2(xv.w), acc, false);
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
        const float lg = (float) (f16) acc;                     // nn.Linear in fp16, then .float() (model.py:1077-1080)
        if (lane == 0) logits[row] = lg;
        if (lg > best) { best = lg; best_row = row; }           // rows ascend: the lowest index wins a tie
    }
    if (blk_best) {
        __syncthreads();                                       // red4 (norm statistics) is free again
        if (lane == 0) { red4[wave] = best; red4[4 + wave] = __builtin_bit_cast(float, best_row); }
        __syncthreads();
        if (tid == 0) {
#pragma unroll
            for (int w = 1; w < 4; ++w) {
                const int ir = __builtin_bit_cast(int, red4[4 + w]);
                if (red4[w] > best || (red4[w] == best && ir < best_row)) { best = red4[w]; best_row = ir; }
            }
            blk_best[blockIdx.x] = make_float2(best, __builtin_bit_cast(float, best_row));
        }
    }
    if (advance && blockIdx.x == 0 && tid == 0) *pos_dev += 1;
}

// ---------------------------------------------------------------------------------------------------------------
// K7 (greedy generation only): argmax of the logits -> the next token, written where the next step reads its input and
// into history[position of that token].  Lowest index wins ties; NaNs never win.  The head kernel leaves (largest logit,
// row) per block, so this kernel only scans ~1000 candidates (scanning the 128 KB of logits from one block took 14 us).
// Replaces the host-driven torch.argmax + two small copies per token of the reference's loop
// (test_benchmark_inference.py:188-191).
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void dec_argmax_kernel(const float2* __restrict__ blk_best, int nblk, int64_t* __restrict__ token_io,
                                                          int64_t* __restrict__ history, const int32_t* __restrict__ pos_dev)
{
    __shared__ float bv[16];
    __shared__ int bi[16];
    const int tid = threadIdx.x;
    float best = -INFINITY;
    int idx = 0x7fffffff;
    for (int i = tid; i < nblk; i += 1024) {                             // (largest logit, its row) of every head-kernel block
        const float2 e = blk_best[i];
        const int ei = __builtin_bit_cast(int, e.y);
        if (e.x > best || (e.x == best && ei < idx)) { best = e.x; idx = ei; }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float ov = __shfl_xor(best, off, 64);
        const int oi = __shfl_xor(idx, off, 64);
        if (ov > best || (ov == best && oi < idx)) { best = ov; idx = oi; }
    }
    if ((tid & 63) == 0) { bv[tid >> 6] = best; bi[tid >> 6] = idx; }
    __syncthreads();
    if (tid == 0) {
#pragma unroll
        for (int w = 1; w < 16; ++w)
            if (bv[w] > best || (bv[w] == best && bi[w] < idx)) { best = bv[w]; idx = bi[w]; }
        if (idx == 0x7fffffff) idx = 0;                                  // all NaN / -inf: token 0
        *token_io = idx;
        if (history) history[*pos_dev] = idx;                            // the head kernel has already advanced the position
    }
}

// ---------------------------------------------------------------------------------------------------------------
// LoRA inside the token step (reference: exllama_ext.cpp:245-324 q4_matmul_lora and the LoRA operands of q4_attn / q4_mlp,
// :424-602): out = W x + (x A) B per projection.  Two small launches behind each GEMV launch of a layer that carries adapters:
//   dec_lora_down_kernel   t_i = x A_i for the <= 3 matrices of the launch: K cut over DEC_LORA_PARTS blocks per matrix (a single
//                          block walking K = 4096 .. 11008 costs 60-170 us; the 7 products of a layer made decoding with an
//                          adapter 11 x slower than without), fp32 partial sums per block; x is RMSNormed first where the GEMV
//                          launch norms its input (every block recomputes the row's sum of squares: 8 KiB from the L2);
//   dec_lora_up_kernel     t_i = h(sum of the partials) (the reference's fp16 temporary), out_i[n] = h(out_i[n] + sum_j t_i[j] B_i[j][n])
//                          -- or, for gate / up, act[n] = silu(g[n] + ..) * (u[n] + ..) on the two un-fused products.
// ---------------------------------------------------------------------------------------------------------------
#define DEC_LORA_PARTS 32
#define DEC_LORA_MAXR 64
struct DecLoraArgs {
    const f16* x; const f16* norm_w; float eps; int K, nmat, kslice;
    const f16* a[3]; const f16* b[3]; int r[3];
    f16* out[3]; int n[3];
    float* part;
    int silu; f16* act;                                               // silu: out[0] / out[1] are the gate / up products, the result goes to act
};

__global__ __launch_bounds__(256) void dec_lora_down_kernel(const DecLoraArgs a)
{
    __shared__ float red[4][8][8];
    __shared__ float ssq[4];
    const int mi = blockIdx.y, p = blockIdx.x;
    const int r = mi == 0 ? a.r[0] : mi == 1 ? a.r[1] : a.r[2];      // (static selects: no run-time index into the argument struct)
    if (r <= 0) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int nc = (r + 7) >> 3;                                             // column chunks of 8, padded to a power of two <= 8
    nc = nc <= 1 ? 1 : nc <= 2 ? 2 : nc <= 4 ? 4 : 8;
    const int c = lane % nc, kl = lane / nc, rpw = 64 / nc;
    const int k0 = p * a.kslice, k1 = min(a.K, k0 + a.kslice);
    const f16* A = mi == 0 ? a.a[0] : mi == 1 ? a.a[1] : a.a[2];
    const bool vec = (r & 7) == 0;                                      // 16-byte rows of A (else element by element)
    // 4 rows per lane and batch, requested together -- and the FIRST batch before the RMSNorm reduction below (its loads do not
    // depend on it): a load per iteration would serialise the L2 / HBM latency
    f16 xv[4], nw[4];
    f16x8 wv[4];
    auto request = [&](int kb) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int k = kb + u * 4 * rpw;
            const bool ok = k < k1;
            xv[u] = ok ? a.x[k] : (f16) 0.f;
            nw[u] = (ok && a.norm_w) ? a.norm_w[k] : (f16) 1.f;
            wv[u] = (f16x8){0, 0, 0, 0, 0, 0, 0, 0};
            if (ok && vec && c * 8 < r) wv[u] = *(const f16x8*) (A + (size_t) k * r + c * 8);
            else if (ok && !vec) for (int j = 0; j < 8 && c * 8 + j < r; ++j) wv[u][j] = A[(size_t) k * r + c * 8 + j];
        }
    };
    const int kfirst = k0 + wave * rpw + kl;
    request(kfirst);
    float rm = 1.f;
    if (a.norm_w) {                                                   // rms_norm.cu: fp32 sum of squares, r rounded to fp16, two fp16 multiplies
        float sq = 0.f;
        for (int k = tid * 8; k < a.K; k += 256 * 8) {
            const f16x8 v = *(const f16x8*) (a.x + k);
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float f = (float) v[j]; sq = fmaf(f, f, sq); }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) sq += __shfl_xor(sq, off, 64);
        if (lane == 0) ssq[wave] = sq;
        __syncthreads();
        rm = 1.0f / sqrtf((ssq[0] + ssq[1] + ssq[2] + ssq[3]) * (1.0f / (float) a.K) + a.eps);
    }
    const f16 rmh = (f16) rm;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int kb = kfirst; kb < k1; kb += 16 * rpw) {
        if (kb != kfirst) request(kb);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            f16 x1 = xv[u];
            if (a.norm_w) { const f16 m = x1 * rmh; x1 = m * nw[u]; }
            const float xf = (float) x1;
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = fmaf(xf, (float) wv[u][j], acc[j]);
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        float v = acc[j];
        for (int off = nc; off < 64; off <<= 1) v += __shfl_xor(v, off, 64);
        if (kl == 0) red[wave][c][j] = v;
    }
    __syncthreads();
    if (tid < nc * 8 && tid < ((r + 7) & ~7)) {
        const int cc = tid >> 3, j = tid & 7;
        if (cc * 8 + j < r)
            a.part[((size_t) mi * DEC_LORA_PARTS + p) * DEC_LORA_MAXR + cc * 8 + j] = red[0][cc][j] + red[1][cc][j] + red[2][cc][j] + red[3][cc][j];
    }
}

// The rows j0 .. j0 + 31 of B at columns n, n + 1 (requested together: one L2 / HBM latency for the lot)
struct LoraRows { f16x2 v[32]; };
__device__ __forceinline__ void lora_rows_load(LoraRows& R, const f16* __restrict__ B, int N, int n, int r, int j0)
{
#pragma unroll
    for (int u = 0; u < 32; ++u) R.v[u] = j0 + u < r ? *(const f16x2*) (B + (size_t) (j0 + u) * N + n) : (f16x2){(f16) 0.f, (f16) 0.f};
}
__device__ __forceinline__ void lora_rows_fma(const LoraRows& R, const float* t, int r, int j0, float& c0, float& c1)
{
#pragma unroll
    for (int u = 0; u < 32; ++u) {
        const float tj = j0 + u < r ? t[j0 + u] : 0.f;
        c0 = fmaf(tj, (float) R.v[u][0], c0); c1 = fmaf(tj, (float) R.v[u][1], c1);
    }
}

__global__ __launch_bounds__(128) void dec_lora_up_kernel(const DecLoraArgs a, int nparts)
{
    // one thread = 2 consecutive columns (the rows of B are read 256 bytes per wave and rank), 256 columns per block: a launch is
    // 16 .. 48 blocks.  Everything a thread reads -- its first 32 rows of B, the product(s) it adds to, the partial sums of t -- is
    // requested BEFORE anything is waited for: the launch is a latency chain (rocprofv3, earlier versions: 27 us with one load per
    // loop iteration, 22 us with batches of 8 behind the t reduction), not a stream.
    __shared__ float t[3][DEC_LORA_MAXR];
    const int tid = threadIdx.x;
    const int n2 = (blockIdx.x * 128 + tid) * 2;
    // which matrix / column this thread works on (static selects: no run-time index into the by-value argument struct)
    int mi = 0, n0 = n2;
    if (!a.silu) {
        if (n0 >= a.n[0]) { n0 -= a.n[0]; mi = 1; if (a.nmat > 1 && n0 >= a.n[1]) { n0 -= a.n[1]; mi = 2; } }
    }
    const int N = mi == 0 ? a.n[0] : mi == 1 ? a.n[1] : a.n[2];
    const int r = mi == 0 ? a.r[0] : mi == 1 ? a.r[1] : a.r[2];
    const f16* B = mi == 0 ? a.b[0] : mi == 1 ? a.b[1] : a.b[2];
    f16* out = mi == 0 ? a.out[0] : mi == 1 ? a.out[1] : a.out[2];
    const bool live = mi < a.nmat && n0 < N;
    LoraRows R0, R1;                                                     // silu: gate rows / up rows; plain: rows of the one matrix
    f16x2 prev0 = {(f16) 0.f, (f16) 0.f}, prev1 = prev0;
    const int r1 = a.silu ? a.r[1] : 0;
    if (live) {
        lora_rows_load(R0, B, N, n0, r, 0);
        prev0 = *(const f16x2*) (out + n0);
        if (a.silu) { lora_rows_load(R1, a.b[1], a.n[1], n0, r1, 0); prev1 = *(const f16x2*) (a.out[1] + n0); }
    }
    // t_i[j] = h(sum over the K parts), fixed order; all <= 32 partial sums of an element requested together
    for (int i = tid; i < a.nmat * DEC_LORA_MAXR; i += 128) {
        const int ti = i / DEC_LORA_MAXR, j = i % DEC_LORA_MAXR;
        const int r_ti = ti == 0 ? a.r[0] : ti == 1 ? a.r[1] : a.r[2];
        float pv[DEC_LORA_PARTS];
#pragma unroll
        for (int p = 0; p < DEC_LORA_PARTS; ++p)
            pv[p] = (j < r_ti && p < nparts) ? a.part[((size_t) ti * DEC_LORA_PARTS + p) * DEC_LORA_MAXR + j] : 0.f;
        float v = 0.f;
#pragma unroll
        for (int p = 0; p < DEC_LORA_PARTS; ++p) v += pv[p];
        t[ti][j] = (float) (f16) v;
    }
    __syncthreads();
    if (!live) return;
    const float* tm = mi == 0 ? t[0] : mi == 1 ? t[1] : t[2];
    float c0 = 0.f, c1 = 0.f, d0 = 0.f, d1 = 0.f;
    lora_rows_fma(R0, tm, r, 0, c0, c1);
    if (r > 32) { lora_rows_load(R0, B, N, n0, r, 32); lora_rows_fma(R0, tm, r, 32, c0, c1); }
    if (a.silu) {
        lora_rows_fma(R1, t[1], r1, 0, d0, d1);
        if (r1 > 32) { lora_rows_load(R1, a.b[1], a.n[1], n0, r1, 32); lora_rows_fma(R1, t[1], r1, 32, d0, d1); }
        // h(product + h(adapter)): the adapter's own fp16 result, then one add; then silu(gate) * up as the fused launch does
        const f16 gh0 = r > 0 ? (f16) ((float) prev0[0] + (float) (f16) c0) : prev0[0], gh1 = r > 0 ? (f16) ((float) prev0[1] + (float) (f16) c1) : prev0[1];
        const f16 uh0 = r1 > 0 ? (f16) ((float) prev1[0] + (float) (f16) d0) : prev1[0], uh1 = r1 > 0 ? (f16) ((float) prev1[1] + (float) (f16) d1) : prev1[1];
        *(f16x2*) (a.act + n0) = (f16x2){silu_mul_f16(gh0, uh0), silu_mul_f16(gh1, uh1)};
        return;
    }
    if (r <= 0) return;
    prev0[0] = (f16) ((float) prev0[0] + (float) (f16) c0);
    prev0[1] = (f16) ((float) prev0[1] + (float) (f16) c1);
    *(f16x2*) (out + n0) = prev0;
}

// The down / up pair behind a GEMV launch; `a` is complete except for the partial-sum buffer.
// (Round 5 measured the two halves as ONE launch -- every block computing t = x A itself, in full, from an LDS copy of x, 1024
// columns per block: 305 instead of 336 tokens/s at rank 16, 135 instead of 189 at rank 64, profiles/r05_lora.json.  One CU keeps
// ~32 KiB of requests in flight, so a block streams its 128-352 KiB of A in 4-10 round trips; cutting K over 32 blocks, as here,
// is what keeps the down half at one round trip.  The kernel was removed again.)
static int dec_lora_launch(DecLoraArgs& a, int cols, float* part, hipStream_t s)
{
    int nparts = (a.K + 255) / 256;
    if (nparts > DEC_LORA_PARTS) nparts = DEC_LORA_PARTS;
    a.kslice = ((a.K + nparts - 1) / nparts + 7) & ~7;
    nparts = (a.K + a.kslice - 1) / a.kslice;
    a.part = part;
    hipLaunchKernelGGL(dec_lora_down_kernel, dim3(nparts, a.nmat), dim3(256), 0, s, a);
    EXL_LAUNCH_CHECK();
    hipLaunchKernelGGL(dec_lora_up_kernel, dim3((cols / 2 + 127) / 128), dim3(128), 0, s, a, nparts);
    EXL_LAUNCH_CHECK();
    return 0;
}

// A decoder stage without the head kernel advances the device-side position itself.
__global__ void dec_advance_kernel(int32_t* pos_dev) { *pos_dev += 1; }

// ---------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------
struct DecLayer {
    Q4Matrix *q, *k, *v, *o, *gate, *up, *down;
    const f16 *in_norm, *post_norm;
    f16 *kc, *vc;
    bool set;
    // act-order: 16-bit gather maps (and their inverses) of the seven matrices, in the decoder's own memory; NULL = no act-order
    const uint16_t *map_q, *map_k, *map_v, *map_o, *map_gate, *map_up, *map_down, *inv_o, *inv_down;
    // LoRA operands of the seven projections (exl_decoder_set_lora; order q, k, v, o, gate, up, down): A [in, r], B [r, out] fp16, rank 0 = none
    const f16 *lora_a[7], *lora_b[7];
    int lora_r[7];
    bool lora_any(int lo, int hi) const { for (int i = lo; i <= hi; ++i) if (lora_r[i] > 0) return true; return false; }
};

struct Decoder {
    uint32_t magic;
    int device, L, h, inter, heads, kv_heads, hd, vocab, max_seq;
    float eps;
    const f16 *embed, *final_norm, *lm_head, *sin, *cos;
    std::vector<DecLayer> layers;
    f16 *hid, *qbuf, *kbuf, *vbuf, *attn_out, *act;
    float* partial;
    float2* head_best;            // (largest logit, row) per head-kernel block, for exl_decoder_step_greedy
    float* probs;                 // [vocab] scratch of the sampler (exl_decoder_step_sample)
    int nsplit;                   // KV splits of the attention kernel in use (<= nsplit_max)
    int nsplit_max;
    int max_blocks;               // persistent GEMV grid: blocks per CU x CUs
    void* block;                  // one hipMalloc
    uint16_t* maps;               // act-order maps of all layers: per layer 2 x (6 hidden + inter) entries
    f16* zero_res;                // [h] zeros: the residual a tensor-parallel rank that does not own it adds (exl_decoder_set_tp)
    f16 *gbuf, *ubuf;             // [inter] each: gate / up products of a layer with LoRA operands, before the adapter and SiLU * mul
    float* lora_part;             // [3][DEC_LORA_PARTS][DEC_LORA_MAXR] fp32: partial x @ A sums of the launch in flight
    bool lora_o;                  // some layer has an o_proj adapter: the attention output is materialised (no merge fold)
    int ring, ring_fence, ring_depth, ring_wide;   // exl_decoder_set_option: rolling-ring weight stream (decode_ring.hip) / its start-up barrier
    bool residual_owner;          // tensor parallel: only one rank adds the residual stream to its partial o_proj / down_proj sums
    int qd() const { return heads * hd; }     // width of q / attention output: = h, or this rank's heads of a tensor-parallel shard
    bool has_embed() const { return embed != nullptr; }
    bool has_head() const { return lm_head != nullptr; }
};
#define DEC_MAGIC 0x44454332u

extern "C" int exl_decoder_create(int device, int n_layers, int hidden, int inter, int heads, int kv_heads, int head_dim,
                                  int vocab, int max_seq_len, float eps, const void* embed, const void* final_norm,
                                  const void* lm_head, const void* sin, const void* cos, void** out)
{
    EXL_REQUIRE(out, EXL_E_INVALID, "decoder_create: out is null");
    *out = nullptr;
    EXL_REQUIRE(device >= 0 && device < EXL_MAX_DEVICES, EXL_E_INVALID, "decoder_create: invalid device %d", device);
    EXL_REQUIRE(head_dim == 128, EXL_E_UNSUPPORTED, "decoder: head_dim must be 128 (got %d)", head_dim);
    // heads * head_dim < hidden: one rank's shard of a tensor-parallel model (exllama_amd/tp.py): its own heads, the full residual stream
    EXL_REQUIRE(hidden % 128 == 0 && heads >= 1 && heads * head_dim <= hidden && heads % kv_heads == 0, EXL_E_UNSUPPORTED, "decoder: bad head geometry");
    EXL_REQUIRE(inter % 128 == 0, EXL_E_UNSUPPORTED, "decoder: intermediate size must be a multiple of 128 (got %d)", inter);
    EXL_REQUIRE(hidden <= 8192 && inter <= 32768, EXL_E_UNSUPPORTED, "decoder: hidden (%d) / intermediate (%d) size too large", hidden, inter);
    // A decoder may be one STAGE of a layer-split model (reference: ExLlamaDeviceMap, model.py:636-668): embed == NULL ->
    // the step reads the residual stream the caller put into exl_decoder_hidden(); lm_head == NULL -> no final norm / head,
    // the residual stream is left there for the next stage.
    EXL_REQUIRE(sin && cos, EXL_E_INVALID, "decoder_create: null pointer");
    EXL_REQUIRE((final_norm != nullptr) == (lm_head != nullptr), EXL_E_INVALID, "decoder_create: final_norm and lm_head go together");
    EXL_REQUIRE(n_layers >= 1, EXL_E_INVALID, "decoder_create: a decoder needs at least one layer");
    Decoder* d = new Decoder();
    d->magic = DEC_MAGIC;
    d->device = device; d->L = n_layers; d->h = hidden; d->inter = inter; d->heads = heads; d->kv_heads = kv_heads;
    d->hd = head_dim; d->vocab = vocab; d->max_seq = max_seq_len; d->eps = eps;
    d->embed = (const f16*) embed; d->final_norm = (const f16*) final_norm; d->lm_head = (const f16*) lm_head;
    d->sin = (const f16*) sin; d->cos = (const f16*) cos;
    d->layers.resize(n_layers);
    for (auto& l : d->layers) { l.set = false; for (int i = 0; i < 7; ++i) { l.lora_a[i] = l.lora_b[i] = nullptr; l.lora_r[i] = 0; } }
    // KV splits of the deepest bucket: one 8-wave block per CU and head-split with up to 320 keys per pass (long contexts), or
    // one 4-wave block per CU with up to 160.  (Round 2 ran 512 four-wave blocks of <= 160 keys: the o_proj kernel that merges the
    // splits reads every partial in every block, and halving their number paid more than the attention kernel lost.)
    // Wider models (hidden > 4096) merge in a kernel of their own and keep the 512-block form.
    const bool folds = heads * head_dim <= DEC_THREADS * 8;
    int ns = (max_seq_len > 1280 && !folds ? 512 : 256) / heads;
    if (const char* env = getenv("EXL_DEC_NSPLIT")) ns = atoi(env);  // measurement aid
    if (ns < 1) ns = 1;
    if (ns > DEC_MAX_NSPLIT) ns = DEC_MAX_NSPLIT;
    while ((max_seq_len + ns - 1) / ns + 32 > DEC_ATT_MAX_KEYS) ++ns;   // chunks of up to 32 keys, round robin: a split holds <= keys / ns + 32 slots
    if (ns > DEC_MAX_NSPLIT) { delete d; EXL_FAIL(EXL_E_UNSUPPORTED, "decoder: max_seq_len %d too long (max %d)", max_seq_len, DEC_MAX_NSPLIT * (DEC_ATT_MAX_KEYS - 32)); }
    d->nsplit = ns;
    d->nsplit_max = ns;
    const int kvd = kv_heads * head_dim;
    size_t bytes = 0;
    auto carve = [&](size_t n) { const size_t off = bytes; bytes += (n + 255) & ~(size_t) 255; return off; };
    const size_t o_hid = carve((size_t) hidden * 2), o_q = carve((size_t) hidden * 2);
    const size_t o_k = carve((size_t) kvd * 2), o_v = carve((size_t) kvd * 2);
    const size_t o_ao = carve((size_t) hidden * 2), o_act = carve((size_t) inter * 2);
    const size_t o_p = carve((size_t) heads * ns * 130 * 4);
    const size_t o_hb = carve((size_t) ((vocab + 31) / 32) * sizeof(float2));
    const size_t o_pr = carve((size_t) vocab * sizeof(float));
    const size_t maps_per_layer = 2 * ((size_t) 6 * hidden + inter);                 // map + inverse of q, k, v, o, gate, up (K = hidden) and down (K = inter)
    const size_t o_maps = carve(maps_per_layer * n_layers * sizeof(uint16_t));
    const size_t o_zero = carve((size_t) hidden * 2);
    const size_t o_gb = carve((size_t) inter * 2), o_ub = carve((size_t) inter * 2);
    const size_t o_lp = carve((size_t) 3 * DEC_LORA_PARTS * DEC_LORA_MAXR * sizeof(float));
    int prev = 0;
    hipError_t e = hipGetDevice(&prev);
    if (e == hipSuccess) e = hipSetDevice(device);
    int cus = 0;
    if (e == hipSuccess) e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device);
    if (e == hipSuccess) e = hipMalloc(&d->block, bytes);
    if (e == hipSuccess) e = hipMemset((unsigned char*) d->block + o_zero, 0, (size_t) hidden * 2);
    (void) hipSetDevice(prev);
    if (e != hipSuccess) { delete d; EXL_FAIL((int) e, "decoder_create: %s", hipGetErrorString(e)); }
    if (lm_head) {                                                   // the sampler's whole-vocabulary workspace (top_k = 0 / > 1024): a captured
        void* sw = nullptr;                                          // exl_decoder_step_sample cannot allocate it
        const int rs = vocab <= SMP_BIG_LIMIT ? exl_sampler_workspace(device, smp_big_bytes(vocab), &sw) : 0;
        if (rs) { (void) hipFree(d->block); delete d; return rs; }
    }
    unsigned char* b = (unsigned char*) d->block;
    d->hid = (f16*) (b + o_hid); d->qbuf = (f16*) (b + o_q);
    d->kbuf = (f16*) (b + o_k); d->vbuf = (f16*) (b + o_v); d->attn_out = (f16*) (b + o_ao); d->act = (f16*) (b + o_act);
    d->partial = (float*) (b + o_p);
    d->head_best = (float2*) (b + o_hb);
    d->probs = (float*) (b + o_pr);
    d->maps = (uint16_t*) (b + o_maps);
    d->zero_res = (f16*) (b + o_zero);
    d->gbuf = (f16*) (b + o_gb); d->ubuf = (f16*) (b + o_ub);
    d->lora_part = (float*) (b + o_lp);
    d->lora_o = false;
    d->residual_owner = true;
    d->ring = getenv("EXL_DEC_RING") ? atoi(getenv("EXL_DEC_RING")) : 15;
    d->ring_fence = getenv("EXL_DEC_RING_FENCE") ? atoi(getenv("EXL_DEC_RING_FENCE")) : 1;
    d->ring_depth = getenv("EXL_DEC_RING_DEPTH") ? atoi(getenv("EXL_DEC_RING_DEPTH")) : 3;
    d->ring_wide = getenv("EXL_DEC_RING_WIDE") ? atoi(getenv("EXL_DEC_RING_WIDE")) : 1;
    int bpc = 2;
    if (const char* env = getenv("EXL_DEC_BLOCKS_PER_CU")) { bpc = atoi(env); if (bpc < 1) bpc = 1; if (bpc > 4) bpc = 4; }
    d->max_blocks = (cus > 0 ? cus : 256) * bpc;
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
    EXL_REQUIRE(l.q->height == d->h && l.q->width == d->qd() && l.k->height == d->h && l.k->width == kvd &&
                l.v->height == d->h && l.v->width == kvd && l.o->height == d->qd() && l.o->width == d->h &&
                l.gate->height == d->h && l.gate->width == d->inter && l.up->height == d->h && l.up->width == d->inter &&
                l.down->height == d->inter && l.down->width == d->h, EXL_E_INVALID, "decoder_set_layer: matrix shapes do not match the model");
    for (Q4Matrix* m : {l.q, l.k, l.v, l.o, l.gate, l.up, l.down}) {
        EXL_REQUIRE(m->device == d->device, EXL_E_INVALID, "decoder_set_layer: matrix lives on another device");
        EXL_REQUIRE(m->layout == EXL_LAYOUT_T16, EXL_E_UNSUPPORTED, "decoder_set_layer: matrix is not in the T16 layout");
    }
    EXL_REQUIRE(l.gate->groupsize == l.up->groupsize && (l.gate->x_map == nullptr) == (l.up->x_map == nullptr), EXL_E_UNSUPPORTED,
                "decoder_set_layer: gate and up projections must share group size and act-order mode");
    EXL_REQUIRE(in_norm && post_norm && key_cache && value_cache, EXL_E_INVALID, "decoder_set_layer: null pointer");
    l.in_norm = (const f16*) in_norm; l.post_norm = (const f16*) post_norm;
    l.kc = (f16*) key_cache; l.vc = (f16*) value_cache;
    // act-order: every gather map once more as 16-bit indices, plus its inverse (decoder-owned; built on the default stream)
    {
        EXL_REQUIRE(d->h < 65536 && d->inter < 65536, EXL_E_UNSUPPORTED, "decoder: act-order maps are 16-bit");
        uint16_t* base = d->maps + (size_t) index * 2 * ((size_t) 6 * d->h + d->inter);
        Q4Matrix* ms[7] = {l.q, l.k, l.v, l.o, l.gate, l.up, l.down};
        const uint16_t** mp[7] = {&l.map_q, &l.map_k, &l.map_v, &l.map_o, &l.map_gate, &l.map_up, &l.map_down};
        const uint16_t* inv[7] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
        int prev = 0;
        EXL_HIP(hipGetDevice(&prev));
        if (prev != d->device) EXL_HIP(hipSetDevice(d->device));
        hipError_t e = hipSuccess;
        for (int i = 0; i < 7 && e == hipSuccess; ++i) {
            const int K = ms[i]->height;
            uint16_t* m16 = base; uint16_t* i16 = base + K;
            base += 2 * (size_t) K;
            *mp[i] = nullptr;
            if (i == 6 && !ms[i]->x_map && l.gate->x_map) {
                // act-order gate / up in front of a down_proj WITHOUT a map (its permutation folded into their column order at load,
                // model.py: _fold_act_order_down_proj): the gathering launch stores through a map by construction (decode_ring.hip:
                // its requests are part of the hand-counted stream) -- here the identity, i.e. the natural order
                hipLaunchKernelGGL(dec_ident16_kernel, dim3((K + 255) / 256), dim3(256), 0, (hipStream_t) 0, i16, K);
                e = hipGetLastError();
                inv[i] = i16;
                continue;
            }
            if (!ms[i]->x_map) continue;
            hipLaunchKernelGGL(dec_map16_kernel, dim3((K + 255) / 256), dim3(256), 0, (hipStream_t) 0, ms[i]->x_map, m16, i16, K);
            e = hipGetLastError();
            *mp[i] = m16; inv[i] = i16;
        }
        if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t) 0);
        if (prev != d->device) (void) hipSetDevice(prev);
        if (e != hipSuccess) EXL_FAIL((int) e, "decoder_set_layer: %s", hipGetErrorString(e));
        l.inv_o = inv[3]; l.inv_down = inv[6];
    }
    l.set = true;
    return 0;
}

// LoRA operands of one layer (order q, k, v, o, gate, up, down; a7[i] == NULL or rank7[i] == 0: no adapter on that projection; all
// NULL clears the layer).  A [in_features, r], B [r, out_features] fp16 in device memory that outlives the decoder's graphs, alpha / r
// folded into B (what exllama_amd.lora / the reference's lora.py hold).  The adapter products then ride inside the token step
// (two small launches behind the GEMV launch of each class that has adapters) instead of forcing the step onto the op-by-op path.
// Not for layers whose o_proj or down_proj gathers through an act-order map (their inputs are stored permuted): EXL_E_UNSUPPORTED,
// the caller keeps the op path.  Set before the step is captured; changing adapters means capturing again.
extern "C" int exl_decoder_set_lora(void* dec, int index, const void* const* a7, const void* const* b7, const int* rank7)
{
    Decoder* d = dec_from(dec);
    EXL_REQUIRE(d && index >= 0 && index < d->L && d->layers[index].set, EXL_E_INVALID, "decoder_set_lora: invalid decoder / layer");
    DecLayer& l = d->layers[index];
    bool any = false;
    for (int i = 0; i < 7; ++i) {
        const int r = (a7 && b7 && rank7 && a7[i] && b7[i]) ? rank7[i] : 0;
        EXL_REQUIRE(r >= 0 && r <= DEC_LORA_MAXR, EXL_E_UNSUPPORTED, "decoder_set_lora: rank %d beyond %d", r, DEC_LORA_MAXR);
        any = any || r > 0;
    }
    if (any) {
        EXL_REQUIRE(!l.o->x_map && !l.down->x_map, EXL_E_UNSUPPORTED,
                    "decoder_set_lora: o_proj / down_proj of layer %d gather through an act-order map (fold down_proj at load; o_proj adapters stay on the op path)", index);
        EXL_REQUIRE(d->residual_owner && d->qd() == d->h, EXL_E_UNSUPPORTED, "decoder_set_lora: not on a tensor-parallel shard");
    }
    for (int i = 0; i < 7; ++i) {
        const int r = (a7 && b7 && rank7 && a7[i] && b7[i]) ? rank7[i] : 0;
        l.lora_a[i] = r ? (const f16*) a7[i] : nullptr;
        l.lora_b[i] = r ? (const f16*) b7[i] : nullptr;
        l.lora_r[i] = r;
    }
    d->lora_o = false;
    for (const DecLayer& x : d->layers) d->lora_o = d->lora_o || x.lora_r[3] > 0;
    return 0;
}

extern "C" int exl_decoder_set_kv_splits(void* dec, int nsplit, int* max_context)
{
    Decoder* d = dec_from(dec);
    EXL_REQUIRE(d, EXL_E_INVALID, "decoder_set_kv_splits: invalid decoder");
    if (nsplit <= 0) nsplit = d->nsplit_max;
    EXL_REQUIRE(nsplit <= d->nsplit_max, EXL_E_INVALID, "decoder_set_kv_splits: %d splits requested, at most %d", nsplit, d->nsplit_max);
    d->nsplit = nsplit;
    if (max_context) {                                               // keys visible = context + 1 must fit nsplit * (MAX_KEYS - 32)
        const long cap = (long) nsplit * (DEC_ATT_MAX_KEYS - 32) - 1;
        *max_context = (int) (cap < d->max_seq - 1 ? cap : d->max_seq - 1);
    }
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

// exl_decoder_plan: when set, the launch helpers record the configuration they would launch and return without launching
static thread_local int* g_plan = nullptr;

// (U, NP) by row-blocks per wave; G16 by group size; NV by K
template <int PNORM, int EMODE, int NV>
static int launch_dec_gemv_cfg(bool g16, int rbw, dim3 grid, size_t smem, hipStream_t s, const DecGemvArgs& a, bool two_per_cu)
{
    // The opt-in for more than 64 KiB of dynamic LDS is a per-device function attribute: tracked per (kernel, device).
#define DEC_LAUNCH1(U, NP, G) do { auto kfn = dec_stream_kernel<U, NP, G, PNORM, EMODE, NV>;                                   \
        if (g_plan) { g_plan[0] = 1; g_plan[1] = U; g_plan[2] = NP; g_plan[3] = G ? 1 : 0; g_plan[4] = PNORM; g_plan[5] = EMODE;  \
                      g_plan[6] = NV; g_plan[7] = (int) grid.x; g_plan[8] = (int) smem; g_plan[9] = a.xs_images; return 0; }    \
        static bool big[EXL_MAX_DEVICES] = {};                                                                                \
        int dev_ = 0;                                                                                                         \
        if (smem > 64 * 1024) {                       /* more than the default dynamic-LDS limit: opt in once per device */   \
            EXL_HIP(hipGetDevice(&dev_));                                                                                     \
            if (dev_ >= 0 && dev_ < EXL_MAX_DEVICES && !big[dev_]) {                                                          \
                EXL_HIP(hipFuncSetAttribute((const void*) kfn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));      \
                big[dev_] = true;                                                                                             \
            }                                                                                                                 \
        }                                                                                                                     \
        hipLaunchKernelGGL(kfn, grid, dim3(DEC_THREADS), smem, s, a); } while (0)
#define DEC_LAUNCH(U, NP) do { if (g16) DEC_LAUNCH1(U, NP, true); else DEC_LAUNCH1(U, NP, false); } while (0)
    // Row-blocks per wave -> (U, NP): U 16-byte loads in flight per lane and pass, NP passes, U * NP >= rbw with as few idle slots
    // as possible.  Idle slots are not free: the branch-free stream re-reads a valid row-block for each (dec_unit_issue), so the
    // old four-entry table cost 13B 37 % extra weight traffic in qkv (5 row-blocks in 8 slots), 33B 46 % in gate/up (13 in 24).
    // The candidates a kernel class can meet are bounded by its K (NV = ceil(K / 4096)): only those are instantiated.
#ifdef EXL_DEC_FAST_BUILD                                            /* ISA inspection builds: 7B instantiations only */
    if (rbw <= 4)       DEC_LAUNCH1(4, 1, true);
    else if (rbw <= 8)  DEC_LAUNCH1(4, 2, true);
    else                DEC_LAUNCH1(6, 2, true);
#else
    if constexpr (NV <= 2) {                                         // K = hidden size <= 8192
        if (rbw <= 4)       DEC_LAUNCH(4, 1);
        else if (rbw == 5)  DEC_LAUNCH(5, 1);
        else if (rbw == 6)  DEC_LAUNCH(6, 1);
        else if (rbw <= 8)  DEC_LAUNCH(4, 2);                        // (7 loads at once and a single pass measured slower than one idle slot: 33B qkv 33 vs 27 us)
        else {
            if constexpr (EMODE == 2) {                              // gate/up: 4 waves per tile, up to 16 row-blocks per wave
                if (rbw <= 10)      DEC_LAUNCH(5, 2);
                else if (rbw <= 12) DEC_LAUNCH(6, 2);
                else if (rbw <= 14) DEC_LAUNCH(7, 2);
                else if (rbw <= 16) DEC_LAUNCH(4, 4);
                else EXL_FAIL(EXL_E_UNSUPPORTED, "decoder: hidden size too large for the gate/up kernel");
            } else EXL_FAIL(EXL_E_UNSUPPORTED, "decoder: in_features too large for this kernel class");
        }
    } else {                                                         // down_proj: K = intermediate size (> 8192)
        if constexpr (PNORM == 0 && EMODE == 1) {
            // More tiles than CUs (13B: 320, 65B / 70B: 512): two blocks per CU must be co-resident or the second round runs with a
            // quarter of the chip -- that needs <= 128 VGPRs, which only the U = 4 streams meet (6 loads in flight: 138-179).
            if (two_per_cu && g16 && rbw > 12 && rbw <= 16)      DEC_LAUNCH(4, 4);   // (groupsize 32 / 64 streams fit 128 VGPRs as they are)
            else if (two_per_cu && g16 && rbw > 16 && rbw <= 20) DEC_LAUNCH(4, 5);
            else if (two_per_cu && g16 && rbw > 20 && rbw <= 24) DEC_LAUNCH(4, 6);
            else if (two_per_cu && g16 && rbw > 24 && rbw <= 28) DEC_LAUNCH(4, 7);
            else
            if (rbw <= 10)      DEC_LAUNCH(5, 2);
            else if (rbw <= 12) DEC_LAUNCH(6, 2);
            else if (rbw <= 14) DEC_LAUNCH(7, 2);
            else if (rbw <= 16) DEC_LAUNCH(4, 4);
            else if (rbw <= 18) DEC_LAUNCH(6, 3);
            else if (rbw <= 20) DEC_LAUNCH(5, 4);
            else if (rbw <= 24) DEC_LAUNCH(6, 4);
            else if (rbw <= 28) DEC_LAUNCH(7, 4);
            else if (rbw <= 30) DEC_LAUNCH(6, 5);
            else                DEC_LAUNCH(6, 6);
        } else EXL_FAIL(EXL_E_UNSUPPORTED, "decoder: in_features too large for this kernel class");
    }
#endif
#undef DEC_LAUNCH
#undef DEC_LAUNCH1
    EXL_LAUNCH_CHECK();
    return 0;
}

// pnorm / emode as in dec_gemv_kernel.  mats: nmat matrices sharing K (emode 2: gate, up).
// maps: the 16-bit gather map of every matrix (NULL entries / NULL array: the activation is read linearly -- no act-order, or a
// producer already stored it gathered); out_perm: inverse gather map of the CONSUMER of this launch's output (EMODE 2).
static int launch_dec_gemv(const Decoder* dcfg, int cls, int pnorm, int emode, const f16* vec, const int64_t* tok, const f16* norm_w, float eps, f16* hid_copy,
                           int nmat, Q4Matrix* const* mats, f16* const* outs, f16* hid_io, hipStream_t s,
                           const float* att_ml = nullptr, int att_nsplit = 0, const uint16_t* const* maps = nullptr,
                           const uint16_t* out_perm = nullptr, const f16* res_in = nullptr)
{
    const int max_blocks = dcfg->max_blocks;
    DecGemvArgs a;
    a.vec = vec; a.tok = tok; a.norm_w = norm_w; a.eps = eps; a.hid_copy = hid_copy; a.nmat = nmat; a.hid_io = hid_io;
    a.res_in = res_in ? res_in : hid_io;
    a.att_ml = att_ml; a.att_nsplit = att_nsplit;
    int tiles = 0;
    bool any_map = false;
    for (int i = 0; i < DEC_MAX_MATS; ++i) {
        if (i < nmat) {
            a.mat[i] = t16_view(mats[i]);
            a.out[i] = outs ? outs[i] : nullptr;
            if (emode != 2 || i == 0) tiles += mats[i]->width / 16;
            a.map16[i] = maps ? maps[i] : nullptr;
            any_map = any_map || a.map16[i] != nullptr;
        } else {
            a.mat[i] = a.mat[0];
            a.out[i] = nullptr;
            a.map16[i] = a.map16[0];
        }
        a.tile_end[i] = tiles;
    }
    const int K = mats[0]->height, RB = K / 128;
    const int wpt = emode == 2 ? DEC_WAVES / 2 : DEC_WAVES;
    const int rbw = (RB + wpt - 1) / wpt;
    EXL_REQUIRE(rbw <= (pnorm == 0 && emode == 1 ? 36 : 24), EXL_E_UNSUPPORTED, "decoder: in_features %d too large", K);
    a.rb_per_wave = rbw;
    const bool g16 = mats[0]->groupsize % 128 == 0;
    for (int i = 1; i < nmat; ++i)
        EXL_REQUIRE((mats[i]->groupsize % 128 == 0) == g16 && mats[i]->height == K, EXL_E_UNSUPPORTED, "decoder: fused matrices must share K and group-size class");
    a.xs_images = any_map ? nmat : 1;                                // act-order: every matrix gathers x through its own map
    static const int ablate = getenv("EXL_DEC_ABLATE") ? atoi(getenv("EXL_DEC_ABLATE")) : 0;
    a.ablate = ablate;
    const size_t smem = (size_t) a.xs_images * (K / 8) * 16 + (2 * DEC_WAVES * 16 + DEC_WAVES + 16) * sizeof(float) + (any_map ? (size_t) K * 2 : 0);
    EXL_REQUIRE(smem <= 160 * 1024, EXL_E_UNSUPPORTED, "decoder: activation stage (%zu bytes of LDS) exceeds the 160 KiB of a CU", smem);
    const int nv = (K / 8 + DEC_THREADS - 1) / DEC_THREADS;
    for (int i = 1; i < nmat; ++i)
        EXL_REQUIRE((a.map16[i] == nullptr) == (a.map16[0] == nullptr), EXL_E_UNSUPPORTED,
                    "decoder: matrices fused into one launch must agree on act-order");
    EXL_REQUIRE(pnorm != 0 || !any_map, EXL_E_INVALID, "decoder: a plain-vector launch takes its activation already in row order");
    a.out_perm = out_perm;
    dim3 grid(tiles < max_blocks ? tiles : max_blocks);
    const bool two_per_cu = (int) grid.x > max_blocks / 2;            // max_blocks = 2 x CUs (EXL_DEC_BLOCKS_PER_CU)
    a.nblocks = (int) grid.x;
    a.units_lo = tiles / (int) grid.x;
    a.units_rem = tiles % (int) grid.x;
    // The rolling-ring stream (decode_ring.hip) takes every launch it covers (exl_decoder_set_option / EXL_DEC_RING=0 keep
    // dec_stream_kernel for A/B; EXL_DEC_RING_FENCE=0 drops the barrier between a block's activation requests and its first
    // weight requests).
    const int ring = (dcfg->ring >> cls) & 1;                        // one bit per GEMV class: q/k/v, o_proj, gate/up, down_proj
    a.ring_flags = dcfg->ring_fence ? 1 : 0;
    if (ring) {
        const int rr = launch_dec_ring(pnorm, emode, g16, K, (int) grid.x, dcfg->ring_depth, dcfg->ring_wide && !two_per_cu, a, s, g_plan);
        if (rr != 1) return rr;
    }
    // NV (8-half activation vectors per thread) instantiations by kernel class: the normed / merged inputs have K = hidden
    // <= 8192 (NV <= 2; the merge fold only exists for hidden <= 4096), only o_proj / down_proj see K = intermediate size
#define DEC_GO(P, E, N) launch_dec_gemv_cfg<P, E, N>(g16, rbw, grid, smem, s, a, two_per_cu)
    EXL_REQUIRE(nv <= 8, EXL_E_UNSUPPORTED, "decoder: in_features %d too large", K);
#ifdef EXL_DEC_FAST_BUILD
    if (pnorm == 1 && emode == 0) return DEC_GO(1, 0, 1);
    if (pnorm == 1 && emode == 2) return DEC_GO(1, 2, 1);
    if (pnorm == 0 && emode == 1) return nv <= 1 ? DEC_GO(0, 1, 1) : DEC_GO(0, 1, 3);
    if (pnorm == 3 && emode == 1) return DEC_GO(3, 1, 1);
#else
    if (pnorm == 1 && emode == 0) { EXL_REQUIRE(nv <= 2, EXL_E_UNSUPPORTED, "decoder: hidden size too large"); return nv <= 1 ? DEC_GO(1, 0, 1) : DEC_GO(1, 0, 2); }
    if (pnorm == 1 && emode == 2) { EXL_REQUIRE(nv <= 2, EXL_E_UNSUPPORTED, "decoder: hidden size too large"); return nv <= 1 ? DEC_GO(1, 2, 1) : DEC_GO(1, 2, 2); }
    if (pnorm == 3 && emode == 1) { EXL_REQUIRE(nv <= 2, EXL_E_UNSUPPORTED, "decoder: hidden size too large"); return nv <= 1 ? DEC_GO(3, 1, 1) : DEC_GO(3, 1, 2); }
    if (pnorm == 0 && emode == 1)
        return nv <= 1 ? DEC_GO(0, 1, 1) : nv <= 2 ? DEC_GO(0, 1, 2) : nv <= 3 ? DEC_GO(0, 1, 3) : nv <= 6 ? DEC_GO(0, 1, 6) : DEC_GO(0, 1, 8);
#endif
#undef DEC_GO
    EXL_FAIL(EXL_E_INVALID, "decoder: unsupported kernel combination");
}

// ---------------------------------------------------------------------------------------------------------------------------
// The executor's fused GEMV launches behind the reference's fused OPS (api.hip: exl_q4_attn / exl_q4_attn_2 / exl_q4_mlp with one
// row and no LoRA -- what the reference's unmodified model.py calls per token and layer, model.py:254-289): RMSNorm + q / k / v in
// one launch, o_proj + residual in one, RMSNorm + gate / up + SiLU in one, down_proj + residual in one -- 4 launches where the
// op-by-op form needs 11 (norm, three GEMVs, ..., norm, two GEMVs, SiLU, GEMV).  Returns 1 when the shapes are not covered
// (act-order without the executor's 16-bit maps, a layout other than T16, K beyond the kernel classes): the caller keeps its path.
// ---------------------------------------------------------------------------------------------------------------------------
int dec_op_gemv(int device, int cls, int pnorm, int emode, const f16* vec, const f16* norm_w, float eps, int nmat,   // cls: ring class bit (0 q/k/v, 1 o_proj, 2 gate/up, 3 down_proj)
                Q4Matrix* const* mats, f16* const* outs, f16* hid_io, hipStream_t s)
{
    static Decoder cfg[EXL_MAX_DEVICES];
    static bool ready[EXL_MAX_DEVICES] = {};
    static std::mutex init_mutex;                                    // (a loader thread next to a serving thread: the per-device configuration is built once)
    if (device < 0 || device >= EXL_MAX_DEVICES) return 1;
    static const bool off = getenv("EXL_OPS_UNFUSED") != nullptr;    // A/B switch: the op-by-op launches
    if (off) return 1;
    std::lock_guard<std::mutex> init_lock(init_mutex);
    if (!ready[device]) {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess) return 1;
        Decoder& d = cfg[device];
        d.device = device;
        d.ring = getenv("EXL_DEC_RING") ? atoi(getenv("EXL_DEC_RING")) : 15;
        d.ring_fence = getenv("EXL_DEC_RING_FENCE") ? atoi(getenv("EXL_DEC_RING_FENCE")) : 1;
        d.ring_depth = getenv("EXL_DEC_RING_DEPTH") ? atoi(getenv("EXL_DEC_RING_DEPTH")) : 3;
        d.ring_wide = getenv("EXL_DEC_RING_WIDE") ? atoi(getenv("EXL_DEC_RING_WIDE")) : 1;
        d.max_blocks = (cus > 0 ? cus : 256) * 2;
        ready[device] = true;
    }
    const int K = mats[0]->height;
    if (K % 128 != 0 || K >= 65536) return 1;
    const int wpt = emode == 2 ? DEC_WAVES / 2 : DEC_WAVES;
    const int rbw = (K / 128 + wpt - 1) / wpt;
    const int nv = (K / 8 + DEC_THREADS - 1) / DEC_THREADS;
    if (rbw > (pnorm == 0 && emode == 1 ? 36 : 24) || nv > (pnorm == 0 ? 8 : 2)) return 1;
    for (int i = 0; i < nmat; ++i) {
        const Q4Matrix* m = mats[i];
        if (m->device != device || m->layout != EXL_LAYOUT_T16 || m->x_map != nullptr || m->height != K || m->width % 16 != 0) return 1;
        if (m->groupsize % 32 != 0 || (m->groupsize % 128 == 0) != (mats[0]->groupsize % 128 == 0)) return 1;
    }
    return launch_dec_gemv(&cfg[device], cls, pnorm, emode, vec, nullptr, norm_w, eps, nullptr, nmat, mats, outs, hid_io, s);
}

// The adapter launches of the executor (dec_lora_down_kernel / dec_lora_up_kernel) behind an op-level fused launch at ONE row: what
// exl_q4_attn / exl_q4_attn_2 / exl_q4_mlp run when the reference's model.py passes LoRA operands per token (model.py:254-289) --
// 2 launches per class instead of 3 per projection (x A as partial + finish, then (x A) B).  outs[i] (+)= (x A_i) B_i; silu: outs[0] /
// outs[1] hold the gate / up products and act receives silu(gate + ..) * (up + ..).  The partial sums live in the device's workspace.
int dec_op_lora(int device, int nmat, const f16* x, const f16* norm_w, float eps, int K, const f16* const* a3, const f16* const* b3, const int* r3,
                f16* const* outs, const int* widths, int silu, f16* act, hipStream_t s)
{
    DecLoraArgs a = {};
    a.x = x; a.norm_w = norm_w; a.eps = eps; a.K = K; a.nmat = nmat;
    int total = 0;
    for (int i = 0; i < nmat; ++i) {
        EXL_REQUIRE(r3[i] >= 0 && r3[i] <= DEC_LORA_MAXR, EXL_E_UNSUPPORTED, "LoRA rank %d beyond %d", r3[i], DEC_LORA_MAXR);
        EXL_REQUIRE(widths[i] % 2 == 0, EXL_E_UNSUPPORTED, "LoRA: odd output width %d", widths[i]);
        a.a[i] = r3[i] > 0 ? a3[i] : nullptr; a.b[i] = r3[i] > 0 ? b3[i] : nullptr; a.r[i] = r3[i];
        a.out[i] = outs[i]; a.n[i] = widths[i];
        total += widths[i];
    }
    float* ws = nullptr;
    EXL_TRY(exl_workspace(device, (size_t) 3 * DEC_LORA_PARTS * DEC_LORA_MAXR, &ws));
    a.silu = silu; a.act = act;
    return dec_lora_launch(a, silu ? widths[0] : total, ws, s);
}

// The split merge rides in the o_proj prologue when a thread owns ONE 8-dim vector of the attention output (hidden <= 4096:
// 16 split loads = 64 registers); wider models keep it as its own kernel (with 2+ vectors per thread hipcc keeps every
// vector's loads live and o_proj, which needs two blocks per CU from hidden 5120 on, drops to one; a rolled loop under a
// 128-register cap spills instead.  Same box, 13B: 177 tokens/s folded vs 193.5 with the merge kernel; 33B: 76.9 vs 81.1),
// where the boundary is also a smaller share of the layer.
// (Round 2 re-test with a ROLLED merge loop writing straight to LDS, no spill: 13B o_proj 7.3 + 2.8 (merge kernel) -> 14.7 us
// folded, 65B 10.0 + 3.0 -> 19.7: every one of the 320-512 o_proj blocks re-reads all split partials (two dependent round
// trips of 16 x 16 bytes per thread); the switch that kept the wide fold reachable is gone.)
static bool dec_folds_merge(const Decoder* d)
{
    return d->qd() <= DEC_THREADS * 8 && !d->lora_o;                  // (an o_proj adapter reads the merged attention output: it must exist)
}

// The adapter launches behind one GEMV launch (dec_lora_down_kernel / dec_lora_up_kernel).  idx: the launch's matrices in the
// order of DecLayer::lora_* (q k v = 0 1 2, o = 3, gate up = 4 5, down = 6).
static int dec_lora(Decoder* d, const DecLayer& l, int first, int nmat, const f16* x, const f16* norm_w, int K, f16* const* outs, const int* widths,
                    int silu, hipStream_t s)
{
    DecLoraArgs a = {};
    a.x = x; a.norm_w = norm_w; a.eps = d->eps; a.K = K; a.nmat = nmat;
    int total = 0;
    for (int i = 0; i < nmat; ++i) {
        a.a[i] = l.lora_a[first + i]; a.b[i] = l.lora_b[first + i]; a.r[i] = l.lora_r[first + i];
        a.out[i] = outs[i]; a.n[i] = widths[i];
        total += widths[i];
    }
    a.silu = silu; a.act = d->act;
    return dec_lora_launch(a, silu ? widths[0] : total, d->lora_part, s);
}

// One kernel class of one layer (EXL_DEC_* in include/exl_amd.h); EXL_DEC_HEAD ignores `i`.
static int dec_launch(Decoder* d, int cls, int i, const int64_t* token_dev, int32_t* pos_dev, float* logits_out, int advance,
                      hipStream_t s)
{
    const DecLayer& l = d->layers[cls == EXL_DEC_HEAD ? 0 : i];
    switch (cls) {
    case EXL_DEC_QKV: {
        Q4Matrix* qkv[3] = {l.q, l.k, l.v};
        f16* qkv_out[3] = {d->qbuf, d->kbuf, d->vbuf};
        const bool emb = i == 0 && d->has_embed();                   // later stages of a layer split start from d->hid
        const f16* xin = emb ? d->embed : d->hid;
        const int64_t* tk = emb ? token_dev : nullptr;
        f16* hc = emb ? d->hid : nullptr;
        const uint16_t* maps[3] = {l.map_q, l.map_k, l.map_v};
        const int rc = launch_dec_gemv(d, 0, 1, 0, xin, tk, l.in_norm, d->eps, hc, 3, qkv, qkv_out, nullptr, s, nullptr, 0, maps);
        if (rc || g_plan || !l.lora_any(0, 2)) return rc;
        const int widths[3] = {l.q->width, l.k->width, l.v->width};   // (d->hid holds the layer's input by now, also behind the embedding lookup)
        return dec_lora(d, l, 0, 3, d->hid, l.in_norm, d->h, qkv_out, widths, 0, s);
    }
    case EXL_DEC_ATTN: {
        const float scale = 1.0f / sqrtf((float) d->hd);
        if (g_plan) { g_plan[0] = 1; g_plan[1] = d->nsplit; g_plan[7] = d->nsplit * d->heads; return 0; }
        // 8 waves in the deepest bucket: where a split can exceed the 160 keys a 4-wave pass holds (<= 4096-wide models, 8 splits per
        // head), and for the wider models (12-16 splits per head of ~128-170 keys: twice the waves per block request twice the rows at a
        // time; round 3, same box: 13B attention 12.96 -> 10.61 us, 407 -> 419 tokens/s; 65B 15.5 -> 14.5 us)
        static const int attn_waves_env = getenv("EXL_DEC_ATTN_WAVES") ? atoi(getenv("EXL_DEC_ATTN_WAVES")) : 0;
        const bool deepest = d->nsplit > 1 && d->nsplit == d->nsplit_max;
        const int attn_waves = attn_waves_env ? attn_waves_env
                             : (deepest && (d->qd() > DEC_THREADS * 8 || (d->max_seq + d->nsplit - 1) / d->nsplit > DEC_ATT_CHUNK)) ? 8 : 4;
#define DEC_ATTN_LAUNCH(SH, NWV, GRID, NS, OUT) hipLaunchKernelGGL((dec_attn_kernel<SH, NWV>), dim3(GRID), dim3(NWV * 64), 0, s, d->qbuf, d->kbuf, \
            d->vbuf, l.kc, l.vc, d->sin, d->cos, d->partial, pos_dev, d->heads, d->kv_heads, d->max_seq, NS, scale, OUT, l.inv_o)
#define DEC_ATTN_SHORT(NWV, GRID, OUT) hipLaunchKernelGGL((dec_attn_short_kernel<NWV>), dim3(GRID), dim3(NWV * 64), 0, s, d->qbuf, d->kbuf, \
            d->vbuf, l.kc, l.vc, d->sin, d->cos, d->partial, pos_dev, d->heads, d->kv_heads, d->max_seq, 1, scale, OUT, l.inv_o)
        if (d->nsplit == 1) {
            if (attn_waves == 8) DEC_ATTN_SHORT(8, d->heads, d->attn_out); else DEC_ATTN_SHORT(4, d->heads, d->attn_out);
        } else {
            if (attn_waves == 8) DEC_ATTN_LAUNCH(false, 8, d->nsplit * d->heads, d->nsplit, (f16*) nullptr);
            else DEC_ATTN_LAUNCH(false, 4, d->nsplit * d->heads, d->nsplit, (f16*) nullptr);
        }
#undef DEC_ATTN_LAUNCH
#undef DEC_ATTN_SHORT
        EXL_LAUNCH_CHECK();
        return 0;
    }
    case EXL_DEC_MERGE:
        // the split merge runs inside the o_proj kernel's prologue (PNORM 3); the stand-alone kernel is the A/B reference
        if (d->nsplit == 1 || dec_folds_merge(d)) return 0;
        if (g_plan) { g_plan[0] = 1; g_plan[1] = d->nsplit; g_plan[7] = d->heads; return 0; }
        hipLaunchKernelGGL(dec_attn_merge_kernel, dim3(d->heads), dim3(128), 0, s, d->partial, d->attn_out, d->nsplit, d->heads, l.inv_o);
        EXL_LAUNCH_CHECK();
        return 0;
    case EXL_DEC_O: {
        Q4Matrix* om[1] = {l.o};
        const f16* res = d->residual_owner ? d->hid : d->zero_res;
        if (d->nsplit > 1 && dec_folds_merge(d)) {                   // merged in this kernel's prologue, then gathered through o_proj's own map
            const uint16_t* maps[1] = {l.map_o};
            return launch_dec_gemv(d, 1, 3, 1, (const f16*) d->partial, nullptr, nullptr, 0.f, nullptr, 1, om, nullptr, d->hid, s,
                                   d->partial + (size_t) d->heads * d->nsplit * 64, d->nsplit, maps, nullptr, res);
        }
        // the attention (one split) / merge kernel stored its output through inv_o: already in o_proj's row order, nothing to gather
        const int rc = launch_dec_gemv(d, 1, 0, 1, d->attn_out, nullptr, nullptr, 0.f, nullptr, 1, om, nullptr, d->hid, s, nullptr, 0, nullptr,
                                       nullptr, res);
        if (rc || g_plan || l.lora_r[3] <= 0) return rc;
        f16* outs[1] = {d->hid};
        const int widths[1] = {l.o->width};
        return dec_lora(d, l, 3, 1, d->attn_out, nullptr, l.o->height, outs, widths, 0, s);
    }
    case EXL_DEC_GATE_UP: {
        Q4Matrix* gu[2] = {l.gate, l.up};
        f16* gu_out[2] = {d->act, nullptr};
        const uint16_t* maps[2] = {l.map_gate, l.map_up};
        if (l.lora_any(4, 5) && !g_plan) {
            // adapters on gate / up act BEFORE SiLU * mul: the two products go out un-fused (the q / k / v form of the launch), the
            // adapter kernels add theirs and apply the activation
            f16* raw[2] = {d->gbuf, d->ubuf};
            EXL_TRY(launch_dec_gemv(d, 2, 1, 0, d->hid, nullptr, l.post_norm, d->eps, nullptr, 2, gu, raw, nullptr, s, nullptr, 0, maps));
            const int widths[2] = {l.gate->width, l.up->width};
            return dec_lora(d, l, 4, 2, d->hid, l.post_norm, d->h, raw, widths, 1, s);
        }
        return launch_dec_gemv(d, 2, 1, 2, d->hid, nullptr, l.post_norm, d->eps, nullptr, 2, gu, gu_out, nullptr, s, nullptr, 0, maps,
                               l.inv_down);                          // the activation is stored in down_proj's row order
    }
    case EXL_DEC_DOWN: {
        Q4Matrix* dm[1] = {l.down};
        const int rc = launch_dec_gemv(d, 3, 0, 1, d->act, nullptr, nullptr, 0.f, nullptr, 1, dm, nullptr, d->hid, s, nullptr, 0, nullptr, nullptr,
                                       d->residual_owner ? d->hid : d->zero_res);
        if (rc || g_plan || l.lora_r[6] <= 0) return rc;
        f16* outs[1] = {d->hid};
        const int widths[1] = {l.down->width};
        return dec_lora(d, l, 6, 1, d->act, nullptr, l.down->height, outs, widths, 0, s);
    }
    case EXL_DEC_HEAD: {
        if (!d->has_head()) return 0;
        const int rows_per_block = 32;
        const int blocks = (d->vocab + rows_per_block - 1) / rows_per_block;
        if (g_plan) { g_plan[0] = 1; g_plan[7] = blocks; return 0; }
        const size_t smem = (size_t) d->h * 2 + 8 * sizeof(float);
        hipLaunchKernelGGL(dec_head_kernel, dim3(blocks), dim3(256), smem, s, d->hid, d->final_norm, d->eps, d->h, d->lm_head,
                           d->vocab, logits_out, rows_per_block, pos_dev, advance, d->head_best);
        EXL_LAUNCH_CHECK();
        return 0;
    }
    }
    EXL_FAIL(EXL_E_INVALID, "decoder: unknown kernel class %d", cls);
}

extern "C" int exl_decoder_step(void* dec, const int64_t* token_dev, int32_t* pos_dev, float* logits_out, int advance,
                                void* stream)
{
    Decoder* d = dec_from(dec);
    EXL_REQUIRE(d, EXL_E_INVALID, "decoder_step: invalid decoder");
    EXL_REQUIRE(pos_dev && (token_dev || !d->has_embed()) && (logits_out || !d->has_head()), EXL_E_INVALID, "decoder_step: null pointer");
    for (const DecLayer& l : d->layers) EXL_REQUIRE(l.set, EXL_E_INVALID, "decoder_step: a layer was not set");
    hipStream_t s = (hipStream_t) stream;
    int prev = 0;
    EXL_HIP(hipGetDevice(&prev));
    if (prev != d->device) EXL_HIP(hipSetDevice(d->device));
    int rc = 0;
    for (int i = 0; i < d->L && rc == 0; ++i)
        for (int cls = EXL_DEC_QKV; cls <= EXL_DEC_DOWN && rc == 0; ++cls)
            rc = dec_launch(d, cls, i, token_dev, pos_dev, logits_out, advance, s);
    if (rc == 0) rc = dec_launch(d, EXL_DEC_HEAD, 0, token_dev, pos_dev, logits_out, advance, s);
    if (rc == 0 && !d->has_head() && advance) {                      // a stage without the head advances its own position
        hipLaunchKernelGGL(dec_advance_kernel, dim3(1), dim3(1), 0, s, pos_dev);
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) { exl_set_error("decoder_step: %s", hipGetErrorString(e)); rc = (int) e; }
    }
    if (prev != d->device) (void) hipSetDevice(prev);
    return rc;
}

// Tensor parallel (exllama_amd/tp.py): a token step in pieces, so that the caller can all-reduce the residual stream between
// them.  part 0 = attention half of `layer` (RMSNorm + q/k/v (+ embedding lookup for layer 0 of a first stage), attention,
// o_proj); part 1 = MLP half (RMSNorm + gate/up + SiLU*mul, down_proj); part 2 = final norm + head (+ position advance).
// After parts 0 and 1 every rank's residual stream (exl_decoder_hidden) holds ITS partial sum -- plus the incoming residual
// on the one rank that owns it (exl_decoder_set_tp) -- and the sum over ranks is the new residual stream.
extern "C" int exl_decoder_step_part(void* dec, int layer, int part, const int64_t* token_dev, int32_t* pos_dev, float* logits_out,
                                     int advance, void* stream)
{
    Decoder* d = dec_from(dec);
    EXL_REQUIRE(d, EXL_E_INVALID, "decoder_step_part: invalid decoder");
    EXL_REQUIRE(part >= 0 && part <= 2, EXL_E_INVALID, "decoder_step_part: unknown part %d", part);
    EXL_REQUIRE(part == 2 || (layer >= 0 && layer < d->L && d->layers[layer].set), EXL_E_INVALID, "decoder_step_part: layer %d not set", layer);
    EXL_REQUIRE(pos_dev && (part != 0 || layer != 0 || token_dev || !d->has_embed()) && (part != 2 || logits_out || !d->has_head()),
                EXL_E_INVALID, "decoder_step_part: null pointer");
    hipStream_t s = (hipStream_t) stream;
    int prev = 0;
    EXL_HIP(hipGetDevice(&prev));
    if (prev != d->device) EXL_HIP(hipSetDevice(d->device));
    int rc = 0;
    if (part == 0) for (int cls = EXL_DEC_QKV; cls <= EXL_DEC_O && rc == 0; ++cls) rc = dec_launch(d, cls, layer, token_dev, pos_dev, logits_out, advance, s);
    else if (part == 1) for (int cls = EXL_DEC_GATE_UP; cls <= EXL_DEC_DOWN && rc == 0; ++cls) rc = dec_launch(d, cls, layer, token_dev, pos_dev, logits_out, advance, s);
    else {
        rc = dec_launch(d, EXL_DEC_HEAD, 0, token_dev, pos_dev, logits_out, advance, s);
        if (rc == 0 && !d->has_head() && advance) {
            hipLaunchKernelGGL(dec_advance_kernel, dim3(1), dim3(1), 0, s, pos_dev);
            const hipError_t e = hipGetLastError();
            if (e != hipSuccess) { exl_set_error("decoder_step_part: %s", hipGetErrorString(e)); rc = (int) e; }
        }
    }
    if (prev != d->device) (void) hipSetDevice(prev);
    return rc;
}

extern "C" int exl_decoder_set_option(void* dec, int option, int value)
{
    Decoder* d = dec_from(dec);
    EXL_REQUIRE(d, EXL_E_INVALID, "decoder_set_option: invalid decoder");
    if (option == EXL_DEC_OPT_RING) d->ring = value;
    else if (option == EXL_DEC_OPT_RING_FENCE) d->ring_fence = value;
    else if (option == EXL_DEC_OPT_RING_DEPTH) d->ring_depth = value;
    else if (option == EXL_DEC_OPT_RING_WIDE) d->ring_wide = value;
    else EXL_FAIL(EXL_E_INVALID, "decoder_set_option: unknown option %d", option);
    return 0;
}

extern "C" int exl_decoder_set_tp(void* dec, int residual_owner)
{
    Decoder* d = dec_from(dec);
    EXL_REQUIRE(d, EXL_E_INVALID, "decoder_set_tp: invalid decoder");
    d->residual_owner = residual_owner != 0;
    return 0;
}

extern "C" int exl_decoder_set_hidden(void* dec, void* hidden_dev)
{
    Decoder* d = dec_from(dec);
    EXL_REQUIRE(d && hidden_dev, EXL_E_INVALID, "decoder_set_hidden: invalid argument");
    d->hid = (f16*) hidden_dev;
    return 0;
}

extern "C" int exl_decoder_hidden(void* dec, void** out)
{
    Decoder* d = dec_from(dec);
    EXL_REQUIRE(d && out, EXL_E_INVALID, "decoder_hidden: invalid argument");
    *out = d->hid;
    return 0;
}

extern "C" int exl_decoder_plan(void* dec, int cls, int* out10)
{
    Decoder* d = dec_from(dec);
    EXL_REQUIRE(d && out10, EXL_E_INVALID, "decoder_plan: invalid argument");
    EXL_REQUIRE(cls >= 0 && cls < EXL_DEC_NCLASS, EXL_E_INVALID, "decoder_plan: unknown kernel class %d", cls);
    for (const DecLayer& l : d->layers) EXL_REQUIRE(l.set, EXL_E_INVALID, "decoder_plan: a layer was not set");
    for (int i = 0; i < 10; ++i) out10[i] = 0;
    g_plan = out10;
    const int rc = dec_launch(d, cls, d->L - 1, nullptr, nullptr, nullptr, 0, nullptr);   // records, launches nothing
    g_plan = nullptr;
    return rc;
}

extern "C" int exl_decoder_step_greedy(void* dec, int64_t* token_io_dev, int32_t* pos_dev, float* logits_out,
                                       int64_t* history_dev, void* stream)
{
    Decoder* d = dec_from(dec);
    EXL_REQUIRE(d, EXL_E_INVALID, "decoder_step_greedy: invalid decoder");
    EXL_REQUIRE(d->has_head(), EXL_E_INVALID, "decoder_step_greedy: this decoder stage has no lm_head");
    const int rc = exl_decoder_step(dec, token_io_dev, pos_dev, logits_out, 1, stream);
    if (rc) return rc;
    int prev = 0;
    EXL_HIP(hipGetDevice(&prev));
    if (prev != d->device) EXL_HIP(hipSetDevice(d->device));
    hipLaunchKernelGGL(dec_argmax_kernel, dim3(1), dim3(1024), 0, (hipStream_t) stream, d->head_best, (d->vocab + 31) / 32, token_io_dev, history_dev, pos_dev);
    const hipError_t e = hipGetLastError();
    if (prev != d->device) (void) hipSetDevice(prev);
    if (e != hipSuccess) EXL_FAIL((int) e, "decoder_step_greedy: %s", hipGetErrorString(e));
    return 0;
}

extern "C" int exl_decoder_step_sample(void* dec, int64_t* token_io_dev, int32_t* pos_dev, float* logits_out, int64_t* history_dev,
                                       const ExlSampler* smp, const float* uniforms_dev, void* stream)
{
    Decoder* d = dec_from(dec);
    EXL_REQUIRE(d && smp && history_dev, EXL_E_INVALID, "decoder_step_sample: invalid argument");
    EXL_REQUIRE(d->has_head(), EXL_E_INVALID, "decoder_step_sample: this decoder stage has no lm_head");
    const int rc = exl_decoder_step(dec, token_io_dev, pos_dev, logits_out, 1, stream);
    if (rc) return rc;
    int prev = 0;
    EXL_HIP(hipGetDevice(&prev));
    if (prev != d->device) EXL_HIP(hipSetDevice(d->device));
    const int r2 = launch_dec_sample(logits_out, d->probs, history_dev, token_io_dev, pos_dev, uniforms_dev, nullptr, d->vocab, smp, (hipStream_t) stream);
    if (prev != d->device) (void) hipSetDevice(prev);
    return r2;
}

// Measurement aid: for each kernel class, `reps` passes over ALL layers' launches of that class back to back (so the
// weights stream from HBM exactly as in a real step: one pass touches every layer's matrices once) between two events on
// `stream`.  class_ms_host[c] = mean time of one pass (= that class' share of one token).  The data flowing through is
// whatever the buffers hold; the K/V slot at *pos_dev is overwritten.  Synchronises.
extern "C" int exl_decoder_step_timed(void* dec, const int64_t* token_dev, int32_t* pos_dev, float* logits_out, int reps,
                                      void* stream, float* class_ms_host)
{
    Decoder* d = dec_from(dec);
    EXL_REQUIRE(d && class_ms_host && pos_dev && (token_dev || !d->has_embed()) && (logits_out || !d->has_head()), EXL_E_INVALID,
                "decoder_step_timed: invalid argument");
    for (const DecLayer& l : d->layers) EXL_REQUIRE(l.set, EXL_E_INVALID, "decoder_step_timed: a layer was not set");
    if (reps < 1) reps = 1;
    hipStream_t s = (hipStream_t) stream;
    int prev = 0;
    EXL_HIP(hipGetDevice(&prev));
    if (prev != d->device) EXL_HIP(hipSetDevice(d->device));
    hipEvent_t e0, e1;
    EXL_HIP(hipEventCreate(&e0));
    EXL_HIP(hipEventCreate(&e1));
    int rc = 0;
    for (int cls = 0; cls < EXL_DEC_NCLASS && rc == 0; ++cls) {
        const int nl = cls == EXL_DEC_HEAD ? 1 : d->L;
        for (int i = 0; i < nl && rc == 0; ++i) rc = dec_launch(d, cls, i, token_dev, pos_dev, logits_out, 0, s);   // warm-up pass
        if (rc) break;
        (void) hipEventRecord(e0, s);
        for (int r = 0; r < reps && rc == 0; ++r)
            for (int i = 0; i < nl && rc == 0; ++i) rc = dec_launch(d, cls, i, token_dev, pos_dev, logits_out, 0, s);
        (void) hipEventRecord(e1, s);
        hipError_t e = hipEventSynchronize(e1);
        if (rc == 0 && e != hipSuccess) { exl_set_error("decoder_step_timed: %s", hipGetErrorString(e)); rc = (int) e; }
        float ms = 0.f;
        if (rc == 0) (void) hipEventElapsedTime(&ms, e0, e1);
        class_ms_host[cls] = ms / (float) reps;
    }
    (void) hipEventDestroy(e0);
    (void) hipEventDestroy(e1);
    if (prev != d->device) (void) hipSetDevice(prev);
    return rc;
}
