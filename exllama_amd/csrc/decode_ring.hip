// Rolling-ring weight stream of the native decode executor (the four GEMV launches of a layer: q/k/v, o_proj, gate/up, down_proj;
// reference: the q4_matmul calls of q4_attn.cu:74-228 and q4_mlp.cu:100-199 at one token).
//
// Same arithmetic, same tile/unit decomposition and the same results, bit for bit, as dec_stream_kernel (decode_fused.hip); what
// changes is WHEN the vector-memory requests are issued.  In dec_stream_kernel the loads are ordinary C++ loads: hipcc places the
// `s_waitcnt vmcnt` itself, and at every control-flow join it has to assume the path with the fewest loads in flight -- the
// disassembly shows `vmcnt(1)` / `vmcnt(0)` in front of the consume step right after the next pass was requested, i.e. the pass
// that was meant to stay in flight is drained first, and the second half of a tile is only requested after the block-wide
// activation barrier.  The phase probe put the consequence at 35-50 % of a launch: the memory system idles while waves sit in
// barriers and dequantise.
//
// Here every vector-memory instruction of a wave is inline asm and every wait is counted by hand:
//   * a wave keeps a RING of U 16-byte-per-lane loads (U KiB) in flight at all times: step (p, q) of a unit waits for slot q
//     (`vmcnt(U - 1)`: exactly the U - 1 younger ring loads stay outstanding), consumes it (dequantise + 4 MFMA) and immediately
//     re-requests slot q with the row-block U steps ahead -- of this unit's next pass, or of the block's NEXT unit, whose
//     scale / zero entries travel just ahead of its first weight load;
//   * the start-up order is what the in-order memory pipe of a CU wants: activation loads of ALL waves first (an optional bare
//     s_barrier keeps the first weight requests of the early waves from queueing ahead of the late waves' activation loads),
//     then the small entry loads, then the ring -- the RMSNorm / split merge / LDS-DMA copy then completes while the first U
//     KiB per wave stream in, and nothing else is ever waited for;
//   * steady units (a next unit exists) and the last unit of a block are two code paths with their own static wait counts, so
//     no load is ever conditional between its issue and its wait.
// Counting rules (checked mechanically by scripts/isa_lint.py over the built library): a register an asm load is still writing
// is an operand ("+v") of the wait that covers it; compiler-visible stores are ignored by the counts (a store in flight can
// only make a wait stricter); there is no compiler-visible vector load after the first asm load.
//
// Covered: group sizes that are multiples of 128 (the scale applies to the fp32 sum of a row-block), no gather map on the
// launch (act-order inputs arrive gathered or the launch falls back to dec_stream_kernel).
#include "decode_args.h"

#include <type_traits>
#include <stdlib.h>

namespace {

template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f)
{
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for<I + 1, N>(f); }
}

// ---- hand-counted vector memory -------------------------------------------------------------------------------------------
__device__ __forceinline__ void rg_ldw(u32x4& d, uint32_t voff, const void* sbase)        // weights: uniform base + lane offset, streaming
{
    // (the base IS wave-uniform; when hipcc has moved its arithmetic to the vector ALU under SGPR pressure, the "s" operand
    // would be handed a VGPR pair: make the scalar form explicit -- a no-op where the value already lives in SGPRs)
    const uint64_t b = (uint64_t) sbase;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t) b), hi = __builtin_amdgcn_readfirstlane((uint32_t) (b >> 32));
    const void* sb = (const void*) (((uint64_t) hi << 32) | lo);
    asm volatile("global_load_dwordx4 %0, %1, %2 nt" : "=v"(d) : "v"(voff), "s"(sb) : "memory");
}
__device__ __forceinline__ void rg_ld16(u32x4& d, const void* p) { asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(d) : "v"(p) : "memory"); }
__device__ __forceinline__ void rg_ld8(u32x2& d, const void* p) { asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(d) : "v"(p) : "memory"); }
__device__ __forceinline__ void rg_ld4(uint32_t& d, const void* p) { asm volatile("global_load_dword %0, %1, off" : "=v"(d) : "v"(p) : "memory"); }
__device__ __forceinline__ void rg_ld2(uint32_t& d, const void* p) { asm volatile("global_load_ushort %0, %1, off" : "=v"(d) : "v"(p) : "memory"); }
// 1 KiB of global memory straight into LDS (lds_dst: wave-uniform LDS byte address; lane l lands at lds_dst + 16 l)
__device__ __forceinline__ void rg_dma16(uint32_t lds_dst, const void* gsrc)
{
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
template <int N> __device__ __forceinline__ void rg_wait() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }
template <int N> __device__ __forceinline__ void rg_wait(u32x4& a) { asm volatile("s_waitcnt vmcnt(%1)" : "+v"(a) : "n"(N) : "memory"); }
// after a wait: the value is only defined from here on (no consumer may be scheduled above the wait)
__device__ __forceinline__ void rg_tie(u32x4& a) { asm volatile("" : "+v"(a) :: "memory"); }
__device__ __forceinline__ void rg_tie(u32x2& a) { asm volatile("" : "+v"(a) :: "memory"); }
__device__ __forceinline__ void rg_tie(uint32_t& a) { asm volatile("" : "+v"(a) :: "memory"); }
// block barrier that knows nothing about vector memory: LDS traffic of this wave done, then s_barrier
__device__ __forceinline__ void rg_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ uint32_t rg_lds_addr(const void* p)
{
    return (uint32_t) (uintptr_t) (__attribute__((address_space(3))) const unsigned char*) p;
}

struct RingUnit {                  // wave-uniform description of one unit of this wave
    const unsigned char* wbase;    // first byte of the 16-column tile
    const uint32_t* qzeros;
    const uint16_t* scales;
    int N, gshift, gprows;
    int n0;                        // first column of the tile
    int mi;                        // matrix index (EMODE 0: which output)
};

}  // namespace

// OCC: blocks per CU the register budget is cut for (2 -> at most 128 VGPRs, 1 -> 256).
template <int U, int NP, int PNORM, int EMODE, int NV, int OCC>
__global__ __launch_bounds__(DEC_THREADS, 2 * OCC) void dec_ring_kernel(const DecGemvArgs a)
{
    constexpr int UL = U * NP;                                       // row-block slots per wave and unit
    constexpr int NSLOT = (U + 3) / 4;                               // entry words per lane and PASS (one per 4 row-blocks)
    constexpr int WPT = EMODE == 2 ? DEC_WAVES / 2 : DEC_WAVES;      // waves per tile
    constexpr int EL = 2 * NSLOT + (EMODE == 1 ? 1 : 0);             // small loads that travel ahead of a pass' first weight load
    constexpr int IMG_ROWS = WPT * UL * 16 > NV * DEC_THREADS ? WPT * UL * 16 : NV * DEC_THREADS;   // packed rows of the image (zero padded)
    constexpr int MS = PNORM == 3 ? DEC_MAX_NSPLIT : 1;
    static_assert(PNORM != 3 || NV == 1, "the merge prologue holds one 8-dim vector per thread");
    static_assert(U + EL + 2 * NV + MS + 1 < 60, "vmcnt is a 6-bit counter");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    // ---- 0. kernel arguments into SGPRs (one batch of scalar loads) ----------------------------------------------------------
    T16Matrix M0 = a.mat[0], M1 = a.mat[1], M2 = a.mat[2];
    dec_pin(M0); dec_pin(M1); dec_pin(M2);
    int K = M0.K;
    const f16* a_vec = dec_pin_ptr(a.vec); const f16* a_norm_w = dec_pin_ptr(a.norm_w); const int64_t* a_tok = dec_pin_ptr(a.tok);
    const f16* a_res = dec_pin_ptr(a.res_in);
    int te0 = a.tile_end[0], te1 = a.tile_end[1], te2 = a.tile_end[2], a_nmat = a.nmat;
    int a_rbw = a.rb_per_wave, nb = a.nblocks, units_lo = a.units_lo, units_rem = a.units_rem, flags = a.ring_flags;
    DEC_PIN_S(K); DEC_PIN_S(te0); DEC_PIN_S(te1); DEC_PIN_S(te2); DEC_PIN_S(a_nmat); DEC_PIN_S(a_rbw);
    DEC_PIN_S(nb); DEC_PIN_S(units_lo); DEC_PIN_S(units_rem); DEC_PIN_S(flags);
    const int RB = M0.RB;
    uint4* xs = (uint4*) smem;                                       // [IMG_ROWS]
    float* red = (float*) (smem + (size_t) IMG_ROWS * 16);           // [2][DEC_WAVES][16] + [DEC_WAVES]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rsub = lane >> 4, col = lane & 15;
    const uint32_t lane16 = (uint32_t) lane * 16u;
    const int nunits = EMODE == 2 ? te0 : (a_nmat == 1 ? te0 : a_nmat == 2 ? te1 : te2);
    const int b = blockIdx.x;
    const int n_my = units_lo + (b < units_rem ? 1 : 0);
    const bool remap = (nunits & 7) == 0 && (nb & 7) == 0;           // XCD x (= b % 8) walks one contiguous eighth of the tiles
    const int per = nunits >> 3;
    const int rb_lo = (wave % WPT) * a_rbw;
    const int rb_hi = min(RB, rb_lo + a_rbw);
    const int nvec = K >> 3;

    auto describe = [&](int i) {
        const int v = b + i * nb;
        const int g = remap ? (v & 7) * per + (v >> 3) : v;
        int mi = 0, tile = g;
        if constexpr (EMODE == 2) {
            mi = wave / WPT;                                         // waves 0-3: gate tile g, waves 4-7: up tile g
        } else {
            if (a_nmat > 1 && g >= te0) { mi = 1; tile = g - te0; }
            if (a_nmat > 2 && g >= te1) { mi = 2; tile = g - te1; }
        }
        const T16Matrix m = dec_pick(M0, M1, M2, mi);
        RingUnit u;
        u.wbase = (const unsigned char*) m.qw + (size_t) (uint32_t) tile * (uint32_t) m.RB * 1024u;
        u.qzeros = m.qzeros; u.scales = (const uint16_t*) m.scales;
        u.N = m.N; u.gshift = m.gshift; u.gprows = m.gprows;
        u.n0 = tile * 16; u.mi = mi;
        return u;
    };

    // ---- ring state ---------------------------------------------------------------------------------------------------------
    u32x4 ring[U];
    uint32_t rz[NSLOT], rs[NSLOT], rres = 0;                         // raw scale / zero words (and the residual) of the pass in flight
    uint32_t ent[NSLOT];                                             // entries of the pass being consumed
    auto issue_w = [&](const RingUnit& u, auto qc, int li) {        // slot q <- row-block rb_lo + li of unit u (clamped: always a valid address)
        constexpr int q = decltype(qc)::value;
        const int rb = min(rb_lo + li, RB - 1);
        rg_ldw(ring[q], lane16, u.wbase + (size_t) (uint32_t) rb * 1024u);
    };
    // The scale / zero words of pass p of unit u -- slot h of lane (rsub, col): row-block p U + 4 h + rsub, column col -- and the
    // residual value of the unit's column: EL loads, in this order, issued just ahead of the pass' first weight load.
    auto issue_entries = [&](const RingUnit& u, int p) {
        const int n = u.n0 + col;
#pragma unroll
        for (int h = 0; h < NSLOT; ++h) {
            const int rb = min(rb_lo + p * U + 4 * h + rsub, RB - 1);
            const int g = u.gshift >= 0 ? ((rb * 16) >> u.gshift) : ((rb * 16) / u.gprows);
            rg_ld4(rz[h], u.qzeros + (size_t) g * (u.N >> 3) + (n >> 3));
            rg_ld2(rs[h], u.scales + (size_t) g * u.N + n);
        }
        if constexpr (EMODE == 1) rg_ld2(rres, (const uint16_t*) a_res + n);
    };
    float res_cur = 0.f;
    auto combine_entries = [&](int p) {                              // after the wait that covers the raw words
#pragma unroll
        for (int h = 0; h < NSLOT; ++h) {
            rg_tie(rz[h]); rg_tie(rs[h]);
            const uint32_t e = (rs[h] & 0xFFFFu) | ((0xE401u + ((rz[h] >> (uint32_t) ((col & 7) * 4)) & 0xFu)) << 16);
            ent[h] = (rb_lo + p * U + 4 * h + rsub < rb_hi) ? e : 0u;   // rows past the wave's range: scale 0 -> contribute nothing
        }
        if constexpr (EMODE == 1) { rg_tie(rres); if (p == 0) res_cur = (float) __builtin_bit_cast(f16, (uint16_t) rres); }
    };

    // ---- 1. activation loads (all waves), then entries + ring of the first unit ---------------------------------------------
    const f16* src = a_vec;
    if constexpr (PNORM == 1) { if (a_tok) src = a_vec + (size_t) (*a_tok) * K; }
    u32x4 xraw[NV], wraw[NV];
    u32x4 praw[MS];
    u32x2 pml = {0u, 0u};
    if constexpr (PNORM == 1) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int idx = tid + i * DEC_THREADS;
            const int ci = idx < nvec ? idx : 0;
            rg_ld16(xraw[i], src + ci * 8);
        }
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int idx = tid + i * DEC_THREADS;
            const int ci = idx < nvec ? idx : 0;
            rg_ld16(wraw[i], a_norm_w + ci * 8);
        }
    } else if constexpr (PNORM == 3) {
        // 16 consecutive 8-dim vectors = one head; lane (l & 15) also fetches (max, sum) of split l & 15 of that head
        const int ci = tid < nvec ? tid : 0;
        const int hd = ci >> 4, sp = min(lane & 15, a.att_nsplit - 1);
        rg_ld8(pml, a.att_ml + ((size_t) hd * a.att_nsplit + sp) * 2);
        const uint4* base = (const uint4*) (src + (size_t) hd * a.att_nsplit * 128 + (ci & 15) * 8);
#pragma unroll
        for (int sp2 = 0; sp2 < MS; ++sp2) rg_ld16(praw[sp2], base + min(sp2, a.att_nsplit - 1) * 16);   // splits beyond nsplit re-read the last one (coefficient 0)
    } else {
        const uint32_t xs_lds = rg_lds_addr(xs);
#pragma unroll
        for (int i = 0; i < NV; ++i) {                              // PNORM 0: plain copy, 1 KiB per wave instruction straight into LDS
            const int idx0 = wave * 64 + i * DEC_THREADS;            // first packed row of this wave's piece (uniform)
            const int idx = idx0 + lane;
            const int ci = idx < nvec ? idx : 0;                     // rows past the end copy row 0 into the padding (finite; never weighted)
            rg_dma16(xs_lds + (uint32_t) idx0 * 16u, src + ci * 8);
        }
    }
    if (flags & 1) asm volatile("s_barrier" ::: "memory");           // every wave's activation request is queued before any weight request
    RingUnit cur = describe(0);
    issue_entries(cur, 0);
    static_for<0, U>([&](auto qc) { issue_w(cur, qc, decltype(qc)::value); });
    // zero padding of the image: slots past a wave's range read it (finite x, scale 0)
    for (int idx = tid; idx < IMG_ROWS; idx += DEC_THREADS)
        if (idx >= nvec && (PNORM != 0 || idx >= NV * DEC_THREADS)) xs[idx] = make_uint4(0u, 0u, 0u, 0u);

    // ---- 2. activation image --------------------------------------------------------------------------------------------------
    rg_wait<EL + U>();                                               // everything older than the entries has landed
    if constexpr (PNORM == 1) {
        f16x8 xv[NV];
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            rg_tie(xraw[i]); rg_tie(wraw[i]);
            const int idx = tid + i * DEC_THREADS;
            xv[i] = __builtin_bit_cast(f16x8, xraw[i]);
            if (idx < nvec) {
                if (a_tok && a.hid_copy && b == 0) *(f16x8*) (a.hid_copy + idx * 8) = xv[i];
#pragma unroll
                for (int j = 0; j < 8; ++j) { const float f = (float) xv[i][j]; ss = fmaf(f, f, ss); }
            }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) ss += __shfl_xor(ss, off, 64);
        if (lane == 0) red[2 * DEC_WAVES * 16 + wave] = ss;
        rg_barrier();
        float total = 0.f;
#pragma unroll
        for (int i = 0; i < DEC_WAVES; ++i) total += red[2 * DEC_WAVES * 16 + i];
        const f16 rm = (f16) (1.0f / sqrtf(total * (1.0f / (float) K) + a.eps));
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const f16x8 nw = __builtin_bit_cast(f16x8, wraw[i]);
#pragma unroll
            for (int j = 0; j < 8; ++j) { const f16 t = xv[i][j] * rm; xv[i][j] = t * nw[j]; }
            const int idx = tid + i * DEC_THREADS;
            if (idx < nvec) xs[idx] = __builtin_bit_cast(uint4, xv[i]);
        }
    } else if constexpr (PNORM == 3) {
        // log-sum-exp merge of the attention splits, per head inside its 16-lane group (dec_stream_kernel, PNORM 3)
        rg_tie(pml);
#pragma unroll
        for (int sp2 = 0; sp2 < MS; ++sp2) rg_tie(praw[sp2]);
        const bool live = (lane & 15) < a.att_nsplit;
        const float pm = live ? __builtin_bit_cast(float, pml[0]) : -INFINITY;
        const float pl = live ? __builtin_bit_cast(float, pml[1]) : 0.f;
        float M = pm;
#pragma unroll
        for (int off = 1; off < 16; off <<= 1) M = fmaxf(M, __shfl_xor(M, off, 64));
        const float lw = pm > -INFINITY ? pl * __expf(pm - M) : 0.f;
        float L = lw;
#pragma unroll
        for (int off = 1; off < 16; off <<= 1) L += __shfl_xor(L, off, 64);
        const float coef = lw / L;
        float acc8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int sp2 = 0; sp2 < MS; ++sp2) {
            const float cf = __shfl(coef, sp2, 16);
            const f16x8 o8 = __builtin_bit_cast(f16x8, praw[sp2]);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc8[j] = fmaf((float) o8[j], cf, acc8[j]);
        }
        f16x8 r;
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] = (f16) acc8[j];
        if (tid < nvec) xs[tid] = __builtin_bit_cast(uint4, r);
    }
    rg_barrier();

    // ---- 3. walk the units --------------------------------------------------------------------------------------------------------
    const uint32_t magic = t16_magic();
    const uint4* xrow = xs + rb_lo * 16 + rsub * 4;
    auto unit_body = [&](auto last_tag, const RingUnit& uc, const RingUnit& un, int i) {
        constexpr bool LAST = decltype(last_tag)::value;
        f32x4 c = {0.f, 0.f, 0.f, 0.f};
        static_for<0, UL>([&](auto lic) {
            constexpr int li = decltype(lic)::value;
            constexpr int p = li / U, q = li % U;
            // younger vector-memory instructions than the load of (p, q): the other U - 1 ring loads, plus -- past slot 0 -- the
            // entry loads that went out at slot 0 of this round (next pass / next unit); in the last pass of a block's last
            // unit the ring runs empty
            constexpr bool reissue = p + 1 < NP || !LAST;
            constexpr int cnt = reissue ? U - 1 + (q >= 1 ? EL : 0) : U - 1 - q;
            rg_wait<cnt>(ring[q]);
            if constexpr (q == 0) combine_entries(p);
            const uint32_t e = (uint32_t) __shfl((int) ent[q >> 2], ((q & 3) << 4) | col, 64);
            const uint4 w = make_uint4(ring[q][0], ring[q][1], ring[q][2], ring[q][3]);
            t16_rowblock<true>(w, e, magic, xrow + li * 16, c);
            if constexpr (p + 1 < NP) {
                if constexpr (q == 0) issue_entries(uc, p + 1);
                issue_w(uc, std::integral_constant<int, q>{}, li + U);
            } else if constexpr (!LAST) {
                if constexpr (q == 0) issue_entries(un, 0);
                issue_w(un, std::integral_constant<int, q>{}, q);
            }
        });
        float* rp = red + (i & 1) * DEC_WAVES * 16;
        const float res = res_cur;                                   // (the next unit's entries may already be on their way: res_cur is this unit's)
        if (lane < 16) rp[wave * 16 + lane] = c[0];
        rg_barrier();
        if (tid < 16) {
            const int n = uc.n0 + tid;
            if constexpr (EMODE == 2) {
                float g = 0.f, u = 0.f;
#pragma unroll
                for (int k = 0; k < WPT; ++k) { g += rp[k * 16 + tid]; u += rp[(WPT + k) * 16 + tid]; }
                a.out[0][n] = silu_mul_f16((f16) g, (f16) u);
            } else {
                float v = 0.f;
#pragma unroll
                for (int k = 0; k < DEC_WAVES; ++k) v += rp[k * 16 + tid];
                if constexpr (EMODE == 0) a.out[uc.mi][n] = (f16) v;
                else a.hid_io[n] = (f16) (v + res);
            }
        }
    };
    int i = 0;
    for (; i + 1 < n_my; ++i) {
        const RingUnit nxt = describe(i + 1);
        unit_body(std::false_type{}, cur, nxt, i);
        cur = nxt;
    }
    unit_body(std::true_type{}, cur, cur, i);
}

// ---------------------------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------------------------
size_t dec_ring_smem(int U, int NP, int emode, int nv)
{
    const int wpt = emode == 2 ? DEC_WAVES / 2 : DEC_WAVES;
    const int rows = wpt * U * NP * 16 > nv * DEC_THREADS ? wpt * U * NP * 16 : nv * DEC_THREADS;
    return (size_t) rows * 16 + (2 * DEC_WAVES * 16 + DEC_WAVES) * sizeof(float);
}

template <int U, int NP, int PNORM, int EMODE, int NV, int OCC>
static int ring_go(int grid, const DecGemvArgs& a, hipStream_t s, int* plan)
{
    auto kfn = dec_ring_kernel<U, NP, PNORM, EMODE, NV, OCC>;
    const size_t smem = dec_ring_smem(U, NP, EMODE, NV);
    if (plan) {                                                      // exl_decoder_plan: [0] launched, [1] U, [2] NP, [3] 2 = ring kernel, [4] PNORM, [5] EMODE, [6] NV
        plan[0] = 1; plan[1] = U; plan[2] = NP; plan[3] = 2; plan[4] = PNORM; plan[5] = EMODE; plan[6] = NV;
        plan[7] = grid; plan[8] = (int) smem; plan[9] = 1;
        return 0;
    }
    static bool big[EXL_MAX_DEVICES] = {};
    if (smem > 64 * 1024) EXL_TRY(exl_lds_opt_in((const void*) kfn, big));
    hipLaunchKernelGGL(kfn, dim3(grid), dim3(DEC_THREADS), smem, s, a);
    EXL_LAUNCH_CHECK();
    return 0;
}

// (U, NP): U loads in flight per lane, NP passes; U * NP >= rbw with as few idle slots as possible (an idle slot re-reads a
// valid row-block with scale 0: real traffic).  two_per_cu: the grid has more blocks than CUs, so two blocks must be co-resident
// (128 VGPRs); otherwise one block per CU may take 256.  Only the (U, NP) a kernel class can meet are instantiated: NV fixes
// the range of K, hence of the row-blocks per wave.
template <int PNORM, int EMODE, int NV>
static int ring_cfg(int rbw, int grid, bool two_per_cu, const DecGemvArgs& a, hipStream_t s, int* plan)
{
    constexpr int WPT = EMODE == 2 ? DEC_WAVES / 2 : DEC_WAVES;
    constexpr int NVP = NV == 1 ? 0 : NV == 2 ? 1 : NV == 3 ? 2 : NV == 6 ? 3 : 6;   // the next smaller instantiated NV
    constexpr int RBW_HI = NV * 32 / WPT, RBW_LO = NVP * 32 / WPT + 1;               // K in (NVP * 4096, NV * 4096]
    // rbw in (TLO, THI] -> <U, NP, OCC>
#define RING(TLO, THI, U, NP, OCC) if constexpr ((THI) >= RBW_LO && (TLO) < RBW_HI) { if (rbw > (TLO) && rbw <= (THI)) return ring_go<U, NP, PNORM, EMODE, NV, OCC>(grid, a, s, plan); }
#ifdef EXL_DEC_FAST_BUILD                                            /* ISA inspection builds: 7B instantiations only */
    if (two_per_cu) { RING(0, 4, 4, 1, 2) } else { RING(0, 4, 4, 1, 1) }
    RING(4, 8, 8, 1, 2)
    if (!two_per_cu) { RING(8, 11, 11, 1, 1) }
    return 1;
#else
    if constexpr (PNORM != 0) {                                      // K = hidden size (<= 8192): q/k/v, o_proj with the merge, gate/up
        if (two_per_cu) { RING(0, 4, 4, 1, 2) RING(4, 5, 5, 1, 2) } else { RING(0, 4, 4, 1, 1) RING(4, 5, 5, 1, 1) }
        RING(5, 6, 6, 1, 2)
        RING(6, 8, 8, 1, 2)
        if constexpr (EMODE == 2) {
            RING(8, 10, 5, 2, 2)
            RING(10, 12, 6, 2, 2)
            RING(12, 14, 7, 2, 2)
            RING(14, 16, 8, 2, 2)
        }
        return 1;
    } else {                                                         // o_proj behind the merge kernel (K = hidden), down_proj (K = intermediate size)
        if (two_per_cu) { RING(0, 4, 4, 1, 2) RING(4, 5, 5, 1, 2) RING(5, 8, 8, 1, 2) }
        else {
            RING(0, 4, 4, 1, 1) RING(4, 5, 5, 1, 1) RING(5, 8, 8, 1, 1)
            RING(8, 11, 11, 1, 1)
            RING(11, 12, 12, 1, 1)
            RING(12, 14, 14, 1, 1)
            RING(21, 22, 11, 2, 1)
        }
        RING(8, 12, 6, 2, 2)
        RING(12, 14, 7, 2, 2)
        RING(14, 16, 8, 2, 2)
        RING(16, 18, 6, 3, 2)
        return 1;                                                    // (deeper two-per-CU streams spill under 128 VGPRs: dec_stream_kernel keeps them)
    }
#endif
#undef RING
}

int launch_dec_ring(int pnorm, int emode, bool g16, int rbw, int nv, int grid, bool two_per_cu, const DecGemvArgs& a, hipStream_t s,
                    int* plan)
{
    if (!g16 || a.out_perm) return 1;
    for (int i = 0; i < DEC_MAX_MATS; ++i)
        if (a.map16[i]) return 1;
#define RING_GO(P, E, N) return ring_cfg<P, E, N>(rbw, grid, two_per_cu, a, s, plan)
#ifdef EXL_DEC_FAST_BUILD
    if (pnorm == 1 && emode == 0 && nv == 1) RING_GO(1, 0, 1);
    if (pnorm == 1 && emode == 2 && nv == 1) RING_GO(1, 2, 1);
    if (pnorm == 3 && emode == 1 && nv == 1) RING_GO(3, 1, 1);
    if (pnorm == 0 && emode == 1) { if (nv <= 1) RING_GO(0, 1, 1); if (nv <= 3) RING_GO(0, 1, 3); }
#else
    if (pnorm == 1 && emode == 0) { if (nv <= 1) RING_GO(1, 0, 1); if (nv <= 2) RING_GO(1, 0, 2); }
    if (pnorm == 1 && emode == 2) { if (nv <= 1) RING_GO(1, 2, 1); if (nv <= 2) RING_GO(1, 2, 2); }
    if (pnorm == 3 && emode == 1 && nv <= 1) RING_GO(3, 1, 1);
    if (pnorm == 0 && emode == 1) {
        if (nv <= 1) RING_GO(0, 1, 1);
        if (nv <= 2) RING_GO(0, 1, 2);
        if (nv <= 3) RING_GO(0, 1, 3);
        if (nv <= 6) RING_GO(0, 1, 6);
        if (nv <= 8) RING_GO(0, 1, 8);
    }
#endif
#undef RING_GO
    return 1;
}
