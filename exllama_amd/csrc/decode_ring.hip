// Rolling-ring weight stream of the native decode executor (the four GEMV launches of a layer: q/k/v, o_proj, gate/up, down_proj;
// reference: the q4_matmul calls of q4_attn.cu:74-228 and q4_mlp.cu:100-199 at one token).
//
// Same arithmetic, same tile/unit decomposition and the same results, bit for bit, as dec_stream_kernel (decode_fused.hip); what
// changes is WHEN the vector-memory requests are issued.  In dec_stream_kernel the loads are ordinary C++ loads: hipcc places the
// `s_waitcnt vmcnt` itself, and at every control-flow join it has to assume the path with the fewest loads in flight -- the
// disassembly shows `vmcnt(1)` / `vmcnt(0)` in front of the consume step right after the next pass was requested, i.e. the pass
// that was meant to stay in flight is drained first, and the second half of a tile is only requested after the block-wide
// activation barrier.
//
// Here every vector-memory instruction of a wave is inline asm and every wait is counted by hand:
//   * a wave keeps a RING of U 16-byte-per-lane loads in flight: step l of a unit (UL row-blocks of one 16-column tile) waits for
//     slot l % U, consumes it (dequantise + 4 MFMA) and immediately re-requests that slot with the row-block U steps ahead -- of
//     this unit, or of the block's NEXT unit, whose scale / zero words travel just ahead of the weight load that first needs them
//     (one pair of small loads per 4 row-blocks);
//   * the ring is SHALLOW (U <= 4): a CU accepts about 2 KiB of outstanding requests per wave (measured, profiles/HISTORY.md section 3: with 8
//     loads per wave the last wave of a block is still queueing its start-up requests when the first 120 KiB have arrived, and the
//     block-wide activation barrier -- hence all arithmetic -- waits for it); deeper rings only move the waiting from `s_waitcnt` to
//     the issue stage, where it also blocks the barrier;
//   * the start-up order is what the in-order memory pipe of a CU wants: activation loads of ALL waves first (an optional bare
//     s_barrier keeps the first weight requests of the early waves from queueing ahead of the late waves' activation loads),
//     then the entry loads, then the ring;
//   * steady units (a next unit exists) and the last unit of a block are two code paths, so no load is ever conditional between
//     its issue and its wait; the wait counts are not written down by hand but computed at compile time by replaying the issue
//     order (ring_younger below).
// Counting rules (checked mechanically by scripts/isa_lint.py over the built library): a register an asm load is still writing
// is an operand ("+v") of the wait that covers it; compiler-visible stores are ignored by the counts (a store in flight can
// only make a wait stricter); there is no compiler-visible vector load after the first asm load; no scratch.
//
// Covered: group sizes that are multiples of 128 (the scale applies to the fp32 sum of a row-block), no gather map on the
// launch (act-order inputs arrive gathered or the launch falls back to dec_stream_kernel).
#include "decode_ring.h"


// Phase attribution (probe builds only, -DEXL_RING_PROBE; scripts/probe_ring.sh): shader cycles since block start at 7 points,
// per kernel class [0 q/k/v, 1 o_proj + merge, 2 gate/up, 3 plain vector (o_proj behind the merge kernel, down_proj)].
// This file is compiled twice (the critical path of a parallel build is one translation unit): as itself -- the launcher and the
// kernels for group size % 128 == 0 -- and through decode_ring_gm.hip with EXL_RING_PART = 1 -- the group-size 32 / 64 kernels.
#ifndef EXL_RING_PART
#define EXL_RING_PART 0
#endif
#if defined(EXL_RING_PROBE) && EXL_RING_PART == 1
#undef EXL_RING_PROBE                                               /* the stamps live in part 0 */
#endif
#ifdef EXL_RING_PROBE
__device__ unsigned long long g_ring_probe[4 * 512 * 8];
#define RP_CLK(i) rp_t[i] = __builtin_readcyclecounter()
extern "C" int exl_debug_ring_probe(int cls, unsigned long long* out8)     // sums over blocks; out8[7] = number of blocks that reported
{
    static unsigned long long h[512 * 8];
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_ring_probe), sizeof(h), (size_t) cls * sizeof(h)) != hipSuccess) return -1;
    for (int i = 0; i < 8; ++i) out8[i] = 0;
    for (int b = 0; b < 512; ++b) {
        if (!h[b * 8 + 6]) continue;
        for (int i = 0; i < 7; ++i) out8[i] += h[b * 8 + i];
        out8[7] += 1;
    }
    return 0;
}
#else
#define RP_CLK(i) do { } while (0)
#endif


// NW: waves per block (8; 16 for the plain-vector launches that leave one block per CU: twice the requests in flight per CU, half
// the serial chain per wave).  Register budget: two 8-wave blocks or one 16-wave block per CU (128 VGPRs); the longest unrolled
// units (UL >= 24) and the merge prologue (hidden <= 4096: never more blocks than CUs) are allowed 256.
template <int U, int UL, int PNORM, int EMODE, int NV, int NW, int PRE, int GM>
__global__ __launch_bounds__(NW * 64, ((UL >= 24 && NW == 8) || PNORM == 3) ? 2 : 4) void dec_ring_kernel(const DecGemvArgs a)
{
    static_assert(PRE >= 0 && PRE <= U, "ring requests ahead of the activation image");
    constexpr int NT = NW * 64;
    static_assert(ring_valid(U, UL, GM != 0), "one raw entry set: see ring_valid");
    static_assert(GM == 0 || PNORM != 3, "group sizes 32 / 64: the merge stays its own kernel");
    constexpr int NRAW = GM ? U : 1;                                 // raw scale / zero words in flight: per ring slot (GM) or one chunk
    constexpr int WPT = EMODE == 2 ? NW / 2 : NW;                    // waves per tile
    constexpr bool ACT = PNORM == 2;                                 // RMSNorm + act-order: one gathered image per matrix, permuted gate/up store
    constexpr int EL0 = 2 + (EMODE == 1 ? 1 : 0);                    // entry loads of chunk 0 (with the residual value of the column)
    constexpr int MAXP = 4;                                          // ACT, EMODE 2: units per block whose store indices are fetched in the prologue
    constexpr int NIMG = !ACT ? 1 : EMODE == 2 ? 2 : 3;              // act-order: q / k / v (gate / up) each gather through their own map
    constexpr int GT = (ACT && EMODE == 2) ? WPT * 64 : NT;          // threads that build one image (gate: waves 0-3, up: waves 4-7)
    constexpr int GV = NV * (NT / GT);                               // packed rows per thread and image
    constexpr int NMAP = !ACT ? 1 : EMODE == 2 ? GV : 3 * GV;
    constexpr int IMG_ROWS = WPT * UL * 16 > NV * NT ? WPT * UL * 16 : NV * NT;   // packed rows of the image (zero padded)
    constexpr int MS = PNORM == 3 ? DEC_MAX_NSPLIT : 1;
    static_assert(PNORM != 3 || NV == 1, "the merge prologue holds one 8-dim vector per thread");
    static_assert(!ACT || EMODE != 1, "a residual launch reads a producer-permuted vector: nothing to gather");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
#ifdef EXL_RING_PROBE
    unsigned long long rp_t[7] = {0, 0, 0, 0, 0, 0, 0};
    const unsigned long long rp_t0 = __builtin_readcyclecounter();
#endif

    // ---- 0. the few kernel arguments the activation requests need; the matrix views follow once those requests are out (each
    // group of pinned scalars is a scalar-cache round trip: with all three matrix views first, q/k/v issued its activation loads
    // 3,000 cycles into the block)
    // Round 3, second pass: EVERY scalar argument the kernel uses is requested in ONE batch.  The pins used to be one asm
    // statement per field; hipcc then loaded the kernarg segment lazily, group by group, with an `s_waitcnt lgkmcnt(0)` between
    // the groups: five dependent round trips to a scalar cache that is cold for every dispatch (the phase stamps put the ring
    // issue of q/k/v 2,500 cycles after its activation requests).  One asm statement with all fields as operands leaves the
    // compiler no place to wait in between.
    int K = a.mat[0].K, flags = a.ring_flags;
    T16Matrix M0 = a.mat[0], M1 = a.mat[1], M2 = a.mat[2];
    int te0 = a.tile_end[0], te1 = a.tile_end[1], te2 = a.tile_end[2], a_nmat = a.nmat;
    int a_rbw = a.rb_per_wave, nb = a.nblocks, units_lo = a.units_lo, units_rem = a.units_rem;
    uint64_t p_vec = (uint64_t) a.vec, p_nw = (uint64_t) a.norm_w, p_tok = (uint64_t) a.tok, p_res = (uint64_t) a.res_in;
    uint64_t p_q0 = (uint64_t) M0.qw, p_z0 = (uint64_t) M0.qzeros, p_s0 = (uint64_t) M0.scales;
    uint64_t p_q1 = (uint64_t) M1.qw, p_z1 = (uint64_t) M1.qzeros, p_s1 = (uint64_t) M1.scales;
    uint64_t p_q2 = (uint64_t) M2.qw, p_z2 = (uint64_t) M2.qzeros, p_s2 = (uint64_t) M2.scales;
    asm volatile("; kernel arguments: one batch"
                 : "+s"(K), "+s"(flags), "+s"(p_vec), "+s"(p_nw), "+s"(p_tok), "+s"(p_res),
                   "+s"(p_q0), "+s"(p_z0), "+s"(p_s0), "+s"(M0.N), "+s"(M0.RB), "+s"(M0.gprows), "+s"(M0.gshift),
                   "+s"(p_q1), "+s"(p_z1), "+s"(p_s1), "+s"(M1.N), "+s"(M1.RB), "+s"(M1.gprows), "+s"(M1.gshift),
                   "+s"(p_q2), "+s"(p_z2), "+s"(p_s2), "+s"(M2.N), "+s"(M2.RB), "+s"(M2.gprows), "+s"(M2.gshift),
                   "+s"(te0), "+s"(te1), "+s"(te2));
    uint64_t p_g0 = (uint64_t) a.map16[0], p_g1 = (uint64_t) a.map16[1], p_g2 = (uint64_t) a.map16[2], p_op = (uint64_t) a.out_perm;
    asm volatile("" : "+s"(a_nmat), "+s"(a_rbw), "+s"(nb), "+s"(units_lo), "+s"(units_rem), "+s"(p_g0), "+s"(p_g1), "+s"(p_g2), "+s"(p_op));
#define RING_GPTR(T, v) ((T) (std::remove_pointer_t<T> __attribute__((address_space(1)))*) (v))
    const f16* a_vec = RING_GPTR(const f16*, p_vec); const f16* a_norm_w = RING_GPTR(const f16*, p_nw);
    const int64_t* a_tok = RING_GPTR(const int64_t*, p_tok); const f16* a_res = RING_GPTR(const f16*, p_res);
    const uint16_t* a_g0 = RING_GPTR(const uint16_t*, p_g0); const uint16_t* a_g1 = RING_GPTR(const uint16_t*, p_g1);
    const uint16_t* a_g2 = RING_GPTR(const uint16_t*, p_g2); const uint16_t* a_operm = RING_GPTR(const uint16_t*, p_op);
    M0.qw = RING_GPTR(const uint4*, p_q0); M0.qzeros = RING_GPTR(const uint32_t*, p_z0); M0.scales = RING_GPTR(const f16*, p_s0);
    M1.qw = RING_GPTR(const uint4*, p_q1); M1.qzeros = RING_GPTR(const uint32_t*, p_z1); M1.scales = RING_GPTR(const f16*, p_s1);
    M2.qw = RING_GPTR(const uint4*, p_q2); M2.qzeros = RING_GPTR(const uint32_t*, p_z2); M2.scales = RING_GPTR(const f16*, p_s2);
#undef RING_GPTR
    uint4* xs = (uint4*) smem;                                       // [NIMG][IMG_ROWS]
    float* red = (float*) (smem + (size_t) NIMG * IMG_ROWS * 16);    // [2][NW][16] + [NW]
    uint4* xlin = (uint4*) (smem + (size_t) NIMG * IMG_ROWS * 16 + (2 * NW * 16 + NW) * sizeof(float));   // ACT: the normalised vector in its own order, [K / 8]
    uint4* zpad = xlin + (ACT ? K >> 3 : 0);                          // GM: 64 bytes of zeros (the A rows of the lanes that carry no k-group)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rsub = lane >> 4, col = lane & 15;
    const uint32_t lane16 = (uint32_t) lane * 16u;
    const int nvec = K >> 3;

    // ---- 1. activation loads (all waves), then the ring of the first unit -----------------------------------------------------
    const f16* src = a_vec;
    if constexpr (PNORM == 1 || ACT) { if (a_tok) src = a_vec + (size_t) (*a_tok) * K; }
    u32x4 xraw[NV], wraw[NV], mraw[NMAP];
    u32x4 praw[MS];
    uint64_t pml = 0;
    if constexpr (PNORM == 1 || ACT) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int idx = tid + i * NT;
            const int ci = idx < nvec ? idx : 0;
            rg_ld16(xraw[i], src + ci * 8);
        }
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int idx = tid + i * NT;
            const int ci = idx < nvec ? idx : 0;
            rg_ld16(wraw[i], a_norm_w + ci * 8);
        }
        if constexpr (ACT) {                                         // the gather maps of the images this thread builds: 8 16-bit indices per packed row
            if constexpr (EMODE == 2) {
                const uint16_t* gm = (wave / WPT) ? a_g1 : a_g0;
#pragma unroll
                for (int i = 0; i < GV; ++i) {
                    const int idx = tid % GT + i * GT;
                    rg_ld16(mraw[i], gm + (idx < nvec ? idx : 0) * 8);
                }
            } else {
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const uint16_t* gm = k == 0 ? a_g0 : k == 1 ? a_g1 : a_g2;
#pragma unroll
                    for (int i = 0; i < GV; ++i) {
                        const int idx = tid + i * NT;
                        rg_ld16(mraw[k * GV + i], gm + (idx < nvec ? idx : 0) * 8);
                    }
                }
            }
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
            const int idx0 = wave * 64 + i * NT;            // first packed row of this wave's piece (uniform)
            const int idx = idx0 + lane;
            const int ci = idx < nvec ? idx : 0;                     // rows past the end copy row 0 into the padding (finite; never weighted)
            rg_dma16(xs_lds + (uint32_t) idx0 * 16u, src + ci * 8);
        }
    }
    RP_CLK(0);                                                       // activation requests issued
    const int RB = M0.RB;
    const int nunits = EMODE == 2 ? te0 : (a_nmat == 1 ? te0 : a_nmat == 2 ? te1 : te2);
    const int b = blockIdx.x;
    const int n_my = units_lo + (b < units_rem ? 1 : 0);
    const bool remap = (nunits & 7) == 0 && (nb & 7) == 0;           // XCD x (= b % 8) walks one contiguous eighth of the tiles
    const int per = nunits >> 3;
    const int rb_lo = (wave % WPT) * a_rbw;
    const int rb_hi = min(RB, rb_lo + a_rbw);

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
    uint32_t rz[NRAW], rs[NRAW], rres = 0;                           // raw zero / scale words in flight (and the residual)
#pragma unroll
    for (int q = 0; q < NRAW; ++q) { rz[q] = 0; rs[q] = 0; }
    uint32_t pr[MAXP] = {0u, 0u, 0u, 0u};                            // ACT, EMODE 2: out_perm[n] of this lane's column in unit i (fetched once, ahead of the ring:
                                                                     // a per-unit request in flight across the loop back-edge made hipcc copy the register it lands in)
    uint32_t ent = 0;                                                // entries of the chunk being consumed: lane (rsub, col) holds row-block 4 c + rsub
    float res_cur = 0.f;
    auto issue_entries = [&](const RingUnit& u, int chunk) {         // ring_entry_loads(4 * chunk) loads, in this order
        const int n = u.n0 + col;
        const int rb = min(rb_lo + 4 * chunk + rsub, RB - 1);
        const int g = u.gshift >= 0 ? ((rb * 16) >> u.gshift) : ((rb * 16) / u.gprows);
        rg_ld4(rz[0], u.qzeros + (size_t) g * (u.N >> 3) + (n >> 3));
        rg_ld2(rs[0], u.scales + (size_t) g * u.N + n);
        if (EMODE == 1 && chunk == 0) rg_ld2(rres, (const uint16_t*) a_res + n);
    };
    // GM (group size 32 / 64, gshift 2 / 3): the pair of lane (k-group rsub, column col) for step t, into slot t % U.  Group of the
    // lane = first group of the row-block (uniform) + (4 rsub >> gshift): uniform base per step, 32-bit lane offset per unit.
    auto issue_entries_gm = [&](const RingUnit& u, auto tc) {
        constexpr int t = decltype(tc)::value;
        const int rb = min(rb_lo + t, RB - 1);
        const uint32_t gb = (uint32_t) (rb * 16) >> (uint32_t) u.gshift;     // uniform
        const uint32_t gl = (uint32_t) (rsub * 4) >> (uint32_t) u.gshift;
        const uint32_t n8 = (uint32_t) u.N >> 3;
        rg_ld4s(rz[t % NRAW], (gl * n8 + ((uint32_t) col >> 3)) * 4u, u.qzeros + (size_t) gb * n8 + ((uint32_t) u.n0 >> 3));
        rg_ld2s(rs[t % NRAW], (gl * (uint32_t) u.N + (uint32_t) col) * 2u, u.scales + (size_t) gb * (uint32_t) u.N + (uint32_t) u.n0);
        if (EMODE == 1 && t == 0) rg_ld2(rres, (const uint16_t*) a_res + u.n0 + col);
    };
    auto combine_entries = [&](int chunk) {                          // after the wait that covers the raw words
        rg_tie(rz[0]); rg_tie(rs[0]);
        const uint32_t e = (rs[0] & 0xFFFFu) | ((0xE401u + ((rz[0] >> (uint32_t) ((col & 7) * 4)) & 0xFu)) << 16);
        ent = (rb_lo + 4 * chunk + rsub < rb_hi) ? e : 0u;           // rows past the wave's range: scale 0 -> contribute nothing
        if (EMODE == 1 && chunk == 0) { rg_tie(rres); res_cur = (float) __builtin_bit_cast(f16, (uint16_t) rres); }
    };
    // request step `t` (a compile-time constant) of unit u into its slot, entries of its chunk first when it opens one
    auto issue_step = [&](const RingUnit& u, auto tc) {
        constexpr int t = decltype(tc)::value;
        if constexpr (GM != 0) issue_entries_gm(u, tc);
        else if constexpr (t % 4 == 0) issue_entries(u, t / 4);
        const int rb = min(rb_lo + t, RB - 1);                        // clamped: always a valid address
        rg_ldw(ring[t % U], lane16, u.wbase + (size_t) (uint32_t) rb * 1024u);
    };

    if constexpr (ACT && EMODE == 2) {                               // column n of silu(gate) * up is stored at out_perm[n] (the consumer's inverse gather map)
#pragma unroll
        for (int i = 0; i < MAXP; ++i) {
            const RingUnit u = describe(i < n_my ? i : 0);
            rg_ld2(pr[i], a_operm + u.n0 + col);
        }
    }
    if (flags & 1) asm volatile("s_barrier" ::: "memory");           // every wave's activation request is queued before any weight request
    RingUnit cur = describe(0);
    // PRE of the U ring requests go out now; the rest follows the image barrier: a wave that is still queueing requests cannot
    // reach that barrier, and nothing is consumed before it (PRE = U: the whole ring first)
    static_for<0, PRE>([&](auto jc) { issue_step(cur, std::integral_constant<int, (UL - U + decltype(jc)::value) % U>{}); });
    RP_CLK(1);                                                       // first ring requests issued
    // zero padding of the image: slots past a wave's range read it (finite x, scale 0)
    for (int idx = tid; idx < IMG_ROWS; idx += NT)
        if (idx >= nvec && (PNORM != 0 || idx >= NV * NT)) {
#pragma unroll
            for (int k = 0; k < NIMG; ++k) xs[k * IMG_ROWS + idx] = make_uint4(0u, 0u, 0u, 0u);
        }

    // ---- 2. activation image --------------------------------------------------------------------------------------------------
    if constexpr (GM != 0) { if (tid < 4) zpad[tid] = make_uint4(0u, 0u, 0u, 0u); }
    rg_wait<ring_prologue_ops(U, UL, EL0, PRE, GM != 0)>();          // everything older than the ring requests has landed
    RP_CLK(2);                                                       // activation landed
    if constexpr (ACT && EMODE == 2) {
#pragma unroll
        for (int i = 0; i < MAXP; ++i) rg_tie(pr[i]);
    }
    if constexpr (PNORM == 1 || ACT) {
        f16x8 xv[NV];
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            rg_tie(xraw[i]); rg_tie(wraw[i]);
            const int idx = tid + i * NT;
            xv[i] = __builtin_bit_cast(f16x8, xraw[i]);
            if (idx < nvec) {
                if (a_tok && a.hid_copy && b == 0) *(f16x8*) (a.hid_copy + idx * 8) = xv[i];
#pragma unroll
                for (int j = 0; j < 8; ++j) { const float f = (float) xv[i][j]; ss = fmaf(f, f, ss); }
            }
        }
        ss = dec_wave_sum(ss);
        if (lane == 0) red[2 * NW * 16 + wave] = ss;
        rg_barrier();
        float total = 0.f;
#pragma unroll
        for (int i = 0; i < NW; ++i) total += red[2 * NW * 16 + i];
        const f16 rm = (f16) (1.0f / sqrtf(total * (1.0f / (float) K) + a.eps));
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const f16x8 nw = __builtin_bit_cast(f16x8, wraw[i]);
#pragma unroll
            for (int j = 0; j < 8; ++j) { const f16 t = xv[i][j] * rm; xv[i][j] = t * nw[j]; }
            const int idx = tid + i * NT;
            if (idx < nvec) (ACT ? xlin : xs)[idx] = __builtin_bit_cast(uint4, xv[i]);
        }
        if constexpr (ACT) {                                         // reference: column_remap.cu:7-36, x'[c] = x[x_map[c]], per matrix
            rg_barrier();
            const f16* xl = (const f16*) xlin;
            auto gather_row = [&](u32x4& m, uint4* dst) {
                rg_tie(m);
                f16x8 g;
                g[0] = xl[m[0] & 0xFFFFu]; g[1] = xl[m[0] >> 16]; g[2] = xl[m[1] & 0xFFFFu]; g[3] = xl[m[1] >> 16];
                g[4] = xl[m[2] & 0xFFFFu]; g[5] = xl[m[2] >> 16]; g[6] = xl[m[3] & 0xFFFFu]; g[7] = xl[m[3] >> 16];
                *dst = __builtin_bit_cast(uint4, g);
            };
            if constexpr (EMODE == 2) {
                uint4* img = xs + (wave / WPT) * IMG_ROWS;
#pragma unroll
                for (int i = 0; i < GV; ++i) {
                    const int idx = tid % GT + i * GT;
                    if (idx < nvec) gather_row(mraw[i], img + idx);
                }
            } else {
#pragma unroll
                for (int k = 0; k < 3; ++k)
#pragma unroll
                    for (int i = 0; i < GV; ++i) {
                        const int idx = tid + i * NT;
                        if (idx < nvec) gather_row(mraw[k * GV + i], xs + k * IMG_ROWS + idx);
                    }
            }
        }
    } else if constexpr (PNORM == 3) {
        // log-sum-exp merge of the attention splits, per head inside its 16-lane group (dec_stream_kernel, PNORM 3)
        rg_tie(pml);
#pragma unroll
        for (int sp2 = 0; sp2 < MS; ++sp2) rg_tie(praw[sp2]);
        const bool live = (lane & 15) < a.att_nsplit;
        const float pm = live ? __uint_as_float((uint32_t) pml) : -INFINITY;
        const float pl = live ? __uint_as_float((uint32_t) (pml >> 32)) : 0.f;
        const float M = dec_row_max(pm);
        const float lw = pm > -INFINITY ? pl * __expf(pm - M) : 0.f;
        const float L = dec_row_sum(lw);
        const float coef = lw / L;
        float acc8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        static_for<0, MS>([&](auto sc) {
            constexpr int sp2 = decltype(sc)::value;
            const float cf = dec_dpp<0x150 + sp2>(coef);              // row_newbcast: lane sp2 of this 16-lane row
            const f16x8 o8 = __builtin_bit_cast(f16x8, praw[sp2]);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc8[j] = fmaf((float) o8[j], cf, acc8[j]);
        });
        f16x8 r;
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] = (f16) acc8[j];
        if (tid < nvec) xs[tid] = __builtin_bit_cast(uint4, r);
    }
    rg_barrier();
    RP_CLK(3);                                                       // image staged
    static_for<PRE, U>([&](auto jc) { issue_step(cur, std::integral_constant<int, (UL - U + decltype(jc)::value) % U>{}); });

    // ---- 3. walk the units --------------------------------------------------------------------------------------------------------
    const uint32_t magic = t16_magic();
    const uint4* xrow = xs + rb_lo * 16 + rsub * 4;
    auto unit_body = [&](auto last_tag, const RingUnit& uc, const RingUnit& un, int i) {
        constexpr bool LAST = decltype(last_tag)::value;
        f32x4 c = {0.f, 0.f, 0.f, 0.f};
        const uint4* xr = ACT ? xrow + uc.mi * IMG_ROWS : xrow;      // act-order: the image gathered through this matrix' map
        // GM: the A operand carries the activation in four masked rows -- lane (rsub, col) with col == 4 rsub reads its k-group, every other
        // lane the zero pad (gemv_t16.h: t16_rowblock_groups); D[4 g][col] is then the exact-integer sum of k-group g for the lane that holds its scale
        const bool live = col == 4 * rsub;
        const uint4* xg = live ? xr : zpad;
        const int xstep = live ? 16 : 0;
        static_for<0, UL>([&](auto lic) {
            constexpr int li = decltype(lic)::value;
            rg_wait<ring_younger(U, UL, EL0, LAST, li, GM != 0)>(ring[li % U]);
            uint32_t e;
            if constexpr (GM != 0) {
                rg_tie(rz[li % NRAW]); rg_tie(rs[li % NRAW]);
                const uint32_t eg = (rs[li % NRAW] & 0xFFFFu) | ((0xE401u + ((rz[li % NRAW] >> (uint32_t) ((col & 7) * 4)) & 0xFu)) << 16);
                e = rb_lo + li < rb_hi ? eg : 0u;                    // (uniform) rows past the wave's range: scale 0
                if constexpr (EMODE == 1 && li == 0) { rg_tie(rres); res_cur = (float) __builtin_bit_cast(f16, (uint16_t) rres); }
            } else {
                if constexpr (li % 4 == 0) combine_entries(li / 4);
                e = (uint32_t) __shfl((int) ent, ((li & 3) << 4) | col, 64);
            }
            const uint4 w = make_uint4(ring[li % U][0], ring[li % U][1], ring[li % U][2], ring[li % U][3]);
#ifdef EXL_RING_ABLATE                                               /* measurement builds: the loads without the arithmetic */
            c[0] += __uint_as_float((w.x ^ w.w) & 0x3fffffffu) + __uint_as_float(e & 0x3fffffffu);
#else
            if constexpr (GM != 0) t16_rowblock_groups(w, e, magic, xg + li * xstep, c);
            else t16_rowblock<true>(w, e, magic, xr + li * 16, c);
#endif
            if constexpr (li + U < UL) issue_step(uc, std::integral_constant<int, li + U>{});
            else if constexpr (!LAST) issue_step(un, std::integral_constant<int, li % U>{});
        });
        float* rp = red + (i & 1) * NW * 16;
        if constexpr (GM != 0) { c[0] += __shfl_xor(c[0], 16, 64); c[0] += __shfl_xor(c[0], 32, 64); }   // the four k-groups of a column (same order as dec_stream_kernel)
        const float res = res_cur;                                   // (the next unit's entries may already be on their way: res_cur is this unit's)
        const uint32_t perm = (i == 0 ? pr[0] : i == 1 ? pr[1] : i == 2 ? pr[2] : pr[3]) & 0xFFFFu;
        if (lane < 16) rp[wave * 16 + lane] = c[0];
        if (i == 0) RP_CLK(4);                                       // unit 0 consumed
        rg_barrier();
        if (i == 0) RP_CLK(5);                                       // unit 0: all waves through
        if (tid < 16) {
            const int n = uc.n0 + tid;
            if constexpr (EMODE == 2) {
                float g = 0.f, u = 0.f;
#pragma unroll
                for (int k = 0; k < WPT; ++k) { g += rp[k * 16 + tid]; u += rp[(WPT + k) * 16 + tid]; }
                a.out[0][ACT ? (int) perm : n] = silu_mul_f16((f16) g, (f16) u);
            } else {
                float v = 0.f;
#pragma unroll
                for (int k = 0; k < NW; ++k) v += rp[k * 16 + tid];
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
#ifdef EXL_RING_PROBE
    RP_CLK(6);
    if (tid == 0 && b < 512) {
        constexpr int cls = PNORM == 3 ? 1 : EMODE == 2 ? 2 : (PNORM == 1 || ACT) ? 0 : 3;
        unsigned long long* dst = g_ring_probe + ((size_t) cls * 512 + b) * 8;
#pragma unroll
        for (int q = 0; q < 7; ++q) dst[q] = rp_t[q] - rp_t0;
        dst[7] = (unsigned long long) n_my;
    }
#endif
}

// ---------------------------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------------------------
static size_t dec_ring_smem(int UL, int pnorm, int emode, int nv, int nw, int K, int gm = 0)
{
    if (gm) return dec_ring_smem(UL, pnorm, emode, nv, nw, K) + 64;
    const int wpt = emode == 2 ? nw / 2 : nw;
    const int rows = wpt * UL * 16 > nv * nw * 64 ? wpt * UL * 16 : nv * nw * 64;
    const int nimg = pnorm != 2 ? 1 : emode == 2 ? 2 : 3;
    return (size_t) nimg * rows * 16 + (2 * nw * 16 + nw) * sizeof(float) + (pnorm == 2 ? (size_t) K * 2 : 0);
}

template <int U, int UL, int PNORM, int EMODE, int NV, int NW, int GM>
static int ring_go(int grid, int rbw, const DecGemvArgs& a0, hipStream_t s, int* plan)
{
    // (PRE < U -- part of the ring requested only after the activation image is staged -- measured within noise of PRE = U on
    // every 7B class, round 3: the template parameter stays, one value is instantiated)
    auto kfn = dec_ring_kernel<U, UL, PNORM, EMODE, NV, NW, U, GM>;
    const size_t smem = dec_ring_smem(UL, PNORM, EMODE, NV, NW, a0.mat[0].K, GM);
    if (smem > 160 * 1024 / (NW == 8 ? 2 : 1) && PNORM == 2) return 1;   // two blocks per CU must fit: otherwise the compiler stream
    if (plan) {                                                      // exl_decoder_plan: [0] launched, [1] U, [2] UL, [3] 2 = ring kernel, [4] PNORM, [5] EMODE, [6] NV, [9] waves per block
        plan[0] = 1; plan[1] = U; plan[2] = UL; plan[3] = 2; plan[4] = PNORM; plan[5] = EMODE; plan[6] = NV;
        plan[7] = grid; plan[8] = (int) smem; plan[9] = NW;
        return 0;
    }
    DecGemvArgs a = a0;
    a.rb_per_wave = rbw;                                             // (the caller's value is for 8 waves per block)
    static bool big[EXL_MAX_DEVICES] = {};
    if (smem > 64 * 1024) EXL_TRY(exl_lds_opt_in((const void*) kfn, big));
    hipLaunchKernelGGL(kfn, dim3(grid), dim3(NW * 64), smem, s, a);
    EXL_LAUNCH_CHECK();
    return 0;
}

// Row-blocks per wave -> the unit length UL the kernel is instantiated for: exact wherever a Llama shape asks for it (no idle slot:
// an idle slot re-reads a valid row-block with scale 0, real traffic), the next listed value otherwise; 0 = not covered.
// Only the lengths a kernel class can meet are instantiated: NV fixes the range of K, hence of the row-blocks per wave.
#ifdef EXL_DEC_FAST_BUILD                                            /* ISA inspection / experiment builds: the 7B shapes */
#define RING_ULS(X) X(2) X(4) X(6) X(8) X(11)
#else
#define RING_ULS(X) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(16) X(18) X(20) X(22) X(24) X(28) X(32)
#endif
template <int PNORM, int EMODE, int NV, int NW, int GM>
static int ring_cfg(int depth, int rbw, int grid, const DecGemvArgs& a, hipStream_t s, int* plan)
{
    constexpr int WPT = EMODE == 2 ? NW / 2 : NW;
    constexpr int NVP = NV <= 3 ? NV - 1 : (NV == 4 || NV == 6) ? 3 : 6;              // the next smaller instantiated NV (8 waves: 1 2 3 6 8; 16 waves: 1 2 3 4)
    constexpr int KU = NW * 64 * 8;                                                    // K covered by one vector per thread
    constexpr int RBW_HI = NV * KU / 128 / WPT, RBW_LO = NVP * KU / 128 / WPT + 1;     // K in (NVP * KU, NV * KU]
    int prev = 0;
    // (units shorter than the requested depth run with the ring as deep as the unit)
#define RING_ONE(ULV)                                                                                                          \
    if constexpr (((ULV) >= RBW_LO && (ULV) <= RBW_HI) || ((ULV) <= 4 && RBW_LO <= 4 && (ULV) >= 2)) {                         \
        if (rbw > prev && rbw <= (ULV)) {                                                                                      \
            constexpr int UMAX = (ULV) < 4 ? (ULV) : 4;                                                                        \
            if (depth <= 2) return ring_go<2, ULV, PNORM, EMODE, NV, NW, GM>(grid, rbw, a, s, plan);                          \
            if constexpr (UMAX >= 3 && ring_valid(3, ULV, GM != 0)) { if (depth == 3) return ring_go<3, ULV, PNORM, EMODE, NV, NW, GM>(grid, rbw, a, s, plan); } \
            return ring_go<UMAX, ULV, PNORM, EMODE, NV, NW, GM>(grid, rbw, a, s, plan);                                        \
        }                                                                                                                      \
        prev = (ULV);                                                                                                          \
    }
    RING_ULS(RING_ONE)
#undef RING_ONE
    return 1;
}

#define RING_GM(P, E, N) return ring_cfg<P, E, N, 8, 1>(depth, rbw, grid, a, s, plan)
#if EXL_RING_PART == 1
// group sizes 32 / 64 (decode_ring_gm.hip)
int launch_dec_ring_gm(int pnorm, int emode, int nv, int depth, int rbw, int grid, const DecGemvArgs& a, hipStream_t s, int* plan)
{
#ifndef EXL_DEC_FAST_BUILD
    if (pnorm == 1 && emode == 0) { if (nv <= 1) RING_GM(1, 0, 1); if (nv <= 2) RING_GM(1, 0, 2); }
    if (pnorm == 1 && emode == 2) { if (nv <= 1) RING_GM(1, 2, 1); if (nv <= 2) RING_GM(1, 2, 2); }
    if (pnorm == 2 && emode == 0) { if (nv <= 1) RING_GM(2, 0, 1); if (nv <= 2) RING_GM(2, 0, 2); }
    if (pnorm == 2 && emode == 2) { if (nv <= 1) RING_GM(2, 2, 1); if (nv <= 2) RING_GM(2, 2, 2); }
    if (pnorm == 0 && emode == 1) { if (nv <= 1) RING_GM(0, 1, 1); if (nv <= 2) RING_GM(0, 1, 2); if (nv <= 3) RING_GM(0, 1, 3); if (nv <= 6) RING_GM(0, 1, 6); }
#endif
    return 1;
}
#else
int launch_dec_ring_gm(int pnorm, int emode, int nv, int depth, int rbw, int grid, const DecGemvArgs& a, hipStream_t s, int* plan);

int launch_dec_ring(int pnorm, int emode, bool g16, int K, int grid, int depth, bool wide_blocks, const DecGemvArgs& a, hipStream_t s, int* plan)
{
    const bool gm = !g16;                                            // group sizes 32 / 64: per-piece scale / zero pairs
    if (gm) {
        for (int i = 0; i < a.nmat && i < DEC_MAX_MATS; ++i)
            if (a.mat[i].gshift != 2 && a.mat[i].gshift != 3) return 1;     // other group sizes: the compiler stream
    }
    bool any_map = false, all_maps = true;
    for (int i = 0; i < a.nmat && i < DEC_MAX_MATS; ++i) { any_map = any_map || a.map16[i]; all_maps = all_maps && a.map16[i]; }
    if (any_map || a.out_perm) {
        // act-order: the RMSNorm launches gather one image per matrix (q / k / v: 3, gate / up: 2 + the permuted store for down_proj);
        // everything else (a permuted store without a gather, a partial set of maps) stays on the compiler stream
        const bool qkv = emode == 0 && a.nmat == 3 && !a.out_perm, gate_up = emode == 2 && a.nmat == 2 && a.out_perm;
        if (pnorm != 1 || !all_maps || !(qkv || gate_up)) return 1;
        if (gate_up && a.units_lo + (a.units_rem ? 1 : 0) > 4) return 1;      // the store indices of <= 4 units per block are fetched in the prologue
        pnorm = 2;
    }
    const int RB = K / 128;
    const bool wide = wide_blocks && pnorm == 0 && emode == 1 && RB >= 32 && !gm;   // 16 waves: at least two row-blocks per wave
    const int nw = wide ? 16 : DEC_WAVES;
    const int wpt = emode == 2 ? nw / 2 : nw;
    const int rbw = (RB + wpt - 1) / wpt;
    const int nv = (K / 8 + nw * 64 - 1) / (nw * 64);
#define RING_GO(P, E, N, W) return ring_cfg<P, E, N, W, 0>(depth, rbw, grid, a, s, plan)
    if (gm) return launch_dec_ring_gm(pnorm, emode, nv, depth, rbw, grid, a, s, plan);
#ifdef EXL_DEC_FAST_BUILD
    if (pnorm == 1 && emode == 0 && nv == 1) RING_GO(1, 0, 1, 8);
    if (pnorm == 1 && emode == 2 && nv == 1) RING_GO(1, 2, 1, 8);
    if (pnorm == 3 && emode == 1 && nv == 1) RING_GO(3, 1, 1, 8);
    if (pnorm == 0 && emode == 1 && !wide) { if (nv <= 1) RING_GO(0, 1, 1, 8); if (nv <= 3) RING_GO(0, 1, 3, 8); }
    if (pnorm == 0 && emode == 1 && wide) { if (nv <= 1) RING_GO(0, 1, 1, 16); if (nv <= 2) RING_GO(0, 1, 2, 16); }
#else
    if (pnorm == 1 && emode == 0) { if (nv <= 1) RING_GO(1, 0, 1, 8); if (nv <= 2) RING_GO(1, 0, 2, 8); }
    if (pnorm == 1 && emode == 2) { if (nv <= 1) RING_GO(1, 2, 1, 8); if (nv <= 2) RING_GO(1, 2, 2, 8); }
    if (pnorm == 2 && emode == 0) { if (nv <= 1) RING_GO(2, 0, 1, 8); if (nv <= 2) RING_GO(2, 0, 2, 8); }
    if (pnorm == 2 && emode == 2) { if (nv <= 1) RING_GO(2, 2, 1, 8); if (nv <= 2) RING_GO(2, 2, 2, 8); }
    if (pnorm == 3 && emode == 1 && nv <= 1) RING_GO(3, 1, 1, 8);
    if (pnorm == 0 && emode == 1 && !wide) {
        if (nv <= 1) RING_GO(0, 1, 1, 8);
        if (nv <= 2) RING_GO(0, 1, 2, 8);
        if (nv <= 3) RING_GO(0, 1, 3, 8);
        if (nv <= 6) RING_GO(0, 1, 6, 8);
        if (nv <= 8) RING_GO(0, 1, 8, 8);
    }
    if (pnorm == 0 && emode == 1 && wide) {
        if (nv <= 1) RING_GO(0, 1, 1, 16);
        if (nv <= 2) RING_GO(0, 1, 2, 16);
        if (nv <= 3) RING_GO(0, 1, 3, 16);
        if (nv <= 4) RING_GO(0, 1, 4, 16);
    }
#endif
#undef RING_GO
    return 1;
}
#endif
#undef RING_GM
