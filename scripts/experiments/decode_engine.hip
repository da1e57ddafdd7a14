// The MLP half of a decode layer as ONE launch: RMSNorm -> gate / up -> SiLU * mul -> down_proj -> + residual
// (reference: q4_mlp.cu:100-199 at one token; here until round 5: two launches, dec_ring_kernel<.., EMODE 2> and <.., EMODE 1>).
//
// Why one launch: at 7B shapes a GEMV launch is 1.7 us of boundary + ~1.5 us of start-up (arguments, first requests, latency) in
// front of 5-7 us of streaming, and the down_proj launch cannot request a byte before gate / up has drained.  Inside one launch
// down_proj's weights do not depend on anything: a wave requests its share of the block's down_proj tile as the LAST requests of
// its gate / up ring (the ring rolls from the block's last gate / up unit into the down_proj tile exactly as it rolls from unit
// to unit), so the tile is in registers when the activation vector is complete.  What is left exposed is the edge itself.
//
// The edge (every block needs all `inter` activations; MI355X_MICROARCH.md rows handoff-flag / fanin / barrier-xcd;
// cdna_hip_programming.md guideline 16, recipe R1):
//   * producers: the wave of a half-block that stores the 16 activations of each unit stores them WRITE-THROUGH (sc1), drains its
//     own stores after its last unit (s_waitcnt vmcnt(0)) and adds 1 to one of 8 arrival counters (agent scope, one per XCD-sized
//     share of the half-blocks, 256 bytes apart: 64 arrivals per counter instead of 512 on one word);
//   * consumers: ONE wave per block polls the 8 counters (one relaxed sc1 load of 8 lanes, s_sleep between polls, bounded), the
//     block passes a barrier, then every wave copies its part of the activation vector global -> LDS with sc1 loads (agent-scope
//     loads: no fence, no L1 / L2 invalidate);
//   * the counters are monotonic and never reset: a block reads them when it starts (it has not arrived itself, so fewer than one
//     launch's arrivals can be ahead of the last multiple of `arrivals`) and waits for the next multiple.  Nothing to zero between
//     launches or graph replays; a counter wraps at 2^32 = a multiple of `arrivals` (a power of two).
// Requirements the launcher checks: one block per CU, all blocks co-resident (grid = number of CUs; 1024 threads and <= 128 VGPRs
// admit exactly one block per CU); nothing else may occupy CUs of the device for the duration of the launch -- a second process
// on the same device can delay it, two ENGINE launches of two processes interleaved can wedge it: the poll is bounded
// (`spin_limit`), a block that gives up sets the decoder's `wedged` word and finishes on whatever it has (exl_decoder_engine_status
// reports it; the executor then runs the five-launch layer).
//
// Block = 16 waves = two HALVES of 8 waves that walk gate / up units exactly like two 8-wave blocks of dec_ring_kernel (half h of
// block b is "virtual block" v = b + h * gridDim: same unit -> block mapping, same wave split of K, same order of additions: the
// activations are bit-identical to the two-launch form), sharing one activation image and one barrier per unit; then all 16 waves
// split K of ONE down_proj tile like the 16-wave form of dec_ring_kernel (bit-identical again).
#include "decode_ring.h"

#include <mutex>

struct DecMlpArgs {
    const f16* x;                 // residual stream [hidden]
    const f16* norm_w;
    float eps;
    T16Matrix gate, up, down;
    int units;                    // 16-column tiles of gate (= of up)
    int tiles_b;                  // 16-column tiles of down_proj (<= gridDim)
    f16* act;                     // [inter]: silu(gate) * up, stored sc1, read sc1
    f16* hid_io;                  // residual stream out
    const f16* res_in;            // residual in (= hid_io, or zeros on the tensor-parallel ranks that do not own it)
    uint32_t* sync;               // [8] arrival counters at a stride of 64 dwords, [512] wedged flag
    int nblocks;                  // = gridDim.x
    int units_lo, units_rem;      // units per half-block: units_lo + (v < units_rem), v = b + half * nblocks
    int arrivals;                 // half-blocks per counter and launch (2 * nblocks / 8, a power of two)
    int ring_flags;               // bit 0: barrier between the activation requests and the first weight requests
    uint32_t spin_limit;          // polls before a block gives up
};

#ifdef EXL_ENGINE_PROBE
__device__ unsigned long long g_engine_probe[256 * 8];
#define EP_CLK(i) ep_t[i] = __builtin_readcyclecounter()
extern "C" int exl_debug_engine_probe(unsigned long long* out8)       // sums over blocks; out8[7] = blocks that reported
{
    static unsigned long long h[256 * 8];
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_engine_probe), sizeof(h)) != hipSuccess) return -1;
    for (int i = 0; i < 8; ++i) out8[i] = 0;
    for (int b = 0; b < 256; ++b) {
        if (!h[b * 8 + 6]) continue;
        for (int i = 0; i < 7; ++i) out8[i] += h[b * 8 + i];
        out8[7] += 1;
    }
    return 0;
}
#else
#define EP_CLK(i) do { } while (0)
#endif

namespace {
// agent-scope (sc1) forms: see the header comment
__device__ __forceinline__ void en_ld4_sc1(uint32_t& d, const void* p) { asm volatile("global_load_dword %0, %1, off sc1" : "=v"(d) : "v"(p) : "memory"); }
__device__ __forceinline__ void en_st2_sc1(void* p, uint32_t v) { asm volatile("global_store_short %0, %1, off sc1" :: "v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ void en_st4_sc1(void* p, uint32_t v) { asm volatile("global_store_dword %0, %1, off sc1" :: "v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ void en_add_sc1(void* p, uint32_t v) { asm volatile("global_atomic_add %0, %1, off sc1" :: "v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ void en_dma16_sc1(uint32_t lds_dst, const void* gsrc)
{
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off sc1\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
}  // namespace

// U: ring depth of the gate / up stream, UL: row-blocks per wave and gate / up unit (hidden / 128 / 4), ULB: row-blocks per wave of the
// down_proj tile (ceil(inter / 128 / 16)).
template <int U, int UL, int ULB>
__global__ __launch_bounds__(1024) void dec_mlp_kernel(const DecMlpArgs a)
{
    constexpr int NW = 16, NT = NW * 64, WPT = 4, EL0 = 2;
    constexpr int NCHB = (ULB + 3) / 4;                               // entry chunks of the down_proj tile
    static_assert(ring_valid(U, UL), "one raw entry set: see ring_valid");
    static_assert(ULB >= U && ULB <= 8, "the ring's tail requests the first U steps of the down_proj tile");
    constexpr int IMG_A = WPT * UL * 16 > NT ? WPT * UL * 16 : NT;   // packed rows (16 bytes) of the gate / up image
    constexpr int IMG_B = NW * ULB * 16;                              // ... of the down_proj image (zero / finite padded)
    constexpr int IMG = IMG_A > IMG_B ? IMG_A : IMG_B;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
#ifdef EXL_ENGINE_PROBE
    unsigned long long ep_t[7] = {0, 0, 0, 0, 0, 0, 0};
    const unsigned long long ep_t0 = __builtin_readcyclecounter();
#endif
    // ---- 0. every scalar argument in one batch (decode_ring.hip: why) ----------------------------------------------------------
    T16Matrix MG = a.gate, MU = a.up, MD = a.down;
    int K = a.gate.K, flags = a.ring_flags, units = a.units, tiles_b = a.tiles_b, nb = a.nblocks, units_lo = a.units_lo, units_rem = a.units_rem;
    int arrivals = a.arrivals;
    uint64_t p_x = (uint64_t) a.x, p_nw = (uint64_t) a.norm_w, p_act = (uint64_t) a.act, p_res = (uint64_t) a.res_in, p_sync = (uint64_t) a.sync;
    uint64_t p_q0 = (uint64_t) MG.qw, p_z0 = (uint64_t) MG.qzeros, p_s0 = (uint64_t) MG.scales;
    uint64_t p_q1 = (uint64_t) MU.qw, p_z1 = (uint64_t) MU.qzeros, p_s1 = (uint64_t) MU.scales;
    uint64_t p_q2 = (uint64_t) MD.qw, p_z2 = (uint64_t) MD.qzeros, p_s2 = (uint64_t) MD.scales;
    asm volatile("; kernel arguments: one batch"
                 : "+s"(K), "+s"(flags), "+s"(p_x), "+s"(p_nw), "+s"(p_act), "+s"(p_res), "+s"(p_sync),
                   "+s"(p_q0), "+s"(p_z0), "+s"(p_s0), "+s"(MG.N), "+s"(MG.RB), "+s"(MG.gprows), "+s"(MG.gshift),
                   "+s"(p_q1), "+s"(p_z1), "+s"(p_s1),
                   "+s"(p_q2), "+s"(p_z2), "+s"(p_s2), "+s"(MD.N), "+s"(MD.RB), "+s"(MD.gprows), "+s"(MD.gshift),
                   "+s"(units), "+s"(tiles_b), "+s"(nb), "+s"(units_lo), "+s"(units_rem));
    asm volatile("" : "+s"(arrivals));
#define EN_GPTR(T, v) ((T) (std::remove_pointer_t<T> __attribute__((address_space(1)))*) (v))
    const f16* a_x = EN_GPTR(const f16*, p_x); const f16* a_norm_w = EN_GPTR(const f16*, p_nw);
    f16* a_act = EN_GPTR(f16*, p_act); const f16* a_res = EN_GPTR(const f16*, p_res); uint32_t* a_sync = EN_GPTR(uint32_t*, p_sync);
    MG.qw = EN_GPTR(const uint4*, p_q0); MG.qzeros = EN_GPTR(const uint32_t*, p_z0); MG.scales = EN_GPTR(const f16*, p_s0);
    MU.qw = EN_GPTR(const uint4*, p_q1); MU.qzeros = EN_GPTR(const uint32_t*, p_z1); MU.scales = EN_GPTR(const f16*, p_s1);
    MD.qw = EN_GPTR(const uint4*, p_q2); MD.qzeros = EN_GPTR(const uint32_t*, p_z2); MD.scales = EN_GPTR(const f16*, p_s2);
#undef EN_GPTR
    uint4* xs = (uint4*) smem;                                        // [IMG]: the normalised x (gate / up), later the activation vector (down_proj)
    float* red = (float*) (smem + (size_t) IMG * 16);                 // [2][NW][16] + [NW]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = wave >> 3, w8 = wave & 7, mi = w8 >> 2;          // waves 0-3 of a half: gate tile, 4-7: up tile
    const int rsub = lane >> 4, col = lane & 15;
    const uint32_t lane16 = (uint32_t) lane * 16u;
    const int nvec = K >> 3;
    const int b = blockIdx.x;

    // ---- 1. activation requests (all waves), the arrival counters as they stand, then the ring of the first unit -------------------
    u32x4 xraw, wraw;
    uint32_t c0;
    {
        const int ci = tid < nvec ? tid : 0;
        rg_ld16(xraw, a_x + ci * 8);
        rg_ld16(wraw, a_norm_w + ci * 8);
        en_ld4_sc1(c0, a_sync + (lane & 7) * 64);                     // (every wave: the issue order is the same in all of them; wave 15 polls)
    }
    EP_CLK(0);
    const int RB = MG.RB;
    const int v = b + half * nb;                                      // the 8-wave block of the two-launch form this half stands for
    const int n_my = units_lo + (v < units_rem ? 1 : 0);
    const int n_rounds = units_lo + (b < units_rem ? 1 : 0);          // = n_my of half 0 (>= that of half 1)
    const int nb2 = nb * 2;
    const bool remap = (units & 7) == 0 && (nb2 & 7) == 0;            // XCD x (= b % 8 = v % 8) walks one contiguous eighth of the tiles
    const int per = units >> 3;
    const int rb_lo = (w8 & 3) * UL;
    const int rb_hi = min(RB, rb_lo + UL);

    auto describe = [&](int i) {
        const int vv = v + i * nb2;
        const int g = remap ? (vv & 7) * per + (vv >> 3) : vv;
        const T16Matrix m = mi ? MU : MG;
        RingUnit u;
        u.wbase = (const unsigned char*) m.qw + (size_t) (uint32_t) g * (uint32_t) m.RB * 1024u;
        u.qzeros = m.qzeros; u.scales = (const uint16_t*) m.scales;
        u.N = MG.N; u.gshift = MG.gshift; u.gprows = MG.gprows;      // (gate and up share shape and group size: the launcher checks)
        u.n0 = g * 16; u.mi = mi;
        return u;
    };
    // the block's down_proj tile
    const bool remap_b = (tiles_b & 7) == 0 && (nb & 7) == 0;
    const int tb0 = remap_b ? (b & 7) * (tiles_b >> 3) + (b >> 3) : b;
    const int tile_b = tb0 < tiles_b ? tb0 : tiles_b - 1;             // blocks beyond the tiles recompute the last one and store nothing
    const int RBB = MD.RB;
    const int rbb_lo = wave * ULB, rbb_hi = min(RBB, rbb_lo + ULB);
    const unsigned char* wbase_b = (const unsigned char*) MD.qw + (size_t) (uint32_t) tile_b * (uint32_t) RBB * 1024u;
    const int n0b = tile_b * 16;

    // ---- ring state ---------------------------------------------------------------------------------------------------------------
    u32x4 ring[U];
    uint32_t rz = 0, rs = 0;
    uint32_t ent = 0;
    auto issue_entries = [&](const RingUnit& u, int chunk) {
        const int n = u.n0 + col;
        const int rb = min(rb_lo + 4 * chunk + rsub, RB - 1);
        const int g = u.gshift >= 0 ? ((rb * 16) >> u.gshift) : ((rb * 16) / u.gprows);
        rg_ld4(rz, u.qzeros + (size_t) g * (u.N >> 3) + (n >> 3));
        rg_ld2(rs, u.scales + (size_t) g * u.N + n);
    };
    auto combine_entries = [&](int chunk) {
        rg_tie(rz); rg_tie(rs);
        const uint32_t e = (rs & 0xFFFFu) | ((0xE401u + ((rz >> (uint32_t) ((col & 7) * 4)) & 0xFu)) << 16);
        ent = (rb_lo + 4 * chunk + rsub < rb_hi) ? e : 0u;
    };
    auto issue_step = [&](const RingUnit& u, auto tc) {
        constexpr int t = decltype(tc)::value;
        if constexpr (t % 4 == 0) issue_entries(u, t / 4);
        const int rb = min(rb_lo + t, RB - 1);
        rg_ldw(ring[t % U], lane16, u.wbase + (size_t) (uint32_t) rb * 1024u);
    };
    // the down_proj tile: every step has its own registers (ULB <= 8: 32 VGPRs), every entry chunk its own raw words
    u32x4 bw[ULB];
    uint32_t rzb[NCHB], rsb[NCHB], rresb = 0;
#pragma unroll
    for (int q = 0; q < NCHB; ++q) { rzb[q] = 0; rsb[q] = 0; }
    auto issue_entries_b = [&](auto cc) {                             // 2 loads, like a gate / up chunk
        constexpr int chunk = decltype(cc)::value;
        const int n = n0b + col;
        const int rb = min(rbb_lo + 4 * chunk + rsub, RBB - 1);
        const int g = MD.gshift >= 0 ? ((rb * 16) >> MD.gshift) : ((rb * 16) / MD.gprows);
        rg_ld4(rzb[chunk], MD.qzeros + (size_t) g * (MD.N >> 3) + (n >> 3));
        rg_ld2(rsb[chunk], (const uint16_t*) MD.scales + (size_t) g * MD.N + n);
    };
    auto issue_step_b = [&](auto tc) {
        constexpr int t = decltype(tc)::value;
        if constexpr (t % 4 == 0) issue_entries_b(std::integral_constant<int, t / 4>{});
        const int rb = min(rbb_lo + t, RBB - 1);
        rg_ldw(bw[t], lane16, wbase_b + (size_t) (uint32_t) rb * 1024u);
    };

    if (flags & 1) asm volatile("s_barrier" ::: "memory");            // every wave's activation request is queued before any weight request
    RingUnit cur = describe(0);
    static_for<0, U>([&](auto jc) { issue_step(cur, std::integral_constant<int, (UL - U + decltype(jc)::value) % U>{}); });
    for (int idx = tid; idx < IMG_A; idx += NT)                       // zero padding of the image: slots past a wave's range read it (finite x, scale 0)
        if (idx >= nvec) xs[idx] = make_uint4(0u, 0u, 0u, 0u);

    // ---- 2. the gate / up image: RMSNorm (same order of additions as the 8-wave kernel: threads beyond the vector add zeros) -------
    rg_wait<ring_prologue_ops(U, UL, EL0, U, false)>();
    {
        rg_tie(xraw); rg_tie(wraw);
        f16x8 xv = __builtin_bit_cast(f16x8, xraw);
        float ss = 0.f;
        if (tid < nvec) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float f = (float) xv[j]; ss = fmaf(f, f, ss); }
        }
        ss = dec_wave_sum(ss);
        if (lane == 0) red[2 * NW * 16 + wave] = ss;
        rg_barrier();
        float total = 0.f;
#pragma unroll
        for (int i = 0; i < NW; ++i) total += red[2 * NW * 16 + i];
        const f16 rm = (f16) (1.0f / sqrtf(total * (1.0f / (float) K) + a.eps));
        const f16x8 nw = __builtin_bit_cast(f16x8, wraw);
#pragma unroll
        for (int j = 0; j < 8; ++j) { const f16 t = xv[j] * rm; xv[j] = t * nw[j]; }
        if (tid < nvec) xs[tid] = __builtin_bit_cast(uint4, xv);
    }
    rg_barrier();
    EP_CLK(1);                                                        // image staged

    // ---- 3. gate / up units of this half --------------------------------------------------------------------------------------------
    const uint32_t magic = t16_magic();
    const uint4* xrow = xs + rb_lo * 16 + rsub * 4;
    // TOB: the unit is the half's last one -- the ring rolls into the block's down_proj tile (same issue order and counts as a roll
    // into the next gate / up unit: step 0 carries two entry loads, every step one weight load)
    auto unit_body = [&](auto tob_tag, const RingUnit& uc, const RingUnit& un, int i) {
        constexpr bool TOB = decltype(tob_tag)::value;
        f32x4 c = {0.f, 0.f, 0.f, 0.f};
        static_for<0, UL>([&](auto lic) {
            constexpr int li = decltype(lic)::value;
            rg_wait<ring_younger(U, UL, EL0, false, li, false)>(ring[li % U]);
            if constexpr (li % 4 == 0) combine_entries(li / 4);
            const uint32_t e = (uint32_t) __shfl((int) ent, ((li & 3) << 4) | col, 64);
            const uint4 w = make_uint4(ring[li % U][0], ring[li % U][1], ring[li % U][2], ring[li % U][3]);
            t16_rowblock<true>(w, e, magic, xrow + li * 16, c);
            if constexpr (li + U < UL) issue_step(uc, std::integral_constant<int, li + U>{});
            else if constexpr (!TOB) issue_step(un, std::integral_constant<int, li % U>{});
            else issue_step_b(std::integral_constant<int, li % U>{});
        });
        float* rp = red + (i & 1) * NW * 16;
        if (lane < 16) rp[wave * 16 + lane] = c[0];
        rg_barrier();
        if ((tid & 511) < 16) {                                       // wave 0 of each half
            const int t = tid & 15;
            const float* hp = rp + half * 8 * 16;
            float g = 0.f, u = 0.f;
#pragma unroll
            for (int k = 0; k < WPT; ++k) { g += hp[k * 16 + t]; u += hp[(WPT + k) * 16 + t]; }
            const f16 r = silu_mul_f16((f16) g, (f16) u);
            en_st2_sc1(a_act + uc.n0 + t, (uint32_t) __builtin_bit_cast(uint16_t, r));
        }
    };
    int i = 0;
    for (; i + 1 < n_my; ++i) {
        const RingUnit nxt = describe(i + 1);
        unit_body(std::false_type{}, cur, nxt, i);
        cur = nxt;
    }
    unit_body(std::true_type{}, cur, cur, i);
    ++i;
    EP_CLK(2);                                                        // this half's gate / up units done
    // publish: the storing wave drains its write-through stores (and the first down_proj requests, long since issued), then arrives
    if (w8 == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) en_add_sc1(a_sync + (v & 7) * 64, 1u);
    }
    // the rest of the down_proj tile (nothing is counted from here on: the next wait is vmcnt(0))
    static_for<U, ULB>([&](auto tc) { issue_step_b(tc); });
    rg_ld2(rresb, (const uint16_t*) a_res + n0b + col);
    EP_CLK(3);
    for (; i < n_rounds; ++i) rg_barrier();                           // the other half is still walking units: its barriers are the block's

    // ---- 4. the edge: all gate / up units of all blocks stored ----------------------------------------------------------------------
    if (wave == NW - 1 && !(flags & 4)) {                             // (the wave with the fewest down_proj rows; flags bit 2: measurement only, no wait)
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(c0) :: "memory");
        const uint32_t target = (c0 & ~(uint32_t) (arrivals - 1)) + (uint32_t) arrivals;
        uint32_t spins = 0;
        for (;;) {
            uint32_t cn;
            en_ld4_sc1(cn, a_sync + (lane & 7) * 64);
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(cn) :: "memory");
            const bool ok = (int32_t) (cn - target) >= 0;
            if (__builtin_amdgcn_ballot_w64(ok) == ~0ull) break;     // (lanes 8-63 repeat lanes 0-7)
            if (++spins > a.spin_limit) {                             // wedged (see the header comment): say so and go on
                if (lane == 0) en_st4_sc1(a_sync + 512, 1u);
                break;
            }
            asm volatile("s_sleep 8" ::: "memory");
        }
    }
    rg_barrier();
    EP_CLK(4);                                                        // edge passed
    if (flags & 2) asm volatile("buffer_inv sc1" ::: "memory");       // A/B: an agent-scope acquire in front of the activation loads
    // the activation vector, global -> LDS, agent-scope loads (1 KiB per wave instruction)
    {
        const uint32_t xs_lds = rg_lds_addr(xs);
        const int nvb = MD.K >> 3;
#pragma unroll
        for (int q = 0; q < (IMG_B + NT - 1) / NT; ++q) {
            const int idx0 = wave * 64 + q * NT;                      // (uniform)
            if (idx0 < IMG_B) {
                const int idx = idx0 + lane;
                const int ci = idx < nvb ? idx : 0;                   // rows past the end copy row 0 into the padding (finite; never weighted)
                en_dma16_sc1(xs_lds + (uint32_t) idx0 * 16u, a_act + ci * 8);
            }
        }
    }
    static_for<0, ULB>([&](auto tc) { rg_wait<0>(bw[decltype(tc)::value]); });
    rg_barrier();
    EP_CLK(5);                                                        // activation vector staged, the whole tile in registers

    // ---- 5. down_proj: 16 waves split K of one tile (dec_ring_kernel, NW = 16: same split, same order) --------------------------------
    {
        uint32_t entb[NCHB];
#pragma unroll
        for (int q = 0; q < NCHB; ++q) {
            rg_tie(rzb[q]); rg_tie(rsb[q]);
            const uint32_t e = (rsb[q] & 0xFFFFu) | ((0xE401u + ((rzb[q] >> (uint32_t) ((col & 7) * 4)) & 0xFu)) << 16);
            entb[q] = (rbb_lo + 4 * q + rsub < rbb_hi) ? e : 0u;
        }
        rg_tie(rresb);
        const float res = (float) __builtin_bit_cast(f16, (uint16_t) rresb);
        const uint4* xrb = xs + rbb_lo * 16 + rsub * 4;
        f32x4 c = {0.f, 0.f, 0.f, 0.f};
        static_for<0, ULB>([&](auto lic) {
            constexpr int li = decltype(lic)::value;
            const uint32_t e = (uint32_t) __shfl((int) entb[li >> 2], ((li & 3) << 4) | col, 64);
            const uint4 w = make_uint4(bw[li][0], bw[li][1], bw[li][2], bw[li][3]);
            t16_rowblock<true>(w, e, magic, xrb + li * 16, c);
        });
        float* rp = red + (n_rounds & 1) * NW * 16;
        if (lane < 16) rp[wave * 16 + lane] = c[0];
        rg_barrier();
        if (tid < 16 && tb0 < tiles_b) {
            float vsum = 0.f;
#pragma unroll
            for (int k = 0; k < NW; ++k) vsum += rp[k * 16 + tid];
            a.hid_io[n0b + tid] = (f16) (vsum + res);
        }
    }
#ifdef EXL_ENGINE_PROBE
    EP_CLK(6);
    if (tid == 0 && b < 256) {
        unsigned long long* dst = g_engine_probe + (size_t) b * 8;
#pragma unroll
        for (int q = 0; q < 7; ++q) dst[q] = ep_t[q] - ep_t0;
        dst[7] = (unsigned long long) n_rounds;
    }
#endif
}

// ---------------------------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------------------------
static size_t dec_mlp_smem(int UL, int ULB)
{
    const int img_a = 4 * UL * 16 > 1024 ? 4 * UL * 16 : 1024, img_b = 16 * ULB * 16;
    return (size_t) (img_a > img_b ? img_a : img_b) * 16 + (2 * 16 * 16 + 16) * sizeof(float);
}

template <int U, int UL, int ULB>
static int mlp_go(const DecMlpArgs& a, hipStream_t s, int* plan)
{
    auto kfn = dec_mlp_kernel<U, UL, ULB>;
    const size_t smem = dec_mlp_smem(UL, ULB);
    if (plan) {                                                       // exl_decoder_plan: [3] = 3: the fused gate/up -> down launch
        plan[0] = 1; plan[1] = U; plan[2] = UL; plan[3] = 3; plan[4] = 1; plan[5] = 2; plan[6] = ULB; plan[7] = a.nblocks; plan[8] = (int) smem; plan[9] = 16;
        return 0;
    }
    hipLaunchKernelGGL(kfn, dim3(a.nblocks), dim3(1024), smem, s, a);
    EXL_LAUNCH_CHECK();
    return 0;
}

// Returns 1 when the shapes are not covered (the caller launches gate / up and down_proj separately), 0 on success.
int launch_dec_mlp(const T16Matrix& gate, const T16Matrix& up, const T16Matrix& down, const f16* x, const f16* norm_w, float eps, f16* act,
                   f16* hid_io, const f16* res_in, uint32_t* sync, int cus, int depth, int ring_flags, uint32_t spin_limit, hipStream_t s, int* plan)
{
    if (gate.K != up.K || gate.N != up.N || gate.gprows != up.gprows || down.K != gate.N || down.N != gate.K) return 1;
    if (gate.gprows % 16 != 0 || down.gprows % 16 != 0) return 1;                     // group sizes that are multiples of 128
    if (gate.K % 512 != 0 || gate.K > 8192 || gate.N % 16 != 0 || down.N % 16 != 0) return 1;
    const int nb = cus;
    if (nb < 8 || nb > 256 || (nb & (nb - 1)) != 0) return 1;                         // arrivals per counter: a power of two
    const int units = gate.N / 16, tiles_b = down.N / 16;
    if (units < 2 * nb || tiles_b > nb) return 1;                                     // every half-block walks >= 1 unit; one down_proj tile per block
    const int ul = gate.RB / 4, ulb = (down.RB + 15) / 16;
    DecMlpArgs a;
    a.x = x; a.norm_w = norm_w; a.eps = eps; a.gate = gate; a.up = up; a.down = down;
    a.units = units; a.tiles_b = tiles_b; a.act = act; a.hid_io = hid_io; a.res_in = res_in ? res_in : hid_io; a.sync = sync;
    a.nblocks = nb; a.units_lo = units / (2 * nb); a.units_rem = units % (2 * nb);
    a.arrivals = 2 * nb / 8; a.ring_flags = ring_flags; a.spin_limit = spin_limit;
    const int u = depth <= 2 ? 2 : 3;
#define MLP_B(UV, ULV) switch (ulb) {                                                                     \
        case 3: return mlp_go<UV, ULV, 3>(a, s, plan);  case 4: return mlp_go<UV, ULV, 4>(a, s, plan);     \
        case 5: return mlp_go<UV, ULV, 5>(a, s, plan);  case 6: return mlp_go<UV, ULV, 6>(a, s, plan);     \
        case 7: return mlp_go<UV, ULV, 7>(a, s, plan);  case 8: return mlp_go<UV, ULV, 8>(a, s, plan);     \
        default: return 1; }
    if (ul == 8) { if (u == 2) MLP_B(2, 8) else MLP_B(3, 8) }
    if (ul == 4) { MLP_B(3, 4) }
#undef MLP_B
    return 1;
}
