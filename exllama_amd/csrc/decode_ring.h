// Shared pieces of the hand-counted weight streams (decode_ring.hip: the rolling-ring GEMV launches; decode_engine.hip: the fused
// gate/up -> down_proj launch): the inline-asm vector-memory primitives, the wave-uniform unit description and the compile-time
// replay of a wave's issue order from which every `s_waitcnt vmcnt(N)` is instantiated.
#pragma once
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
    // s_nop 4: a VALU write of an SGPR (the v_readfirstlane above, when it is real) needs 5 wait states before a vector-memory
    // instruction reads that SGPR as its base; hipcc pads its own instructions, never the inside of an asm statement
    asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2 nt" : "=v"(d) : "v"(voff), "s"(sb) : "memory");
}
// wave-uniform base + 32-bit lane offset for the small loads too (GM: three requests per step; 64-bit per-lane addresses for each of
// them, computed ahead by the scheduler, were what spilled the long units)
__device__ __forceinline__ const void* rg_uniform(const void* p)
{
    const uint64_t b = (uint64_t) p;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t) b), hi = __builtin_amdgcn_readfirstlane((uint32_t) (b >> 32));
    return (const void*) (((uint64_t) hi << 32) | lo);
}
__device__ __forceinline__ void rg_ld4s(uint32_t& d, uint32_t voff, const void* sbase) { asm volatile("s_nop 4\n\tglobal_load_dword %0, %1, %2" : "=v"(d) : "v"(voff), "s"(rg_uniform(sbase)) : "memory"); }
__device__ __forceinline__ void rg_ld2s(uint32_t& d, uint32_t voff, const void* sbase) { asm volatile("s_nop 4\n\tglobal_load_ushort %0, %1, %2" : "=v"(d) : "v"(voff), "s"(rg_uniform(sbase)) : "memory"); }
__device__ __forceinline__ void rg_ld16(u32x4& d, const void* p) { asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(d) : "v"(p) : "memory"); }
// (8 bytes travel as one uint64_t: hipcc 7.2 reads element 0 for BOTH elements of a 2 x 32-bit ext_vector that comes out of an asm)
__device__ __forceinline__ void rg_ld8(uint64_t& d, const void* p) { asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(d) : "v"(p) : "memory"); }
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
__device__ __forceinline__ void rg_tie(u32x4& a) { asm volatile("; tie %0" : "+v"(a) :: "memory"); }
__device__ __forceinline__ void rg_tie(uint64_t& a) { asm volatile("; tie %0" : "+v"(a) :: "memory"); }
__device__ __forceinline__ void rg_tie(uint32_t& a) { asm volatile("; tie %0" : "+v"(a) :: "memory"); }
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

// ---- the issue order of a wave, replayed at compile time ------------------------------------------------------------------------
// Unit = UL steps (row-blocks rb_lo + l of one tile); ring slot of step l = l % U.  After consuming step l a wave requests, into the
// same slot, step l + U of the same unit or -- from the last use of a slot on -- step l % U of the NEXT unit (none in a block's
// last unit).  The first U steps of a unit are therefore requested in the order of the previous unit's last U steps (the
// prologue uses the same order for the first unit).  A request for a step that opens a chunk of 4 row-blocks (step % 4 == 0) is
// preceded by the chunk's entry loads: 2, plus the residual load with chunk 0 when the epilogue adds the residual (EL0).
// GM (group sizes 32 / 64: every lane fetches the scale / zero pair of its own k-group with every piece): 2 per step instead.
constexpr int ring_entry_loads(int step, int el0, bool gm) { return gm ? 2 + (step == 0 ? el0 - 2 : 0) : step % 4 == 0 ? (step == 0 ? el0 : 2) : 0; }
// One raw entry set: the words of a chunk must have been combined before the next chunk's words are requested.  Inside a unit that
// holds for every U <= 4 (chunk c + 1 is requested at step 4 (c + 1) - U >= 4 c); across units the next unit's chunk 0 goes out at
// the last use of slot 0, which must not come before the last chunk of this unit is combined.
// (GM keeps one raw pair per ring slot: every U <= 4 is valid)
constexpr bool ring_valid(int U, int UL, bool gm = false) { return U >= 1 && U <= 4 && UL >= U && (gm || ((UL - 1) / U) * U >= ((UL - 1) / 4) * 4); }
// vector-memory instructions the first `n` ring requests of the prologue make up (entries included)
constexpr int ring_prologue_ops(int U, int UL, int el0, int n, bool gm)
{
    int ops = 0;
    for (int j = 0; j < n; ++j) ops += ring_entry_loads((UL - U + j) % U, el0, gm) + 1;
    return ops;
}
// vector-memory instructions issued after the load of step `li` and before step `li` is consumed = the N of its `s_waitcnt vmcnt(N)`
constexpr int ring_younger(int U, int UL, int el0, bool last, int li, bool gm)
{
    int issued = 0;                 // instructions issued so far
    int pos = -1;                   // issue index of the load of (this unit, li)
    for (int j = 0; j < U; ++j) {   // the previous unit's tail (or the prologue): this unit's steps (UL - U + j) % U
        const int t = (UL - U + j) % U;
        issued += ring_entry_loads(t, el0, gm);
        if (t == li) pos = issued;
        issued += 1;
    }
    for (int s = 0; s < UL; ++s) {  // this unit's own steps, up to the consumption of li
        if (s == li) return issued - (pos + 1);
        const int T = s + U;
        if (T < UL) {
            issued += ring_entry_loads(T, el0, gm);
            if (T == li) pos = issued;
            issued += 1;
        } else if (!last) {
            issued += ring_entry_loads(s % U, el0, gm) + 1;      // next unit's step s % U
        }
    }
    return 0;
}
