// Declarations shared by the two translation units of the native decode executor (decode_fused.hip: host side, attention,
// head and the compiler-scheduled weight stream; decode_ring.hip: the hand-counted rolling-ring weight stream).
#pragma once
#include "gemv_t16.h"

#define DEC_MAX_MATS 3
#define DEC_MAX_NSPLIT 16
#define DEC_ATT_MAX_KEYS 1024
#define DEC_WAVES 8
#define DEC_THREADS (DEC_WAVES * 64)

T16Matrix t16_view(const Q4Matrix* m);          // q4_gemv.hip

struct DecGemvArgs {
    // ---- prologue: the activation vector of length K staged in LDS ----
    const f16* vec;               // PNORM 0: plain fp16 [K];  PNORM 1: residual stream [h] (or the embedding table when tok)
    const int64_t* tok;           // PNORM 1, layer 0: token id (device)
    const f16* norm_w;            // PNORM 1
    float eps;
    f16* hid_copy;                // PNORM 1 with tok: block 0 stores the embedding row here (start of the residual stream)
    const float* att_ml;          // PNORM 3: (max, sum) of every (head, split) of the attention kernel; vec = their fp16 outputs
    int att_nsplit;               // PNORM 3
    // ---- matrices; 16-column tiles are numbered across them in order ----
    int nmat;
    T16Matrix mat[DEC_MAX_MATS];
    int tile_end[DEC_MAX_MATS];   // cumulative tile counts (EMODE 2: tiles of mat[0]; mat[1] is walked in lock-step)
    // ---- epilogue ----
    f16* out[DEC_MAX_MATS];       // EMODE 0: out[mi][n] = h(y);  EMODE 2: out[0][n] = silu(h(y_gate)) * h(y_up)
    f16* hid_io;                  // EMODE 1: hid_io[n] = h(res_in[n] + y)
    const f16* res_in;            // EMODE 1: where the residual is read (= hid_io, or a zero vector on the tensor-parallel ranks that do not own it)
    int rb_per_wave;
    int xs_images;                // 1, or 2 when gate and up carry different act-order maps (EMODE 2)
    int ablate;                   // measurement only (EXL_DEC_ABLATE): 1 = skip the dequant + MFMA work, 3 = also the scale / zero loads, 4 = also the activation loads
    int nblocks;                  // = gridDim.x (passed explicitly: the implicit-argument load is one more scalar round trip)
    int units_lo, units_rem;      // unit count per block: units_lo + (block < units_rem)
    int ring_flags;               // decode_ring.hip: bit 0 = barrier between the activation loads and the first weight loads of a block
    // act-order (reference: column_remap.cu:7-36 gathers x through x_map before every matmul):
    const uint16_t* map16[DEC_MAX_MATS];   // gather maps of the matrices of this launch as 16-bit indices (K < 65536), or NULL:
                                           // NULL with an act-order matrix means its input arrives ALREADY gathered (out_perm below)
    const uint16_t* out_perm;     // EMODE 0 (mat 0 only) / EMODE 2: the consumer's inverse gather map -- column n is stored at out_perm[n],
                                  // so that the next kernel reads its activation linearly and gathers nothing
};

// Every field of a matrix view the streaming loop touches, forced into SGPRs at the top of the kernel: the compiler
// otherwise indexes the kernarg segment dynamically (a.mat[mi] with a computed mi), i.e. issues a scalar load, waits,
// computes, issues the next -- six dependent round trips to a cold scalar cache (~1.4 us) before the first weight load.
#define DEC_PIN_S(x) asm volatile("" : "+s"(x))
// pointers: pinned as integers and rebuilt as GLOBAL pointers (an opaque generic pointer would turn every access into a
// flat_load, which counts against both the vector-memory and the LDS wait counters)
template <typename T>
__device__ __forceinline__ T* dec_pin_ptr(T* p)
{
    uint64_t v = (uint64_t) p;
    asm volatile("" : "+s"(v));
    return (T*) (T __attribute__((address_space(1)))*) v;
}
__device__ __forceinline__ void dec_pin(T16Matrix& m)
{
    m.qw = dec_pin_ptr(m.qw); m.qzeros = dec_pin_ptr(m.qzeros); m.scales = dec_pin_ptr(m.scales); m.x_map = dec_pin_ptr(m.x_map);
    DEC_PIN_S(m.N); DEC_PIN_S(m.RB); DEC_PIN_S(m.gprows); DEC_PIN_S(m.gshift);
}
__device__ __forceinline__ T16Matrix dec_pick(const T16Matrix& m0, const T16Matrix& m1, const T16Matrix& m2, int mi)
{
    T16Matrix m = m0;                                                // mi is wave-uniform: scalar selects
    if (mi == 1) m = m1;
    if (mi == 2) m = m2;
    return m;
}

__device__ __forceinline__ f16 silu_mul_f16(f16 x, f16 y)
{
    const f16 e = (f16) __expf((float) (f16) (-x));
    const f16 sm = (f16) 1.0f + e;
    const f16 rc = (f16) (1.0f / (float) sm);
    const f16 v = x * rc;
    return v * y;
}


// ---- cross-lane reductions on the vector ALU (DPP): no LDS-pipe round trips (a ds_bpermute butterfly is ~150 cycles per step
// when the CU is streaming).  Both stream kernels use the same ORDER of additions, so their results stay bit-identical:
// pairs, quads (quad_perm), 8 and 16 lanes (row_half_mirror / row_mirror: after each step the lanes of a group hold the same
// value, so the mirrored partner carries the other group's sum), then the four 16-lane rows as (r0 + r1) + (r2 + r3).
template <int CTRL>
__device__ __forceinline__ float dec_dpp(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float dec_row_sum(float v)                // sum over the 16 lanes of a row, in every lane
{
    v += dec_dpp<0xB1>(v); v += dec_dpp<0x4E>(v); v += dec_dpp<0x141>(v); v += dec_dpp<0x140>(v);
    return v;
}
__device__ __forceinline__ float dec_row_max(float v)
{
    v = fmaxf(v, dec_dpp<0xB1>(v)); v = fmaxf(v, dec_dpp<0x4E>(v)); v = fmaxf(v, dec_dpp<0x141>(v)); v = fmaxf(v, dec_dpp<0x140>(v));
    return v;
}
__device__ __forceinline__ float dec_lane(float v, int lane) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane)); }
// v + (the same lane of the other 16-lane row of its pair) and then + (the other half of the wave): gfx950's v_permlane16_swap /
// v_permlane32_swap on the vector ALU instead of two ds_bpermute round trips.  Association per lane l < 16: (v_l + v_l+16) + (v_l+32 + v_l+48).
__device__ __forceinline__ float dec_rows_sum(float v)               // every lane: sum over the 4 lanes with its (lane % 16)
{
    const unsigned u = __builtin_bit_cast(unsigned, v);
    const auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    const float s = __builtin_bit_cast(float, (unsigned) a[0]) + __builtin_bit_cast(float, (unsigned) a[1]);
    const unsigned w = __builtin_bit_cast(unsigned, s);
    const auto b = __builtin_amdgcn_permlane32_swap(w, w, false, false);
    return __builtin_bit_cast(float, (unsigned) b[0]) + __builtin_bit_cast(float, (unsigned) b[1]);
}
// block barrier for LDS hand-offs only: waits for this wave's LDS traffic, NOT for its outstanding global loads (__syncthreads
// drains vmcnt as well, which parks a wave until every row it has requested is back)
__device__ __forceinline__ void dec_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ float dec_wave_sum(float v)               // sum over the 64 lanes, wave-uniform
{
    v = dec_row_sum(v);
    return (dec_lane(v, 0) + dec_lane(v, 16)) + (dec_lane(v, 32) + dec_lane(v, 48));
}

// decode_ring.hip: the rolling-ring stream kernels.  depth: loads in flight per lane (2 .. 4); wide_blocks: 16-wave blocks for the
// plain-vector launches (o_proj behind the merge / one KV split, down_proj) whose grid leaves one block per CU.  Returns 1 when the launch is not
// covered (the caller falls back to dec_stream_kernel), 0 on success, an error code otherwise.  plan: exl_decoder_plan's record (or NULL).
int launch_dec_ring(int pnorm, int emode, bool g16, int K, int grid, int depth, bool wide_blocks, const DecGemvArgs& a, hipStream_t s, int* plan);
