// Whole-sequence lm_head of the prompt pass: out[M, N] fp32 = float(half(x[M, K] @ w[N, K]^T)) -- the reference's fp16 nn.Linear
// followed by .float() (/root/reference/model.py:1077-1078; the `-ppl` evaluation and validation read every row's logits,
// perplexity.py:121-138).  Both operands are "k contiguous" (activations [row][k], head weights [vocab][k]), i.e. exactly the
// [rows][64 k] tile layout the MFMA fragment reads want, so BOTH tiles travel global -> LDS by LDS-DMA (global_load_lds_dwordx4, no
// VGPR round trip), swizzled through the per-lane source address like the activation tile of q4_gemm.hip:
//   * 128 x 128 x 64 block tile, 4 waves (2 x 2), 64 x 64 per wave = 4 x 4 v_mfma_f32_16x16x32_f16 tiles, fp32 accumulate;
//   * 3-slot LDS ring for A and B (96 KiB); per K step a wave issues its 8 one-KiB DMA pieces of tile t + 2 and waits with a
//     hand-counted s_waitcnt vmcnt(8): tile t + 1 may still be in flight while tile t feeds the MFMAs; one barrier per K step;
//   * MFMA roles swapped (A operand = head rows) so a lane owns 4 consecutive vocabulary columns of one activation row: 16-byte
//     fp32 stores; the result is rounded to fp16 first (the reference's fp16 Linear), then widened.
// K % 64 == 0, K >= 128; any M, N (rows beyond the edge re-read the last valid row and are not stored).
#include "common.h"

#define HN_BM 128
#define HN_BN 128
#define HN_BK 64
#define HN_TILE_BYTES (128 * 64 * 2)

__device__ __forceinline__ int hn_off(int r, int c) { return r * 128 + ((c ^ (r & 7)) << 4); }   // byte offset of 16-byte chunk c of row r

__global__ __launch_bounds__(256) void half_gemm_nt_kernel(const f16* __restrict__ x, const f16* __restrict__ w, float* __restrict__ out,
                                                           int M, int K, int N, int mtiles, int ntiles)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];       // [3][A tile][B tile]
    // XCD-aware order: the blocks of one XCD walk the row tiles of the same vocabulary tile (the head rows are read once per XCD L2)
    const int b = blockIdx.x;
    const int xcd = b & 7, idx = b >> 3;
    const int nl = idx / mtiles, mt = idx - nl * mtiles;
    const int nt = nl * 8 + xcd;
    if (nt >= ntiles) return;
    const int m0 = mt * HN_BM, n0 = nt * HN_BN;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int nk = K / HN_BK;

    // DMA pieces: 16 per tile (8 rows x 128 bytes each); wave w stages pieces 4 w .. 4 w + 3 of A and of B
    uint32_t a_off[4], b_off[4];                                       // element offsets (M * K, N * K < 2^31 checked on the host)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = (wave * 4 + i) * 8 + (lane >> 3), slot = lane & 7;
        a_off[i] = (uint32_t) min(m0 + row, M - 1) * (uint32_t) K + ((slot ^ (row & 7)) << 3);
        b_off[i] = (uint32_t) min(n0 + row, N - 1) * (uint32_t) K + ((slot ^ (row & 7)) << 3);
    }
    auto stage = [&](int slot3, int k0) {
        unsigned char* base = lds + (size_t) slot3 * 2 * HN_TILE_BYTES + (wave * 4) * 1024;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*) (x + (size_t) (a_off[i] + (uint32_t) k0)),
                                             (__attribute__((address_space(3))) unsigned char*) (base + i * 1024), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*) (w + (size_t) (b_off[i] + (uint32_t) k0)),
                                             (__attribute__((address_space(3))) unsigned char*) (base + HN_TILE_BYTES + i * 1024), 16, 0, 0);
    };

    f32x4 acc[4][4];                                                   // [n-tile][m-tile]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int fr = lane & 15, fk = lane >> 4;
    const int fx0 = hn_off(wm * 64 + fr, fk), fw0 = hn_off(wn * 64 + fr, fk);

    stage(0, 0);
    stage(1, HN_BK);                                                   // nk >= 2
    int slot = 0;
    for (int t = 0; t < nk; ++t) {
        // tile t has landed when at most the 8 pieces of tile t + 1 are still in flight (the last step has nothing behind it)
        if (t + 1 < nk) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                                  // everyone's pieces of tile t are in LDS; everyone is done reading tile t - 1
        if (t + 2 < nk) stage(slot == 0 ? 2 : slot - 1, (t + 2) * HN_BK);   // = (slot + 2) % 3: the slot tile t - 1 was read from
        const unsigned char* at = lds + (size_t) slot * 2 * HN_TILE_BYTES;
        const unsigned char* bt = at + HN_TILE_BYTES;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            f16x8 fx[4], fw[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) fx[i] = *(const f16x8*) (at + (fx0 ^ (kk << 6)) + i * 2048);
#pragma unroll
            for (int i = 0; i < 4; ++i) fw[i] = *(const f16x8*) (bt + (fw0 ^ (kk << 6)) + i * 2048);
#pragma unroll
            for (int in = 0; in < 4; ++in)
#pragma unroll
                for (int im = 0; im < 4; ++im)
                    acc[in][im] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fw[in], fx[im], acc[in][im], 0, 0, 0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");             // this wave's fragment reads of tile t are done before the slot is recycled
        slot = slot == 2 ? 0 : slot + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                   // (nothing is in flight here; states it for scripts/isa_lint.py, whose dataflow merges the two waits above)
    // acc[in][im][j]: activation row m0 + wm * 64 + im * 16 + fr, vocabulary column n0 + wn * 64 + in * 16 + fk * 4 + j
#pragma unroll
    for (int im = 0; im < 4; ++im) {
        const int row = m0 + wm * 64 + im * 16 + fr;
        if (row >= M) continue;
#pragma unroll
        for (int in = 0; in < 4; ++in) {
            const int n = n0 + wn * 64 + in * 16 + fk * 4;
            float* op = out + (size_t) row * N + n;
            if (n + 4 <= N) {
                *(f32x4*) op = (f32x4){(float) (f16) acc[in][im][0], (float) (f16) acc[in][im][1], (float) (f16) acc[in][im][2], (float) (f16) acc[in][im][3]};
            } else {
                for (int j = 0; j < 4 && n + j < N; ++j) op[j] = (float) (f16) acc[in][im][j];
            }
        }
    }
}

// 1 = not covered (the caller keeps its own GEMM): K must be a multiple of 64, >= 128, 32-bit element offsets, 16-byte aligned rows.
int launch_head_gemm(const f16* x, const f16* w, float* out, int rows, int hidden, int vocab, hipStream_t s)
{
    if (rows <= 0 || vocab <= 0) return 0;
    if (hidden % HN_BK != 0 || hidden < 2 * HN_BK || vocab % 4 != 0 || (uint64_t) rows * hidden >= (1ull << 31) || (uint64_t) vocab * hidden >= (1ull << 31) ||
        (((uintptr_t) x | (uintptr_t) w | (uintptr_t) out) & 15) != 0)
        return 1;
    const int mtiles = (rows + HN_BM - 1) / HN_BM, ntiles = (vocab + HN_BN - 1) / HN_BN;
    const int grid = 8 * ((ntiles + 7) / 8) * mtiles;
    const size_t smem = 3 * 2 * (size_t) HN_TILE_BYTES;                // 96 KiB
    static bool big[EXL_MAX_DEVICES] = {};
    EXL_TRY(exl_lds_opt_in((const void*) half_gemm_nt_kernel, big));
    hipLaunchKernelGGL(half_gemm_nt_kernel, dim3(grid), dim3(256), smem, s, x, w, out, rows, hidden, vocab, mtiles, ntiles);
    EXL_LAUNCH_CHECK();
    return 0;
}
