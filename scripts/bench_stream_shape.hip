// How much per-byte ALU work does the decode GEMV's weight stream tolerate before it stops being HBM-bound?
//   hipcc -O3 --offload-arch=gfx950 scripts/bench_stream_shape.hip -o build/bench_stream_shape
// A skeleton with the real kernel's STRUCTURE -- 512 blocks x 8 waves (2 blocks per CU), a dependent 8 KB activation vector
// staged in LDS first, then 64 KB tiles: each wave streams its 8 KB slice of a tile as 1 KiB wave-loads (16 bytes per lane,
// nt), U in flight while U are consumed, one block barrier + 16-lane store per tile -- and a dial for the work per 16-byte
// piece: V packed-fp16 VALU instructions per dword (the dequantisation is 9 per dword + ~5 per piece) and optionally the 4
// MFMAs per piece.  Chains of 80 dependent kernels walk through a 4 GiB buffer (no cache reuse), captured in a hipGraph.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s (line %d)\n", #x, hipGetErrorString(e), __LINE__); exit(1); } } while (0)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int V, bool MFMA, int U, bool SYNC, bool ILV = false, bool RBMAJOR = false>
__global__ __launch_bounds__(512) void tile_stream(const u32x4* __restrict__ w, int ntiles, int rbw, float* vec, _Float16* out)
{
    __shared__ u32x4 xs[1024];                       // activation image (16 KB)
    __shared__ float red[2][8][16];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // dependent vector -> LDS (what the RMSNorm prologue does, without the arithmetic)
    const float4 xv = *(const float4*) (vec + tid * 4);
    xs[tid] = __builtin_bit_cast(u32x4, xv);
    xs[tid + 512] = __builtin_bit_cast(u32x4, xv);
    __syncthreads();
    const f16x2 k1 = {(_Float16) 0.0625f, (_Float16) 0.0625f};
    int par = 0;
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x, par ^= 1) {
        // ILV: the 8 waves of a block read 8 CONSECUTIVE KiB per step (wave w takes row-blocks w, w + 8, ...) instead of 8
        // separate 1 KiB pieces 8 KiB apart (wave w takes the contiguous slice [w * rbw, (w + 1) * rbw))
        // RBMAJOR: the pieces of one row-block of ALL tiles are contiguous in memory ([rb][tile] instead of [tile][rb]); with the
        // interleaved wave assignment one step of the whole grid then reads one contiguous 4 MB window, like a grid-strided sweep
        const u32x4* base = RBMAJOR ? w + ((size_t) wave * ntiles + t) * 64 + lane
                          : ILV ? w + ((size_t) t * 8 * rbw + wave) * 64 + lane : w + ((size_t) t * 8 + wave) * (size_t) rbw * 64 + lane;
        const size_t RS = RBMAJOR ? (size_t) 8 * ntiles * 64 : ILV ? 8 * 64 : 64;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        u32x4 buf[2][U];
#pragma unroll
        for (int i = 0; i < U; ++i) buf[0][i] = __builtin_nontemporal_load(base + i * RS);
        const int npass = rbw / U;
#pragma unroll 2
        for (int p = 0; p < npass; ++p) {
            if (p + 1 < npass) {
#pragma unroll
                for (int i = 0; i < U; ++i) buf[(p + 1) & 1][i] = __builtin_nontemporal_load(base + ((p + 1) * U + i) * RS);
            }
#pragma unroll
            for (int i = 0; i < U; ++i) {
                const u32x4 d = buf[p & 1][i];
                const int rb = p * U + i;
                f16x8 b8;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    f16x2 a = __builtin_bit_cast(f16x2, d[j]);
                    f16x2 b = __builtin_bit_cast(f16x2, d[j] >> 8);
#pragma unroll
                    for (int v = 0; v < V / 2; ++v) { a = a * k1 + b; b = b * k1 + a; }      // V packed VALU per dword
                    b8[2 * j] = a[0] + b[0]; b8[2 * j + 1] = a[1] + b[1];
                }
                if constexpr (MFMA) {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, xs[(rb * 16 + (lane >> 4) * 4 + j) & 1023]), b8, acc, 0, 0, 0);
                } else {
                    acc[0] += (float) b8[0] + (float) b8[3] + (float) b8[5] + (float) b8[6];
                }
            }
        }
        if (lane < 16) red[par][wave][lane] = acc[0];
        if (SYNC) __syncthreads();
        if (tid < 16) {
            float v = 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) v += red[par][k][tid];
            out[t * 16 + tid] = (_Float16) v;
        }
    }
    if (blockIdx.x == 0) vec[tid * 4] = 1e-30f * (float) tid;
}

template <int V, bool MFMA, int U, bool SYNC, bool ILV = false, bool RBMAJOR = false>
static void run(const char* name, const u32x4* w, float* vec, _Float16* out, int grid)
{
    const int ntiles = 688, rbw = 8;                           // 688 tiles x 64 KB = 45.1 MB per kernel (the 7B gate_up launch)
    const size_t pieces = (size_t) ntiles * 8 * rbw * 64;
    hipStream_t s; CK(hipStreamCreate(&s));
    hipGraph_t g; hipGraphExec_t ge;
    const int n = 80;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    for (int i = 0; i < n; ++i)
        hipLaunchKernelGGL((tile_stream<V, MFMA, U, SYNC, ILV, RBMAJOR>), dim3(grid), dim3(512), 0, s, w + (size_t) i * (pieces + 4096), ntiles, rbw, vec, out);
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
    CK(hipEventRecord(e0, s));
    for (int r = 0; r < 10; ++r) CK(hipGraphLaunch(ge, s));
    CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const float t = ms * 1000.f / (10 * n);
    printf("%-46s grid %4d: %6.2f us per kernel = %.2f TB/s\n", name, grid, t, pieces * 16 / t * 1e-6);
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g)); CK(hipStreamDestroy(s));
}

int main()
{
    float* vec; _Float16* out; u32x4* w;
    const size_t wbytes = (size_t) 1 << 32;
    CK(hipMalloc(&vec, 8192)); CK(hipMalloc(&out, 1 << 20)); CK(hipMalloc(&w, wbytes + (1 << 20)));
    CK(hipMemset(vec, 0, 8192)); CK(hipMemset(w, 0x11, wbytes));
    printf("tile-structured stream, 45.1 MB per kernel, dependent chain in a hipGraph (us include the launch boundary)\n");
    run<0, false, 4, true>("loads only, U=4, barrier per tile", w, vec, out, 512);
    run<0, false, 4, false>("loads only, U=4, no barrier", w, vec, out, 512);
    run<0, false, 4, true, true>("loads only, U=4, waves interleaved in K", w, vec, out, 512);
    run<0, false, 4, true, true, true>("loads only, U=4, row-block-major layout", w, vec, out, 512);
    run<0, false, 4, true, true, true>("loads only, U=4, row-block-major, 688 blocks", w, vec, out, 688);
    run<8, true, 4, true, true, true>("4 MFMA + 8 pk VALU, row-block-major layout", w, vec, out, 512);
    run<0, false, 2, true>("loads only, U=2", w, vec, out, 512);
    run<0, false, 2, true, true>("loads only, U=2, waves interleaved in K", w, vec, out, 512);
    run<0, false, 4, true>("loads only, U=4, 1 block per CU", w, vec, out, 256);
    run<0, false, 4, true>("loads only, U=4, one tile per block", w, vec, out, 688);
    run<0, true, 4, true>("4 MFMA per piece, no VALU", w, vec, out, 512);
    run<4, true, 4, true>("4 MFMA + 4 pk VALU per dword", w, vec, out, 512);
    run<8, true, 4, true>("4 MFMA + 8 pk VALU per dword (~ the dequant)", w, vec, out, 512);
    run<12, true, 4, true>("4 MFMA + 12 pk VALU per dword", w, vec, out, 512);
    run<16, true, 4, true>("4 MFMA + 16 pk VALU per dword", w, vec, out, 512);
    run<8, false, 4, true>("8 pk VALU per dword, no MFMA", w, vec, out, 512);
    run<8, true, 4, true, true>("4 MFMA + 8 pk VALU, waves interleaved in K", w, vec, out, 512);
    return 0;
}
