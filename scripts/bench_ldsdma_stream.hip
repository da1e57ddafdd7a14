// Does the decode GEMV's weight stream run faster THROUGH LDS (global_load_lds_dwordx4, no VGPR return path) than through registers?
//   hipcc -O3 --offload-arch=gfx950 scripts/bench_ldsdma_stream.hip -o build/bench_ldsdma_stream
// Round 3 found that a CU accepts ~32 KiB of outstanding vector-memory requests that return to VGPRs: with 16 waves x 3-4 loads in flight
// the chip holds ~8 MB in the air, which at ~1.7 us of loaded latency is 4.6 TB/s -- about what the ring kernels reach.
// MI355X_MICROARCH.md prices an LDS-DMA stream (one loader wave, 8 x 16 KiB ring per CU) at 6.4-6.8 TB/s chip-wide.  This skeleton keeps
// the REAL kernel's structure (bench_stream_shape.hip: 512 blocks x 8 waves, a dependent 8 KB vector staged first, 64 KB tiles, every wave
// streams its 8 KiB slice of a tile as 1 KiB wave-loads, one block barrier + 16-lane store per tile, dial for the work per piece) and
// changes ONE thing: the pieces travel global -> LDS by DMA into a per-wave ring of D 1-KiB slots that runs across tile boundaries, and
// are read back with ds_read_b128.  Register variant (U = 4, double-buffered) beside it as the baseline, same binary, same buffers.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s (line %d)\n", #x, hipGetErrorString(e), __LINE__); exit(1); } } while (0)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t lds_addr(const void* p) { return (uint32_t) (uintptr_t) (__attribute__((address_space(3))) const unsigned char*) p; }
// 1 KiB of global memory straight into LDS (lds_dst wave-uniform; lane l lands at lds_dst + 16 l); nt: streamed once
__device__ __forceinline__ void dma16(uint32_t lds_dst, const void* gsrc)
{
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }

template <int V, bool MFMA>
__device__ __forceinline__ void work(const u32x4 d, int rb, int lane, const u32x4* xs, f32x4& acc)
{
    const f16x2 k1 = {(_Float16) 0.0625f, (_Float16) 0.0625f};
    f16x8 b8;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        f16x2 a = __builtin_bit_cast(f16x2, d[j]);
        f16x2 b = __builtin_bit_cast(f16x2, d[j] >> 8);
#pragma unroll
        for (int v = 0; v < V / 2; ++v) { a = a * k1 + b; b = b * k1 + a; }
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

// ---- baseline: registers, U = 4 in flight while 4 are consumed (bench_stream_shape.hip's tile_stream) ---------------------------------
template <int V, bool MFMA>
__global__ __launch_bounds__(512) void reg_stream(const u32x4* __restrict__ w, int ntiles, int rbw, float* vec, _Float16* out)
{
    constexpr int U = 4;
    __shared__ u32x4 xs[1024];
    __shared__ float red[2][8][16];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const float4 xv = *(const float4*) (vec + tid * 4);
    xs[tid] = __builtin_bit_cast(u32x4, xv);
    xs[tid + 512] = __builtin_bit_cast(u32x4, xv);
    __syncthreads();
    int par = 0;
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x, par ^= 1) {
        const u32x4* base = w + ((size_t) t * 8 + wave) * (size_t) rbw * 64 + lane;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        u32x4 buf[2][U];
#pragma unroll
        for (int i = 0; i < U; ++i) buf[0][i] = __builtin_nontemporal_load(base + i * 64);
        const int npass = rbw / U;
#pragma unroll 2
        for (int p = 0; p < npass; ++p) {
            if (p + 1 < npass) {
#pragma unroll
                for (int i = 0; i < U; ++i) buf[(p + 1) & 1][i] = __builtin_nontemporal_load(base + ((p + 1) * U + i) * 64);
            }
#pragma unroll
            for (int i = 0; i < U; ++i) work<V, MFMA>(buf[p & 1][i], p * U + i, lane, xs, acc);
        }
        if (lane < 16) red[par][wave][lane] = acc[0];
        __syncthreads();
        if (tid < 16) {
            float v = 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) v += red[par][k][tid];
            out[t * 16 + tid] = (_Float16) v;
        }
    }
    if (blockIdx.x == 0) vec[tid * 4] = 1e-30f * (float) tid;
}

// ---- LDS-DMA: per-wave ring of D 1-KiB slots, running across the block's tiles --------------------------------------------------------
// Step s of a wave = row-block s % 8 of its tile s / 8 (tiles blockIdx.x, + gridDim.x, ...).  Slot = s % D.  Steady state: D DMAs in
// flight; wait for the oldest (vmcnt(D - 1)), read the slot, re-issue it for step s + D.  The epilogue store of a tile also counts in
// vmcnt on gfx9 (it can only make a wait stricter).  The last D steps wait for everything.
template <int D, int V, bool MFMA>
__global__ __launch_bounds__(512) void dma_stream(const u32x4* __restrict__ w, int ntiles, int rbw, float* vec, _Float16* out)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    u32x4* xs = (u32x4*) smem;                                   // [1024] activation image (16 KB)
    float* red = (float*) (smem + 16384);                        // [2][8][16]
    u32x4* ring = (u32x4*) (smem + 16384 + 1024);                // [8 waves][D][64]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int my_tiles = blockIdx.x < ntiles ? (ntiles - 1 - (int) blockIdx.x) / (int) gridDim.x + 1 : 0;
    const int nsteps = my_tiles * 8;                             // (rbw == 8)
    u32x4* myring = ring + (size_t) wave * D * 64;
    const uint32_t ring_lds = lds_addr(myring);
    auto src = [&](int s) -> const u32x4* {
        const int t = blockIdx.x + (s >> 3) * gridDim.x;
        return w + ((size_t) t * 8 + wave) * (size_t) rbw * 64 + (size_t) (s & 7) * 64 + lane;
    };
    // the ring first (weights do not depend on the activation), then the dependent vector -- the real kernels' order
#pragma unroll
    for (int i = 0; i < D; ++i) if (i < nsteps) dma16(ring_lds + i * 1024, src(i));
    const float4 xv = *(const float4*) (vec + tid * 4);
    xs[tid] = __builtin_bit_cast(u32x4, xv);
    xs[tid + 512] = __builtin_bit_cast(u32x4, xv);
    __syncthreads();                                             // (drains vmcnt once: the first D slots have landed too)
    int par = 0;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < nsteps; ++s) {
        const int slot = s % D;
        // DMAs younger than step s's: min(D - 1, nsteps - 1 - s); a compile-time count, so the tail waits for everything
        if (nsteps - 1 - s >= D - 1) wait_vm<D - 1>(); else wait_vm<0>();
        const u32x4 d = myring[slot * 64 + lane];                // ds_read_b128
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // the read has left LDS before the slot is refilled
        if (s + D < nsteps) dma16(ring_lds + slot * 1024, src(s + D));
        work<V, MFMA>(d, s & 7, lane, xs, acc);
        if ((s & 7) == 7) {                                      // tile done: reduce + store, as the real kernels do
            const int t = blockIdx.x + (s >> 3) * gridDim.x;
            if (lane < 16) red[(par * 8 + wave) * 16 + lane] = acc[0];
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // LDS-only barrier: does not drain the DMA queue
            if (tid < 16) {
                float v = 0.f;
#pragma unroll
                for (int k = 0; k < 8; ++k) v += red[(par * 8 + k) * 16 + tid];
                out[t * 16 + tid] = (_Float16) v;
            }
            par ^= 1;
            acc = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    }
    if (blockIdx.x == 0) vec[tid * 4] = 1e-30f * (float) tid;
}

// ---- LDS-DMA with ONE LOADER WAVE per block (wave 8 of 9): the eight consumer waves issue no vector-memory instruction at all ----------
// (in the prompt GEMM that split was decisive: a wave stalled on a VMEM issue cannot issue the MFMAs behind it).  Ring of S steps x 8 KiB
// per block (step g = row-block g % 8 of the block's tile g / 8: one 1-KiB piece per consumer wave).  One block barrier per STEP orders
// both directions: the loader arrives once group g has landed, the consumers once they have finished reading step g - 1, whose slot the
// loader then refills with group g + S - 1.  No polling, nothing that can hang.
template <int S, int V, bool MFMA>
__global__ __launch_bounds__(576) void loader_stream(const u32x4* __restrict__ w, int ntiles, int rbw, float* vec, _Float16* out)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    u32x4* xs = (u32x4*) smem;                                   // [1024]
    float* red = (float*) (smem + 16384);                        // [2][8][16]
    u32x4* ring = (u32x4*) (smem + 16384 + 1024);                // [S][8 waves][64]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int my_tiles = blockIdx.x < ntiles ? (ntiles - 1 - (int) blockIdx.x) / (int) gridDim.x + 1 : 0;
    const int nsteps = my_tiles * 8;
    const uint32_t ring_lds = lds_addr(ring);
    if (tid < 512) {
        const float4 xv = *(const float4*) (vec + tid * 4);
        xs[tid] = __builtin_bit_cast(u32x4, xv);
        xs[tid + 512] = __builtin_bit_cast(u32x4, xv);
    }
    if (wave == 8) {
        // ---------------- loader ----------------
        auto issue = [&](int g) {
            const int t = blockIdx.x + (g >> 3) * gridDim.x;
            const u32x4* src = w + (size_t) t * 8 * (size_t) rbw * 64 + (size_t) (g & 7) * 64 + lane;
            const uint32_t dst = ring_lds + (uint32_t) (g % S) * 8192u;
#pragma unroll
            for (int c = 0; c < 8; ++c) dma16(dst + c * 1024, src + (size_t) c * rbw * 64);
        };
#pragma unroll
        for (int g = 0; g < S - 1; ++g) if (g < nsteps) issue(g);
        __syncthreads();                                         // (the image barrier; the consumers' xs stores)
        for (int g = 0; g < nsteps; ++g) {
            // groups younger than g in flight: min(S - 2, nsteps - 1 - g)
            constexpr int YOUNGER = 8 * (S - 2) > 63 ? 63 : 8 * (S - 2);     // (vmcnt is a 6-bit counter: deeper rings wait a little early)
            if (nsteps - 1 - g >= S - 2) wait_vm<YOUNGER>(); else wait_vm<0>();
            asm volatile("s_barrier" ::: "memory");              // step g: data landed / step g - 1 read
            if (g + S - 1 < nsteps) issue(g + S - 1);
        }
        asm volatile("s_barrier" ::: "memory");                  // (the consumers' last reduction barrier)
        return;
    }
    // ---------------- consumers ----------------
    __syncthreads();
    int par = 0;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int g = 0; g < nsteps; ++g) {
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if ((g & 7) == 0 && g > 0 && tid < 16) {                 // the tile finished at step g - 1: its partial sums are visible now
            const int t = blockIdx.x + ((g - 1) >> 3) * gridDim.x;
            float v = 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) v += red[((par ^ 1) * 8 + k) * 16 + tid];
            out[t * 16 + tid] = (_Float16) v;
        }
        const u32x4 d = ring[((size_t) (g % S) * 8 + wave) * 64 + lane];
        work<V, MFMA>(d, g & 7, lane, xs, acc);
        if ((g & 7) == 7) {
            if (lane < 16) red[(par * 8 + wave) * 16 + lane] = acc[0];
            par ^= 1;
            acc = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (nsteps > 0 && tid < 16) {
        const int t = blockIdx.x + ((nsteps - 1) >> 3) * gridDim.x;
        float v = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) v += red[((par ^ 1) * 8 + k) * 16 + tid];
        out[t * 16 + tid] = (_Float16) v;
    }
    if (blockIdx.x == 0 && tid < 512) vec[tid * 4] = 1e-30f * (float) tid;
}

template <typename K>
static void run(const char* name, K kernel, size_t smem, const u32x4* w, float* vec, _Float16* out, int grid, int threads = 512)
{
    const int ntiles = 688, rbw = 8;                           // 688 tiles x 64 KB = 45.1 MB per kernel (the 7B gate/up launch)
    const size_t pieces = (size_t) ntiles * 8 * rbw * 64;
    if (smem > 64 * 1024) CK(hipFuncSetAttribute((const void*) kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int) smem));
    hipStream_t s; CK(hipStreamCreate(&s));
    hipGraph_t g; hipGraphExec_t ge;
    const int n = 80;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    for (int i = 0; i < n; ++i)
        hipLaunchKernelGGL(kernel, dim3(grid), dim3(threads), smem, s, w + (size_t) i * (pieces + 4096), ntiles, rbw, vec, out);
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
    CK(hipEventRecord(e0, s));
    for (int r = 0; r < 10; ++r) CK(hipGraphLaunch(ge, s));
    CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const float t = ms * 1000.f / (10 * n);
    printf("%-58s grid %4d: %6.2f us per kernel = %.2f TB/s\n", name, grid, t, pieces * 16 / t * 1e-6);
    fflush(stdout);
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g)); CK(hipStreamDestroy(s));
}

// correctness of the ring bookkeeping: every piece consumed exactly once (sum of a known pattern)
template <int D>
static void check(u32x4* w, float* vec, _Float16* out)
{
    const int ntiles = 688, rbw = 8;
    const size_t smem = 16384 + 1024 + (size_t) 8 * D * 1024;
    auto k = dma_stream<D, 0, false>;
    if (smem > 64 * 1024) CK(hipFuncSetAttribute((const void*) k, hipFuncAttributeMaxDynamicSharedMemorySize, (int) smem));
    hipLaunchKernelGGL(k, dim3(512), dim3(512), smem, 0, w, ntiles, rbw, vec, out);
    hipLaunchKernelGGL((reg_stream<0, false>), dim3(512), dim3(512), 0, 0, w, ntiles, rbw, vec, out + 688 * 16);
    CK(hipDeviceSynchronize());
    static _Float16 h[2 * 688 * 16];
    CK(hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost));
    int bad = 0;
    for (int i = 0; i < 688 * 16; ++i) if ((float) h[i] != (float) h[688 * 16 + i]) ++bad;
    printf("ring depth %2d: results differing from the register stream: %d of %d\n", D, bad, 688 * 16);
}

__global__ void fill(uint32_t* p, size_t n)
{
    for (size_t i = blockIdx.x * (size_t) blockDim.x + threadIdx.x; i < n; i += (size_t) gridDim.x * blockDim.x) {
        // small fp16 pairs (0 .. 3.75 in steps of 0.25) so that sums are exact and order-independent
        const uint32_t a = (uint32_t) (i * 2654435761u) >> 28, b = (uint32_t) (i * 40503u + 7u) & 15u;
        const _Float16 x = (_Float16) (0.25f * (float) a), y = (_Float16) (0.25f * (float) b);
        p[i] = (uint32_t) __builtin_bit_cast(uint16_t, x) | ((uint32_t) __builtin_bit_cast(uint16_t, y) << 16);
    }
}

int main()
{
    float* vec; _Float16* out; u32x4* w;
    const size_t wbytes = (size_t) 1 << 32;
    CK(hipMalloc(&vec, 8192)); CK(hipMalloc(&out, 1 << 20)); CK(hipMalloc(&w, wbytes + (1 << 20)));
    CK(hipMemset(vec, 0, 8192));
    fill<<<4096, 256>>>((uint32_t*) w, wbytes / 4);
    CK(hipDeviceSynchronize());
    check<4>(w, vec, out); check<8>(w, vec, out); check<6>(w, vec, out);
    {
        const size_t smem = 16384 + 1024 + (size_t) 4 * 8192;
        hipLaunchKernelGGL((loader_stream<4, 0, false>), dim3(512), dim3(576), smem, 0, w, 688, 8, vec, out);
        hipLaunchKernelGGL((reg_stream<0, false>), dim3(512), dim3(512), 0, 0, w, 688, 8, vec, out + 688 * 16);
        CK(hipDeviceSynchronize());
        static _Float16 h[2 * 688 * 16];
        CK(hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost));
        int bad = 0;
        for (int i = 0; i < 688 * 16; ++i) if ((float) h[i] != (float) h[688 * 16 + i]) ++bad;
        printf("loader wave, ring of 4 steps: results differing from the register stream: %d of %d\n", bad, 688 * 16);
    }
    printf("tile-structured stream, 45.1 MB per kernel, dependent chain in a hipGraph (us include the launch boundary)\n");
#define SM(D) (16384 + 1024 + (size_t) 8 * (D) * 1024)
    run("registers, loads only, U=4", reg_stream<0, false>, 0, w, vec, out, 512);
    run("LDS-DMA,   loads only, ring 2 KiB/wave", dma_stream<2, 0, false>, SM(2), w, vec, out, 512);
    run("LDS-DMA,   loads only, ring 4 KiB/wave", dma_stream<4, 0, false>, SM(4), w, vec, out, 512);
    run("LDS-DMA,   loads only, ring 6 KiB/wave (128 KiB/CU)", dma_stream<6, 0, false>, SM(6), w, vec, out, 512);
    run("LDS-DMA,   loads only, ring 8 KiB/wave, 1 block per CU", dma_stream<8, 0, false>, SM(8), w, vec, out, 256);
    run("LDS-DMA,   loads only, ring 16 KiB/wave, 1 block per CU", dma_stream<16, 0, false>, SM(16), w, vec, out, 256);
    run("registers, 4 MFMA + 8 pk VALU per dword, U=4", reg_stream<8, true>, 0, w, vec, out, 512);
    run("LDS-DMA,   4 MFMA + 8 pk VALU, ring 4 KiB/wave", dma_stream<4, 8, true>, SM(4), w, vec, out, 512);
    run("LDS-DMA,   4 MFMA + 8 pk VALU, ring 6 KiB/wave", dma_stream<6, 8, true>, SM(6), w, vec, out, 512);
    run("LDS-DMA,   4 MFMA + 8 pk VALU, ring 8 KiB/wave, 1 block/CU", dma_stream<8, 8, true>, SM(8), w, vec, out, 256);
    run("LDS-DMA,   4 MFMA + 8 pk VALU, ring 16 KiB/wave, 1 block/CU", dma_stream<16, 8, true>, SM(16), w, vec, out, 256);
    run("registers, 4 MFMA + 8 pk VALU, U=4 (again)", reg_stream<8, true>, 0, w, vec, out, 512);
#define SL(S) (16384 + 1024 + (size_t) (S) * 8192)
    run("loader wave, loads only, ring 3 steps (24 KiB/block)", loader_stream<3, 0, false>, SL(3), w, vec, out, 512, 576);
    run("loader wave, loads only, ring 4 steps (32 KiB/block)", loader_stream<4, 0, false>, SL(4), w, vec, out, 512, 576);
    run("loader wave, loads only, ring 6 steps (48 KiB/block)", loader_stream<6, 0, false>, SL(6), w, vec, out, 512, 576);
    run("loader wave, loads only, ring 8 steps, 1 block per CU", loader_stream<8, 0, false>, SL(8), w, vec, out, 256, 576);
    run("loader wave, loads only, ring 16 steps, 1 block per CU", loader_stream<16, 0, false>, SL(16), w, vec, out, 256, 576);
    run("loader wave, 4 MFMA + 8 pk VALU, ring 3 steps", loader_stream<3, 8, true>, SL(3), w, vec, out, 512, 576);
    run("loader wave, 4 MFMA + 8 pk VALU, ring 4 steps", loader_stream<4, 8, true>, SL(4), w, vec, out, 512, 576);
    run("loader wave, 4 MFMA + 8 pk VALU, ring 6 steps", loader_stream<6, 8, true>, SL(6), w, vec, out, 512, 576);
    run("loader wave, 4 MFMA + 8 pk VALU, ring 8 steps, 1 block/CU", loader_stream<8, 8, true>, SL(8), w, vec, out, 256, 576);
    run("loader wave, 4 MFMA + 8 pk VALU, ring 16 steps, 1 block/CU", loader_stream<16, 8, true>, SL(16), w, vec, out, 256, 576);
    run("LDS-DMA,   4 MFMA + 8 pk VALU, ring 4 KiB/wave (again)", dma_stream<4, 8, true>, SM(4), w, vec, out, 512);
    return 0;
}
