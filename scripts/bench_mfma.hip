// MFMA calibration for the prefill GEMM: what does the loop STRUCTURE cost on this chip, arithmetic aside?
//   hipcc -O3 --offload-arch=gfx950 scripts/bench_mfma.hip -o build/bench_mfma
// mode 0: 32 independent v_mfma_f32_16x16x32_f16 per trip, nothing else          (the achievable MFMA peak at the clock the chip holds)
// mode 1: + one s_barrier per trip                                              (8 waves of a block in lock step)
// mode 2: + 16 ds_read_b128 per trip feeding the MFMAs (the fragment reads of a 64x64 wave tile)
// mode 3: mode 2 with the reads of the NEXT trip's first half issued before the barrier-free second half (software pipelined)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int MODE>
__global__ __launch_bounds__(512) void k(float* out, int trips)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 48 * 1024 / 4; i += 512) ((float*) lds)[i] = (float) (i & 7) * 0.001f;
    __syncthreads();
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0, 0, 0, 0};
    f16x8 a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { a[i] = (f16x8){(_Float16) lane, 1, 2, 3, 4, 5, 6, 7}; b[i] = (f16x8){(_Float16) i, 1, 0, 1, 0, 1, 0, 1}; }
    const unsigned char* at = lds + (wave >> 1) * 8192 + lane * 16;
    const unsigned char* bt = lds + 32768 + (wave & 1) * 8192 + lane * 16;
    const long long c0 = clock64();
    for (int t = 0; t < trips; ++t) {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            if constexpr (MODE >= 2) {
#pragma unroll
                for (int i = 0; i < 4; ++i) { a[i] = *(const f16x8*) (at + kk * 4096 + i * 1024); b[i] = *(const f16x8*) (bt + kk * 4096 + i * 1024); }
            }
#pragma unroll
            for (int in = 0; in < 4; ++in)
#pragma unroll
                for (int im = 0; im < 4; ++im) acc[in][im] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[in], a[im], acc[in][im], 0, 0, 0);
        }
        if constexpr (MODE >= 1) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[1024 * 512] = (float) (clock64() - c0);
}

template <int MODE> void run(const char* name, int blocks, float* out)
{
    const int trips = 40000;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipFuncSetAttribute((const void*) k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
    k<MODE><<<blocks, 512, 64 * 1024>>>(out, 100);
    CK(hipDeviceSynchronize());
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        k<MODE><<<blocks, 512, 64 * 1024>>>(out, trips);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double flop = (double) blocks * 8 * trips * 32 * 2.0 * 16 * 16 * 32;
        float ticks; CK(hipMemcpy(&ticks, out + 1024 * 512, 4, hipMemcpyDeviceToHost));
        printf("%-44s blocks %4d: %8.3f ms  %7.1f TFLOP/s  %.0f s_memtime ticks per trip -> %.2f GHz if a tick is a shader cycle\n", name, blocks, ms,
               flop / ms / 1e9, ticks / trips, ticks / (ms * 1e6));
    }
}
int main()
{
    float* out; CK(hipMalloc(&out, 1024 * 512 * 4 + 64));
    for (int blocks : {256, 512}) {
        run<0>("mfma only", blocks, out);
        run<1>("mfma + barrier/trip", blocks, out);
        run<2>("mfma + barrier + 16 ds_read_b128/trip", blocks, out);
    }
    return 0;
}
