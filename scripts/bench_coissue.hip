// How much VALU work can a loader wave issue beside two MFMA waves on the same SIMD?
//   hipcc -O3 --offload-arch=gfx950 scripts/bench_coissue.hip -o build/bench_coissue
// Block = 12 waves: waves 0-7 run 32 x v_mfma_f32_16x16x32_f16 (SHAPE 0) or 16 x v_mfma_f32_32x32x16_f16 (SHAPE 1) per
// trip, waves 8-11 run NV dependent-free packed-f16 VALU instructions per trip; one s_barrier per trip.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int SHAPE, int NV>
__global__ __launch_bounds__(768) void k(float* out, int trips, int seed)
{
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float res = 0.f;
    if (wave < 8) {
        f16x8 a[4], b[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { a[i][e] = (_Float16) (((lane * 7 + i * 3 + e + seed) % 13) * 0.125f - 0.7f); b[i][e] = (_Float16) (((lane * 5 + i + e * 3 + seed) % 11) * 0.25f - 1.f); }
        }
        if constexpr (SHAPE == 0) {
            f32x4 acc[4][4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0, 0, 0, 0};
            for (int t = 0; t < trips; ++t) {
#pragma unroll
                for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                    for (int in = 0; in < 4; ++in)
#pragma unroll
                        for (int im = 0; im < 4; ++im) acc[in][im] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[in], a[im], acc[in][im], 0, 0, 0);
                __builtin_amdgcn_s_barrier();
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) res += acc[i][j][0] + acc[i][j][3];
        } else {
            f32x16 acc[2][2];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
            for (int t = 0; t < trips; ++t) {
#pragma unroll
                for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                    for (int in = 0; in < 2; ++in)
#pragma unroll
                        for (int im = 0; im < 2; ++im) acc[in][im] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b[(in + kk) & 3], a[(im + kk) & 3], acc[in][im], 0, 0, 0);
                __builtin_amdgcn_s_barrier();
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) res += acc[i][j][0] + acc[i][j][15];
        }
    } else {
        f16x2 v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = (f16x2){(_Float16) (lane * 0.01f + i), (_Float16) (i * 0.5f)};
        const f16x2 m = {(_Float16) 1.0009765625f, (_Float16) 0.99951171875f};
        for (int t = 0; t < trips; ++t) {
#pragma unroll
            for (int i = 0; i < NV; ++i) v[i & 7] = v[i & 7] * m + v[(i + 3) & 7];      // 8 independent chains of v_pk_fma_f16
            __builtin_amdgcn_s_barrier();
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) res += (float) v[i][0] + (float) v[i][1];
    }
    out[blockIdx.x * 768 + threadIdx.x] = res;
}

template <int SHAPE, int NV> void run(float* out)
{
    const int trips = 20000, blocks = 256;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    k<SHAPE, NV><<<blocks, 768>>>(out, 100, 1);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    k<SHAPE, NV><<<blocks, 768>>>(out, trips, 2);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double flop = (double) blocks * 8 * trips * 32 * 2.0 * 16 * 16 * 32;
    printf("%s  loader VALU/trip %3d : %8.3f ms  %7.1f TFLOP/s  %.0f ns per trip\n", SHAPE ? "32x32x16" : "16x16x32", NV, ms, flop / ms / 1e9, ms * 1e6 / trips);
}
int main()
{
    float* out; CK(hipMalloc(&out, 256 * 768 * 4));
    run<0, 0>(out); run<0, 32>(out); run<0, 64>(out); run<0, 96>(out); run<0, 128>(out); run<0, 192>(out);
    run<1, 0>(out); run<1, 32>(out); run<1, 64>(out); run<1, 96>(out); run<1, 128>(out); run<1, 192>(out);
    return 0;
}
