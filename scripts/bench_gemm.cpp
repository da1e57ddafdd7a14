// Stand-alone prefill-GEMM driver through the C ABI: times exl_q4_matmul_gemm for the Llama-7B shapes at M rows.
//   hipcc -O2 -std=c++17 --offload-arch=gfx950 scripts/bench_gemm.cpp -Iinclude -Lexllama_amd -lexl_amd -Wl,-rpath,'$ORIGIN/../exllama_amd' -o build/bench_gemm
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>
#include "exl_amd.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
#define EX(x) do { int r = (x); if (r) { printf("%s -> %d: %s\n", #x, r, exl_last_error()); exit(1); } } while (0)
__global__ void fill_u32(uint32_t* p, size_t n, uint32_t seed)
{
    for (size_t i = blockIdx.x * (size_t) blockDim.x + threadIdx.x; i < n; i += (size_t) gridDim.x * blockDim.x) {
        uint32_t x = (uint32_t) i * 2654435761u + seed; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13; x *= 3266489917u; x ^= x >> 16; p[i] = x;
    }
}
__global__ void fill_f16(_Float16* p, size_t n, float lo, float hi, uint32_t seed)
{
    for (size_t i = blockIdx.x * (size_t) blockDim.x + threadIdx.x; i < n; i += (size_t) gridDim.x * blockDim.x) {
        uint32_t x = (uint32_t) i * 2654435761u + seed; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
        p[i] = (_Float16) (lo + (hi - lo) * ((x & 0xFFFF) / 65535.0f));
    }
}
int main(int argc, char** argv)
{
    const int M = argc > 1 ? atoi(argv[1]) : 2048;
    const int reps = argc > 2 ? atoi(argv[2]) : 20;
    const int shapes[3][2] = {{4096, 4096}, {4096, 11008}, {11008, 4096}};
    CK(hipSetDevice(0));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    double tot_flop = 0, tot_ms = 0;
    {   // scratch the short-prompt path re-tiles the activations into (the reference's temp_state)
        const size_t ts = (size_t) 2048 * 11008, tm = 1024;
        void *t0, *t1, *t2, *t3;
        CK(hipMalloc(&t0, ts * 2)); CK(hipMalloc(&t1, tm * 2)); CK(hipMalloc(&t2, 1024 * 4)); CK(hipMalloc(&t3, 1024 * 2));
        EX(exl_prepare_buffers(0, t0, ts, t1, tm, t2, 1024, t3, 1024));
    }
    for (auto& sh : shapes) {
        const int K = sh[0], N = sh[1], gs = 128, G = K / gs, NB = 6;     // NB weight copies: beyond the 256 MB cache together
        std::vector<void*> hs(NB);
        for (int b = 0; b < NB; ++b) {
            uint32_t *qw, *qz; _Float16* sc;
            CK(hipMalloc(&qw, (size_t) K / 8 * N * 4)); CK(hipMalloc(&qz, (size_t) G * N / 8 * 4)); CK(hipMalloc(&sc, (size_t) G * N * 2));
            fill_u32<<<1024, 256>>>(qw, (size_t) K / 8 * N, b + 1);
            CK(hipMemset(qz, 0x77, (size_t) G * N / 8 * 4));
            fill_f16<<<256, 256>>>(sc, (size_t) G * N, 0.002f, 0.006f, b + 7);
            EX(exl_make_q4(0, K, N, G, qw, qz, (uint16_t*) sc, nullptr, nullptr, &hs[b]));
        }
        _Float16 *x, *out;
        CK(hipMalloc(&x, (size_t) M * K * 2)); CK(hipMalloc(&out, (size_t) M * N * 2));
        fill_f16<<<1024, 256>>>(x, (size_t) M * K, -1.f, 1.f, 3);
        for (int i = 0; i < NB; ++i) EX(exl_q4_matmul_gemm(hs[i], x, M, out, 0, nullptr));
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        for (int i = 0; i < reps; ++i) EX(exl_q4_matmul_gemm(hs[i % NB], x, M, out, 0, nullptr));
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms * 1e3 / reps, tf = 2.0 * M * K * N / us / 1e6;
        printf("M %d K %5d N %5d : %8.2f us  %7.1f TFLOP/s\n", M, K, N, us, tf);
        const int cnt = (K == 4096 && N == 4096) ? 4 : (N == 11008 ? 2 : 1);
        tot_flop += cnt * 2.0 * M * K * N; tot_ms += cnt * us / 1e3;
    }
    printf("7B layer linear part: %.3f ms per layer at M=%d -> %.1f TFLOP/s\n", tot_ms, M, tot_flop / tot_ms / 1e9);
    return 0;
}
