// Stall attribution of the pipelined dual (gate/up) prefill GEMM: compiles csrc/q4_gemm.hip with EXL_GEMM_PROBE and reports, per K
// step, the cycles a wave spends in the first half (4 MFMA groups + the vmcnt wait), the second half (4 groups + dequant / store)
// and the barrier, averaged over blocks, for the older (0-3) and younger (4-7) wave of each SIMD.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -Iinclude -Iexllama_amd/csrc -DEXL_GEMM_PROBE scripts/probe_dual.hip -o build/probe_dual
#include "../exllama_amd/csrc/q4_gemm.hip"
#include <vector>
Q4Matrix* q4_from_handle(void*) { return nullptr; }
int launch_column_remap(const f16*, f16*, int, int, const uint32_t*, hipStream_t) { return 0; }
int launch_gemm_t16s(const Q4Matrix*, const f16*, int, f16*, int, hipStream_t) { return 1; }
void exl_set_error(const char*, ...) {}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
__global__ void fill_u32(uint32_t* p, size_t n, uint32_t seed)
{
    for (size_t i = blockIdx.x * (size_t) blockDim.x + threadIdx.x; i < n; i += (size_t) gridDim.x * blockDim.x) {
        uint32_t x = (uint32_t) i * 2654435761u + seed; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13; p[i] = x;
    }
}
__global__ void fill_f16(f16* p, size_t n, float lo, float hi)
{
    for (size_t i = blockIdx.x * (size_t) blockDim.x + threadIdx.x; i < n; i += (size_t) gridDim.x * blockDim.x) {
        uint32_t x = (uint32_t) i * 2654435761u; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
        p[i] = (f16) (lo + (hi - lo) * ((x & 0xFFFF) / 65535.0f));
    }
}
int main(int argc, char** argv)
{
    const int M = argc > 1 ? atoi(argv[1]) : 2048;
    const int K = 4096, N = 11008, gs = 128, G = K / gs;
    Q4Matrix w[2];
    for (int i = 0; i < 2; ++i) {
        w[i] = Q4Matrix{};
        w[i].magic = EXL_Q4_MAGIC; w[i].device = 0; w[i].height = K; w[i].width = N; w[i].groups = G; w[i].groupsize = gs; w[i].layout = EXL_LAYOUT_T16;
        CK(hipMalloc(&w[i].qweight, (size_t) K / 8 * N * 4)); CK(hipMalloc(&w[i].qzeros, (size_t) G * N / 8 * 4)); CK(hipMalloc(&w[i].scales, (size_t) G * N * 2));
        fill_u32<<<1024, 256>>>(w[i].qweight, (size_t) K / 8 * N, 1 + i);
        CK(hipMemset(w[i].qzeros, 0x77, (size_t) G * N / 8 * 4));
        fill_f16<<<256, 256>>>(w[i].scales, (size_t) G * N, 0.002f, 0.006f);
    }
    f16 *x, *out;
    CK(hipMalloc(&x, (size_t) M * K * 2)); CK(hipMalloc(&out, (size_t) M * N * 2));
    fill_f16<<<1024, 256>>>(x, (size_t) M * K, -1.f, 1.f);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 200; ++i) launch_q4_gemm_dual(&w[0], &w[1], x, M, out, nullptr, 1, nullptr);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    const int reps = 200;
    for (int i = 0; i < reps; ++i) launch_q4_gemm_dual(&w[0], &w[1], x, M, out, nullptr, 1, nullptr);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / reps;
    const int nblk = 8 * ((N / 128 + 7) / 8) * ((M + 255) / 256);
    std::vector<unsigned long long> h((size_t) 1024 * 8 * 4);
    CK(hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(g_gemm_probe), h.size() * 8));
    const int steps = K / 64;
    double c[2][4] = {{0}}; int n = 0;
    for (int b = 0; b < nblk && b < 1024; ++b) {
        if (!h[(size_t) b * 32]) continue;
        ++n;
        for (int wv = 0; wv < 8; ++wv)
            for (int q = 0; q < 4; ++q) c[wv >> 2][q] += h[((size_t) b * 8 + wv) * 4 + q] / 4.0;
    }
    printf("dual M %d K %d N %d: %.1f us %.1f TFLOP/s (probe build)\n", M, K, N, us, 4.0 * M * K * N / us / 1e6);
    for (int g = 0; g < 2; ++g)
        printf("  waves %d-%d per K step: total %6.0f  first half + vmcnt wait %6.0f  second half + stores %6.0f  barrier %6.0f   (cycles, %d blocks)\n",
               4 * g, 4 * g + 3, c[g][0] / n / steps, c[g][1] / n / steps, c[g][2] / n / steps, c[g][3] / n / steps, n);
    return 0;
}
