// Stall attribution of the pipelined prefill GEMM: compiles csrc/q4_gemm.hip with EXL_GEMM_PROBE and reports, per K step,
// the cycles a wave spends in the VMEM wait / dequant + LDS store / barrier (averaged over blocks and waves).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -Iinclude -Iexllama_amd/csrc -DEXL_GEMM_PROBE scripts/probe_gemm.hip -o build/probe_gemm
#include "../exllama_amd/csrc/q4_gemm.hip"
#include <vector>
// the pieces of the library this translation unit references but does not exercise
Q4Matrix* q4_from_handle(void*) { return nullptr; }
int launch_gemm_t16s(const Q4Matrix*, const f16*, int, f16*, int, hipStream_t) { return 1; }
int launch_column_remap(const f16*, f16*, int, int, const uint32_t*, hipStream_t) { return 0; }
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
    const int shapes[3][2] = {{4096, 4096}, {4096, 11008}, {11008, 4096}};
    for (auto& sh : shapes) {
        const int K = sh[0], N = sh[1], gs = 128, G = K / gs;
        Q4Matrix w{};
        w.magic = EXL_Q4_MAGIC; w.device = 0; w.height = K; w.width = N; w.groups = G; w.groupsize = gs; w.layout = EXL_LAYOUT_T16;
        CK(hipMalloc(&w.qweight, (size_t) K / 8 * N * 4)); CK(hipMalloc(&w.qzeros, (size_t) G * N / 8 * 4)); CK(hipMalloc(&w.scales, (size_t) G * N * 2));
        fill_u32<<<1024, 256>>>(w.qweight, (size_t) K / 8 * N, 1);
        CK(hipMemset(w.qzeros, 0x77, (size_t) G * N / 8 * 4));
        fill_f16<<<256, 256>>>(w.scales, (size_t) G * N, 0.002f, 0.006f);
        f16 *x, *out;
        CK(hipMalloc(&x, (size_t) M * K * 2)); CK(hipMalloc(&out, (size_t) M * N * 2));
        fill_f16<<<1024, 256>>>(x, (size_t) M * K, -1.f, 1.f);
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int i = 0; i < 3; ++i) launch_q4_gemm(&w, x, M, out, 0, nullptr, 0, nullptr);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        const int reps = 10;
        for (int i = 0; i < reps; ++i) launch_q4_gemm(&w, x, M, out, 0, nullptr, 0, nullptr);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms * 1e3 / reps;
        const int nblk = 8 * ((N / 128 + 7) / 8) * ((M + 255) / 256);
        std::vector<unsigned long long> h((size_t) 1024 * 8 * 4);
        CK(hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(g_gemm_probe), h.size() * 8));
        const int steps = K / 64;
        double c[4] = {0}, pr[4] = {0}; int n = 0;
        for (int b = 0; b < nblk && b < 1024; ++b) {
            if (!h[(size_t) b * 32]) continue;
            ++n;
            for (int wv = 0; wv < 4; ++wv)
                for (int q = 0; q < 4; ++q) { c[q] += h[((size_t) b * 8 + wv) * 4 + q] / 4.0; pr[q] += h[((size_t) b * 8 + 4 + wv) * 4 + q] / 4.0; }
        }
        printf("M %d K %5d N %5d: %7.1f us %6.1f TFLOP/s | consumer per K step: total %6.0f (loop %.1f us by the 100 MHz clock -> %.2f GHz)  mfma part %6.0f  barrier %6.0f | producer per step: issue %6.0f  wait %6.0f  dequant+store %6.0f  barrier %6.0f\n",
               M, K, N, us, 2.0 * M * K * N / us / 1e6, c[0] / n / steps, c[2] / n / 100.0, c[0] / (c[2] * 10.0), c[1] / n / steps, c[3] / n / steps,
               pr[0] / n / steps, pr[1] / n / steps, pr[2] / n / steps, pr[3] / n / steps);
    }
    return 0;
}
