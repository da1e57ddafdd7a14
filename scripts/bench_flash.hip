// Stand-alone driver of the prompt-pass attention kernel (csrc/flash_prefill.hip is compiled INTO this program, so probe / variant
// builds need no second library): S x S causal attention of H heads, head_dim 128, checked against a plain fp32 kernel on the same
// fp16 inputs, timed with HIP events.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 [-DEXL_FLASH_PROBE] scripts/bench_flash.hip -Iinclude -Iexllama_amd/csrc -o build/bench_flash
// (NOT linked against libexl_amd.so: the library registers a kernel of the same name, and which of the two a launch reaches would
// depend on the order of the registrations)
//   build/bench_flash [S] [heads] [kv_heads] [past] [reps]
#include "../exllama_amd/csrc/flash_prefill.hip"
#include <stdio.h>
#include <vector>
#include <algorithm>
#include <stdarg.h>
static char g_err[512];
void exl_set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof(g_err), fmt, ap); va_end(ap); }
extern "C" const char* exl_last_error(void) { return g_err; }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void fill_f16(f16* p, size_t n, float lo, float hi, uint32_t seed)
{
    for (size_t i = blockIdx.x * (size_t) blockDim.x + threadIdx.x; i < n; i += (size_t) gridDim.x * blockDim.x) {
        uint32_t x = (uint32_t) i * 2654435761u + seed; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13; x *= 3266489917u; x ^= x >> 16;
        p[i] = (f16) (lo + (hi - lo) * ((x & 0xFFFF) / 65535.0f));
    }
}

// one block = one (query row, head); thread d owns output dimension d; scores through LDS in chunks of 256 keys
__global__ __launch_bounds__(128) void naive_attn(const f16* q, const f16* kc, const f16* vc, float* out, int q_len, int heads, int kv_heads,
                                                   int max_seq, int past, float scale)
{
    __shared__ float qs[128], sc[256], red[2];
    const int row = blockIdx.x, h = blockIdx.y, d = threadIdx.x;
    const int kvh = h / (heads / kv_heads);
    qs[d] = (float) q[((size_t) row * heads + h) * 128 + d];
    __syncthreads();
    const int vis = past + row + 1;
    float m = -INFINITY, l = 0.f, o = 0.f;
    for (int k0 = 0; k0 < vis; k0 += 256) {
        const int n = min(256, vis - k0);
        for (int j = d; j < n; j += 128) {
            const f16* kr = kc + ((size_t) kvh * max_seq + k0 + j) * 128;
            float s = 0.f;
            for (int e = 0; e < 128; ++e) s = fmaf(qs[e], (float) kr[e], s);
            sc[j] = s * scale;
        }
        __syncthreads();
        float mx = m;
        for (int j = 0; j < n; ++j) mx = fmaxf(mx, sc[j]);
        const float a = __expf(m - mx);
        l *= a; o *= a;
        for (int j = 0; j < n; ++j) {
            const float p = __expf(sc[j] - mx);
            l += p;
            o = fmaf(p, (float) vc[((size_t) kvh * max_seq + k0 + j) * 128 + d], o);
        }
        m = mx;
        __syncthreads();
    }
    out[((size_t) row * heads + h) * 128 + d] = o / l;
}

int main(int argc, char** argv)
{
    const int S = argc > 1 ? atoi(argv[1]) : 2048, H = argc > 2 ? atoi(argv[2]) : 32, KVH = argc > 3 ? atoi(argv[3]) : H;
    const int past = argc > 4 ? atoi(argv[4]) : 0, reps = argc > 5 ? atoi(argv[5]) : 20;
    const int max_seq = past + S;
    CK(hipSetDevice(0));
    f16 *q, *kc, *vc, *out; float* ref;
    CK(hipMalloc(&q, (size_t) S * H * 128 * 2)); CK(hipMalloc(&out, (size_t) S * H * 128 * 2)); CK(hipMalloc(&ref, (size_t) S * H * 128 * 4));
    CK(hipMalloc(&kc, (size_t) KVH * max_seq * 128 * 2)); CK(hipMalloc(&vc, (size_t) KVH * max_seq * 128 * 2));
    fill_f16<<<1024, 256>>>(q, (size_t) S * H * 128, -2.f, 2.f, 1);
    fill_f16<<<1024, 256>>>(kc, (size_t) KVH * max_seq * 128, -2.f, 2.f, 2);
    fill_f16<<<1024, 256>>>(vc, (size_t) KVH * max_seq * 128, -1.f, 1.f, 3);
    CK(hipMemset(out, 0xFF, (size_t) S * H * 128 * 2));
    if (launch_flash_prefill(q, kc, vc, out, 1, S, H, KVH, 128, max_seq, past, nullptr)) { printf("launch failed: %s\n", exl_last_error()); return 1; }
    naive_attn<<<dim3(S, H), 128>>>(q, kc, vc, ref, S, H, KVH, max_seq, past, 1.0f / sqrtf(128.f));
    CK(hipDeviceSynchronize());
    std::vector<f16> ho((size_t) S * H * 128); std::vector<float> hr((size_t) S * H * 128);
    CK(hipMemcpy(ho.data(), out, ho.size() * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(hr.data(), ref, hr.size() * 4, hipMemcpyDeviceToHost));
    double worst = 0, scale = 0; size_t at = 0; int bad = 0;
    for (size_t i = 0; i < ho.size(); ++i) {
        const double g = (double) (float) ho[i], r = hr[i];
        if (!(g == g)) { ++bad; continue; }
        scale = std::max(scale, fabs(r));
        if (fabs(g - r) > worst) { worst = fabs(g - r); at = i; }
    }
    printf("S %d heads %d kv %d past %d: max |out - fp32 ref| %.3e at row %zu head %zu d %zu (scale %.3f), NaN %d\n", S, H, KVH, past, worst,
           at / ((size_t) H * 128), (at / 128) % H, at % 128, scale, bad);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) launch_flash_prefill(q, kc, vc, out, 1, S, H, KVH, 128, max_seq, past, nullptr);
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) launch_flash_prefill(q, kc, vc, out, 1, S, H, KVH, 128, max_seq, past, nullptr);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / reps;
    const double flops = 4.0 * H * 128 * ((double) S * past + (double) S * (S + 1) / 2);
    printf("flash prefill: %.1f us  %.1f TFLOP/s (causal flops)\n", us, flops / us / 1e6);
#ifdef EXL_FLASH_PROBE
    if (!getenv("EXL_FLASH_4WAVE")) {
        static unsigned long long h[FA_PROBE_BLOCKS * 2 * 8];
        CK(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_flash_probe8), sizeof(h)));
        for (int b : {0, 1, 255, 256, 511}) {
            if (b >= (S + 127) / 128 * H) continue;
            for (int set = 0; set < 2; ++set) {
                const unsigned long long* p = h + ((size_t) b * 2 + set) * 8;
                if (!p[5]) continue;
                printf("probe block %3d set %d: %llu steps, %6.0f cycles per step = barrier(even set) %5.0f + softmax, P V %5.0f + barrier(odd set) %5.0f + S, DMA issue %5.0f\n",
                       b, set, p[5], (double) p[0] / p[5], (double) p[1] / p[5], (double) p[2] / p[5], (double) p[3] / p[5], (double) p[4] / p[5]);
            }
        }
    } else {
        static unsigned long long h[FA_PROBE_BLOCKS * 8];
        CK(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_flash_probe), sizeof(h)));
        int bi = 0;
        for (int b = 0; b < FA_PROBE_BLOCKS; ++b) if (h[b * 8 + 0] > h[bi * 8 + 0]) bi = b;
        const unsigned long long* p = h + (size_t) bi * 8;
        printf("probe (4-wave kernel), slowest block %d (wave 0): %llu tiles, %.0f cycles per tile = wait + barrier %.0f, LDS store %.0f, barrier %.0f, load issue %.0f, compute %.0f\n",
               bi, p[6], (double) p[0] / p[6], (double) p[1] / p[6], (double) p[2] / p[6], (double) p[3] / p[6], (double) p[4] / p[6], (double) p[5] / p[6]);
    }
#endif
    return worst < 2e-2 * std::max(scale, 1.0) && bad == 0 ? 0 : 2;
}
