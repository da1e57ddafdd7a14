// Stand-alone driver of the native decode executor through the C ABI only (no Python, no torch): builds a synthetic
// Llama-7B-shaped GPTQ model in HBM, then reports per-kernel-class times (exl_decoder_step_timed) and the hipGraph
// replay rate at two context lengths.  Doubles as a C-caller example of include/exl_amd.h.
//   hipcc -O2 -std=c++17 --offload-arch=gfx950 scripts/bench_decoder.cpp -Iinclude -Lexllama_amd -lexl_amd -Wl,-rpath,'$ORIGIN/../exllama_amd' -o build/bench_decoder
//   build/bench_decoder [layers=32] [ctx=2048] [groupsize=128] [contexts=2: ctx and 4; 1: ctx only (counter passes)]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <vector>
#include "exl_amd.h"
#ifdef EXL_RING_PROBE
extern "C" int exl_debug_ring_probe(int cls, unsigned long long* out8);
#endif
#ifdef EXL_ATTN_PROBE
extern "C" int exl_debug_attn_probe(unsigned long long* out8);
extern "C" int exl_debug_stream_probe(int cls, unsigned long long* out12);
#endif

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
#define EX(x) do { int r = (x); if (r) { printf("%s -> %d: %s\n", #x, r, exl_last_error()); exit(1); } } while (0)

__global__ void fill_u32(uint32_t* p, size_t n, uint32_t seed)
{
    for (size_t i = blockIdx.x * (size_t) blockDim.x + threadIdx.x; i < n; i += (size_t) gridDim.x * blockDim.x) {
        uint32_t x = (uint32_t) i * 2654435761u + seed;
        x ^= x >> 15; x *= 2246822519u; x ^= x >> 13; x *= 3266489917u; x ^= x >> 16;
        x |= (~(x | (x >> 1) | (x >> 2) | (x >> 3)) & 0x11111111u) << 3;   // every 0 nibble -> 8: weights symmetric about the zero point 8
        p[i] = x;                                                          // (uniform nibbles bias every weight by -0.5 steps: a deep model overflows fp16)
    }
}
__global__ void fill_f16(_Float16* p, size_t n, float lo, float hi, uint32_t seed)
{
    for (size_t i = blockIdx.x * (size_t) blockDim.x + threadIdx.x; i < n; i += (size_t) gridDim.x * blockDim.x) {
        uint32_t x = (uint32_t) i * 2654435761u + seed;
        x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
        p[i] = (_Float16) (lo + (hi - lo) * ((x & 0xFFFF) / 65535.0f));
    }
}

struct Lin { uint32_t *qw, *qz; _Float16* sc; void* h; };

static Lin make_lin(int K, int N, int gs, uint32_t seed)
{
    Lin l;
    const int G = K / gs;
    CK(hipMalloc(&l.qw, (size_t) K / 8 * N * 4));
    CK(hipMalloc(&l.qz, (size_t) G * N / 8 * 4));
    CK(hipMalloc(&l.sc, (size_t) G * N * 2));
    fill_u32<<<1024, 256>>>(l.qw, (size_t) K / 8 * N, seed);
    CK(hipMemset(l.qz, 0x77, (size_t) G * N / 8 * 4));
    const float s = 0.02f * sqrtf(4096.f / K) / 4.6f;
    fill_f16<<<256, 256>>>(l.sc, (size_t) G * N, 0.5f * s, 1.5f * s, seed ^ 0x9e3779b9u);
    EX(exl_make_q4(0, K, N, G, l.qw, l.qz, (uint16_t*) l.sc, nullptr, nullptr, &l.h));
    return l;
}

int main(int argc, char** argv)
{
    const int L = argc > 1 ? atoi(argv[1]) : 32;
    const int ctx = argc > 2 ? atoi(argv[2]) : 2048;
    const int gs = argc > 3 ? atoi(argv[3]) : 128;
    const int nctx = argc > 4 ? atoi(argv[4]) : 2;
    const int h = 4096, I = 11008, heads = 32, kvh = 32, hd = 128, V = 32000, maxseq = ctx + 160;
    CK(hipSetDevice(0));
    _Float16 *embed, *lm_head, *fnorm, *sin, *cos;
    CK(hipMalloc(&embed, (size_t) V * h * 2)); CK(hipMalloc(&lm_head, (size_t) V * h * 2)); CK(hipMalloc(&fnorm, h * 2));
    CK(hipMalloc(&sin, (size_t) maxseq * hd * 2)); CK(hipMalloc(&cos, (size_t) maxseq * hd * 2));
    fill_f16<<<1024, 256>>>(embed, (size_t) V * h, -0.04f, 0.04f, 1);
    fill_f16<<<1024, 256>>>(lm_head, (size_t) V * h, -0.04f, 0.04f, 2);
    fill_f16<<<16, 256>>>(fnorm, h, 1.f, 1.f, 3);
    fill_f16<<<256, 256>>>(sin, (size_t) maxseq * hd, -1.f, 1.f, 4);
    fill_f16<<<256, 256>>>(cos, (size_t) maxseq * hd, -1.f, 1.f, 5);
    void* dec;
    EX(exl_decoder_create(0, L, h, I, heads, kvh, hd, V, maxseq, 1e-6f, embed, fnorm, lm_head, sin, cos, &dec));
    for (int i = 0; i < L; ++i) {
        Lin q = make_lin(h, h, gs, 10 * i + 1), k = make_lin(h, kvh * hd, gs, 10 * i + 2), v = make_lin(h, kvh * hd, gs, 10 * i + 3);
        Lin o = make_lin(h, h, gs, 10 * i + 4), g = make_lin(h, I, gs, 10 * i + 5), u = make_lin(h, I, gs, 10 * i + 6);
        Lin d = make_lin(I, h, gs, 10 * i + 7);
        _Float16 *n1, *n2, *kc, *vc;
        CK(hipMalloc(&n1, h * 2)); CK(hipMalloc(&n2, h * 2));
        fill_f16<<<16, 256>>>(n1, h, 0.9f, 1.1f, 7); fill_f16<<<16, 256>>>(n2, h, 0.9f, 1.1f, 8);
        const size_t cb = (size_t) kvh * maxseq * hd;
        CK(hipMalloc(&kc, cb * 2)); CK(hipMalloc(&vc, cb * 2));
        fill_f16<<<1024, 256>>>(kc, cb, -1.f, 1.f, 100 + i); fill_f16<<<1024, 256>>>(vc, cb, -1.f, 1.f, 200 + i);
        EX(exl_decoder_set_layer(dec, i, q.h, k.h, v.h, o.h, g.h, u.h, d.h, n1, n2, kc, vc));
    }
    int64_t* tok; int32_t* pos; float* logits;
    CK(hipMalloc(&tok, 8)); CK(hipMalloc(&pos, 4)); CK(hipMalloc(&logits, (size_t) V * 4));
    const int64_t t0 = 17;
    CK(hipMemcpy(tok, &t0, 8, hipMemcpyHostToDevice));
    hipStream_t s;
    CK(hipStreamCreate(&s));
    static const char* names[EXL_DEC_NCLASS] = {"qkv", "attn", "merge", "o_proj", "gate_up", "down", "head"};
    const int ctxs[2] = {ctx, 4};
    for (int c = 0; c < nctx && c < 2; ++c) {
        const int32_t p0 = ctxs[c];
        CK(hipMemcpy(pos, &p0, 4, hipMemcpyHostToDevice));
        float ms[EXL_DEC_NCLASS];
        EX(exl_decoder_step_timed(dec, tok, pos, logits, 8, s, ms));
        float sum = 0;
        printf("ctx %5d  per-launch us:", p0);
        for (int k = 0; k < EXL_DEC_NCLASS; ++k) { printf(" %s %.2f", names[k], ms[k] * 1e3 / (k == EXL_DEC_HEAD ? 1 : L)); sum += ms[k]; }
        printf("  | sum %.3f ms/token\n", sum);
        // hipGraph replay of whole tokens
        EX(exl_decoder_step(dec, tok, pos, logits, 0, s));
        CK(hipStreamSynchronize(s));
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
        EX(exl_decoder_step(dec, tok, pos, logits, 0, s));
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int i = 0; i < 3; ++i) CK(hipGraphLaunch(ge, s));
        CK(hipStreamSynchronize(s));
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        const int reps = 30;
        CK(hipEventRecord(e0, s));
        for (int i = 0; i < reps; ++i) CK(hipGraphLaunch(ge, s));
        CK(hipEventRecord(e1, s));
        CK(hipEventSynchronize(e1));
        float tms = 0;
        CK(hipEventElapsedTime(&tms, e0, e1));
        printf("ctx %5d  graph replay: %.4f ms/token = %.1f tokens/s (x%d layers)\n", p0, tms / reps, 1e3 * reps / tms, L);
#ifdef EXL_RING_PROBE
        {
            static const char* rn[4] = {"qkv", "o_proj + merge", "gate_up", "plain vector (last launched: down)"};
            for (int cls = 0; cls < 4; ++cls) {
                unsigned long long pr[8];
                if (exl_debug_ring_probe(cls, pr) != 0 || !pr[7]) continue;
                const double nb = (double) pr[7];
                printf("ctx %5d  ring kernel [%s], mean cycles over %llu blocks: activation requested %.0f  ring issued %.0f  activation landed %.0f  image staged %.0f  unit 0 consumed %.0f  unit 0 reduced %.0f  end %.0f\n",
                       p0, rn[cls], pr[7], pr[0] / nb, pr[1] / nb, pr[2] / nb, pr[3] / nb, pr[4] / nb, pr[5] / nb, pr[6] / nb);
            }
        }
#endif
#ifdef EXL_ATTN_PROBE
        {
            unsigned long long pr[12];
            if (exl_debug_attn_probe(pr) == 0 && pr[7]) {
                static const char* ph[7] = {"pos known", "loads issued", "rope done", "scores done", "softmax done", "PV done", "end"};
                printf("ctx %5d  attention kernel, mean cycles since block start over %llu blocks:", p0, pr[7]);
                for (int i = 0; i < 7; ++i) printf("  %s %.0f", ph[i], (double) pr[i] / (double) pr[7]);
                printf("\n");
            }
            static const char* cn[8] = {"-", "plain vector + residual (the LAST launched: down)", "qkv", "gate_up", "-", "-", "-", "o_proj + split merge"};
            for (int cls = 0; cls < 8; ++cls) {
                if (exl_debug_stream_probe(cls, pr) != 0 || !pr[7]) continue;
                printf("ctx %5d  stream kernel class %d [%s], mean cycles over %llu blocks (%.2f units/block): args arrived %.0f  activation loads issued %.0f  unit described %.0f  weights issued %.0f  entries issued %.0f  image staged %.0f  unit 0 consumed %.0f  unit 0 reduced %.0f  all units done %.0f\n",
                       p0, cls, cn[cls], pr[7], (double) pr[5] / pr[7], (double) pr[6] / pr[7], (double) pr[8] / pr[7], (double) pr[9] / pr[7], (double) pr[10] / pr[7], (double) pr[0] / pr[7], (double) pr[1] / pr[7], (double) pr[2] / pr[7], (double) pr[3] / pr[7], (double) pr[4] / pr[7]);
            }
        }
#endif
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }
    float l0[4];
    CK(hipMemcpy(l0, logits, 16, hipMemcpyDeviceToHost));
    printf("logits[0..3] = %g %g %g %g\n", l0[0], l0[1], l0[2], l0[3]);
    return 0;
}
