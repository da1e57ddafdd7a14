// What does ONE dependent kernel boundary cost inside a replayed hipGraph on MI355X, and which launch attribute moves it?
//   hipcc -O3 --offload-arch=gfx950 scripts/bench_boundary.hip -o build/bench_boundary
// Round 1 measured "3.1-3.3 us per empty dependent kernel" with 2000 EAGER launches -- a host-bound figure (the host
// needs ~3.3-3.8 us per launch).  Here every chain of N kernels is captured ONCE into a hipGraph and replayed; the time of a
// replay / N is the device-side cost per kernel = launch boundary + the kernel's own start-up.  One attribute varies at a
// time around the decode GEMV's real geometry (512 threads, 2 blocks per CU, ~9 KB dynamic LDS, a 328-byte by-value
// argument struct):
//   grid      256 / 512 / 1024 blocks           threads   256 / 512
//   lds       0 / 9 KB / 24 KB / 64 KB dynamic  kernarg   8 B / 328 B (all of it read by the kernel)
//   body      empty | reads its arguments | + one dependent 8 KB vector read (what the previous kernel wrote) |
//             + 46 MB streamed (each block reads its 1/512 share with 16-byte nt loads, 4 in flight per lane)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s (line %d)\n", #x, hipGetErrorString(e), __LINE__); exit(1); } } while (0)

struct Big { const float* p[8]; int v[66]; };           // 8 * 8 + 66 * 4 = 328 bytes, like DecGemvArgs

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// BODY 0: empty.  1: reads every kernel argument.  2: + each thread reads 16 bytes of the vector the previous kernel wrote
// and block 0 rewrites it.  3: + streams `bytes` of weights (grid-strided 16-byte nt loads, 4 in flight per lane).
template <int BODY, typename ARGS>
__global__ __launch_bounds__(512) void kern(const ARGS a, float* vec, const u32x4* w, size_t pieces, float* sink)
{
    extern __shared__ float lds[];
    if (BODY == 0) return;
    float acc = 0.f;
    if constexpr (sizeof(ARGS) > 16) {
#pragma unroll
        for (int i = 0; i < 66; ++i) acc += (float) a.v[i];
    } else {
        acc = (float) a.v[0];
    }
    if (BODY >= 2) {
        const float4 x = *(const float4*) (vec + (threadIdx.x & 511) * 4);
        acc += x.x + x.y + x.z + x.w;
        lds[threadIdx.x] = acc;
        __syncthreads();
        acc += lds[(threadIdx.x + 1) & (blockDim.x - 1)];
    }
    if (BODY >= 3) {
        const size_t stride = (size_t) gridDim.x * blockDim.x;
        size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
        u32x4 s = {0, 0, 0, 0};
        for (; i + 3 * stride < pieces; i += 4 * stride) {
            const u32x4 v0 = __builtin_nontemporal_load(w + i), v1 = __builtin_nontemporal_load(w + i + stride);
            const u32x4 v2 = __builtin_nontemporal_load(w + i + 2 * stride), v3 = __builtin_nontemporal_load(w + i + 3 * stride);
            s ^= v0 ^ v1 ^ v2 ^ v3;
        }
        acc += (float) (s[0] ^ s[1] ^ s[2] ^ s[3]);
    }
    if (BODY >= 2 && blockIdx.x == 0) vec[threadIdx.x * 4 % 2048] = acc * 1e-30f;
    if (acc == 12345.678f) sink[0] = acc;
}

struct Small { int v[2]; };

template <int BODY, typename ARGS>
static float run(int grid, int threads, size_t lds, int chain, int reps, float* vec, const u32x4* w, size_t pieces, float* sink)
{
    ARGS a;
    for (size_t i = 0; i < sizeof(a.v) / sizeof(int); ++i) a.v[i] = (int) i;
    auto kfn = kern<BODY, ARGS>;
    if (lds > 48 * 1024) CK(hipFuncSetAttribute((const void*) kfn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipStream_t s;
    CK(hipStreamCreate(&s));
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    for (int i = 0; i < chain; ++i) hipLaunchKernelGGL(kfn, dim3(grid), dim3(threads), lds, s, a, vec, w, pieces, sink);
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
    CK(hipEventRecord(e0, s));
    for (int r = 0; r < reps; ++r) CK(hipGraphLaunch(ge, s));
    CK(hipEventRecord(e1, s));
    CK(hipEventSynchronize(e1));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g)); CK(hipStreamDestroy(s));
    return ms * 1000.f / (float) (reps * chain);
}

int main()
{
    float *vec, *sink; u32x4* w;
    const size_t bytes = 46880256, pieces = bytes / 16;
    const size_t wbytes = (size_t) 1 << 32;                      // 4 GiB: consecutive kernels stream different weights (no MALL reuse)
    CK(hipMalloc(&vec, 8192 * 4)); CK(hipMalloc(&sink, 64)); CK(hipMalloc(&w, wbytes));
    CK(hipMemset(vec, 0, 8192 * 4)); CK(hipMemset(w, 1, wbytes));
    const int chain = 160, reps = 20;
    printf("per-kernel time inside a replayed hipGraph of %d dependent kernels (us)\n", chain);
    printf("empty, 8 B args:   grid 256x256 %.2f | 512x512 %.2f | 1024x512 %.2f | 256x512 %.2f\n",
           run<0, Small>(256, 256, 0, chain, reps, vec, w, pieces, sink), run<0, Small>(512, 512, 0, chain, reps, vec, w, pieces, sink),
           run<0, Small>(1024, 512, 0, chain, reps, vec, w, pieces, sink), run<0, Small>(256, 512, 0, chain, reps, vec, w, pieces, sink));
    printf("empty, 512x512:    lds 0 %.2f | 9 KB %.2f | 24 KB %.2f | 64 KB %.2f\n",
           run<0, Small>(512, 512, 0, chain, reps, vec, w, pieces, sink), run<0, Small>(512, 512, 9 * 1024, chain, reps, vec, w, pieces, sink),
           run<0, Small>(512, 512, 24 * 1024, chain, reps, vec, w, pieces, sink), run<0, Small>(512, 512, 64 * 1024, chain, reps, vec, w, pieces, sink));
    printf("512x512, 9 KB lds: empty 328 B args %.2f | args read 8 B %.2f | args read 328 B %.2f | + dependent 8 KB vector %.2f\n",
           run<0, Big>(512, 512, 9 * 1024, chain, reps, vec, w, pieces, sink), run<1, Small>(512, 512, 9 * 1024, chain, reps, vec, w, pieces, sink),
           run<1, Big>(512, 512, 9 * 1024, chain, reps, vec, w, pieces, sink), run<2, Big>(512, 512, 9 * 1024, chain, reps, vec, w, pieces, sink));
    // streaming bodies: the same 46.9 MB per kernel, a different slice of the 4 GiB buffer per kernel would need per-node
    // arguments; instead the chain is short enough that 46.9 MB x 160 = 7.5 GB >> the 256 MB Infinity Cache only if the
    // slices differ -- so measure both the "same slice" (MALL-resident) and the event-timed eager "different slices" forms.
    {
        const float t = run<3, Big>(512, 512, 9 * 1024, chain, reps, vec, w, pieces, sink);
        printf("512x512 stream 46.9 MB (same slice every kernel: Infinity-Cache resident): %.2f us = %.2f TB/s\n", t, bytes / t * 1e-6);
    }
    {   // different slice per kernel: capture a chain whose nodes walk through the 4 GiB buffer
        hipStream_t s; CK(hipStreamCreate(&s));
        Big a; for (int i = 0; i < 66; ++i) a.v[i] = i;
        for (int variant = 0; variant < 3; ++variant) {
            const int grid = variant == 0 ? 512 : variant == 1 ? 256 : 1024;
            hipGraph_t g; hipGraphExec_t ge;
            CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
            const int n = 80;
            for (int i = 0; i < n; ++i)
                hipLaunchKernelGGL((kern<3, Big>), dim3(grid), dim3(512), 9 * 1024, s, a, vec, w + (size_t) i * (pieces + 4096), pieces, sink);
            CK(hipStreamEndCapture(s, &g));
            CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
            CK(hipEventRecord(e0, s));
            for (int r = 0; r < 10; ++r) CK(hipGraphLaunch(ge, s));
            CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            const float t = ms * 1000.f / (10 * n);
            printf("%4dx512 stream 46.9 MB from HBM (a different slice per kernel) + dependent vector: %.2f us = %.2f TB/s\n", grid, t, bytes / t * 1e-6);
            CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
        }
        CK(hipStreamDestroy(s));
    }
    return 0;
}
