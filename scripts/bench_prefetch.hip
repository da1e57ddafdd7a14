// Can a side stream keep HBM busy across launch boundaries by pulling the NEXT kernel's weights into the 256 MB Infinity
// Cache while the current kernel runs?
//   hipcc -O3 --offload-arch=gfx950 scripts/bench_prefetch.hip -o build/bench_prefetch
// Facts this builds on (profiles/r02_boundary_microbench.txt): a dependent kernel boundary costs 1.6 us during which HBM idles;
// a 46.9 MB stream takes 7.9 us from HBM and 4.8 us when the same bytes sit in the Infinity Cache.  The prefetch is a pure
// hint (the consumer never waits for it): no flags, no correctness or deadlock risk.
// Graph: main chain K(0) -> K(1) -> ... (consumer kernels, each streams its own 46.9 MB slice + a dependent 8 KB vector);
// side branch P(i+1) starts when K(i-1) has finished (i.e. together with K(i)) and touches slice i+1.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s (line %d)\n", #x, hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(512) void consumer(float* vec, const u32x4* w, size_t pieces, float* sink)
{
    __shared__ float lds[512];
    const float4 x = *(const float4*) (vec + (threadIdx.x & 511) * 4);
    float acc = x.x + x.y + x.z + x.w;
    lds[threadIdx.x] = acc;
    __syncthreads();
    acc += lds[(threadIdx.x + 1) & 511];
    const size_t stride = (size_t) gridDim.x * blockDim.x;
    u32x4 s = {0, 0, 0, 0};
    for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i + 3 * stride < pieces; i += 4 * stride) {
        const u32x4 v0 = __builtin_nontemporal_load(w + i), v1 = __builtin_nontemporal_load(w + i + stride);
        const u32x4 v2 = __builtin_nontemporal_load(w + i + 2 * stride), v3 = __builtin_nontemporal_load(w + i + 3 * stride);
        s ^= v0 ^ v1 ^ v2 ^ v3;
    }
    acc += (float) (s[0] ^ s[1] ^ s[2] ^ s[3]);
    if (blockIdx.x == 0) vec[threadIdx.x * 4 % 2048] = acc * 1e-30f;
    if (acc == 12345.678f) sink[0] = acc;
}

// touches `pieces` 16-byte pieces (one 64-byte... every piece is loaded: a load brings its 128-byte line) and drops them
template <int NT>
__global__ __launch_bounds__(256) void prefetcher(const u32x4* w, size_t pieces, float* sink)
{
    const size_t stride = (size_t) gridDim.x * blockDim.x;
    u32x4 s = {0, 0, 0, 0};
    // one 16-byte load per 128-byte line is enough to pull the line: lane l reads piece 8 * l of its wave's 8 KiB span
    for (size_t i = ((size_t) blockIdx.x * blockDim.x + threadIdx.x) * 8; i < pieces; i += stride * 8 * 4) {
        const size_t i1 = i + stride * 8, i2 = i + stride * 16, i3 = i + stride * 24;
        const u32x4 v0 = NT ? __builtin_nontemporal_load(w + i) : w[i];
        const u32x4 v1 = i1 < pieces ? (NT ? __builtin_nontemporal_load(w + i1) : w[i1]) : v0;
        const u32x4 v2 = i2 < pieces ? (NT ? __builtin_nontemporal_load(w + i2) : w[i2]) : v0;
        const u32x4 v3 = i3 < pieces ? (NT ? __builtin_nontemporal_load(w + i3) : w[i3]) : v0;
        s ^= v0 ^ v1 ^ v2 ^ v3;
    }
    if ((s[0] ^ s[1] ^ s[2] ^ s[3]) == 0x12345679u) sink[1] = 1.f;
}

static float run(int mode, int pf_blocks, float frac, float* vec, const u32x4* w, size_t pieces, float* sink)
{
    // mode 0: no prefetch; 1: P(i+1) next to K(i), plain loads; 2: same, nt loads
    hipStream_t a, b; CK(hipStreamCreate(&a)); CK(hipStreamCreate(&b));
    const int n = 64;
    std::vector<hipEvent_t> ev(n + 1);
    for (auto& evt : ev) CK(hipEventCreateWithFlags(&evt, hipEventDisableTiming));
    hipEvent_t joined; CK(hipEventCreateWithFlags(&joined, hipEventDisableTiming));
    hipGraph_t g; hipGraphExec_t ge;
    const size_t slice = pieces + 4096;
    CK(hipStreamBeginCapture(a, hipStreamCaptureModeGlobal));
    CK(hipEventRecord(ev[0], a));
    for (int i = 0; i < n; ++i) {
        if (mode && i + 1 < n) {
            CK(hipStreamWaitEvent(b, ev[i], 0));                      // K(i-1) finished == K(i) starting
            const size_t pp = (size_t) (pieces * frac);
            if (mode == 1) hipLaunchKernelGGL(prefetcher<0>, dim3(pf_blocks), dim3(256), 0, b, w + (size_t) (i + 1) * slice, pp, sink);
            else           hipLaunchKernelGGL(prefetcher<1>, dim3(pf_blocks), dim3(256), 0, b, w + (size_t) (i + 1) * slice, pp, sink);
        }
        hipLaunchKernelGGL(consumer, dim3(512), dim3(512), 0, a, vec, w + (size_t) i * slice, pieces, sink);
        CK(hipEventRecord(ev[i + 1], a));
    }
    if (mode) { CK(hipEventRecord(joined, b)); CK(hipStreamWaitEvent(a, joined, 0)); }
    CK(hipStreamEndCapture(a, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipGraphLaunch(ge, a)); CK(hipStreamSynchronize(a));
    CK(hipEventRecord(e0, a));
    const int reps = 10;
    for (int r = 0; r < reps; ++r) CK(hipGraphLaunch(ge, a));
    CK(hipEventRecord(e1, a)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g)); CK(hipStreamDestroy(a)); CK(hipStreamDestroy(b));
    return ms * 1000.f / (reps * n);
}

int main()
{
    float *vec, *sink; u32x4* w;
    const size_t bytes = 46880256, pieces = bytes / 16;
    CK(hipMalloc(&vec, 8192 * 4)); CK(hipMalloc(&sink, 64)); CK(hipMalloc(&w, (size_t) 1 << 32));
    CK(hipMemset(vec, 0, 8192 * 4)); CK(hipMemset(w, 1, (size_t) 1 << 32));
    printf("64 dependent consumer kernels, 46.9 MB each from a different slice (us per kernel, hipGraph replay)\n");
    printf("no prefetch:                                   %.2f\n", run(0, 0, 0.f, vec, w, pieces, sink));
    for (int mode = 1; mode <= 2; ++mode)
        for (int blocks : {64, 256, 512})
            for (float frac : {0.35f, 1.0f})
                printf("prefetch next slice (%s loads, %3d blocks x 256 threads, %3.0f%% of it): %.2f\n", mode == 1 ? "plain" : "nt   ", blocks, frac * 100,
                       run(mode, blocks, frac, vec, w, pieces, sink));
    return 0;
}
