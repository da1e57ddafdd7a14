// What does COLD CODE cost at the start of a kernel on MI355X?
//   hipcc -O3 --offload-arch=gfx950 scripts/bench_icache.hip -o build/bench_icache
// The decode GEMV's phase probe shows 3,000-6,000 cycles between "kernel arguments arrived" and "first loads issued" for
// ~250 instructions.  Hypothesis: instruction fetch -- every launch starts with a cold instruction cache (five different
// kernels alternate, 46 MB of weights stream through the L2 between two launches of the same code), and a straight-line
// prologue pays one miss per 64-byte line.
// Test: kernels whose body is N straight-line v_add_u32 (4 bytes each), timed with s_memtime from first to last instruction,
//   (a) launched back to back with themselves (code hot in the instruction cache / L2),
//   (b) alternating with a 46 MB streaming kernel and a DIFFERENT code kernel (the decode pattern).
// Geometry as in the decoder: 512 blocks x 512 threads.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s (line %d)\n", #x, hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

#define BODY(N) asm volatile(".rept " #N "\n v_add_u32 %0, %0, %1\n .endr" : "+v"(v) : "v"(one))

template <int N, int VARIANT>
__global__ __launch_bounds__(512) void code_kernel(unsigned long long* cycles, uint32_t* sink)
{
    const unsigned long long t0 = __builtin_readcyclecounter();
    uint32_t v = threadIdx.x, one = 1 + VARIANT;
    if constexpr (N == 64) BODY(64);
    if constexpr (N == 256) BODY(256);
    if constexpr (N == 1024) BODY(1024);
    if constexpr (N == 4096) BODY(4096);
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) atomicAdd(cycles + (N == 64 ? 0 : N == 256 ? 1 : N == 1024 ? 2 : 3), t1 - t0);
    if (v == 0xdeadbeef) sink[0] = v;
}

__global__ __launch_bounds__(512) void stream_kernel(const u32x4* w, size_t pieces, uint32_t* sink)
{
    const size_t stride = (size_t) gridDim.x * blockDim.x;
    u32x4 s = {0, 0, 0, 0};
    for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i + 3 * stride < pieces; i += 4 * stride) {
        const u32x4 v0 = __builtin_nontemporal_load(w + i), v1 = __builtin_nontemporal_load(w + i + stride);
        const u32x4 v2 = __builtin_nontemporal_load(w + i + 2 * stride), v3 = __builtin_nontemporal_load(w + i + 3 * stride);
        s ^= v0 ^ v1 ^ v2 ^ v3;
    }
    if ((s[0] ^ s[1] ^ s[2] ^ s[3]) == 0x12345678) sink[0] = 1;
}

template <int N>
static void run(const u32x4* w, size_t pieces, unsigned long long* cyc, uint32_t* sink, int grid)
{
    hipStream_t s; CK(hipStreamCreate(&s));
    const int idx = N == 64 ? 0 : N == 256 ? 1 : N == 1024 ? 2 : 3;
    unsigned long long h[4];
    double res[2];
    for (int mode = 0; mode < 2; ++mode) {
        hipGraph_t g; hipGraphExec_t ge;
        const int n = 40;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
        for (int i = 0; i < n; ++i) {
            if (mode == 1) {
                hipLaunchKernelGGL(stream_kernel, dim3(512), dim3(512), 0, s, w + (size_t) i * (pieces + 4096), pieces, sink);
                hipLaunchKernelGGL((code_kernel<N, 1>), dim3(grid), dim3(512), 0, s, cyc + 4, sink);       // other code in between
                hipLaunchKernelGGL(stream_kernel, dim3(512), dim3(512), 0, s, w + (size_t) (i + 40) * (pieces + 4096), pieces, sink);
            }
            hipLaunchKernelGGL((code_kernel<N, 0>), dim3(grid), dim3(512), 0, s, cyc, sink);
        }
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
        CK(hipMemset(cyc, 0, 64));
        const int reps = 5;
        for (int r = 0; r < reps; ++r) CK(hipGraphLaunch(ge, s));
        CK(hipStreamSynchronize(s));
        CK(hipMemcpy(h, cyc, 32, hipMemcpyDeviceToHost));
        res[mode] = (double) h[idx] / ((double) reps * n * grid);
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }
    printf("%5d instructions (%6d bytes), grid %4d: back to back %7.0f cycles per block | between 46 MB streams + other code %7.0f cycles (+%.0f = %.0f per 64-byte line)\n",
           N, N * 4, grid, res[0], res[1], res[1] - res[0], (res[1] - res[0]) / (N * 4 / 64.0));
    CK(hipStreamDestroy(s));
}

int main()
{
    u32x4* w; unsigned long long* cyc; uint32_t* sink;
    const size_t bytes = 46880256, pieces = bytes / 16;
    CK(hipMalloc(&w, (size_t) 1 << 32)); CK(hipMalloc(&cyc, 64)); CK(hipMalloc(&sink, 64));
    CK(hipMemset(w, 1, (size_t) 1 << 32));
    for (int grid : {512, 256}) {
        run<64>(w, pieces, cyc, sink, grid);
        run<256>(w, pieces, cyc, sink, grid);
        run<1024>(w, pieces, cyc, sink, grid);
        run<4096>(w, pieces, cyc, sink, grid);
    }
    return 0;
}
