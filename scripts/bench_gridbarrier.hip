// What does a grid-wide barrier cost on MI355X, next to the ~2.5-3 us of a dependent kernel boundary?
//   hipcc -O3 --offload-arch=gfx950 scripts/bench_gridbarrier.hip -o build/bench_gridbarrier
// All blocks are resident (<= 2 per CU); every barrier = block barrier, one device-scope atomic add by thread 0, spin on the
// counter (bounded: a lost block ends the run instead of hanging the GPU), block barrier.
//   mode 0: relaxed atomics only (no data hand-off)
//   mode 1: + release fence before / acquire fence after (what a real producer -> consumer hand-off needs)
//   mode 2: mode 1 + every block writes 8 KB before the barrier and reads 8 KB written by another block after it
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int MODE>
__global__ __launch_bounds__(512) void k(unsigned* counter, float* data, int rounds, unsigned* fail)
{
    const int tid = threadIdx.x, nb = gridDim.x, b = blockIdx.x;
    float acc = 0.f;
    for (int r = 0; r < rounds; ++r) {
        if (MODE == 2) {
#pragma unroll
            for (int i = 0; i < 4; ++i) data[(size_t) b * 2048 + i * 512 + tid] = (float) (r + i) + acc;
        }
        __syncthreads();
        if (tid == 0) {
            if (MODE >= 1) __atomic_thread_fence(__ATOMIC_RELEASE);       // agent scope by default in HIP
            atomicAdd(counter, 1u);
            const unsigned target = (unsigned) (r + 1) * (unsigned) nb;
            int spins = 0;
            while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > 2000000) { *fail = 1; break; }
            }
            if (MODE >= 1) __atomic_thread_fence(__ATOMIC_ACQUIRE);
        }
        __syncthreads();
        if (MODE == 2) {
            const int src = (b + 37) % nb;
#pragma unroll
            for (int i = 0; i < 4; ++i) acc += __builtin_nontemporal_load(&data[(size_t) src * 2048 + i * 512 + tid]);
        }
    }
    if (acc == 12345.f) data[0] = acc;
}

template <int MODE> void run(const char* name, int blocks, unsigned* counter, float* data, unsigned* fail)
{
    const int rounds = 2000;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipMemset(counter, 0, 4));
    k<MODE><<<blocks, 512>>>(counter, data, 10, fail);
    CK(hipDeviceSynchronize());
    CK(hipMemset(counter, 0, 4));
    CK(hipEventRecord(e0));
    k<MODE><<<blocks, 512>>>(counter, data, rounds, fail);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned f; CK(hipMemcpy(&f, fail, 4, hipMemcpyDeviceToHost));
    printf("%-52s blocks %3d: %7.3f us per barrier%s\n", name, blocks, ms * 1e3 / rounds, f ? "  (SPIN LIMIT HIT)" : "");
}
int main()
{
    unsigned *counter, *fail; float* data;
    CK(hipMalloc(&counter, 256)); CK(hipMalloc(&fail, 4)); CK(hipMemset(fail, 0, 4)); CK(hipMalloc(&data, 512 * 2048 * 4));
    for (int blocks : {256, 512}) {
        run<0>("relaxed atomic + spin", blocks, counter, data, fail);
        run<1>("release / acquire fences", blocks, counter, data, fail);
        run<2>("fences + 8 KB written before, 8 KB read after", blocks, counter, data, fail);
    }
    // for comparison: dependent empty kernels in a stream
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipMemset(counter, 0, 4));
    CK(hipEventRecord(e0));
    for (int i = 0; i < 2000; ++i) k<0><<<512, 512>>>(counter, data, 0, fail);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-52s blocks 512: %7.3f us per launch\n", "empty dependent kernels in a stream", ms * 1e3 / 2000);
    return 0;
}
