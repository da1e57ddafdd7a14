// What does a grid-wide barrier cost on MI355X, next to the ~2.5-3 us of a dependent kernel boundary?
//   hipcc -O3 --offload-arch=gfx950 scripts/bench_gridbarrier.hip -o build/bench_gridbarrier
// All blocks are resident (<= 2 per CU); every barrier = block barrier, one device-scope atomic add by thread 0, spin on the
// counter (bounded: a lost block ends the run instead of hanging the GPU), block barrier.
//   mode 0: relaxed atomics only (no data hand-off)
//   mode 1: + release fence before / acquire fence after (what a real producer -> consumer hand-off needs)
//   mode 2: mode 1 + every block writes 8 KB before the barrier and reads 8 KB written by another block after it
//   mode 3: hierarchical barrier, no fences: one counter per XCD (block b is assumed to run on XCD b % 8: speed only), the
//           last arriver of an XCD bumps a global counter, the last of those releases 8 per-XCD flags the blocks spin on
//   mode 4: mode 3 + an 8 KB hand-off per block that BYPASSES the caches instead of fencing them: agent-scope relaxed atomic
//           stores before the barrier (s_waitcnt vmcnt(0) in front of the arrive), agent-scope relaxed atomic loads after
//           it, double-buffered, every value checked (a stale read is counted)
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

template <int MODE>
__global__ __launch_bounds__(512) void kh(unsigned* cx, unsigned* cg, unsigned* flag, float* data, int rounds, unsigned* fail, unsigned* stale)
{
    const int tid = threadIdx.x, nb = gridDim.x, b = blockIdx.x, x = b & 7;
    const unsigned per_xcd = (unsigned) nb / 8u;
    unsigned bad = 0;
    __shared__ unsigned quit;
    for (int r = 0; r < rounds; ++r) {
        if (tid == 0) quit = __hip_atomic_load(fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        if (quit) break;                                                 // block-uniform: a lost barrier ends the run everywhere
        float* buf = data + (size_t) (r & 1) * 512 * 2048;
        if (MODE == 4) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                __hip_atomic_store(&buf[(size_t) b * 2048 + i * 512 + tid], (float) (b * 7 + r + i), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // this thread's stores are acknowledged
        }
        __syncthreads();
        if (tid == 0) {
            const unsigned old = __hip_atomic_fetch_add(&cx[x * 32], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (old == per_xcd * (unsigned) (r + 1) - 1u) {
                const unsigned old2 = __hip_atomic_fetch_add(&cg[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (old2 == 8u * (unsigned) (r + 1) - 1u)
                    for (int i = 0; i < 8; ++i) __hip_atomic_store(&flag[i * 32], (unsigned) (r + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            int spins = 0;
            while (__hip_atomic_load(&flag[x * 32], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned) (r + 1)) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > 200000 || ((spins & 1023) == 0 && __hip_atomic_load(fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
                    __hip_atomic_store(fail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
            }
        }
        __syncthreads();
        if (MODE == 4) {
            const int src = (b + 37) % nb;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float v = __hip_atomic_load(&buf[(size_t) src * 2048 + i * 512 + tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                bad += v != (float) (src * 7 + r + i);
            }
        }
    }
    if (bad) atomicAdd(stale, bad);
}

template <int MODE> void runh(const char* name, int blocks, unsigned* ctr, float* data, unsigned* fail)
{
    const int rounds = 2000;
    unsigned* stale = ctr + 1024;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipMemset(ctr, 0, 8192));
    kh<MODE><<<blocks, 512>>>(ctr, ctr + 256, ctr + 512, data, 10, fail, stale);
    CK(hipDeviceSynchronize());
    CK(hipMemset(ctr, 0, 8192));
    CK(hipEventRecord(e0));
    kh<MODE><<<blocks, 512>>>(ctr, ctr + 256, ctr + 512, data, rounds, fail, stale);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned f, st; CK(hipMemcpy(&f, fail, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&st, stale, 4, hipMemcpyDeviceToHost));
    printf("%-52s blocks %3d: %7.3f us per barrier%s  stale reads: %u\n", name, blocks, ms * 1e3 / rounds, f ? "  (SPIN LIMIT HIT)" : "", st);
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
int main(int argc, char** argv)
{
    unsigned *counter, *fail; float* data;
    CK(hipMalloc(&counter, 8192)); CK(hipMalloc(&fail, 4)); CK(hipMemset(fail, 0, 4)); CK(hipMalloc(&data, 2 * 512 * 2048 * 4));
    const bool quick = argc > 1;                                      // any argument: only the hierarchical modes
    for (int blocks : {256, 512}) {
        if (!quick) {
            run<0>("relaxed atomic + spin", blocks, counter, data, fail);
            run<1>("release / acquire fences", blocks, counter, data, fail);
            run<2>("fences + 8 KB written before, 8 KB read after", blocks, counter, data, fail);
        }
        runh<3>("per-XCD counters, no fences", blocks, counter, data, fail);
        runh<4>("per-XCD counters + 8 KB cache-bypassing hand-off", blocks, counter, data, fail);
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
