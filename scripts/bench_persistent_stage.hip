// Would ONE persistent kernel per pair of dependent decode stages beat two dependent launches?
//   hipcc -O3 --offload-arch=gfx950 scripts/bench_persistent_stage.hip -o build/bench_persistent_stage
// A "stage" is the skeleton of a decode GEMV launch: every block first needs the whole 8 KB activation vector the PREVIOUS
// stage produced (32 bytes per block), then streams its 96 KB share of a 48 MB weight matrix (8 waves x 3 passes x 4
// loads of 16 bytes per lane, next pass in flight while the current one is reduced) and writes its 32 bytes.
//   two launches : stage kernel, stage kernel, ... (dependent launches in one stream, as the product does)
//   persistent   : one kernel runs S stages; between stages a fence-free hierarchical grid barrier (per-XCD counters) and a
//                  cache-bypassing hand-off of the vector (agent-scope relaxed atomics); the FIRST TWO weight batches of the
//                  next stage are requested BEFORE the barrier, so the weights stream while the blocks wait for each other
// The weights rotate over 8 matrices (384 MB, beyond the 256 MB cache).  Results are checked between the two variants.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
#define NB 512
#define NW 8
#define U 4
#define NP 3
#define NMAT 8
static const size_t MAT_U4 = (size_t) NB * NW * NP * U * 64;       // uint4 per matrix (48 MiB)

__device__ __forceinline__ float lane_sum(const uint4& v) { return (float) ((v.x ^ v.y) & 0xFF) + (float) ((v.z ^ v.w) & 0xFF) * 0.5f; }

// one stage for one block; vec_in read by `LD`, result written by `ST`
template <bool BYPASS, bool have_pre>
__device__ __forceinline__ void stage_body(const uint4* __restrict__ w, const float* vec_in, float* vec_out, float* xs, float* red,
                                           uint4 (&pre)[2][U])
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, b = blockIdx.x;
    const uint4* wp = w + ((size_t) (b * NW + wave) * NP * U) * 64 + lane;
    uint4 buf[2][U];
    if (have_pre) {
#pragma unroll
        for (int u = 0; u < U; ++u) { buf[0][u] = pre[0][u]; buf[1][u] = pre[1][u]; }
    } else {
#pragma unroll
        for (int u = 0; u < U; ++u) buf[0][u] = wp[u * 64];
    }
    // the activation vector: 2048 floats, 4 per thread
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int idx = i * 512 + tid;
        xs[idx] = BYPASS ? __hip_atomic_load(&vec_in[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : vec_in[idx];
    }
    __syncthreads();
    float acc = 0.f;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        if (p + 1 < NP && !(have_pre && p == 0)) {
#pragma unroll
            for (int u = 0; u < U; ++u) buf[(p + 1) & 1][u] = wp[((p + 1) * U + u) * 64];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) acc = fmaf(lane_sum(buf[p & 1][u]), xs[(p * U + u) * 64 + lane + wave * 3], acc);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if (lane == 0) red[wave] = acc;
    __syncthreads();
    if (tid < 4) {                                                       // 4 floats per block = 2048 in all
        float v = 0.f;
#pragma unroll
        for (int k = 0; k < NW; ++k) v += red[k];
        v = v * 1e-6f + (float) tid;
        if (BYPASS) __hip_atomic_store(&vec_out[b * 4 + tid], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else vec_out[b * 4 + tid] = v;
    }
}

__global__ __launch_bounds__(512) void stage_kernel(const uint4* __restrict__ w, const float* vec_in, float* vec_out)
{
    __shared__ float xs[2048 + 64];
    __shared__ float red[NW];
    uint4 pre[2][U];
    stage_body<false, false>(w, vec_in, vec_out, xs, red, pre);
}

template <bool PREFETCH>
__global__ __launch_bounds__(512) void persistent_kernel(const uint4* __restrict__ w0, float* vecs, int stages, unsigned* ctr, unsigned* fail)
{
    __shared__ float xs[2048 + 64];
    __shared__ float red[NW];
    __shared__ unsigned quit;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, b = blockIdx.x, x = b & 7;
    unsigned* cx = ctr; unsigned* cg = ctr + 256; unsigned* flag = ctr + 512;
    uint4 pre[2][U];
    if (PREFETCH) {                                                      // stage 0's first two batches
        const uint4* wn = w0 + ((size_t) (b * NW + wave) * NP * U) * 64 + lane;
#pragma unroll
        for (int u = 0; u < U; ++u) { pre[0][u] = wn[u * 64]; pre[1][u] = wn[(U + u) * 64]; }
    }
    for (int s = 0; s < stages; ++s) {
        const uint4* w = w0 + (size_t) (s % NMAT) * MAT_U4;
        stage_body<true, PREFETCH>(w, vecs + (size_t) (s & 1) * 2048, vecs + (size_t) ((s + 1) & 1) * 2048, xs, red, pre);
        if (s + 1 == stages) break;
        if (tid < 4) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this block's 16 bytes are out (all of them come from wave 0)
        if (PREFETCH) {                                                  // next stage's first two batches: they do not depend on the vector
            const uint4* wn = w0 + (size_t) ((s + 1) % NMAT) * MAT_U4 + ((size_t) (b * NW + wave) * NP * U) * 64 + lane;
#pragma unroll
            for (int u = 0; u < U; ++u) { pre[0][u] = wn[u * 64]; pre[1][u] = wn[(U + u) * 64]; }
        }
        // ---- grid barrier by thread 0, then a RAW block barrier (hipcc's __syncthreads would wait for the prefetched loads) ----
        if (tid == 0) {
            const unsigned r1 = (unsigned) (s + 1);
            const unsigned old = __hip_atomic_fetch_add(&cx[x * 32], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (old == (NB / 8) * r1 - 1u) {
                const unsigned old2 = __hip_atomic_fetch_add(&cg[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (old2 == 8u * r1 - 1u)
                    for (int i = 0; i < 8; ++i) __hip_atomic_store(&flag[i * 32], r1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            int spins = 0;
            unsigned q = 0;
            while (__hip_atomic_load(&flag[x * 32], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < r1) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > 200000 || ((spins & 1023) == 0 && __hip_atomic_load(fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
                    __hip_atomic_store(fail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    q = 1;
                    break;
                }
            }
            quit = q;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (quit) break;                                                 // block-uniform
    }
}

__global__ void fill(uint4* p, size_t n)
{
    for (size_t i = blockIdx.x * (size_t) blockDim.x + threadIdx.x; i < n; i += (size_t) gridDim.x * blockDim.x) {
        unsigned v = (unsigned) i * 2654435761u; v ^= v >> 13;
        p[i] = make_uint4(v, v * 3u, v * 7u, v * 11u);
    }
}

int main()
{
    uint4* w; float* vecs; unsigned *ctr, *fail;
    CK(hipMalloc(&w, NMAT * MAT_U4 * 16)); CK(hipMalloc(&vecs, 2 * 2048 * 4)); CK(hipMalloc(&ctr, 8192)); CK(hipMalloc(&fail, 4));
    fill<<<2048, 256>>>(w, NMAT * MAT_U4);
    CK(hipDeviceSynchronize());
    const int stages = 320;                                              // 64 "layers" x 5
    std::vector<float> init(4096, 1.0f), r0(2048), r1(2048);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms;
    // --- dependent launches
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipMemcpy(vecs, init.data(), 16384, hipMemcpyHostToDevice));
        CK(hipEventRecord(e0));
        for (int s = 0; s < stages; ++s)
            stage_kernel<<<NB, 512>>>(w + (size_t) (s % NMAT) * MAT_U4, vecs + (size_t) (s & 1) * 2048, vecs + (size_t) ((s + 1) & 1) * 2048);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
    }
    CK(hipMemcpy(r0.data(), vecs + (size_t) (stages & 1) * 2048, 8192, hipMemcpyDeviceToHost));
    printf("dependent launches      : %7.3f us per stage (48 MiB each -> %.2f TB/s)\n", ms * 1e3 / stages, 48.0 * 1.048576 / (ms * 1e3 / stages));
    // --- persistent
    for (int prefetch = 0; prefetch < 2; ++prefetch) {
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipMemcpy(vecs, init.data(), 16384, hipMemcpyHostToDevice));
            CK(hipMemset(ctr, 0, 8192)); CK(hipMemset(fail, 0, 4));
            CK(hipEventRecord(e0));
            if (prefetch) persistent_kernel<true><<<NB, 512>>>(w, vecs, stages, ctr, fail);
            else          persistent_kernel<false><<<NB, 512>>>(w, vecs, stages, ctr, fail);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1));
        }
        unsigned f; CK(hipMemcpy(&f, fail, 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(r1.data(), vecs + (size_t) (stages & 1) * 2048, 8192, hipMemcpyDeviceToHost));
        int diff = 0;
        for (int i = 0; i < 2048; ++i) diff += r0[i] != r1[i];
        printf("persistent, prefetch %d  : %7.3f us per stage (%.2f TB/s)%s  results differing from the launches: %d of 2048\n", prefetch,
               ms * 1e3 / stages, 48.0 * 1.048576 / (ms * 1e3 / stages), f ? "  (BARRIER LOST)" : "", diff);
    }
    return 0;
}
