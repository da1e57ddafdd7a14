// Which decomposition streams a decode-GEMV-like stage fastest as a chain of dependent launches?  (skeleton of
// scripts/bench_persistent_stage.hip: every block first needs the 8 KB vector the previous stage produced, then streams its
// share of a 48 MiB matrix with U loads of 16 bytes per lane in flight per wave and NP passes, then writes its outputs.)
//   hipcc -O3 --offload-arch=gfx950 scripts/bench_stage_sweep.hip -o build/bench_stage_sweep
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
#define NMAT 8
static const size_t MAT_U4 = (size_t) 48 * 1024 * 1024 / 16;

__device__ __forceinline__ float lane_sum(const uint4& v) { return (float) ((v.x ^ v.y) & 0xFF) + (float) ((v.z ^ v.w) & 0xFF) * 0.5f; }

template <int NW, int U, int NP>
__global__ __launch_bounds__(NW * 64) void stage_kernel(const uint4* __restrict__ w, const float* __restrict__ vec_in, float* __restrict__ vec_out, int nout)
{
    __shared__ float xs[2048 + 1024];
    __shared__ float red[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, b = blockIdx.x;
    const uint4* wp = w + ((size_t) (b * NW + wave) * NP * U) * 64 + lane;
    uint4 buf[2][U];
#pragma unroll
    for (int u = 0; u < U; ++u) buf[0][u] = wp[u * 64];
    for (int idx = tid; idx < 2048; idx += NW * 64) xs[idx] = vec_in[idx];
    __syncthreads();
    float acc = 0.f;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        if (p + 1 < NP) {
#pragma unroll
            for (int u = 0; u < U; ++u) buf[(p + 1) & 1][u] = wp[((p + 1) * U + u) * 64];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) acc = fmaf(lane_sum(buf[p & 1][u]), xs[((p * U + u) & 15) * 64 + lane + wave * 3], acc);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if (lane == 0) red[wave] = acc;
    __syncthreads();
    if (tid < nout) {
        float v = 0.f;
        for (int k = 0; k < NW; ++k) v += red[k];
        vec_out[b * nout + tid] = v * 1e-6f + (float) tid;
    }
}

__global__ void fill(uint4* p, size_t n)
{
    for (size_t i = blockIdx.x * (size_t) blockDim.x + threadIdx.x; i < n; i += (size_t) gridDim.x * blockDim.x) {
        unsigned v = (unsigned) i * 2654435761u; v ^= v >> 13;
        p[i] = make_uint4(v, v * 3u, v * 7u, v * 11u);
    }
}

template <int NW, int U, int NP> void run(const uint4* w, float* vecs)
{
    const int nb = (int) (MAT_U4 / ((size_t) NW * U * NP * 64));
    const int nout = 2048 / nb > 0 ? 2048 / nb : 1;
    const int stages = 320;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(e0));
        for (int s = 0; s < stages; ++s)
            stage_kernel<NW, U, NP><<<nb, NW * 64>>>(w + (size_t) (s % NMAT) * MAT_U4, vecs + (size_t) (s & 1) * 4096, vecs + (size_t) ((s + 1) & 1) * 4096, nout);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
    }
    printf("waves %2d  U %2d  passes %d  blocks %4d : %7.3f us per 48 MiB stage = %.2f TB/s\n", NW, U, NP, nb, ms * 1e3 / stages, 48.0 * 1.048576 / (ms * 1e3 / stages));
}

int main()
{
    uint4* w; float* vecs;
    CK(hipMalloc(&w, NMAT * MAT_U4 * 16)); CK(hipMalloc(&vecs, 2 * 4096 * 4));
    fill<<<2048, 256>>>(w, NMAT * MAT_U4);
    CK(hipMemset(vecs, 0, 2 * 4096 * 4));
    CK(hipDeviceSynchronize());
    run<8, 4, 3>(w, vecs);      // the product's shape for gate_up-sized launches (2 blocks per CU)
    run<8, 6, 2>(w, vecs);
    run<8, 12, 1>(w, vecs);
    run<8, 3, 4>(w, vecs);
    run<8, 2, 6>(w, vecs);
    run<4, 4, 3>(w, vecs);      // 1024 blocks of 4 waves
    run<16, 4, 3>(w, vecs);     // 256 blocks of 16 waves
    run<8, 4, 6>(w, vecs);      // 256 blocks: one per CU
    run<8, 6, 1>(w, vecs);      // 1024 one-shot blocks
    run<8, 3, 1>(w, vecs);      // 2048 one-shot blocks
    run<4, 8, 3>(w, vecs);      // 512 blocks of 4 fat waves
    run<8, 8, 3>(w, vecs);      // 256 blocks, 8 loads in flight
    return 0;
}
