// Stand-alone micro-benchmark: which access pattern streams a packed [R][N] uint32 weight matrix fastest on MI355X?
// hipcc --offload-arch=gfx950 -O3 scripts/bench_gemv.hip -o /tmp/bench_gemv && /tmp/bench_gemv
// Not part of the product; used to choose the decode GEMV decomposition (results recorded in DESIGN.md).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <bool NT>
__device__ __forceinline__ uint4 ld16(const uint4* p)
{
    if (NT) { const u32x4 v = __builtin_nontemporal_load((const u32x4*) p); return make_uint4(v[0], v[1], v[2], v[3]); }
    return *p;
}

__device__ __forceinline__ float dot8(uint32_t w, float acc)
{
    // same instruction mix as the real kernel (4 and_or, 1 shift, 4 pk ops, 4 dot2)
    const f16x2 s = {(_Float16) 0.0625f, (_Float16) 0.0625f};
    const f16x2 z = {(_Float16) -1032.f, (_Float16) -1032.f};
    const f16x2 z1 = {(_Float16) -72.f, (_Float16) -72.f};
    const f16x2 x = {(_Float16) 0.5f, (_Float16) -0.25f};
    const uint32_t w8 = w >> 8;
    const f16x2 d0 = __builtin_bit_cast(f16x2, (w & 0x000F000Fu) | 0x64006400u) + z;
    const f16x2 d1 = __builtin_bit_cast(f16x2, (w & 0x00F000F0u) | 0x64006400u) * s + z1;
    const f16x2 d2 = __builtin_bit_cast(f16x2, (w8 & 0x000F000Fu) | 0x64006400u) + z;
    const f16x2 d3 = __builtin_bit_cast(f16x2, (w8 & 0x00F000F0u) | 0x64006400u) * s + z1;
    acc = __builtin_amdgcn_fdot2(d0, x, acc, false);
    acc = __builtin_amdgcn_fdot2(d1, x, acc, false);
    acc = __builtin_amdgcn_fdot2(d2, x, acc, false);
    acc = __builtin_amdgcn_fdot2(d3, x, acc, false);
    return acc;
}

// Pattern P: TX lanes side by side (TX*16 bytes contiguous), TY = 256/TX row slices, RPT rows per thread.
// CONTIG = true : thread ty owns rows [ty*RPT, +RPT)      (what the product kernel does)
// CONTIG = false: thread ty owns rows ty, ty+TY, ty+2TY.. (adjacent slices touch adjacent rows)
template <int TX, int RPT, bool CONTIG, bool NT>
__global__ __launch_bounds__(256) void gemv_pat(const uint4* __restrict__ w, float* __restrict__ out, int R, int N)
{
    constexpr int TY = 256 / TX;
    const int tid = threadIdx.x, tx = tid % TX, ty = tid / TX;
    const int n4 = N >> 2;
    const int col4 = blockIdx.x * TX + tx;
    const int r0 = blockIdx.y * TY * RPT;
    uint4 v[RPT];
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
        const int r = r0 + (CONTIG ? ty * RPT + i : ty + i * TY);
        v[i] = (r < R && col4 < n4) ? ld16<NT>(w + (size_t) r * n4 + col4) : make_uint4(0, 0, 0, 0);
    }
    float a0 = 0, a1 = 0, a2 = 0, a3 = 0;
#pragma unroll
    for (int i = 0; i < RPT; ++i) { a0 = dot8(v[i].x, a0); a1 = dot8(v[i].y, a1); a2 = dot8(v[i].z, a2); a3 = dot8(v[i].w, a3); }
    // cheap stand-in for the cross-slice reduction
    float s = a0 + a1 + a2 + a3;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    if ((tid & 63) == 0) out[(blockIdx.y * gridDim.x + blockIdx.x) * 4 + (tid >> 6)] = s;
}

// Linear streaming: the whole matrix as a flat array, grid-stride, UN loads in flight per thread.
template <int UN, bool NT>
__global__ __launch_bounds__(256) void stream_flat(const uint4* __restrict__ w, float* __restrict__ out, size_t n16)
{
    const size_t base = (size_t) blockIdx.x * 256 * UN + threadIdx.x;
    uint4 v[UN];
#pragma unroll
    for (int i = 0; i < UN; ++i) { const size_t k = base + (size_t) i * 256; v[i] = k < n16 ? ld16<NT>(w + k) : make_uint4(0, 0, 0, 0); }
    float a = 0;
#pragma unroll
    for (int i = 0; i < UN; ++i) { a = dot8(v[i].x, a); a = dot8(v[i].y, a); a = dot8(v[i].z, a); a = dot8(v[i].w, a); }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) a += __shfl_xor(a, off, 64);
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = a;
}

// Realistic variant: x staged global -> LDS (+barrier), per-group zero/scale loads, 4-wave LDS reduction, slab store.
template <int TX, int RPT>
__global__ __launch_bounds__(256) void gemv_real(const uint4* __restrict__ w, const uint32_t* __restrict__ qz,
                                                 const _Float16* __restrict__ sc, const uint4* __restrict__ x,
                                                 float* __restrict__ out, int R, int N)
{
    constexpr int TY = 256 / TX;
    constexpr int BN = 4 * TX;
    __shared__ uint4 xs[TY * RPT];
    __shared__ float red[4 * BN];
    const int tid = threadIdx.x, tx = tid % TX, ty = tid / TX;
    const int n4 = N >> 2;
    const int col4 = blockIdx.x * TX + tx;
    const int col = col4 * 4;
    const int r0 = blockIdx.y * TY * RPT;
    const int nrows = min(R - r0, TY * RPT);
    // prologue loads first, then the weight stream
    uint4 xr[(TY * RPT + 255) / 256];
#pragma unroll
    for (int i = 0; i < (TY * RPT + 255) / 256; ++i) { const int idx = tid + i * 256; xr[i] = x[r0 + (idx < nrows ? idx : 0)]; }
    uint4 v[RPT];
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
        const int r = r0 + ty * RPT + i;
        v[i] = (r < R && col4 < n4) ? ld16<true>(w + (size_t) r * n4 + col4) : make_uint4(0, 0, 0, 0);
    }
    const int g = (r0 + ty * RPT) / 16;
    const uint32_t zw = qz[(size_t) g * (N >> 3) + (col >> 3)];
    const uint2 s2 = *(const uint2*) (sc + (size_t) g * N + col);
#pragma unroll
    for (int i = 0; i < (TY * RPT + 255) / 256; ++i) { const int idx = tid + i * 256; if (idx < nrows) xs[idx] = xr[i]; }
    __syncthreads();
    float a[4] = {0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
        const uint4 x4 = xs[ty * RPT + i];
        const uint32_t ww[4] = {v[i].x, v[i].y, v[i].z, v[i].w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int z = (int) ((zw >> (((col & 7) + j) * 4)) & 0xF) + 1;
            const _Float16 za = (_Float16) (float) (-(1024 + z)), zb = (_Float16) (float) (-(64 + z));
            const f16x2 zc0 = {za, za}, zc1 = {zb, zb};
            const f16x2 s = {(_Float16) 0.0625f, (_Float16) 0.0625f};
            const uint32_t wq = ww[j], w8 = wq >> 8;
            const f16x2 d0 = __builtin_bit_cast(f16x2, (wq & 0x000F000Fu) | 0x64006400u) + zc0;
            const f16x2 d1 = __builtin_bit_cast(f16x2, (wq & 0x00F000F0u) | 0x64006400u) * s + zc1;
            const f16x2 d2 = __builtin_bit_cast(f16x2, (w8 & 0x000F000Fu) | 0x64006400u) + zc0;
            const f16x2 d3 = __builtin_bit_cast(f16x2, (w8 & 0x00F000F0u) | 0x64006400u) * s + zc1;
            a[j] = __builtin_amdgcn_fdot2(d0, __builtin_bit_cast(f16x2, x4.x), a[j], false);
            a[j] = __builtin_amdgcn_fdot2(d1, __builtin_bit_cast(f16x2, x4.y), a[j], false);
            a[j] = __builtin_amdgcn_fdot2(d2, __builtin_bit_cast(f16x2, x4.z), a[j], false);
            a[j] = __builtin_amdgcn_fdot2(d3, __builtin_bit_cast(f16x2, x4.w), a[j], false);
        }
    }
    const _Float16* sp = (const _Float16*) &s2;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float t = a[j] * (float) sp[j];
#pragma unroll
        for (int off = TX; off < 64; off <<= 1) t += __shfl_xor(t, off, 64);
        a[j] = t;
    }
    const int wave = tid >> 6, lane = tid & 63;
    if (lane < TX) {
#pragma unroll
        for (int j = 0; j < 4; ++j) red[wave * BN + lane * 4 + j] = a[j];
    }
    __syncthreads();
    if (tid < BN) out[(size_t) blockIdx.y * N + blockIdx.x * BN + tid] = red[tid] + red[BN + tid] + red[2 * BN + tid] + red[3 * BN + tid];
}

struct Shape { const char* name; int K, N; };

int main()
{
    const Shape shapes[] = {{"qkv 4096x12288", 4096, 12288}, {"o 4096x4096", 4096, 4096}, {"gate_up 4096x22016", 4096, 22016},
                            {"down 11008x4096", 11008, 4096}};
    const int NBUF = 24;                                     // rotate buffers: total >> 256 MB Infinity Cache
    float* out;
    CHECK(hipMalloc(&out, 1 << 22));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (const Shape& sh : shapes) {
        const int R = sh.K / 8, N = sh.N;
        const size_t bytes = (size_t) R * N * 4;
        std::vector<uint4*> bufs(NBUF);
        for (auto& b : bufs) { CHECK(hipMalloc(&b, bytes)); CHECK(hipMemset(b, 0x5a, bytes)); }
        printf("== %s : %.1f MB per launch ==\n", sh.name, bytes / 1e6);
        auto timeit = [&](const char* label, auto launch) {
            for (int i = 0; i < NBUF; ++i) launch(bufs[i]);
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0));
            const int reps = 3 * NBUF;
            for (int i = 0; i < reps; ++i) launch(bufs[i % NBUF]);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms = 0;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            const double us = ms * 1e3 / reps;
            printf("  %-46s %8.2f us  %7.1f GB/s\n", label, us, bytes / us / 1e3);
        };
#define PAT(TX, RPT, CONTIG, NT) timeit("pattern TX=" #TX " RPT=" #RPT " contig=" #CONTIG " nt=" #NT, [&](uint4* b) { \
            constexpr int TY = 256 / TX; dim3 grid((N / 4 + TX - 1) / TX, (R + TY * RPT - 1) / (TY * RPT)); \
            hipLaunchKernelGGL((gemv_pat<TX, RPT, CONTIG, NT>), grid, dim3(256), 0, 0, b, out, R, N); })
        PAT(4, 8, true, true);
        PAT(4, 4, true, true);
        PAT(8, 16, true, true);
        PAT(8, 16, true, false);
        PAT(8, 16, false, true);
        PAT(8, 8, true, true);
        PAT(8, 4, true, true);
        PAT(16, 16, true, true);
        PAT(16, 8, true, true);
        PAT(16, 8, false, true);
        PAT(32, 8, true, true);
        PAT(32, 8, false, true);
        PAT(64, 8, true, true);
        PAT(64, 8, false, true);
        PAT(64, 4, false, true);
        PAT(64, 16, false, true);
        {
            uint32_t* qz; _Float16* scp; uint4* xp;
            CHECK(hipMalloc(&qz, (size_t) (sh.K / 128) * (N / 8) * 4)); CHECK(hipMemset(qz, 0x77, (size_t) (sh.K / 128) * (N / 8) * 4));
            CHECK(hipMalloc(&scp, (size_t) (sh.K / 128) * N * 2)); CHECK(hipMemset(scp, 0x11, (size_t) (sh.K / 128) * N * 2));
            CHECK(hipMalloc(&xp, (size_t) R * 16)); CHECK(hipMemset(xp, 0x31, (size_t) R * 16));
#define REAL(TX, RPT) timeit("real gemv TX=" #TX " RPT=" #RPT, [&](uint4* b) { \
            constexpr int TY = 256 / TX; dim3 grid((N / 4 + TX - 1) / TX, (R + TY * RPT - 1) / (TY * RPT)); \
            hipLaunchKernelGGL((gemv_real<TX, RPT>), grid, dim3(256), 0, 0, b, qz, scp, xp, out, R, N); })
            REAL(8, 16); REAL(8, 8); REAL(8, 4); REAL(8, 2); REAL(4, 8); REAL(4, 4); REAL(16, 4); REAL(16, 2);
            CHECK(hipFree(qz)); CHECK(hipFree(scp)); CHECK(hipFree(xp));
        }
#define FLAT(UN, NT) timeit("flat stream UN=" #UN " nt=" #NT, [&](uint4* b) { \
            const size_t n16 = bytes / 16; dim3 grid((unsigned) ((n16 + 256 * UN - 1) / (256 * UN))); \
            hipLaunchKernelGGL((stream_flat<UN, NT>), grid, dim3(256), 0, 0, b, out, n16); })
        FLAT(4, true);
        FLAT(8, true);
        FLAT(8, false);
        FLAT(16, true);
        for (auto& b : bufs) CHECK(hipFree(b));
    }
    return 0;
}
