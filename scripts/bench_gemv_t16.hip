// Micro-benchmark, layout v2: 16-column tiles, 16-byte piece = 4 consecutive packed rows (32 k) of ONE column.
//   word(n, r) -> tile t = n/16, col = n%16, row-block rb = r/16, rsub = (r%16)/4, j = r%4
//   uint4 index = (t * RB + rb) * 64 + rsub * 16 + col        (RB = R/16), word j of that uint4
// A wave instruction = lanes (col = l&15, rsub = l>>4) = 1 KiB contiguous = 16 packed rows x 16 columns.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/bench_gemv_t16.hip -o build/bench_gemv_t16
#include "../exllama_amd/csrc/common.h"
#include <vector>
#include <algorithm>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
#define MAGIC 0x64006400u

__device__ __forceinline__ f16x2 h2(uint32_t v) { return __builtin_bit_cast(f16x2, v); }

__device__ __forceinline__ uint4 permute8(uint4 d)
{
    uint4 o;
    o.x = (d.x & 0xFFFFu) | (d.z << 16);
    o.y = (d.x >> 16) | (d.z & 0xFFFF0000u);
    o.z = (d.y & 0xFFFFu) | (d.w << 16);
    o.w = (d.y >> 16) | (d.w & 0xFFFF0000u);
    return o;
}

__device__ __forceinline__ float dot8_exact(uint32_t w, const uint4& x4, f16x2 zc0, f16x2 zc1, float acc)
{
    const f16x2 sixteenth = {(f16) 0.0625f, (f16) 0.0625f};
    const uint32_t w8 = w >> 8;
    const f16x2 d0 = h2((w & 0x000F000Fu) | MAGIC) + zc0;
    const f16x2 d1 = h2((w & 0x00F000F0u) | MAGIC) * sixteenth + zc1;
    const f16x2 d2 = h2((w8 & 0x000F000Fu) | MAGIC) + zc0;
    const f16x2 d3 = h2((w8 & 0x00F000F0u) | MAGIC) * sixteenth + zc1;
    acc = __builtin_amdgcn_fdot2(d0, h2(x4.x), acc, false);
    acc = __builtin_amdgcn_fdot2(d1, h2(x4.y), acc, false);
    acc = __builtin_amdgcn_fdot2(d2, h2(x4.z), acc, false);
    acc = __builtin_amdgcn_fdot2(d3, h2(x4.w), acc, false);
    return acc;
}

// raw magic-number values against pre-scaled x (hi-nibble positions carry x/16); corrected afterwards
__device__ __forceinline__ float dot8_raw(uint32_t w, const uint4& x4, float acc)
{
    const uint32_t w8 = w >> 8;
    acc = __builtin_amdgcn_fdot2(h2((w & 0x000F000Fu) | MAGIC), h2(x4.x), acc, false);
    acc = __builtin_amdgcn_fdot2(h2((w & 0x00F000F0u) | MAGIC), h2(x4.y), acc, false);
    acc = __builtin_amdgcn_fdot2(h2((w8 & 0x000F000Fu) | MAGIC), h2(x4.z), acc, false);
    acc = __builtin_amdgcn_fdot2(h2((w8 & 0x00F000F0u) | MAGIC), h2(x4.w), acc, false);
    return acc;
}

// 8 exact fp16 weights (q - z) of one word, in the order (q0,q4,q1,q5,q2,q6,q3,q7)
__device__ __forceinline__ f16x8 dequant8(uint32_t w, f16x2 zc0, f16x2 zc1)
{
    const f16x2 sixteenth = {(f16) 0.0625f, (f16) 0.0625f};
    const uint32_t w8 = w >> 8;
    const f16x2 d0 = h2((w & 0x000F000Fu) | MAGIC) + zc0;
    const f16x2 d1 = h2((w & 0x00F000F0u) | MAGIC) * sixteenth + zc1;
    const f16x2 d2 = h2((w8 & 0x000F000Fu) | MAGIC) + zc0;
    const f16x2 d3 = h2((w8 & 0x00F000F0u) | MAGIC) * sixteenth + zc1;
    const uint4 u = make_uint4(__builtin_bit_cast(uint32_t, d0), __builtin_bit_cast(uint32_t, d1),
                               __builtin_bit_cast(uint32_t, d2), __builtin_bit_cast(uint32_t, d3));
    return __builtin_bit_cast(f16x8, u);
}

struct Mat { const uint4* qw; const uint32_t* qzeros; const f16* scales; int K, N, R, RB, gshift, G; unsigned long long* tl; };

template <int U, int WAVES, int MODE, bool XCD>   // MODE 0 exact, 1 raw+correction, 2 loads only
__global__ __launch_bounds__(WAVES * 64) void gemv16_kernel(const Mat m, const f16* __restrict__ x, f16* __restrict__ out,
                                                           int rb_per_wave)
{
    constexpr int NTH = WAVES * 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint4* xs = (uint4*) smem;                                           // [R]
    float2* cs = (float2*) (smem + (size_t) m.R * 16);                   // [R/4] (C4, S4)
    uint32_t* tab = (uint32_t*) (smem + (size_t) m.R * 16 + (size_t) m.R * 2);   // [G][16]
    float* red = (float*) (smem + (size_t) m.R * 18 + (size_t) m.G * 64);        // [WAVES][16]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    unsigned long long T0 = 0, T1 = 0, T2 = 0, T3 = 0;
    if (m.tl) T0 = wall_clock64();
    int t = blockIdx.x;
    if (XCD) { const int per = gridDim.x >> 3; t = (t & 7) * per + (t >> 3); }
    const int col = lane & 15, rsub = lane >> 4;
    const int rb0 = wave * rb_per_wave;
    const int rb1 = min(m.RB, rb0 + rb_per_wave);

    // ---- prologue loads first (x, scales, zeros), then the weight stream ----
    constexpr int XV = (1376 + NTH - 1) / NTH;                                                // up to 3 * NTH 8-half vectors (K <= 24 * NTH)
    uint4 xr[XV];
#pragma unroll
    for (int i = 0; i < XV; ++i) { const int idx = tid + i * NTH; xr[i] = *(const uint4*) (x + (idx < m.R ? idx : 0) * 8); }
    constexpr int TV = (86 * 16 + NTH - 1) / NTH;
    uint32_t traw[TV];
#pragma unroll
    for (int i = 0; i < TV; ++i) {
        const int idx = tid + i * NTH;
        const int g = min(idx >> 4, m.G - 1), c = idx & 15;
        const uint32_t zw = m.qzeros[(size_t) g * (m.N >> 3) + t * 2 + (c >> 3)];
        const uint16_t sb = ((const uint16_t*) m.scales)[(size_t) g * m.N + t * 16 + c];
        traw[i] = (uint32_t) sb | ((((zw >> ((c & 7) * 4)) & 0xFu) + 1) << 16);
    }
    const uint4* base = m.qw + (size_t) t * m.RB * 64 + lane;
    uint4 wv[U];
#pragma unroll
    for (int i = 0; i < U; ++i) { const int rb = rb0 + i; wv[i] = nt_load16(base + (size_t) (rb < rb1 ? rb : rb0) * 64); }

#pragma unroll
    for (int i = 0; i < XV; ++i) {
        const int idx = tid + i * NTH;
        if (idx < m.R) {
            uint4 p = permute8(xr[i]);
            if (MODE == 1) {
                const f16x2 s16 = {(f16) 0.0625f, (f16) 0.0625f};
                const f16x2 a = h2(p.x), b = h2(p.y) * s16, c = h2(p.z), d = h2(p.w) * s16;
                p.y = __builtin_bit_cast(uint32_t, b); p.w = __builtin_bit_cast(uint32_t, d);
                const float Cr = 1024.f * ((float) a[0] + (float) a[1] + (float) c[0] + (float) c[1]) +
                                 1024.f * ((float) b[0] + (float) b[1] + (float) d[0] + (float) d[1]);
                const float Sr = (float) a[0] + (float) a[1] + (float) c[0] + (float) c[1] +
                                 16.f * ((float) b[0] + (float) b[1] + (float) d[0] + (float) d[1]);
                // reduce over the 4 rows of a quad: rows idx..idx+3 are 4 adjacent threads
                float C4 = Cr, S4 = Sr;
                C4 += __shfl_xor(C4, 1, 64); S4 += __shfl_xor(S4, 1, 64);
                C4 += __shfl_xor(C4, 2, 64); S4 += __shfl_xor(S4, 2, 64);
                if ((idx & 3) == 0) cs[idx >> 2] = make_float2(C4, S4);
            }
            xs[idx] = p;
        }
    }
#pragma unroll
    for (int i = 0; i < TV; ++i) { const int idx = tid + i * NTH; if (idx < m.G * 16) tab[idx] = traw[i]; }
    __syncthreads();
    if (m.tl) T1 = wall_clock64();

    float acc = 0.f;
    for (int rbase = rb0; rbase < rb1; rbase += U) {
        if (rbase != rb0) {
#pragma unroll
            for (int i = 0; i < U; ++i) { const int rb = rbase + i; wv[i] = nt_load16(base + (size_t) (rb < rb1 ? rb : rb0) * 64); }
        }
#pragma unroll
        for (int i = 0; i < U; ++i) {
            const int rb = rbase + i;
            if (rb < rb1) {
                const int r = rb * 16 + rsub * 4;
                if (MODE == 2) { acc += __builtin_bit_cast(float, wv[i].x ^ wv[i].y ^ wv[i].z ^ wv[i].w); continue; }
                const uint32_t e = tab[(r >> m.gshift) * 16 + col];
                const float s = (float) __builtin_bit_cast(f16, (uint16_t) (e & 0xFFFFu));
                const uint4 x0 = xs[r], x1 = xs[r + 1], x2 = xs[r + 2], x3 = xs[r + 3];
                if (MODE == 3) {
                    const int z = (int) (e >> 16);
                    const f16 a = (f16) (float) (-(1024 + z));
                    const f16x2 zc0 = {a, a};
                    const f16x2 c960 = {(f16) 960.f, (f16) 960.f};
                    const f16x2 zc1 = zc0 + c960;
                    f32x4 c = {0.f, 0.f, 0.f, 0.f};
                    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, x0), dequant8(wv[i].x, zc0, zc1), c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, x1), dequant8(wv[i].y, zc0, zc1), c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, x2), dequant8(wv[i].z, zc0, zc1), c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, x3), dequant8(wv[i].w, zc0, zc1), c, 0, 0, 0);
                    acc = fmaf(s, c[0], acc);
                } else if (MODE == 0) {
                    const int z = (int) (e >> 16);
                    const f16 a = (f16) (float) (-(1024 + z));
                    const f16x2 zc0 = {a, a};
                    const f16x2 c960 = {(f16) 960.f, (f16) 960.f};
                    const f16x2 zc1 = zc0 + c960;
                    float part = dot8_exact(wv[i].x, x0, zc0, zc1, 0.f);
                    part = dot8_exact(wv[i].y, x1, zc0, zc1, part);
                    part = dot8_exact(wv[i].z, x2, zc0, zc1, part);
                    part = dot8_exact(wv[i].w, x3, zc0, zc1, part);
                    acc = fmaf(s, part, acc);
                } else {
                    const float z = (float) (e >> 16);
                    const float2 c4 = cs[r >> 2];
                    float part = dot8_raw(wv[i].x, x0, 0.f);
                    part = dot8_raw(wv[i].y, x1, part);
                    part = dot8_raw(wv[i].z, x2, part);
                    part = dot8_raw(wv[i].w, x3, part);
                    part = fmaf(-z, c4.y, part - c4.x);
                    acc = fmaf(s, part, acc);
                }
            }
        }
    }
    if (m.tl) { asm volatile("" :: "v"(acc)); T2 = wall_clock64(); }
    if (MODE != 3) { acc += __shfl_xor(acc, 16, 64); acc += __shfl_xor(acc, 32, 64); }
    if (lane < 16) red[wave * 16 + lane] = acc;
    __syncthreads();
    if (tid < 16) {
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) v += red[w * 16 + tid];
        out[(size_t) t * 16 + tid] = (f16) v;
    }
    if (m.tl && lane == 0) { T3 = wall_clock64(); unsigned long long* o = m.tl + ((size_t) blockIdx.x * WAVES + wave) * 4; o[0] = T0; o[1] = T1; o[2] = T2; o[3] = T3; }
}

// MODE 4 kernel: prologue FIRST (x -> LDS, barrier), per-lane scale/zero entries distributed by shuffles (no LDS table),
// then the weight stream with MFMA dot products.  ORDER: 0 = weights issued after the barrier, 1 = before the LDS writes.
template <int U, int WAVES, int ORDER>
__global__ __launch_bounds__(WAVES * 64) void gemv16b_kernel(const Mat m, const f16* __restrict__ x, f16* __restrict__ out,
                                                            int rb_per_wave)
{
    constexpr int NTH = WAVES * 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint4* xs = (uint4*) smem;                                           // [R]
    float* red = (float*) (smem + (size_t) m.R * 16);                    // [WAVES][16]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    unsigned long long T0 = 0, T1 = 0, T2 = 0, T3 = 0;
    if (m.tl) T0 = wall_clock64();
    int t = blockIdx.x;
    { const int per = gridDim.x >> 3; t = (t & 7) * per + (t >> 3); }
    const int col = lane & 15, rsub = lane >> 4;
    const int rb0 = wave * rb_per_wave;
    const int rb1 = min(m.RB, rb0 + rb_per_wave);
    constexpr int XV = (1376 + NTH - 1) / NTH;
    uint4 xr[XV];
#pragma unroll
    for (int i = 0; i < XV; ++i) { const int idx = tid + i * NTH; if (i * NTH < m.R) xr[i] = *(const uint4*) (x + (idx < m.R ? idx : 0) * 8); }
    // scale/zero entries: slot h holds group (rb0 + 4h + rsub) (g128: group == row-block), column col
    constexpr int EH = 6;                                                // up to 24 row-blocks per wave
    uint32_t ent[EH];
#pragma unroll
    for (int h = 0; h < EH; ++h) {
        const int g = min(rb0 + 4 * h + rsub, m.G - 1);
        if (4 * h < rb_per_wave) {
            const uint32_t zw = m.qzeros[(size_t) g * (m.N >> 3) + t * 2 + (col >> 3)];
            const uint16_t sb = ((const uint16_t*) m.scales)[(size_t) g * m.N + t * 16 + col];
            ent[h] = (uint32_t) sb | ((((zw >> ((col & 7) * 4)) & 0xFu) + 1) << 16);
        }
    }
    const uint4* base = m.qw + (size_t) t * m.RB * 64 + lane;
    uint4 wv[U];
    if (ORDER == 1) {
#pragma unroll
        for (int i = 0; i < U; ++i) { const int rb = rb0 + i; wv[i] = nt_load16(base + (size_t) (rb < rb1 ? rb : rb0) * 64); }
    }
#pragma unroll
    for (int i = 0; i < XV; ++i) { const int idx = tid + i * NTH; if (i * NTH < m.R && idx < m.R) xs[idx] = permute8(xr[i]); }
    __syncthreads();
    if (m.tl) T1 = wall_clock64();
    if (ORDER == 0) {
#pragma unroll
        for (int i = 0; i < U; ++i) { const int rb = rb0 + i; wv[i] = nt_load16(base + (size_t) (rb < rb1 ? rb : rb0) * 64); }
    }
    float acc = 0.f;
#pragma unroll
    for (int p = 0; p < 24 / U; ++p) {
        const int rbase = rb0 + p * U;
        if (rbase < rb1) {
            if (p > 0) {
#pragma unroll
                for (int i = 0; i < U; ++i) { const int rb = rbase + i; wv[i] = nt_load16(base + (size_t) (rb < rb1 ? rb : rb0) * 64); }
            }
#pragma unroll
            for (int i = 0; i < U; ++i) {
                const int rb = rbase + i;
                const int li = p * U + i;                                  // static index of the row-block within the wave
                if (rb < rb1) {
                    const int r = rb * 16 + rsub * 4;
                    const uint32_t e = (uint32_t) __shfl((int) ent[li >> 2], ((li & 3) << 4) | col, 64);
                    const float s = (float) __builtin_bit_cast(f16, (uint16_t) (e & 0xFFFFu));
                    const int z = (int) (e >> 16);
                    const f16 a = (f16) (float) (-(1024 + z));
                    const f16x2 zc0 = {a, a};
                    const f16x2 c960 = {(f16) 960.f, (f16) 960.f};
                    const f16x2 zc1 = zc0 + c960;
                    const uint4 x0 = xs[r], x1 = xs[r + 1], x2 = xs[r + 2], x3 = xs[r + 3];
                    f32x4 c = {0.f, 0.f, 0.f, 0.f};
                    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, x0), dequant8(wv[i].x, zc0, zc1), c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, x1), dequant8(wv[i].y, zc0, zc1), c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, x2), dequant8(wv[i].z, zc0, zc1), c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, x3), dequant8(wv[i].w, zc0, zc1), c, 0, 0, 0);
                    acc = fmaf(s, c[0], acc);
                }
            }
        }
    }
    if (m.tl) { asm volatile("" :: "v"(acc)); T2 = wall_clock64(); }
    if (lane < 16) red[wave * 16 + lane] = acc;
    __syncthreads();
    if (tid < 16) {
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) v += red[w * 16 + tid];
        out[(size_t) t * 16 + tid] = (f16) v;
    }
    if (m.tl && lane == 0) { T3 = wall_clock64(); unsigned long long* o = m.tl + ((size_t) blockIdx.x * WAVES + wave) * 4; o[0] = T0; o[1] = T1; o[2] = T2; o[3] = T3; }
}

struct Shape { const char* name; int K, N; };

int main(int argc, char** argv)
{
    const int only = argc > 1 ? atoi(argv[1]) : -1;
    int shape_index = -1;
    const Shape shapes[] = {{"qkv 4096x12288", 4096, 12288}, {"o 4096x4096", 4096, 4096}, {"gate_up 4096x22016", 4096, 22016},
                            {"down 11008x4096", 11008, 4096}};
    const int NBUF = 24;
    f16* out; f16* x;
    CHECK(hipMalloc(&out, 1 << 20));
    CHECK(hipMalloc(&x, 1 << 20));
    CHECK(hipMemset(x, 0x31, 1 << 20));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (const Shape& sh : shapes) {
        ++shape_index;
        if (only >= 0 && shape_index != only) continue;
        const int R = sh.K / 8, N = sh.N, G = sh.K / 128, RB = R / 16;
        const size_t bytes = (size_t) R * N * 4;
        std::vector<uint4*> bufs(NBUF);
        for (auto& b : bufs) { CHECK(hipMalloc(&b, bytes)); CHECK(hipMemset(b, 0x5a, bytes)); }
        uint32_t* qz; f16* sc;
        CHECK(hipMalloc(&qz, (size_t) G * (N / 8) * 4)); CHECK(hipMemset(qz, 0x77, (size_t) G * (N / 8) * 4));
        CHECK(hipMalloc(&sc, (size_t) G * N * 2)); CHECK(hipMemset(sc, 0x11, (size_t) G * N * 2));
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
        auto mat = [&](uint4* b) {
            Mat m;
            m.qw = b; m.qzeros = qz; m.scales = sc; m.K = sh.K; m.N = N; m.R = R; m.RB = RB; m.gshift = 4; m.G = G; m.tl = nullptr;
            return m;
        };
#define RUN(U, WAVES, MODE, XCD) timeit("t16 U=" #U " waves=" #WAVES " mode=" #MODE " xcd=" #XCD, [&](uint4* b) { \
            const int rbw = (RB + WAVES - 1) / WAVES; \
             \
            const size_t smem = (size_t) R * 18 + (size_t) G * 64 + WAVES * 16 * 4; \
            hipLaunchKernelGGL((gemv16_kernel<U, WAVES, MODE, XCD>), dim3(N / 16), dim3(WAVES * 64), smem, 0, mat(b), x, out, rbw); })
        RUN(4, 8, 3, true);
#define RUNB(U, WAVES, ORDER) timeit("t16b U=" #U " waves=" #WAVES " order=" #ORDER, [&](uint4* b) { \
            const int rbw = (RB + WAVES - 1) / WAVES; \
            if (rbw > 24 || (24 / U) * U < rbw) { return; } \
            const size_t smem = (size_t) R * 16 + WAVES * 16 * 4; \
            hipLaunchKernelGGL((gemv16b_kernel<U, WAVES, ORDER>), dim3(N / 16), dim3(WAVES * 64), smem, 0, mat(b), x, out, rbw); })
        RUNB(4, 8, 0); RUNB(4, 8, 1); RUNB(8, 4, 0); RUNB(8, 4, 1); RUNB(4, 4, 1); RUNB(12, 8, 0); RUNB(12, 8, 1); RUNB(6, 8, 1); RUNB(6, 16, 1); RUNB(2, 16, 1); RUNB(2, 16, 0);
        if (only >= 0) {
            constexpr int W = 8, UU = 4;
            const int nblk = N / 16;
            unsigned long long* tl; CHECK(hipMalloc(&tl, (size_t) nblk * W * 4 * 8));
            std::vector<unsigned long long> h((size_t) nblk * W * 4);
            for (int rep = 0; rep < 3; ++rep) {
                Mat m = mat(bufs[rep + 3]); m.tl = tl;
                const int rbw = (RB + W - 1) / W;
                const size_t smem = (size_t) R * 18 + (size_t) G * 64 + W * 16 * 4;
                CHECK(hipDeviceSynchronize());
                if (rbw <= 24 && (24 / UU) * UU >= rbw) hipLaunchKernelGGL((gemv16b_kernel<UU, W, 1>), dim3(nblk), dim3(W * 64), smem, 0, m, x, out, rbw);
                CHECK(hipDeviceSynchronize());
                CHECK(hipMemcpy(h.data(), tl, h.size() * 8, hipMemcpyDeviceToHost));
                unsigned long long t0 = ~0ull;
                for (size_t i = 0; i < h.size(); i += 4) if (h[i] < t0) t0 = h[i];
                // percentiles of each stamp, in units of 10 ns
                for (int k = 0; k < 4; ++k) {
                    std::vector<long> v;
                    for (size_t i = 0; i < h.size(); i += 4) v.push_back((long) (h[i + k] - t0));
                    std::sort(v.begin(), v.end());
                    printf("  timeline rep %d stamp %d (x10ns): min %ld p10 %ld p50 %ld p90 %ld max %ld\n", rep, k, v[0], v[v.size() / 10], v[v.size() / 2], v[v.size() * 9 / 10], v.back());
                }
            }
            CHECK(hipFree(tl));
        }
        for (auto& b : bufs) CHECK(hipFree(b));
        CHECK(hipFree(qz)); CHECK(hipFree(sc));
    }
    return 0;
}
