// Stand-alone micro-benchmark of the re-tiled decode GEMV core (exllama_amd/csrc/gemv_tile.h) on MI355X.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -I. scripts/bench_gemv_tile.hip -o build/bench_gemv_tile
// Not part of the product; the kernel below is the same structure as the product's decode kernels (prologue that
// stages x into LDS and builds the constant table while the first pass of weight loads is in flight).
#include "../exllama_amd/csrc/gemv_tile.h"
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// NT tiles per block (1, 2 or 4); waves = 4; wave w -> tile (w % NT), row slice (w / NT) of (4 / NT)
template <int U, int NT, bool XCD, int ABL>
__global__ __launch_bounds__(256) void gemv_tile_kernel(const GtMatrix m, const f16* __restrict__ x, f16* __restrict__ out,
                                                        int rows_per_slice)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint4* xs = (uint4*) smem;                                           // [R]
    GtEntry* tab = (GtEntry*) (smem + (size_t) m.R * 16);                // [NT][G][2]
    float* red = (float*) (smem + (size_t) m.R * 16 + (size_t) NT * m.G * 64);   // [4][8]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    int b = blockIdx.x;
    if (XCD) { const int per = gridDim.x >> 3; b = (b & 7) * per + (b >> 3); }
    const int tl = wave % NT, slice = wave / NT;
    const int t = b * NT + tl;
    const int row0 = slice * rows_per_slice;
    const int row1 = min(m.R, row0 + rows_per_slice);
    uint4 wv[U];
    float acc[1][4] = {{0.f, 0.f, 0.f, 0.f}};
    if constexpr (ABL == 0) {
        gt_issue<U>(m, t, row0, row0, row1, lane, wv);
        for (int i = tid; i < m.R; i += 256) xs[i] = gt_permute(*(const uint4*) (x + i * 8));
        constexpr int TPT = 256 / NT;
        gt_build_table(m, b * NT + tid / TPT, tab + (size_t) (tid / TPT) * m.G * 2, tid % TPT, TPT);
        __syncthreads();
        gt_finish<U, 1>(m, t, row0, row1, xs, 0, tab + (size_t) tl * m.G * 2, lane, wv, acc);
    } else if constexpr (ABL == 1) {
        // prologue loads FIRST (x, scales, zeros into registers), then the weight stream
        uint4 xr[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) { const int idx = tid + i * 256; xr[i] = *(const uint4*) (x + (idx < m.R ? idx : 0) * 8); }
        constexpr int TPT = 256 / NT;
        const int tt = b * NT + tid / TPT, g = tid % TPT;
        const int gc = g < m.G ? g : 0;
        const uint32_t zw = m.qzeros[(size_t) gc * (m.N >> 3) + tt];
        const uint4 sraw = *(const uint4*) (m.scales + (size_t) gc * m.N + tt * 8);
        gt_issue<U>(m, t, row0, row0, row1, lane, wv);
#pragma unroll
        for (int i = 0; i < 2; ++i) { const int idx = tid + i * 256; if (idx < m.R) xs[idx] = gt_permute(xr[i]); }
        if (g < m.G) {
            const f16x8 s8 = __builtin_bit_cast(f16x8, sraw);
            GtEntry* tb = tab + (size_t) (tid / TPT) * m.G * 2;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                GtEntry e; uint32_t zz[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int z = (int) ((zw >> (4 * (q * 4 + j))) & 0xFu) + 1;
                    const f16 a = (f16) (float) (-(1024 + z));
                    const f16x2 p = {a, a};
                    zz[j] = __builtin_bit_cast(uint32_t, p);
                }
                e.z = make_uint4(zz[0], zz[1], zz[2], zz[3]);
                e.s = make_float4((float) s8[q * 4 + 0], (float) s8[q * 4 + 1], (float) s8[q * 4 + 2], (float) s8[q * 4 + 3]);
                tb[g * 2 + q] = e;
            }
        }
        __syncthreads();
        gt_finish<U, 1>(m, t, row0, row1, xs, 0, tab + (size_t) tl * m.G * 2, lane, wv, acc);
    } else {
        // ABL 2: like 1 but constant table entry (no per-row LDS lookups of constants); ABL 3: also no x staging, no barrier;
        // ABL 4: loads + trivial consume only
        gt_issue<U>(m, t, row0, row0, row1, lane, wv);
        if constexpr (ABL == 2) {
            for (int i = tid; i < m.R; i += 256) xs[i] = gt_permute(*(const uint4*) (x + i * 8));
            __syncthreads();
        }
        const f16x2 zc = {(f16) -1032.f, (f16) -1032.f}, zc1 = {(f16) -72.f, (f16) -72.f};
        const int lr = lane >> 1;
        float part[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < U; ++i) {
            if (row0 + i * 32 < row1) {
                if constexpr (ABL == 4) {
                    part[0] += __builtin_bit_cast(float, wv[i].x ^ wv[i].y); part[1] += __builtin_bit_cast(float, wv[i].z ^ wv[i].w);
                } else {
                    const int rc = min(row0 + i * 32 + lr, row1 - 1);
                    const uint4 x4 = ABL == 2 ? xs[rc] : make_uint4(0x3c003c00u, 0x3c003c00u, 0x38003800u, 0x34003400u);
                    part[0] = gt_dot8(wv[i].x, x4, zc, zc1, part[0]);
                    part[1] = gt_dot8(wv[i].y, x4, zc, zc1, part[1]);
                    part[2] = gt_dot8(wv[i].z, x4, zc, zc1, part[2]);
                    part[3] = gt_dot8(wv[i].w, x4, zc, zc1, part[3]);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[0][j] = part[j] * 0.01f;
    }
    gt_wave_reduce<1>(acc);
    if (lane < 2) {
#pragma unroll
        for (int j = 0; j < 4; ++j) red[wave * 8 + lane * 4 + j] = acc[0][j];
    }
    __syncthreads();
    if (tid < NT * 8) {
        const int tt = tid >> 3, c = tid & 7;
        float v = 0.f;
        for (int s = 0; s < 4 / NT; ++s) v += red[(s * NT + tt) * 8 + c];
        out[(size_t) (b * NT + tt) * 8 + c] = (f16) v;
    }
}

struct Shape { const char* name; int K, N; };

int main()
{
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
        const int R = sh.K / 8, N = sh.N, G = sh.K / 128;
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
            GtMatrix m;
            m.qw = b; m.qzeros = qz; m.scales = sc; m.x_map = nullptr; m.K = sh.K; m.N = N; m.R = R; m.gprows = 16; m.gshift = 4; m.G = G;
            return m;
        };
#define RUN(U, NT, XCD, ABL) timeit("tile U=" #U " NT=" #NT " xcd=" #XCD " abl=" #ABL, [&](uint4* b) { \
            const int nsl = 4 / NT; const int rps = ((R + nsl - 1) / nsl + 31) / 32 * 32; \
            if ((rps + 31) / 32 > 3 * U) return; \
            const size_t smem = (size_t) R * 16 + (size_t) NT * G * 64 + 4 * 8 * 4; \
            hipLaunchKernelGGL((gemv_tile_kernel<U, NT, XCD, ABL>), dim3(N / 8 / NT), dim3(256), smem, 0, mat(b), x, out, rps); })
        RUN(4, 1, true, 0); RUN(4, 1, true, 1); RUN(4, 1, true, 2); RUN(4, 1, true, 3); RUN(4, 1, true, 4);
        RUN(8, 2, true, 0); RUN(8, 2, true, 1); RUN(8, 2, true, 2); RUN(8, 2, true, 3); RUN(8, 2, true, 4);
        RUN(12, 1, true, 1); RUN(12, 1, true, 3);
        for (auto& b : bufs) CHECK(hipFree(b));
        CHECK(hipFree(qz)); CHECK(hipFree(sc));
    }
    return 0;
}
