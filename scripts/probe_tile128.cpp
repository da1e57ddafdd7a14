// Hunts the defect of the 128 x 128 tile GEMM (q4_gemm_t16m_kernel<2,2,4,4>) through the C ABI only: every
// buffer the kernel may read sits in one arena pre-filled with fp16 NaNs (so any read outside a buffer shows), the output
// is pre-filled with a sentinel (so any cell the kernel does not write shows), and the result is compared bit for bit with
// the same product from the 256-row kernels (run first without EXL_GEMM_TILE128, which dumps the reference).
//   hipcc -O2 -std=c++17 --offload-arch=gfx950 scripts/probe_tile128.cpp -Iinclude -Lexllama_amd -lexl_amd -Wl,-rpath,'$ORIGIN/../exllama_amd' -o build/probe_tile128
//   build/probe_tile128 ref.bin 400 4096 11008 32 ; EXL_GEMM_TILE128=1 build/probe_tile128 ref.bin 400 4096 11008 32
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <vector>
#include "exl_amd.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
#define EX(x) do { int r = (x); if (r) { printf("%s -> %d: %s\n", #x, r, exl_last_error()); exit(1); } } while (0)
__global__ void fill_u32(uint32_t* p, size_t n, uint32_t seed)
{
    for (size_t i = blockIdx.x * (size_t) blockDim.x + threadIdx.x; i < n; i += (size_t) gridDim.x * blockDim.x) {
        uint32_t x = (uint32_t) i * 2654435761u + seed; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13; x *= 3266489917u; x ^= x >> 16; p[i] = x;
    }
}
__global__ void fill_const(uint32_t* p, size_t n, uint32_t v)
{
    for (size_t i = blockIdx.x * (size_t) blockDim.x + threadIdx.x; i < n; i += (size_t) gridDim.x * blockDim.x) p[i] = v;
}
__global__ void fill_f16(_Float16* p, size_t n, float lo, float hi, uint32_t seed)
{
    for (size_t i = blockIdx.x * (size_t) blockDim.x + threadIdx.x; i < n; i += (size_t) gridDim.x * blockDim.x) {
        uint32_t x = (uint32_t) i * 2654435761u + seed; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
        p[i] = (_Float16) (lo + (hi - lo) * ((x & 0xFFFF) / 65535.0f));
    }
}
int main(int argc, char** argv)
{
    const char* ref_path = argc > 1 ? argv[1] : "ref.bin";
    const int M = argc > 2 ? atoi(argv[2]) : 400, K = argc > 3 ? atoi(argv[3]) : 4096, N = argc > 4 ? atoi(argv[4]) : 11008;
    const int gs = argc > 5 ? atoi(argv[5]) : 32, G = K / gs, NOUT = 6, LAUNCHES = 24;
    const bool tile128 = getenv("EXL_GEMM_TILE128") != nullptr;
    CK(hipSetDevice(0));
    const size_t arena_bytes = (size_t) 1 << 30;
    unsigned char* arena;
    CK(hipMalloc(&arena, arena_bytes));
    fill_const<<<2048, 256>>>((uint32_t*) arena, arena_bytes / 4, 0x7E007E00u);      // fp16 NaN everywhere
    size_t top = 1 << 20;
    auto take = [&](size_t bytes) { void* p = arena + top; top += (bytes + (1 << 16) + 4095) & ~(size_t) 4095; if (top > arena_bytes) { printf("arena\n"); exit(1); } return p; };
    const size_t ts = (size_t) 512 * 11008, tm = 1024;
    void* t0 = take(ts * 2); void* t1 = take(tm * 2); void* t2 = take(1024 * 4); void* t3 = take(1024 * 2);
    EX(exl_prepare_buffers(0, t0, ts, t1, tm, t2, 1024, t3, 1024));
    uint32_t* qw = (uint32_t*) take((size_t) K / 8 * N * 4);
    uint32_t* qz = (uint32_t*) take((size_t) G * N / 8 * 4);
    _Float16* sc = (_Float16*) take((size_t) G * N * 2);
    _Float16* x = (_Float16*) take((size_t) M * K * 2);
    fill_u32<<<1024, 256>>>(qw, (size_t) K / 8 * N, 11);
    fill_u32<<<256, 256>>>(qz, (size_t) G * N / 8, 12);
    fill_f16<<<256, 256>>>(sc, (size_t) G * N, 0.002f, 0.006f, 13);
    fill_f16<<<1024, 256>>>(x, (size_t) M * K, -1.f, 1.f, 3);
    void* h;
    EX(exl_make_q4(0, K, N, G, qw, qz, (uint16_t*) sc, nullptr, nullptr, &h));
    _Float16* outs[NOUT];
    for (int i = 0; i < NOUT; ++i) outs[i] = (_Float16*) take((size_t) M * N * 2);
    const size_t cells = (size_t) M * N;
    std::vector<uint16_t> got(cells), ref(cells);
    bool have_ref = false;
    if (tile128) {
        FILE* f = fopen(ref_path, "rb");
        if (f) { have_ref = fread(ref.data(), 2, cells, f) == cells; fclose(f); }
        if (!have_ref) printf("no reference dump at %s: sentinel check only\n", ref_path);
    }
    int bad_launches = 0;
    for (int l = 0; l < LAUNCHES; ++l) {
        _Float16* out = outs[l % NOUT];
        fill_const<<<1024, 256>>>((uint32_t*) out, cells / 2, 0xFBFFFBFFu);               // sentinel -65504
        EX(exl_q4_matmul_gemm(h, x, M, out, 0, nullptr));
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(got.data(), out, cells * 2, hipMemcpyDeviceToHost));
        size_t sent = 0, nan = 0, diff = 0;
        int r0 = 1 << 30, r1 = -1, c0 = 1 << 30, c1 = -1;
        for (size_t i = 0; i < cells; ++i) {
            const uint16_t v = got[i];
            const bool is_sent = v == 0xFBFF, is_nan = (v & 0x7C00) == 0x7C00 && !is_sent, is_diff = have_ref && v != ref[i];
            sent += is_sent; nan += is_nan; diff += is_diff;
            if (is_sent || is_nan || is_diff) { const int r = (int) (i / N), c = (int) (i % N); r0 = r < r0 ? r : r0; r1 = r > r1 ? r : r1; c0 = c < c0 ? c : c0; c1 = c > c1 ? c : c1; }
        }
        if (sent || nan || diff) {
            ++bad_launches;
            printf("launch %2d: %zu sentinel (unwritten) %zu non-finite %zu differ from the 256-row kernels; rows %d..%d cols %d..%d\n", l, sent, nan, diff, r0, r1, c0, c1);
            // which rows x which 16-column tiles
            int shown = 0;
            for (int r = r0; r <= r1 && shown < 12; ++r) {
                int cnt = 0, first = -1;
                for (int c = c0; c <= c1; ++c) { const uint16_t v = got[(size_t) r * N + c]; if (v == 0xFBFF || (v & 0x7C00) == 0x7C00 || (have_ref && v != ref[(size_t) r * N + c])) { ++cnt; if (first < 0) first = c; } }
                if (cnt) { printf("   row %d: %d cells from col %d, e.g. got %04x ref %04x\n", r, cnt, first, got[(size_t) r * N + first], have_ref ? ref[(size_t) r * N + first] : 0); ++shown; }
            }
        }
        if (!tile128 && l == 0) {
            FILE* f = fopen(ref_path, "wb");
            if (f) { fwrite(got.data(), 2, cells, f); fclose(f); }
        }
    }
    printf("%s  M %d K %d N %d groupsize %d: %d of %d launches bad\n", tile128 ? "128 x 128 tile kernel" : "256-row kernels", M, K, N, gs, bad_launches, LAUNCHES);
    return 0;
}
