// VALU issue-rate probe: cycles per wave64 instruction for the op mix of the int4 dequantisation, at 1..4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int OP>
__global__ __launch_bounds__(256) void probe(uint32_t* out, int iters, long long* cyc)
{
    uint32_t a[8];
    f16x2 h[8];
    float f[8];
    for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x * 2654435761u + i; h[i] = __builtin_bit_cast(f16x2, a[i] & 0x3c003c00u); f[i] = (float) i; }
    const f16x2 c1 = {(_Float16) 0.0625f, (_Float16) 0.0625f}, c2 = {(_Float16) -72.f, (_Float16) -72.f};
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (OP == 0) a[i] = (a[i] & 0x000F000Fu) | 0x64006400u;                       // v_and_or_b32
                if (OP == 1) h[i] = h[i] + c2;                                                // v_pk_add_f16
                if (OP == 2) h[i] = h[i] * c1 + c2;                                           // v_pk_fma_f16
                if (OP == 3) f[i] = __builtin_amdgcn_fdot2(h[i], c1, f[i], false);            // v_dot2_f32_f16
                if (OP == 4) f[i] = fmaf(f[i], 1.0001f, 0.5f);                                // v_fma_f32
                if (OP == 5) a[i] = a[i] >> 8 | (a[i] << 24);                                 // v_alignbit / shift-or
                if (OP == 6) h[i] = h[i] * c1;                                                // v_pk_mul_f16
            }
    }
    const long long t1 = clock64();
    uint32_t s = 0;
    for (int i = 0; i < 8; ++i) s += a[i] + __builtin_bit_cast(uint32_t, h[i]) + (uint32_t) f[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

int main()
{
    uint32_t* out; long long* cyc;
    CK(hipMalloc(&out, 1 << 24)); CK(hipMalloc(&cyc, 8));
    const char* names[] = {"v_and_or_b32", "v_pk_add_f16", "v_pk_fma_f16", "v_dot2_f32_f16", "v_fma_f32", "shift_or", "v_pk_mul_f16"};
    const int iters = 2000;
    for (int op = 0; op < 7; ++op)
        for (int bpc = 1; bpc <= 4; bpc *= 2) {       // 256-thread blocks: 1 wave per SIMD each; bpc blocks per CU
            long long c = 0;
            for (int rep = 0; rep < 2; ++rep) {
                switch (op) {
                case 0: probe<0><<<256 * bpc, 256>>>(out, iters, cyc); break;
                case 1: probe<1><<<256 * bpc, 256>>>(out, iters, cyc); break;
                case 2: probe<2><<<256 * bpc, 256>>>(out, iters, cyc); break;
                case 3: probe<3><<<256 * bpc, 256>>>(out, iters, cyc); break;
                case 4: probe<4><<<256 * bpc, 256>>>(out, iters, cyc); break;
                case 5: probe<5><<<256 * bpc, 256>>>(out, iters, cyc); break;
                case 6: probe<6><<<256 * bpc, 256>>>(out, iters, cyc); break;
                }
                CK(hipDeviceSynchronize());
                CK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
            }
            printf("%-16s waves/SIMD %d : %.2f clock64 ticks per instruction per wave  (%.2f per SIMD-instruction)\n", names[op], bpc,
                   (double) c / (iters * 32.0), (double) c / (iters * 32.0) / bpc);
        }
    return 0;
}
