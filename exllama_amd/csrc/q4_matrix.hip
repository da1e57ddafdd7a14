// Q4 matrix construction, act-order repack, full dequantisation and the activation gather.
//
// Behavioural reference (arithmetic only; the kernels are written for wave64 / 128-bit accesses):
//   /root/reference/exllama_ext/cuda_func/q4_matrix.cu:61-102   make_sequential_kernel
//   /root/reference/exllama_ext/cuda_func/q4_matrix.cu:104-168  Q4Matrix::make_sequential
//   /root/reference/exllama_ext/cuda_func/q4_matrix.cu:170-224  reconstruct_kernel / reconstruct
//   /root/reference/exllama_ext/cuda_func/column_remap.cu:7-61  column_remap
#include "common.h"

// New packed row r takes nibble i from old row x_map[8r + i].  One thread = 4 adjacent columns (128-bit).
__global__ __launch_bounds__(256) void make_sequential_kernel(const uint4* __restrict__ w, uint4* __restrict__ w_new,
                                                              const uint32_t* __restrict__ x_map, int n4)
{
    const int c = blockIdx.x * 256 + threadIdx.x;
    const int r = blockIdx.y;
    if (c >= n4) return;
    uint4 dst = make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const uint32_t src = x_map[r * 8 + i];          // block-uniform -> scalar load
        const uint4 v = w[(size_t) (src >> 3) * n4 + c];
        const int sh = (src & 7) * 4;
        dst.x |= ((v.x >> sh) & 0xFu) << (4 * i);
        dst.y |= ((v.y >> sh) & 0xFu) << (4 * i);
        dst.z |= ((v.z >> sh) & 0xFu) << (4 * i);
        dst.w |= ((v.w >> sh) & 0xFu) << (4 * i);
    }
    w_new[(size_t) r * n4 + c] = dst;
}

int launch_make_sequential(Q4Matrix* m, const uint32_t* x_map_host, hipStream_t s)
{
    const size_t wbytes = (size_t) (m->height / 8) * m->width * sizeof(uint32_t);
    uint32_t* tmp = nullptr;
    EXL_HIP(hipMalloc((void**) &tmp, wbytes));
    // every failure path below releases the temporary; m->x_map is released by the caller's free_matrix
    hipError_t e = hipMalloc((void**) &m->x_map, (size_t) m->height * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMemcpyAsync(m->x_map, x_map_host, (size_t) m->height * sizeof(uint32_t), hipMemcpyHostToDevice, s);
    if (e == hipSuccess) {
        const int n4 = m->width / 4;
        dim3 grid((n4 + 255) / 256, m->height / 8);
        hipLaunchKernelGGL(make_sequential_kernel, grid, dim3(256), 0, s, (const uint4*) m->qweight, (uint4*) tmp, m->x_map, n4);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(m->qweight, tmp, wbytes, hipMemcpyDeviceToDevice, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void) hipFree(tmp);
    if (e != hipSuccess) EXL_FAIL((int) e, "make_sequential: %s", hipGetErrorString(e));
    return 0;
}

// W16[8r + j, n] = half(q - (z + 1)) * scale   (one fp16 multiply, as q4_matrix.cu:207)
__global__ __launch_bounds__(256) void reconstruct_kernel(const uint4* __restrict__ w, f16* __restrict__ out,
                                                          const f16* __restrict__ scales,
                                                          const uint32_t* __restrict__ qzeros, int n4, int width,
                                                          int groupsize)
{
    const int c = blockIdx.x * 256 + threadIdx.x;       // column quad
    const int r = blockIdx.y;                           // packed row
    if (c >= n4) return;
    const int group = (r * 8) / groupsize;
    const int col = c * 4;
    const uint32_t zw = qzeros[(size_t) group * (width / 8) + (col >> 3)];
    const f16x4 sc = *(const f16x4*) (scales + (size_t) group * width + col);
    const uint4 v = w[(size_t) r * n4 + c];
    const uint32_t words[4] = {v.x, v.y, v.z, v.w};
    int z[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) z[j] = (int) ((zw >> (4 * ((col & 7) + j))) & 0xFu) + 1;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        f16x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int q = (int) ((words[j] >> (4 * i)) & 0xFu);
            o[j] = (f16) (q - z[j]) * sc[j];
        }
        *(f16x4*) (out + (size_t) (r * 8 + i) * width + col) = o;
    }
}

// T16 layout: one thread = one 16-byte piece = 4 packed rows x 1 column -> 32 fp16 values of that column.
__global__ __launch_bounds__(256) void reconstruct_t16_kernel(const uint4* __restrict__ w, f16* __restrict__ out,
                                                              const f16* __restrict__ scales,
                                                              const uint32_t* __restrict__ qzeros, int RB, int width,
                                                              int groupsize, size_t npieces)
{
    const size_t p = (size_t) blockIdx.x * 256 + threadIdx.x;
    if (p >= npieces) return;
    const int lane = (int) (p & 63);
    const size_t trb = p >> 6;
    const int rb = (int) (trb % RB), t = (int) (trb / RB);
    const int col = lane & 15, rsub = lane >> 4;
    const int n = t * 16 + col;
    const int r0 = rb * 16 + rsub * 4;
    const int group = (r0 * 8) / groupsize;                  // 4 packed rows never straddle a group (groupsize % 32 == 0)
    const int z = (int) ((qzeros[(size_t) group * (width / 8) + (n >> 3)] >> (4 * (n & 7))) & 0xFu) + 1;
    const f16 sc = scales[(size_t) group * width + n];
    const uint4 v = w[p];
    const uint32_t words[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int q = (int) ((words[j] >> (4 * ((i >> 1) + 4 * (i & 1)))) & 0xFu);      // interleaved nibble of weight i
            out[(size_t) ((r0 + j) * 8 + i) * width + n] = (f16) (q - z) * sc;
        }
}

// nibble k of a GPTQ word -> nibble (k >> 1) + 4 * (k & 1): even weights in the low half-word, odd ones in the high one
__device__ __forceinline__ uint32_t t16_interleave(uint32_t w)
{
    const uint32_t e = w & 0x0F0F0F0Fu, o = (w >> 4) & 0x0F0F0F0Fu;
    const uint32_t ep = (e & 0xFu) | ((e >> 4) & 0xF0u) | ((e >> 8) & 0xF00u) | ((e >> 12) & 0xF000u);
    const uint32_t op = (o & 0xFu) | ((o >> 4) & 0xF0u) | ((o >> 8) & 0xF00u) | ((o >> 12) & 0xF000u);
    return ep | (op << 16);
}

// GPTQ [R][N] words -> T16 pieces.  One thread = one destination piece.
__global__ __launch_bounds__(256) void retile_t16_kernel(const uint32_t* __restrict__ src, uint4* __restrict__ dst, int RB,
                                                         int width, size_t npieces)
{
    const size_t p = (size_t) blockIdx.x * 256 + threadIdx.x;
    if (p >= npieces) return;
    const int lane = (int) (p & 63);
    const size_t trb = p >> 6;
    const int rb = (int) (trb % RB), t = (int) (trb / RB);
    const int n = t * 16 + (lane & 15);
    const int r0 = rb * 16 + (lane >> 4) * 4;
    uint4 v;
    v.x = t16_interleave(src[(size_t) (r0 + 0) * width + n]);
    v.y = t16_interleave(src[(size_t) (r0 + 1) * width + n]);
    v.z = t16_interleave(src[(size_t) (r0 + 2) * width + n]);
    v.w = t16_interleave(src[(size_t) (r0 + 3) * width + n]);
    dst[p] = v;
}

int launch_retile_t16(Q4Matrix* m, hipStream_t s)
{
    const int R = m->height / 8;
    const size_t wbytes = (size_t) R * m->width * sizeof(uint32_t);
    uint32_t* tmp = nullptr;
    EXL_HIP(hipMalloc((void**) &tmp, wbytes));
    hipError_t e = hipMemcpyAsync(tmp, m->qweight, wbytes, hipMemcpyDeviceToDevice, s);
    const size_t npieces = wbytes / 16;
    if (e == hipSuccess) {
        hipLaunchKernelGGL(retile_t16_kernel, dim3((unsigned) ((npieces + 255) / 256)), dim3(256), 0, s, tmp, (uint4*) m->qweight,
                           R / 16, m->width, npieces);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void) hipFree(tmp);
    if (e != hipSuccess) EXL_FAIL((int) e, "retile_t16: %s", hipGetErrorString(e));
    m->layout = EXL_LAYOUT_T16;
    return 0;
}

int launch_reconstruct(const Q4Matrix* m, f16* out, hipStream_t s)
{
    if (m->layout == EXL_LAYOUT_T16) {
        const size_t npieces = (size_t) (m->height / 8) * m->width / 4;
        hipLaunchKernelGGL(reconstruct_t16_kernel, dim3((unsigned) ((npieces + 255) / 256)), dim3(256), 0, s,
                           (const uint4*) m->qweight, out, m->scales, m->qzeros, m->height / 128, m->width, m->groupsize, npieces);
        EXL_LAUNCH_CHECK();
        return 0;
    }
    const int n4 = m->width / 4;
    dim3 grid((n4 + 255) / 256, m->height / 8);
    hipLaunchKernelGGL(reconstruct_kernel, grid, dim3(256), 0, s, (const uint4*) m->qweight, out, m->scales,
                       m->qzeros, n4, m->width, m->groupsize);
    EXL_LAUNCH_CHECK();
    return 0;
}

// x_new[m, c] = x[m, x_map[c]];  one thread = 8 destination columns (one 128-bit store), 8 gathered halves.
__global__ __launch_bounds__(256) void column_remap_kernel(const f16* __restrict__ x, f16* __restrict__ x_new,
                                                           const uint32_t* __restrict__ x_map, int width)
{
    const int c8 = blockIdx.x * 256 + threadIdx.x;
    const int row = blockIdx.y;
    const int c = c8 * 8;
    if (c >= width) return;
    const f16* xr = x + (size_t) row * width;
    if (c + 8 <= width) {
        const uint4 m0 = *(const uint4*) (x_map + c);
        const uint4 m1 = *(const uint4*) (x_map + c + 4);
        f16x8 o;
        o[0] = xr[m0.x]; o[1] = xr[m0.y]; o[2] = xr[m0.z]; o[3] = xr[m0.w];
        o[4] = xr[m1.x]; o[5] = xr[m1.y]; o[6] = xr[m1.z]; o[7] = xr[m1.w];
        *(f16x8*) (x_new + (size_t) row * width + c) = o;
    } else {
        for (int i = c; i < width; ++i) x_new[(size_t) row * width + i] = xr[x_map[i]];
    }
}

// The same gather with the source row staged in LDS: one block per row copies it in with coalesced 16-byte loads and gathers from
// LDS (2-byte reads that cost no cache-line transaction each), so HBM / L2 see one read and one write of the row.  width % 8 == 0,
// width * 2 bytes <= 64 KiB (every Llama shape: 28672 columns = 56 KiB).
__global__ __launch_bounds__(512) void column_remap_lds_kernel(const f16* __restrict__ x, f16* __restrict__ x_new,
                                                               const uint32_t* __restrict__ x_map, int width)
{
    extern __shared__ __attribute__((aligned(16))) f16 srow[];
    const int row = blockIdx.x;
    const int nvec = width >> 3;
    const f16* xr = x + (size_t) row * width;
    for (int i = threadIdx.x; i < nvec; i += 512) *(f16x8*) (srow + i * 8) = *(const f16x8*) (xr + i * 8);
    __syncthreads();
    for (int i = threadIdx.x; i < nvec; i += 512) {
        const uint4 m0 = *(const uint4*) (x_map + i * 8);
        const uint4 m1 = *(const uint4*) (x_map + i * 8 + 4);
        f16x8 o;
        o[0] = srow[m0.x]; o[1] = srow[m0.y]; o[2] = srow[m0.z]; o[3] = srow[m0.w];
        o[4] = srow[m1.x]; o[5] = srow[m1.y]; o[6] = srow[m1.z]; o[7] = srow[m1.w];
        *(f16x8*) (x_new + (size_t) row * width + i * 8) = o;
    }
}

int launch_column_remap(const f16* x, f16* x_new, int height, int width, const uint32_t* x_map, hipStream_t s)
{
    if (height <= 0) return 0;
    if (width % 8 == 0 && (size_t) width * 2 <= 64 * 1024 && width >= 1024) {
        hipLaunchKernelGGL(column_remap_lds_kernel, dim3(height), dim3(512), (size_t) width * 2, s, x, x_new, x_map, width);
        EXL_LAUNCH_CHECK();
        return 0;
    }
    dim3 grid(((width + 7) / 8 + 255) / 256, height);
    hipLaunchKernelGGL(column_remap_kernel, grid, dim3(256), 0, s, x, x_new, x_map, width);
    EXL_LAUNCH_CHECK();
    return 0;
}
