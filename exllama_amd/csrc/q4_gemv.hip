// Decode path: out[M,N] (+)= x[M,K] @ dequant(W[K,N]) for M <= 8 -- a bandwidth-bound wave64 GEMV.
//
// Replaces /root/reference/exllama_ext/cuda_func/q4_matmul.cu:35-212 (q4_matmul_kernel) and the
// dot_product_8* helpers of /root/reference/exllama_ext/matrix.cuh:87-286.  Same algebra
//     out[m,n] = sum_g scale[g,n] * sum_{k in g} x[m,k] * (q[k,n] - (z[g,n] + 1))
// but designed for CDNA4 instead of translated:
//   * packed weights are read straight from the GPTQ layout with 128-bit loads: one lane = one uint4 =
//     4 adjacent output columns x 8 consecutive k;  TX lanes side by side cover full 128-byte lines;
//   * the other lanes of the block walk DIFFERENT k-chunks of the same column tile (TY k-slices), every
//     lane issues its whole chunk of loads before consuming the first (deep memory-level parallelism,
//     no LDS round trip for weights);
//   * nibbles are expanded with the fp16 magic-number trick (0x6400 | q -> 1024 + q) two at a time,
//     zero-point removed exactly in fp16, products accumulated in FP32 with v_dot2_f32_f16;
//   * x (gathered through x_map for act-order weights -- the reference's separate column_remap pass is
//     fused here) is staged once per block in LDS, pre-permuted to the nibble-pair order;
//   * k-slices are combined with wave shuffles + LDS; split-K across blocks writes fp32 slabs that a
//     tiny second kernel sums in a FIXED order.  No atomics anywhere -> bit-reproducible run to run
//     (the reference uses fp16 atomicAdd, q4_matmul.cu:206).
#include "gemv_t16.h"

#define MAGIC_1024 0x64006400u

__device__ __forceinline__ f16x2 as_h2(uint32_t v) { return __builtin_bit_cast(f16x2, v); }

// (h0..h7) -> (h0,h4),(h1,h5),(h2,h6),(h3,h7): the order in which the nibble pairs come out of a packed word
__device__ __forceinline__ uint4 permute_x8(uint4 d)
{
    uint4 o;
    o.x = (d.x & 0xFFFFu) | (d.z << 16);
    o.y = (d.x >> 16) | (d.z & 0xFFFF0000u);
    o.z = (d.y & 0xFFFFu) | (d.w << 16);
    o.w = (d.y >> 16) | (d.w & 0xFFFF0000u);
    return o;
}

// sum_k x[k] * (q[k] - z) over the 8 nibbles of `w`;  x4 = 8 halves in permuted pair order
__device__ __forceinline__ float dot8(uint32_t w, const uint4& x4, f16x2 zc0, f16x2 zc1, float acc)
{
    const f16x2 sixteenth = {(f16) 0.0625f, (f16) 0.0625f};
    const uint32_t w8 = w >> 8;
    const f16x2 d0 = as_h2((w & 0x000F000Fu) | MAGIC_1024) + zc0;                    // (q0 - z, q4 - z)
    const f16x2 d1 = as_h2((w & 0x00F000F0u) | MAGIC_1024) * sixteenth + zc1;        // (q1 - z, q5 - z)
    const f16x2 d2 = as_h2((w8 & 0x000F000Fu) | MAGIC_1024) + zc0;                   // (q2 - z, q6 - z)
    const f16x2 d3 = as_h2((w8 & 0x00F000F0u) | MAGIC_1024) * sixteenth + zc1;       // (q3 - z, q7 - z)
    acc = __builtin_amdgcn_fdot2(d0, as_h2(x4.x), acc, false);
    acc = __builtin_amdgcn_fdot2(d1, as_h2(x4.y), acc, false);
    acc = __builtin_amdgcn_fdot2(d2, as_h2(x4.z), acc, false);
    acc = __builtin_amdgcn_fdot2(d3, as_h2(x4.w), acc, false);
    return acc;
}

// M  : rows of x held per block (1, 2, 4, 8; real row count `rows` <= M)
// TX : lanes across N (uint4 each) -> BN = 4 * TX columns per block; TY = 256 / TX k-slices
// CH : packed rows (8 k each) per chunk; a chunk never straddles a group when groupsize >= 8 * CH
template <int M, int TX, int CH>
__global__ __launch_bounds__(256) void q4_gemv_kernel(const f16* __restrict__ x, const uint4* __restrict__ qweight,
                                                      const uint32_t* __restrict__ qzeros,
                                                      const f16* __restrict__ scales,
                                                      const uint32_t* __restrict__ x_map, f16* __restrict__ out,
                                                      float* __restrict__ slabs, int rows, int K, int N, int groupsize,
                                                      int prows_per_block, int no_zero, int splitk)
{
    constexpr int TY = 256 / TX;
    constexpr int BN = 4 * TX;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x;
    const int tx = tid % TX;
    const int ty = tid / TX;
    const int n4 = N >> 2;
    const int col = blockIdx.x * BN + tx * 4;
    const int prow_total = K >> 3;
    const int r0 = blockIdx.y * prows_per_block;
    const int r1 = min(prow_total, r0 + prows_per_block);
    const int nrows = r1 - r0;                               // packed rows handled by this block

    // ---- stage x[:, 8*r0 .. 8*r1) into LDS, permuted, gathered through x_map when present ----------
    uint4* xs = (uint4*) smem;                               // [M][nrows] uint4 (8 halves each)
    for (int idx = tid; idx < M * nrows; idx += 256) {
        const int m = idx / nrows;
        const int rr = idx - m * nrows;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (m < rows) {
            const f16* xr = x + (size_t) m * K;
            const int k = (r0 + rr) * 8;
            if (x_map) {
                const uint4 m0 = *(const uint4*) (x_map + k);
                const uint4 m1 = *(const uint4*) (x_map + k + 4);
                f16x8 g;
                g[0] = xr[m0.x]; g[1] = xr[m0.y]; g[2] = xr[m0.z]; g[3] = xr[m0.w];
                g[4] = xr[m1.x]; g[5] = xr[m1.y]; g[6] = xr[m1.z]; g[7] = xr[m1.w];
                v = __builtin_bit_cast(uint4, g);
            } else {
                v = *(const uint4*) (xr + k);
            }
        }
        xs[idx] = permute_x8(v);
    }
    __syncthreads();

    float acc[M][4];
#pragma unroll
    for (int m = 0; m < M; ++m)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[m][j] = 0.f;

    const bool col_ok = col < N;
    const int gprows = groupsize >> 3;                       // packed rows per group
    const uint4* wcol = qweight + (col >> 2);

    for (int c0 = ty * CH; c0 < nrows; c0 += TY * CH) {
        // issue the whole chunk's weight loads first
        uint4 wv[CH];
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            const int rr = c0 + i;
            wv[i] = (col_ok && rr < nrows) ? nt_load16(wcol + (size_t) (r0 + rr) * n4)
                                           : make_uint4(0, 0, 0, 0);
        }
        // group bookkeeping without per-row divisions: `until` = packed rows left in the current group
        const int rbase = r0 + c0;
        int g = rbase / gprows;
        int until = gprows - (rbase - g * gprows);
        f16x2 zc0[4], zc1[4];
        float sc[4];
        float part[M][4];
        auto load_group = [&](int grp) {
            uint32_t zw = 0;
            f16x4 s4 = {(f16) 0.f, (f16) 0.f, (f16) 0.f, (f16) 0.f};
            if (col_ok) {
                zw = qzeros[(size_t) grp * (N >> 3) + (col >> 3)];
                s4 = *(const f16x4*) (scales + (size_t) grp * N + col);
            }
            const int sh = (col & 7) * 4;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int z = (int) ((zw >> (sh + 4 * j)) & 0xFu) + 1;
                const f16 a = (f16) (float) (-(1024 + z));
                const f16 b = (f16) (float) (-(64 + z));
                zc0[j] = (f16x2){a, a};
                zc1[j] = (f16x2){b, b};
                sc[j] = (float) s4[j];
            }
#pragma unroll
            for (int m = 0; m < M; ++m)
#pragma unroll
                for (int j = 0; j < 4; ++j) part[m][j] = 0.f;
        };
        auto flush_group = [&]() {
#pragma unroll
            for (int m = 0; m < M; ++m)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[m][j] = fmaf(sc[j], part[m][j], acc[m][j]);
        };
        load_group(g);
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            const int rr = c0 + i;
            if (rr < nrows) {
                if (until == 0) { flush_group(); ++g; load_group(g); until = gprows; }
                --until;
                const uint32_t words[4] = {wv[i].x, wv[i].y, wv[i].z, wv[i].w};
#pragma unroll
                for (int m = 0; m < M; ++m) {
                    const uint4 x4 = xs[m * nrows + rr];
#pragma unroll
                    for (int j = 0; j < 4; ++j) part[m][j] = dot8(words[j], x4, zc0[j], zc1[j], part[m][j]);
                }
            }
        }
        flush_group();
    }

    // ---- combine the TY k-slices: shuffles inside the wave, LDS across the 4 waves -------------------
#pragma unroll
    for (int m = 0; m < M; ++m)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float v = acc[m][j];
#pragma unroll
            for (int off = TX; off < 64; off <<= 1) v += __shfl_xor(v, off, 64);
            acc[m][j] = v;
        }
    __syncthreads();                                          // xs no longer needed: reuse LDS
    float* red = (float*) smem;                               // [4 waves][M][BN]
    const int wave = tid >> 6;
    const int lane = tid & 63;
    if (lane < TX) {
#pragma unroll
        for (int m = 0; m < M; ++m)
#pragma unroll
            for (int j = 0; j < 4; ++j) red[(wave * M + m) * BN + lane * 4 + j] = acc[m][j];
    }
    __syncthreads();
    for (int idx = tid; idx < M * BN; idx += 256) {
        const int m = idx / BN;
        const int c = idx - m * BN;
        const int n = blockIdx.x * BN + c;
        if (m < rows && n < N) {
            const float v = red[(0 * M + m) * BN + c] + red[(1 * M + m) * BN + c] + red[(2 * M + m) * BN + c] +
                            red[(3 * M + m) * BN + c];
            if (splitk == 1) {
                float r = v;
                if (no_zero) r += (float) out[(size_t) m * N + n];
                out[(size_t) m * N + n] = (f16) r;
            } else {
                slabs[((size_t) blockIdx.y * rows + m) * N + n] = v;
            }
        }
    }
}

// out[m,n] = h( sum_s slabs[s,m,n] (+ out[m,n]) ), slices added in ascending order.
__global__ __launch_bounds__(256) void q4_gemv_reduce_kernel(const float* __restrict__ slabs, f16* __restrict__ out,
                                                             int total, int splitk, int no_zero)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    float v = 0.f;
    for (int s = 0; s < splitk; ++s) v += slabs[(size_t) s * total + i];
    if (no_zero) v += (float) out[i];
    out[i] = (f16) v;
}

template <int M, int TX>
static int launch_cfg(int ch, dim3 grid, size_t smem, hipStream_t s, const f16* x, const Q4Matrix* w, f16* out,
                      float* slabs, int rows, int prows_per_block, int no_zero, int splitk)
{
#define GEMV_ARGS x, (const uint4*) w->qweight, w->qzeros, w->scales, w->x_map, out, slabs, rows, w->height, \
                  w->width, w->groupsize, prows_per_block, no_zero, splitk
    if (ch == 16)     hipLaunchKernelGGL((q4_gemv_kernel<M, TX, 16>), grid, dim3(256), smem, s, GEMV_ARGS);
    else if (ch == 8) hipLaunchKernelGGL((q4_gemv_kernel<M, TX, 8>),  grid, dim3(256), smem, s, GEMV_ARGS);
    else              hipLaunchKernelGGL((q4_gemv_kernel<M, TX, 4>),  grid, dim3(256), smem, s, GEMV_ARGS);
#undef GEMV_ARGS
    EXL_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// T16 layout (the product path): one block = one 16-column tile, 8 waves split the K range, MFMA dot products
// (gemv_t16.h).  Up to 16 activation rows ride along in the M dimension of the MFMA for free.
// ---------------------------------------------------------------------------------------------------------------
T16Matrix t16_view(const Q4Matrix* m)
{
    T16Matrix v;
    v.qw = (const uint4*) m->qweight; v.qzeros = m->qzeros; v.scales = m->scales; v.x_map = m->x_map;
    v.K = m->height; v.N = m->width; v.R = m->height / 8; v.RB = m->height / 128;
    v.gprows = m->groupsize / 8;
    v.gshift = -1;
    if ((v.gprows & (v.gprows - 1)) == 0) { v.gshift = 0; while ((1 << v.gshift) < v.gprows) ++v.gshift; }
    v.G = m->groups;
    return v;
}

#define T16_WAVES 8

template <int U, int NP, bool G16, int MR>
__global__ __launch_bounds__(T16_WAVES * 64) void q4_gemv_t16_kernel(const T16Matrix m, const f16* __restrict__ x,
                                                                      f16* __restrict__ out, int rows, int no_zero,
                                                                      int rb_per_wave, int xstride)
{
    constexpr int NTH = T16_WAVES * 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint4* xs = (uint4*) smem;                                           // [rows][xstride]
    float* red = (float*) (smem + (size_t) rows * xstride * 16);         // [WAVES][MR][16]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    int t = blockIdx.x;
    if ((gridDim.x & 7) == 0) { const int per = gridDim.x >> 3; t = (t & 7) * per + (t >> 3); }   // neighbours share an XCD L2
    T16Wave<U, NP, G16> w;
    const int rb0 = wave * rb_per_wave;
    w.init(m, t, lane, rb0, min(m.RB, rb0 + rb_per_wave));
    w.load_entries(m);
    w.issue(m, 0);
    // activation rows -> LDS (gathered through x_map for act-order weights)
    for (int idx = tid; idx < rows * m.R; idx += NTH) {
        const int mm = idx / m.R, r = idx - mm * m.R;
        const f16* xr = x + (size_t) mm * m.K;
        uint4 v;
        if (m.x_map) {
            const uint4 m0 = *(const uint4*) (m.x_map + r * 8);
            const uint4 m1 = *(const uint4*) (m.x_map + r * 8 + 4);
            f16x8 g;
            g[0] = xr[m0.x]; g[1] = xr[m0.y]; g[2] = xr[m0.z]; g[3] = xr[m0.w];
            g[4] = xr[m1.x]; g[5] = xr[m1.y]; g[6] = xr[m1.z]; g[7] = xr[m1.w];
            v = __builtin_bit_cast(uint4, g);
        } else {
            v = *(const uint4*) (xr + r * 8);
        }
        xs[(size_t) mm * xstride + r] = v;
    }
    __syncthreads();
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    const int arow = MR == 1 ? 0 : min(lane & 15, rows - 1);
    w.run(m, xs + (size_t) arow * xstride, c);
    if (MR == 1) {
        if (lane < 16) red[wave * 16 + lane] = c[0];
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) red[(wave * 16 + (lane >> 4) * 4 + j) * 16 + (lane & 15)] = c[j];
    }
    __syncthreads();
    if (tid < rows * 16) {
        const int mm = tid >> 4, cc = tid & 15;
        float v = 0.f;
#pragma unroll
        for (int wv = 0; wv < T16_WAVES; ++wv) v += MR == 1 ? red[wv * 16 + cc] : red[(wv * 16 + mm) * 16 + cc];
        f16* o = out + (size_t) mm * m.N + t * 16 + cc;
        if (no_zero) v += (float) *o;
        *o = (f16) v;
    }
}

template <bool G16, int MR>
static int launch_t16_cfg(int rbw, dim3 grid, size_t smem, hipStream_t s, const T16Matrix& m, const f16* x, f16* out, int rows,
                          int no_zero, int xstride)
{
    // more than 64 KiB of dynamic LDS (in_features > ~28000, or several rows of a wide matrix) is a per-device opt-in
#define T16_LAUNCH(U, NP) do { auto kfn = q4_gemv_t16_kernel<U, NP, G16, MR>;                                                  \
        if (smem > 64 * 1024) {                                                                                               \
            static bool big[EXL_MAX_DEVICES] = {};                                                                            \
            int dev_ = 0;                                                                                                     \
            EXL_HIP(hipGetDevice(&dev_));                                                                                     \
            if (dev_ >= 0 && dev_ < EXL_MAX_DEVICES && !big[dev_]) {                                                          \
                EXL_HIP(hipFuncSetAttribute((const void*) kfn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));      \
                big[dev_] = true;                                                                                             \
            }                                                                                                                 \
        }                                                                                                                     \
        hipLaunchKernelGGL(kfn, grid, dim3(T16_WAVES * 64), smem, s, m, x, out, rows, no_zero, rbw, xstride); } while (0)
    if (rbw <= 4)       T16_LAUNCH(4, 1);
    else if (rbw <= 8)  T16_LAUNCH(8, 1);
    else if (rbw <= 12) T16_LAUNCH(6, 2);
    else if (rbw <= 24) T16_LAUNCH(6, 4);
    else                T16_LAUNCH(6, 6);
#undef T16_LAUNCH
    EXL_LAUNCH_CHECK();
    return 0;
}

// Shapes the decode kernel covers: 8 waves x 36 row-blocks of K, one activation row staged in LDS.  Anything else (no
// Llama shape: Llama-2-70B's down_proj, K = 28672, is covered) is routed to the MFMA GEMM by the callers (q4_gemv_covers).
#define T16_MAX_RBW 36
#define T16_LDS_BUDGET (160 * 1024)
bool q4_gemv_covers(const Q4Matrix* w)
{
    if (w->layout != EXL_LAYOUT_T16) return true;                    // the generic split-K kernel takes any K
    const int R = w->height / 8, RB = R / 16;
    return (RB + T16_WAVES - 1) / T16_WAVES <= T16_MAX_RBW && (size_t) (R + 1) * 16 + 8 * 1024 <= T16_LDS_BUDGET;
}

static int launch_q4_gemv_t16(const Q4Matrix* w, const f16* x, int rows, f16* out, int no_zero, hipStream_t s)
{
    const T16Matrix m = t16_view(w);
    const int rbw = (m.RB + T16_WAVES - 1) / T16_WAVES;
    EXL_REQUIRE(rbw <= T16_MAX_RBW, EXL_E_UNSUPPORTED, "q4 gemv: in_features %d too large for the decode kernel (max %d)", m.K, T16_MAX_RBW * T16_WAVES * 128);
    const bool g16 = m.gprows % 16 == 0;
    const int ntiles = m.N / 16;
    // LDS budget: rows * xstride * 16 + reduction buffer <= 64 KiB (the default dynamic-LDS limit); more rows go in
    // several launches (this is the op-level path for 2..7 rows; single-token decode runs through decode_fused.hip)
    const int xstride = rows == 1 ? m.R : m.R + 1;
    const int max_rows = (int) ((T16_LDS_BUDGET - 8 * 1024) / ((size_t) (m.R + 1) * 16));
    EXL_REQUIRE(max_rows >= 1, EXL_E_UNSUPPORTED, "q4 gemv: in_features %d does not fit the activation stage", m.K);
    for (int r0 = 0; r0 < rows; r0 += max_rows) {
        const int nr = rows - r0 < max_rows ? rows - r0 : max_rows;
        const f16* xp = x + (size_t) r0 * m.K;
        f16* op = out + (size_t) r0 * m.N;
        const int xs = nr == 1 ? m.R : xstride;
        const size_t smem = (size_t) nr * xs * 16 + (size_t) T16_WAVES * (nr == 1 ? 16 : 256) * sizeof(float);
        int rc;
        if (nr == 1) rc = g16 ? launch_t16_cfg<true, 1>(rbw, dim3(ntiles), smem, s, m, xp, op, nr, no_zero, xs)
                              : launch_t16_cfg<false, 1>(rbw, dim3(ntiles), smem, s, m, xp, op, nr, no_zero, xs);
        else         rc = g16 ? launch_t16_cfg<true, 16>(rbw, dim3(ntiles), smem, s, m, xp, op, nr, no_zero, xs)
                              : launch_t16_cfg<false, 16>(rbw, dim3(ntiles), smem, s, m, xp, op, nr, no_zero, xs);
        if (rc) return rc;
    }
    return 0;
}

int launch_q4_gemv(const Q4Matrix* w, const f16* x, int rows, f16* out, int no_zero, float* ws, size_t ws_floats,
                   hipStream_t s)
{
    if (rows <= 0) return 0;
    if (w->layout == EXL_LAYOUT_T16) {
        EXL_REQUIRE(rows <= 16, EXL_E_UNSUPPORTED, "q4 gemv: rows (%d) > 16", rows);
        return launch_q4_gemv_t16(w, x, rows, out, no_zero, s);
    }
    const int K = w->height, N = w->width;
    EXL_REQUIRE(rows <= 8, EXL_E_UNSUPPORTED, "q4 gemv: rows (%d) > 8", rows);
    EXL_REQUIRE(N % 4 == 0 && K % 8 == 0, EXL_E_UNSUPPORTED, "q4 gemv: need N %% 4 == 0 and K %% 8 == 0");
    EXL_REQUIRE(w->groupsize % 8 == 0, EXL_E_UNSUPPORTED, "q4 gemv: groupsize (%d) must be a multiple of 8", w->groupsize);

    constexpr int TX = 8, TY = 32, BN = 32;
    const int M = rows == 1 ? 1 : rows == 2 ? 2 : rows <= 4 ? 4 : 8;
    const int prow_total = K / 8;
    const int nb_n = (N + BN - 1) / BN;

    // chunk length: at most one group, at most 16 packed rows
    int ch = 16;
    while (ch > 4 && (ch * 8 > w->groupsize || ch * TY > prow_total)) ch >>= 1;
    // split K across blocks until the grid holds >= ~2 blocks per CU, keeping every k-slice busy
    int splitk = 1;
    while (nb_n * splitk < 512 && prow_total / (splitk * 2) >= TY * 4 && splitk < 16) {
        splitk *= 2;
        while (ch > 4 && ch * TY * splitk > prow_total) ch >>= 1;
    }
    // LDS bound: M * nrows * 16 B <= 64 KiB
    int prows_per_block = (prow_total + splitk - 1) / splitk;
    prows_per_block = (prows_per_block + ch - 1) / ch * ch;
    while ((size_t) M * prows_per_block * 16 > 64 * 1024) {
        splitk *= 2;
        prows_per_block = (prow_total + splitk - 1) / splitk;
        prows_per_block = (prows_per_block + ch - 1) / ch * ch;
    }
    splitk = (prow_total + prows_per_block - 1) / prows_per_block;

    float* slabs = nullptr;
    if (splitk > 1) {
        const size_t need = (size_t) splitk * rows * N;
        EXL_REQUIRE(ws && ws_floats >= need, EXL_E_TOO_SMALL, "q4 gemv: workspace too small (%zu < %zu floats)",
                    ws_floats, need);
        slabs = ws;
    }
    const size_t smem_x = (size_t) M * prows_per_block * 16;
    const size_t smem_r = (size_t) 4 * M * BN * sizeof(float);
    const size_t smem = smem_x > smem_r ? smem_x : smem_r;
    dim3 grid(nb_n, splitk);
    int r;
    switch (M) {
        case 1:  r = launch_cfg<1, TX>(ch, grid, smem, s, x, w, out, slabs, rows, prows_per_block, no_zero, splitk); break;
        case 2:  r = launch_cfg<2, TX>(ch, grid, smem, s, x, w, out, slabs, rows, prows_per_block, no_zero, splitk); break;
        case 4:  r = launch_cfg<4, TX>(ch, grid, smem, s, x, w, out, slabs, rows, prows_per_block, no_zero, splitk); break;
        default: r = launch_cfg<8, TX>(ch, grid, smem, s, x, w, out, slabs, rows, prows_per_block, no_zero, splitk); break;
    }
    if (r) return r;
    if (splitk > 1) {
        const int total = rows * N;
        hipLaunchKernelGGL(q4_gemv_reduce_kernel, dim3((total + 255) / 256), dim3(256), 0, s, slabs, out, total, splitk,
                           no_zero);
        EXL_LAUNCH_CHECK();
    }
    return 0;
}
