// On-device sampler: the token choice of the reference's generator (generator.py:91-170 sample(), :344-381
// gen_single_token(), cpu_func/rep_penalty.cpp:36-74 apply_rep_penalty) as ONE kernel behind the decode executor's head
// kernel, so that a captured graph samples the next token and feeds it to the following step with no host work in between.
// The reference does all of this on the host with torch ops between two forward passes (doc/TODO.md:19 lists moving it).
//
//   logits[V] fp32 (device, left by dec_head_kernel)
//   1. repetition penalty over the token history in device memory: walking back from the newest token, every DISTINCT token
//      is penalised once with v = penalty_max, v += (1 - penalty_max) / decay after `sustain` steps (rep_penalty.cpp:36-74)
//   2. logits[banned] = -10000 (gen_single_token: the BOS token), logits = logits / temperature + 1e-8, softmax
//   3. top-k: the k most probable tokens, sorted by probability (ties: lower id first), L1-normalised
//   4. top-p with the min-p cut, exactly the reference's loop: keep tokens while the running sum (a double, as `.item()`
//      yields Python floats there) has not exceeded top_p and the next probability is not below min_p; normalise
//   5. locally typical sampling: order by |sum(p log p) - log p| ascending, keep until the running sum exceeds `typical`
//   6. the draw: inverse CDF over the surviving list IN ITS ORDER with one uniform number per token -- either supplied by
//      the caller (uniforms[position]) or Philox4x32-10(seed, position).  (torch.multinomial's exponential-race draw
//      cannot be reproduced from a uniform stream; oracle/sampler_oracle.py states the same rule, so tokens are comparable
//      one to one.)
// One block of 1024 threads.  1 <= top_k <= 1024 (dec_sample_kernel): selection by a 4-pass radix select on the probability
// bits, the candidates sorted by a bitonic network in LDS.  top_k = 0 -- the reference then sorts the WHOLE vocabulary and does
// not renormalise (generator.py:110-111) -- and top_k > 1024 (dec_sample_big_kernel): the whole vocabulary is
// sorted in a per-device workspace by the same network run in LDS-sized chunks, and the reference's sequential loops (top-p /
// min-p, typical, the draw) walk the sorted list chunk by chunk, so any number of survivors is handled.
#include "common.h"

#define SMP_THREADS 1024
#define SMP_MAXK 1024

struct SamplerArgs {
    float* logits;                 // [vocab] in/out (penalty, ban and temperature are applied in place, as the reference does)
    float* probs;                  // [vocab] scratch
    const int64_t* history;        // [>= position + 2]: tokens 0..position are the sequence so far
    int64_t* history_out;          // same buffer, written at position + 1
    int64_t* token_io;             // the sampled token, where the next decode step reads its input
    const int32_t* pos_dev;        // ALREADY advanced by the head kernel: *pos_dev = position of the new token
    const float* uniforms;         // optional [>= position + 2]: the draw for the token at position p is uniforms[p]
    float* prob_out;               // optional: probability of the sampled token in the final distribution
    int vocab;
    ExlSampler s;
};

__device__ __forceinline__ float smp_block_max(float v, float* red)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float r = red[0];
    for (int i = 1; i < SMP_THREADS / 64; ++i) r = fmaxf(r, red[i]);
    return r;
}

__device__ __forceinline__ double smp_block_sum(double v, double* red)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    double r = 0.0;
    for (int i = 0; i < SMP_THREADS / 64; ++i) r += red[i];          // fixed order: bit-reproducible
    return r;
}

// Philox4x32-10 (Salmon et al., SC'11): counter (position, 0, 0, 0), key (seed lo, seed hi) -> first output word
__device__ __forceinline__ uint32_t smp_philox(uint64_t seed, uint32_t ctr)
{
    uint32_t c0 = ctr, c1 = 0, c2 = 0, c3 = 0, k0 = (uint32_t) seed, k1 = (uint32_t) (seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t) 0xD2511F53u * c0, p1 = (uint64_t) 0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t) (p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t) p1, n2 = (uint32_t) (p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t) p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return c0;
}

// descending bitonic sort of 1024 64-bit keys in LDS
__device__ __forceinline__ void smp_bitonic_desc(unsigned long long* key)
{
    const int t = threadIdx.x;
    for (int size = 2; size <= SMP_MAXK; size <<= 1)
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            __syncthreads();
            const int partner = t ^ stride;
            if (partner > t) {
                const bool desc = (t & size) == 0;
                const unsigned long long a = key[t], b = key[partner];
                if (desc ? a < b : a > b) { key[t] = b; key[partner] = a; }
            }
        }
    __syncthreads();
}

// Steps 1 and 2 of both kernels: repetition penalty, ban, temperature, softmax -> a.probs (logits modified in place).
__device__ __forceinline__ void smp_prepare(const SamplerArgs& a, int n, float* redf, double* redd)
{
    const int tid = threadIdx.x, V = a.vocab;
    float* lg = a.logits;
    // ---- 1. repetition penalty (rep_penalty.cpp:36-74) -----------------------------------------------------------------
    if (a.s.rep_penalty_max != 1.0f && n > 0) {
        const int sustain = a.s.rep_sustain < 0 ? n : a.s.rep_sustain;
        const int decay = a.s.rep_decay;
        const float dv = decay ? (1.0f - a.s.rep_penalty_max) / (float) decay : 0.0f;
        int beg = n - sustain - decay;
        if (beg < 0) beg = 0;
        const int W = n - beg;                                       // steps j = 0 .. W - 1 walk back from the newest token
        for (int j = tid; j < W; j += SMP_THREADS) {
            const int64_t t = a.history[n - 1 - j];
            bool seen = false;                                       // a more recent occurrence took the (larger) penalty already
            for (int q = 0; q < j && !seen; ++q) seen = a.history[n - 1 - q] == t;
            if (seen || t < 0 || t >= V) continue;
            float v = a.s.rep_penalty_max;
            for (int q = sustain; q < j; ++q) v += dv;               // the reference's running `v += dv`, same rounding
            const float l = lg[t];
            lg[t] = l > 0.0f ? l / v : l * v;
        }
        __syncthreads();
    }
    if (tid == 0 && a.s.banned_token >= 0 && a.s.banned_token < V) lg[a.s.banned_token] = -10000.0f;
    __syncthreads();

    // ---- 2. temperature, softmax (generator.py:104-108) ----------------------------------------------------------------
    float mx = -INFINITY;
    for (int i = tid; i < V; i += SMP_THREADS) {
        const float x = lg[i] / a.s.temperature + 1e-8f;
        lg[i] = x;
        mx = fmaxf(mx, x);
    }
    mx = smp_block_max(mx, redf);
    double ssum = 0.0;
    for (int i = tid; i < V; i += SMP_THREADS) {
        const float e = expf(lg[i] - mx);
        a.probs[i] = e;
        ssum += (double) e;
    }
    const float inv = (float) (1.0 / smp_block_sum(ssum, redd));
    for (int i = tid; i < V; i += SMP_THREADS) a.probs[i] *= inv;
    __syncthreads();

}

__global__ __launch_bounds__(SMP_THREADS) void dec_sample_kernel(const SamplerArgs a)
{
    __shared__ unsigned long long key[SMP_MAXK];
    __shared__ float cp[SMP_MAXK];
    __shared__ int ci[SMP_MAXK];
    __shared__ unsigned int hist[256];
    __shared__ double redd[SMP_THREADS / 64];
    __shared__ float redf[SMP_THREADS / 64];
    __shared__ int scan[SMP_THREADS];
    __shared__ unsigned int sh_prefix, sh_need;
    __shared__ int sh_count, sh_n;
    const int tid = threadIdx.x, V = a.vocab;
    const int pos_new = *a.pos_dev;                                  // position the sampled token will take
    const int n = pos_new;                                           // sequence so far = history[0 .. n - 1]
    float* lg = a.logits;

    smp_prepare(a, n, redf, redd);

    // ---- 3. top-k: radix select of the k-th largest probability (non-negative floats order like their bits) ----------
    int k = a.s.top_k < 1 ? 1 : a.s.top_k > SMP_MAXK ? SMP_MAXK : a.s.top_k;
    if (k > V) k = V;
    if (tid == 0) { sh_prefix = 0; sh_need = (unsigned) k; }
    for (int pass = 0; pass < 4; ++pass) {
        const int shift = 24 - 8 * pass;
        if (tid < 256) hist[tid] = 0;
        __syncthreads();
        const unsigned prefix = sh_prefix, himask = pass == 0 ? 0u : 0xFFFFFFFFu << (shift + 8);
        for (int i = tid; i < V; i += SMP_THREADS) {
            const unsigned b = __float_as_uint(a.probs[i]);
            if ((b & himask) == prefix) atomicAdd(&hist[(b >> shift) & 255u], 1u);
        }
        __syncthreads();
        if (tid == 0) {
            unsigned need = sh_need, bin = 255;
            for (;; --bin) {                                         // from the top bin down to the one holding the k-th largest
                if (hist[bin] >= need || bin == 0) break;
                need -= hist[bin];
            }
            sh_prefix = prefix | (bin << shift);
            sh_need = need;
        }
        __syncthreads();
    }
    const unsigned thr = sh_prefix;                                  // bits of the k-th largest probability
    const int ties_wanted = (int) sh_need;                           // how many elements == thr belong to the top k
    // candidates: everything above the threshold, then the lowest-index ties
    if (tid == 0) sh_count = 0;
    key[tid] = 0ull;
    __syncthreads();
    const int chunk = (V + SMP_THREADS - 1) / SMP_THREADS;
    const int i0 = tid * chunk, i1 = min(V, i0 + chunk);
    int my_ties = 0;
    for (int i = i0; i < i1; ++i) {
        const unsigned b = __float_as_uint(a.probs[i]);
        if (b > thr) {
            const int slot = atomicAdd(&sh_count, 1);
            if (slot < SMP_MAXK) key[slot] = ((unsigned long long) b << 32) | (unsigned long long) (0xFFFFFFFFu - (unsigned) i);
        } else if (b == thr) ++my_ties;
    }
    scan[tid] = my_ties;
    __syncthreads();
    if (tid == 0) {                                                  // exclusive scan over the threads' contiguous index ranges
        int run = 0;
        for (int t = 0; t < SMP_THREADS; ++t) { const int c = scan[t]; scan[t] = run; run += c; }
    }
    __syncthreads();
    {
        int rank = scan[tid];
        const int above = sh_count;
        for (int i = i0; i < i1 && rank < ties_wanted; ++i)
            if (__float_as_uint(a.probs[i]) == thr) {
                const int slot = above + rank;
                if (slot < SMP_MAXK) key[slot] = ((unsigned long long) thr << 32) | (unsigned long long) (0xFFFFFFFFu - (unsigned) i);
                ++rank;
            }
    }
    smp_bitonic_desc(key);                                           // probability descending, ties: lower id first
    cp[tid] = __uint_as_float((unsigned) (key[tid] >> 32));
    ci[tid] = (int) (0xFFFFFFFFu - (unsigned) (key[tid] & 0xFFFFFFFFu));
    __syncthreads();

    // ---- 4..6: the surviving list is short (k, usually 40): thread 0 walks it exactly as the reference's Python loops do ----
    if (tid == 0) {
        int m = k;
        float s1 = 0.f;
        for (int i = 0; i < m; ++i) s1 += cp[i];
        s1 = fmaxf(s1, 1e-12f);                                      // F.normalize(p = 1): x / max(||x||_1, eps)
        for (int i = 0; i < m; ++i) cp[i] /= s1;
        if (a.s.top_p > 0.0f) {                                      // generator.py:121-134
            int num = 0;
            double cum = (double) cp[0];
            for (;;) {
                ++num;
                if (num == m) break;
                if (cp[num] < a.s.min_p) break;
                cum += (double) cp[num];
                if (cum > (double) a.s.top_p) break;
            }
            m = num;
            float s2 = 0.f;
            for (int i = 0; i < m; ++i) s2 += cp[i];
            s2 = fmaxf(s2, 1e-12f);
            for (int i = 0; i < m; ++i) cp[i] /= s2;
        }
        sh_n = m;
    }
    __syncthreads();
    if (a.s.typical > 0.0f) {                                        // generator.py:138-161
        const int m = sh_n;
        double part = 0.0;
        float lp = 0.f;
        if (tid < m) { lp = logf(cp[tid] + 1e-10f); part = (double) (cp[tid] * lp); }
        const float neg_entropy = (float) smp_block_sum(part, redd);
        // ascending |neg_entropy - log p|; ties keep the current order.  Sorted descending on the complemented key.
        unsigned long long kk = 0ull;
        if (tid < m) {
            const unsigned dev_bits = __float_as_uint(fabsf(neg_entropy - lp));
            kk = ((unsigned long long) (0xFFFFFFFFu - dev_bits) << 32) | (unsigned long long) (0xFFFFFFFFu - (unsigned) tid);
        }
        __syncthreads();
        key[tid] = kk;
        smp_bitonic_desc(key);
        const int src = tid < m ? (int) (0xFFFFFFFFu - (unsigned) (key[tid] & 0xFFFFFFFFu)) : 0;
        const float p_new = tid < m ? cp[src] : 0.f;
        const int i_new = tid < m ? ci[src] : 0;
        __syncthreads();
        cp[tid] = p_new; ci[tid] = i_new;
        __syncthreads();
        if (tid == 0) {
            int num = 0;
            double cum = (double) cp[0];
            for (;;) {
                ++num;
                if (num == m) break;
                cum += (double) cp[num];
                if (cum > (double) a.s.typical) break;
            }
            float s3 = 0.f;
            for (int i = 0; i < num; ++i) s3 += cp[i];
            s3 = fmaxf(s3, 1e-12f);
            for (int i = 0; i < num; ++i) cp[i] /= s3;
            sh_n = num;
        }
        __syncthreads();
    }
    if (tid == 0) {
        const int m = sh_n;
        float u;
        if (a.uniforms) u = a.uniforms[pos_new];
        else u = (float) (smp_philox(a.s.seed, (uint32_t) pos_new) >> 8) * (1.0f / 16777216.0f);
        double cum = 0.0;
        int pick = m - 1;
        for (int i = 0; i < m; ++i) {
            cum += (double) cp[i];
            if ((double) u < cum) { pick = i; break; }
        }
        const int64_t tok = ci[pick];
        *a.token_io = tok;
        if (a.history_out) a.history_out[pos_new] = tok;
        if (a.prob_out) *a.prob_out = cp[pick];
    }
}


// ---------------------------------------------------------------------------------------------------------------
// top_k = 0 / top_k > 1024: the whole vocabulary sorted in global memory (per-device workspace, exl_sampler_workspace).
// ---------------------------------------------------------------------------------------------------------------
#define SMP_CH 4096                                   // keys per LDS chunk of the big sort (32 KiB)
struct SamplerBig { unsigned long long* key; float* cp; int* ci; float* cp2; int* ci2; };

// Descending bitonic sort of np (a power of two) 64-bit keys in global memory by one block: the network of smp_bitonic_desc on the
// global index; every run of stages whose partners lie inside one SMP_CH-aligned chunk is done in LDS.
__device__ void smp_big_sort(unsigned long long* g, int np, unsigned long long* lds)
{
    const int t = threadIdx.x;
    const int ch = np < SMP_CH ? np : SMP_CH;
    auto chunk_stages = [&](int c, int size, int first_stride) {        // strides first_stride .. 1 of network size `size` inside chunk c
        for (int j = t; j < ch; j += SMP_THREADS) lds[j] = g[c * ch + j];
        for (int stride = first_stride; stride > 0; stride >>= 1) {
            __syncthreads();
            for (int p = t; p < ch / 2; p += SMP_THREADS) {
                const int i = ((p / stride) * 2 * stride) + (p % stride), q = i + stride;
                const bool desc = (((c * ch + i) & size) == 0);
                const unsigned long long x = lds[i], y = lds[q];
                if (desc ? x < y : x > y) { lds[i] = y; lds[q] = x; }
            }
        }
        __syncthreads();
        for (int j = t; j < ch; j += SMP_THREADS) g[c * ch + j] = lds[j];
        __syncthreads();
    };
    for (int c = 0; c < np / ch; ++c)                                    // sizes 2 .. ch: whole sub-networks per chunk
        for (int size = 2; size <= ch; size <<= 1) {
            // (one load / store per size keeps the code small; this path is the rare one)
            chunk_stages(c, size, size >> 1);
        }
    for (int size = 2 * ch; size <= np; size <<= 1) {
        for (int stride = size >> 1; stride >= ch; stride >>= 1) {        // partners in different chunks: global memory
            __syncthreads();
            for (int p = t; p < np / 2; p += SMP_THREADS) {
                const int i = ((p / stride) * 2 * stride) + (p % stride), q = i + stride;
                const bool desc = ((i & size) == 0);
                const unsigned long long x = g[i], y = g[q];
                if (desc ? x < y : x > y) { g[i] = y; g[q] = x; }
            }
        }
        __syncthreads();
        for (int c = 0; c < np / ch; ++c) chunk_stages(c, size, ch >> 1);
    }
}

// Thread 0 walks g[0 .. n) IN ORDER through an LDS staging buffer (a single thread reading global memory element by element
// would pay a memory latency per element); f(i, value) returns false to stop.  All threads must call it.
template <class F>
__device__ __forceinline__ void smp_walk(const float* g, int n, float* stage, int* stop, F&& f)
{
    if (threadIdx.x == 0) *stop = 0;
    for (int base = 0; base < n; base += SMP_THREADS) {
        __syncthreads();
        if (*stop) break;
        if (base + (int) threadIdx.x < n) stage[threadIdx.x] = g[base + threadIdx.x];
        __syncthreads();
        if (threadIdx.x == 0) {
            const int m = min(SMP_THREADS, n - base);
            for (int j = 0; j < m; ++j)
                if (!f(base + j, stage[j])) { *stop = 1; break; }
        }
    }
    __syncthreads();
}

__global__ __launch_bounds__(SMP_THREADS) void dec_sample_big_kernel(const SamplerArgs a, const SamplerBig w)
{
    __shared__ unsigned long long chunk[SMP_CH];
    __shared__ float stage[SMP_THREADS];
    __shared__ double redd[SMP_THREADS / 64];
    __shared__ float redf[SMP_THREADS / 64];
    __shared__ int sh_stop, sh_n;
    __shared__ float sh_f;
    __shared__ double sh_d;
    const int tid = threadIdx.x, V = a.vocab;
    const int pos_new = *a.pos_dev;
    const int n = pos_new;
    smp_prepare(a, n, redf, redd);

    // ---- 3. the whole vocabulary, probability descending, ties: lower id first ------------------------------------------
    int np = 1;
    while (np < V) np <<= 1;
    for (int i = tid; i < np; i += SMP_THREADS)
        w.key[i] = i < V ? ((unsigned long long) __float_as_uint(a.probs[i]) << 32) | (unsigned long long) (0xFFFFFFFFu - (unsigned) i) : 0ull;
    __syncthreads();
    smp_big_sort(w.key, np, chunk);
    int m = a.s.top_k == 0 ? V : (a.s.top_k < V ? a.s.top_k : V);
    float* cp = w.cp; int* ci = w.ci; float* cp2 = w.cp2; int* ci2 = w.ci2;
    for (int i = tid; i < m; i += SMP_THREADS) {
        cp[i] = __uint_as_float((unsigned) (w.key[i] >> 32));
        ci[i] = (int) (0xFFFFFFFFu - (unsigned) (w.key[i] & 0xFFFFFFFFu));
    }
    __syncthreads();
    // F.normalize(p = 1) = x / max(sum, 1e-12), the sum in list order in fp32 (the order the oracle and the small kernel use)
    auto normalize = [&](int cnt) {
        if (tid == 0) sh_f = 0.f;
        smp_walk(cp, cnt, stage, &sh_stop, [&](int, float v) { sh_f += v; return true; });
        const float s1 = fmaxf(sh_f, 1e-12f);
        for (int i = tid; i < cnt; i += SMP_THREADS) cp[i] /= s1;
        __syncthreads();
    };
    if (a.s.top_k != 0) normalize(m);                                  // generator.py:113-114; top_k = 0: torch.sort only (:110-111)

    // ---- 4. top-p with the min-p cut (generator.py:118-134) ------------------------------------------------------------
    if (a.s.top_p > 0.0f) {
        if (tid == 0) { sh_n = 0; sh_d = 0.0; }
        const int list = m;
        smp_walk(cp, list, stage, &sh_stop, [&](int i, float v) {
            if (i == 0) { sh_d = (double) v; sh_n = 1; return list > 1; }      // cum = top_probs[0]; num = 1; (num == length -> break)
            if (v < a.s.min_p) return false;
            sh_d += (double) v;
            if (sh_d > (double) a.s.top_p) return false;
            sh_n = i + 1;
            return i + 1 < list;
        });
        m = sh_n;
        __syncthreads();
        normalize(m);
    }

    // ---- 5. locally typical sampling (generator.py:138-161) ------------------------------------------------------------
    if (a.s.typical > 0.0f) {
        double part = 0.0;
        for (int i = tid; i < m; i += SMP_THREADS) { const float lp = logf(cp[i] + 1e-10f); part += (double) (cp[i] * lp); }
        const float neg_entropy = (float) smp_block_sum(part, redd);
        int np2 = 1;
        while (np2 < m) np2 <<= 1;
        for (int i = tid; i < np2; i += SMP_THREADS) {
            unsigned long long kk = 0ull;
            if (i < m) {
                const unsigned dev_bits = __float_as_uint(fabsf(neg_entropy - logf(cp[i] + 1e-10f)));
                kk = ((unsigned long long) (0xFFFFFFFFu - dev_bits) << 32) | (unsigned long long) (0xFFFFFFFFu - (unsigned) i);
            }
            w.key[i] = kk;
        }
        __syncthreads();
        smp_big_sort(w.key, np2, chunk);                               // ascending deviation, ties keep the list order
        for (int i = tid; i < m; i += SMP_THREADS) {
            const int src = (int) (0xFFFFFFFFu - (unsigned) (w.key[i] & 0xFFFFFFFFu));
            cp2[i] = cp[src]; ci2[i] = ci[src];
        }
        __syncthreads();
        { float* tf = cp; cp = cp2; cp2 = tf; int* ti = ci; ci = ci2; ci2 = ti; }
        if (tid == 0) { sh_n = 0; sh_d = 0.0; }
        const int list = m;
        smp_walk(cp, list, stage, &sh_stop, [&](int i, float v) {
            if (i == 0) { sh_d = (double) v; sh_n = 1; return list > 1; }
            sh_d += (double) v;
            if (sh_d > (double) a.s.typical) return false;
            sh_n = i + 1;
            return i + 1 < list;
        });
        m = sh_n;
        __syncthreads();
        normalize(m);
    }

    // ---- 6. the draw: inverse CDF over the surviving list in its order ---------------------------------------------------
    float u;
    if (a.uniforms) u = a.uniforms[pos_new];
    else u = (float) (smp_philox(a.s.seed, (uint32_t) pos_new) >> 8) * (1.0f / 16777216.0f);
    if (tid == 0) { sh_n = m - 1; sh_d = 0.0; }
    smp_walk(cp, m, stage, &sh_stop, [&](int i, float v) {
        sh_d += (double) v;
        if ((double) u < sh_d) { sh_n = i; return false; }
        return true;
    });
    if (tid == 0) {
        const int pick = sh_n;
        const int64_t tok = ci[pick];
        *a.token_io = tok;
        if (a.history_out) a.history_out[pos_new] = tok;
        if (a.prob_out) *a.prob_out = cp[pick];
    }
}

int launch_dec_sample(float* logits, float* probs, int64_t* history, int64_t* token_io, const int32_t* pos_dev, const float* uniforms,
                      float* prob_out, int vocab, const ExlSampler* s, hipStream_t stream)
{
    EXL_REQUIRE(s->top_k >= 0, EXL_E_INVALID, "device sampler: top_k must be >= 0 (0 = the whole vocabulary, generator.py:110-111), got %d", s->top_k);
    EXL_REQUIRE(s->temperature > 0.f, EXL_E_INVALID, "device sampler: temperature must be positive");
    EXL_REQUIRE(s->rep_penalty_max > 0.f && s->rep_decay >= 0, EXL_E_INVALID, "device sampler: bad repetition-penalty settings");
    SamplerArgs a;
    a.logits = logits; a.probs = probs; a.history = history; a.history_out = history; a.token_io = token_io; a.pos_dev = pos_dev;
    a.uniforms = uniforms; a.prob_out = prob_out; a.vocab = vocab; a.s = *s;
    if (s->top_k >= 1 && (s->top_k < vocab ? s->top_k : vocab) <= SMP_MAXK) {
        hipLaunchKernelGGL(dec_sample_kernel, dim3(1), dim3(SMP_THREADS), 0, stream, a);
        EXL_LAUNCH_CHECK();
        return 0;
    }
    // top_k = 0 or beyond the LDS network: the whole-vocabulary sort in the device's sampler workspace
    EXL_REQUIRE(vocab <= SMP_BIG_LIMIT, EXL_E_UNSUPPORTED, "device sampler: top_k = %d needs the whole-vocabulary sort, which holds at most %d entries (vocabulary %d)",
                s->top_k, SMP_BIG_LIMIT, vocab);
    int dev = 0;
    EXL_HIP(hipGetDevice(&dev));
    void* base = nullptr;
    EXL_TRY(exl_sampler_workspace(dev, smp_big_bytes(vocab), &base));   // grows with the vocabulary (NOT capturable: exl_decoder_create does it up front)
    const size_t np = smp_big_np(vocab);
    SamplerBig w;
    w.key = (unsigned long long*) base;
    w.cp = (float*) (w.key + np);
    w.cp2 = w.cp + np;
    w.ci = (int*) (w.cp2 + np);
    w.ci2 = w.ci + np;
    hipLaunchKernelGGL(dec_sample_big_kernel, dim3(1), dim3(SMP_THREADS), 0, stream, a, w);
    EXL_LAUNCH_CHECK();
    return 0;
}

extern "C" int exl_sample(int device, float* logits_dev, float* probs_scratch_dev, int vocab, int64_t* history_dev, int64_t* token_out_dev,
                          const int32_t* pos_new_dev, const float* uniforms_dev, float* prob_out_dev, const ExlSampler* s, void* stream)
{
    EXL_REQUIRE(device >= 0 && device < EXL_MAX_DEVICES, EXL_E_INVALID, "sample: invalid device index %d", device);
    EXL_REQUIRE(logits_dev && probs_scratch_dev && history_dev && token_out_dev && pos_new_dev && s, EXL_E_INVALID, "sample: null pointer");
    EXL_REQUIRE(vocab > 0, EXL_E_INVALID, "sample: bad vocabulary size");
    int prev = 0;
    EXL_HIP(hipGetDevice(&prev));
    if (prev != device) EXL_HIP(hipSetDevice(device));
    const int rc = launch_dec_sample(logits_dev, probs_scratch_dev, history_dev, token_out_dev, pos_new_dev, uniforms_dev, prob_out_dev, vocab, s,
                                     (hipStream_t) stream);
    if (prev != device) (void) hipSetDevice(prev);
    return rc;
}
