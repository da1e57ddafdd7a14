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
// One block of 1024 threads; selection by a 4-pass radix select on the probability bits, candidates (<= 1024) sorted by a
// bitonic network in LDS.  top_k = 0 (sort the whole vocabulary) is not offered on the device: 1 <= top_k <= 1024.
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

int launch_dec_sample(float* logits, float* probs, int64_t* history, int64_t* token_io, const int32_t* pos_dev, const float* uniforms,
                      float* prob_out, int vocab, const ExlSampler* s, hipStream_t stream)
{
    EXL_REQUIRE(s->top_k >= 1 && s->top_k <= SMP_MAXK, EXL_E_UNSUPPORTED,
                "device sampler: top_k must be in 1..%d (got %d; top_k = 0, a sort of the whole vocabulary, stays on the host path)", SMP_MAXK, s->top_k);
    EXL_REQUIRE(s->temperature > 0.f, EXL_E_INVALID, "device sampler: temperature must be positive");
    EXL_REQUIRE(s->rep_penalty_max > 0.f && s->rep_decay >= 0, EXL_E_INVALID, "device sampler: bad repetition-penalty settings");
    SamplerArgs a;
    a.logits = logits; a.probs = probs; a.history = history; a.history_out = history; a.token_io = token_io; a.pos_dev = pos_dev;
    a.uniforms = uniforms; a.prob_out = prob_out; a.vocab = vocab; a.s = *s;
    hipLaunchKernelGGL(dec_sample_kernel, dim3(1), dim3(SMP_THREADS), 0, stream, a);
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
