// C-ABI entry points of libexl_amd.so (declared in include/exl_amd.h): argument checking, handle / buffer
// registries, threshold dispatch and the fused decode compositions.  Mirrors the pybind layer of the
// reference, /root/reference/exllama_ext/exllama_ext.cpp, function by function (citations in the header).
#include "common.h"

#include <math.h>
#include <mutex>
#include <string.h>
#include <unordered_set>
#include <unordered_map>
#include <vector>

// ---- errors -------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";

void exl_set_error(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* exl_last_error(void) { return g_err; }
extern "C" int exl_version(void) { return 100; }

// ---- tuning (reference defaults: model.py:94-103) --------------------------------------------------------
ExlTuning g_tuning = {8, 2, 8, 0, 0, 0, 0, 0, 0};

extern "C" int exl_set_tuning(const ExlTuning* t)
{
    EXL_REQUIRE(t, EXL_E_INVALID, "exl_set_tuning: null");
    g_tuning = *t;
    return 0;
}

extern "C" int exl_get_tuning(ExlTuning* t)
{
    EXL_REQUIRE(t, EXL_E_INVALID, "exl_get_tuning: null");
    *t = g_tuning;
    return 0;
}

// ---- per-device buffers --------------------------------------------------------------------------------
static DeviceBuffers g_buffers[EXL_MAX_DEVICES];
static const size_t kWorkspaceFloats = (size_t) 16 * 1024 * 1024;     // 64 MiB

DeviceBuffers* exl_buffers(int device) { return &g_buffers[device]; }

static int ensure_workspace(int device)
{
    DeviceBuffers* b = &g_buffers[device];
    if (b->workspace) return 0;
    int prev = 0;
    EXL_HIP(hipGetDevice(&prev));
    EXL_HIP(hipSetDevice(device));
    EXL_HIP(hipMalloc((void**) &b->workspace, kWorkspaceFloats * sizeof(float)));
    b->workspace_floats = kWorkspaceFloats;
    EXL_HIP(hipSetDevice(prev));
    return 0;
}

int exl_workspace(int device, size_t floats, float** out)
{
    EXL_REQUIRE(device >= 0 && device < EXL_MAX_DEVICES, EXL_E_INVALID, "invalid device index %d", device);
    EXL_TRY(ensure_workspace(device));      // lazily allocated when prepare_buffers was skipped (NOT capturable)
    DeviceBuffers* b = &g_buffers[device];
    EXL_REQUIRE(b->workspace_floats >= floats, EXL_E_TOO_SMALL, "workspace too small (%zu < %zu floats)",
                b->workspace_floats, floats);
    *out = b->workspace;
    return 0;
}

int exl_sampler_workspace(int device, size_t bytes, void** out)
{
    EXL_REQUIRE(device >= 0 && device < EXL_MAX_DEVICES, EXL_E_INVALID, "invalid device index %d", device);
    DeviceBuffers* b = &g_buffers[device];
    if (b->sampler_ws_bytes < bytes) {
        int prev = 0;
        EXL_HIP(hipGetDevice(&prev));
        EXL_HIP(hipSetDevice(device));
        if (b->sampler_ws) (void) hipFree(b->sampler_ws);
        b->sampler_ws = nullptr; b->sampler_ws_bytes = 0;
        const hipError_t e = hipMalloc(&b->sampler_ws, bytes);
        (void) hipSetDevice(prev);
        if (e != hipSuccess) { (void) hipGetLastError(); EXL_FAIL((int) e, "device sampler: cannot allocate its %zu-byte workspace: %s", bytes, hipGetErrorString(e)); }
        b->sampler_ws_bytes = bytes;
    }
    *out = b->sampler_ws;
    return 0;
}

int exl_gemm_workspace(int device, size_t floats, float** out)
{
    EXL_REQUIRE(device >= 0 && device < EXL_MAX_DEVICES, EXL_E_INVALID, "invalid device index %d", device);
    // QUIET on "no room": every caller treats a non-zero return as "run the variant that needs no workspace" (launch_q4_gemm's probe,
    // plan_gemm_tail, the half GEMM) -- a message left in exl_last_error() here would be attached to some later, unrelated failure.
    // Growth is serialised, and a buffer that has been handed out is never freed while the process lives: another host thread may hold
    // the old pointer without having launched yet (the superseded buffers are kept, at most a handful: each growth is >= 1.25 x).
    static std::mutex grow_lock;
    std::lock_guard<std::mutex> hold(grow_lock);
    DeviceBuffers* b = &g_buffers[device];
    if (b->gemm_ws_floats < floats) {
        if (floats > ((size_t) 512 << 20) / sizeof(float)) return EXL_E_TOO_SMALL;
        int prev = 0;
        EXL_HIP(hipGetDevice(&prev));
        EXL_HIP(hipSetDevice(device));
        const size_t want = floats + floats / 4;             // head room: the next shape rarely needs a new allocation
        float* fresh = nullptr;
        const hipError_t e = hipMalloc((void**) &fresh, want * sizeof(float));
        (void) hipSetDevice(prev);
        if (e != hipSuccess) { (void) hipGetLastError(); return EXL_E_TOO_SMALL; }
        static std::vector<float*> retired;                  // (never freed: see above)
        if (b->gemm_ws) retired.push_back(b->gemm_ws);
        b->gemm_ws = fresh;
        b->gemm_ws_floats = want;
    }
    *out = b->gemm_ws;
    return 0;
}

extern "C" int exl_prepare_buffers(int device, void* temp_state, size_t temp_state_numel, void* temp_mlp,
                                   size_t temp_mlp_numel, void* temp_zeros_float, size_t max_zeros_float,
                                   void* temp_dq, size_t temp_dq_numel)
{
    EXL_REQUIRE(device >= 0, EXL_E_INVALID, "no device index");
    EXL_REQUIRE(device < EXL_MAX_DEVICES, EXL_E_INVALID, "invalid device index");
    DeviceBuffers* b = &g_buffers[device];
    b->temp_state = (f16*) temp_state;   b->temp_state_numel = temp_state_numel;
    b->temp_mlp = (f16*) temp_mlp;       b->temp_mlp_numel = temp_mlp_numel;
    b->temp_zeros_float = (float*) temp_zeros_float; b->max_zeros_float = max_zeros_float;
    b->temp_dq = (f16*) temp_dq;         b->temp_dq_numel = temp_dq_numel;
    b->prepared = true;
    return ensure_workspace(device);
}

struct DeviceGuard {
    int prev; bool ok;
    explicit DeviceGuard(int d) : prev(0), ok(false) {
        if (hipGetDevice(&prev) == hipSuccess && (prev == d || hipSetDevice(d) == hipSuccess)) ok = true;
    }
    ~DeviceGuard() { int cur = 0; if (ok && hipGetDevice(&cur) == hipSuccess && cur != prev) (void) hipSetDevice(prev); }
};

// ---- Q4 handles -----------------------------------------------------------------------------------------
static std::vector<Q4Matrix*> g_matrices;
static std::unordered_set<void*> g_live;
static std::unordered_map<const void*, Q4Matrix*> g_by_qweight;   // last handle that rewrote the tensor at this address in place

// Registry state (handles, re-tiled buffers, device pool) is guarded by g_reg_mutex: the reference is single-threaded by
// contract, a loader thread next to a serving thread must not corrupt the tables all the same.
static std::mutex g_reg_mutex;

Q4Matrix* q4_from_handle(void* h)
{
    std::lock_guard<std::mutex> lock(g_reg_mutex);
    if (!h || g_live.find(h) == g_live.end()) return nullptr;
    Q4Matrix* m = (Q4Matrix*) h;
    return m->magic == EXL_Q4_MAGIC ? m : nullptr;
}

static void free_matrix(Q4Matrix* m)
{
    if (m->x_map) {
        int prev = 0;
        if (hipGetDevice(&prev) == hipSuccess) {
            (void) hipSetDevice(m->device);
            (void) hipFree(m->x_map);                        // the reference leaks this (q4_matrix.cu:55-57,108)
            (void) hipSetDevice(prev);
        }
    }
    m->magic = 0;
    delete m;
}

extern "C" int exl_cleanup(void)
{
    std::lock_guard<std::mutex> lock(g_reg_mutex);
    for (Q4Matrix* m : g_matrices) free_matrix(m);
    g_matrices.clear();
    g_live.clear();
    g_by_qweight.clear();
    for (int d = 0; d < EXL_MAX_DEVICES; ++d) {

        DeviceBuffers* b = &g_buffers[d];
        if (b->workspace || b->gemm_ws || b->sampler_ws) {
            int prev = 0;
            if (hipGetDevice(&prev) == hipSuccess) {
                (void) hipSetDevice(d);
                if (b->workspace) (void) hipFree(b->workspace);
                if (b->gemm_ws) (void) hipFree(b->gemm_ws);
                if (b->sampler_ws) (void) hipFree(b->sampler_ws);
                (void) hipSetDevice(prev);
            }
        }
        memset(b, 0, sizeof(*b));
    }
    return 0;
}

extern "C" int exl_make_q4(int device, int height, int width, int groups, uint32_t* qweight, uint32_t* qzeros,
                           uint16_t* scales, const uint32_t* g_idx_host, void* stream, void** out_handle)
{
    EXL_REQUIRE(out_handle, EXL_E_INVALID, "make_q4: out_handle is null");
    *out_handle = nullptr;
    EXL_REQUIRE(device >= 0 && device < EXL_MAX_DEVICES, EXL_E_INVALID, "make_q4: invalid device index %d", device);
    EXL_REQUIRE(qweight && qzeros && scales, EXL_E_INVALID, "make_q4: null tensor pointer");
    EXL_REQUIRE(height > 0 && width > 0 && groups > 0, EXL_E_INVALID, "make_q4: bad shape");
    EXL_REQUIRE(height % 8 == 0, EXL_E_INVALID, "make_q4: height must be a multiple of 8");
    EXL_REQUIRE(width % 8 == 0, EXL_E_INVALID, "make_q4: width must be a multiple of 8 (qzeros packs 8 columns)");
    EXL_REQUIRE(height % groups == 0, EXL_E_INVALID, "w.shape[-2] must be a multiple of zeros.shape[-2]");
    const int groupsize = height / groups;
    EXL_REQUIRE(groupsize % 8 == 0, EXL_E_UNSUPPORTED, "make_q4: groupsize %d is not a multiple of 8", groupsize);

    Q4Matrix* m = new Q4Matrix();
    m->magic = EXL_Q4_MAGIC;
    m->device = device;
    m->height = height;
    m->width = width;
    m->groups = groups;
    m->groupsize = groupsize;
    m->qweight = qweight;
    m->qzeros = qzeros;
    m->scales = (f16*) scales;
    m->x_map = nullptr;
    m->xmap_hash = 0;
    m->layout = EXL_LAYOUT_GPTQ;
    m->fp_valid = false;
    // make_q4 re-tiles (and, with act-order, repacks -- as the reference does, q4_matrix.cu:159) `qweight` IN PLACE: a second
    // make_q4 on the same tensor would re-tile re-tiled words and silently compute garbage.  The address alone cannot decide (handles
    // live until cleanup() while the caller's allocator recycles the addresses of freed tensors), so a handle keeps a fingerprint of
    // the rewritten tensor -- its first and last 16 bytes -- and a call on an address a live handle rewrote is refused when the
    // tensor STILL holds exactly those bytes (a recycled address with a fresh checkpoint tensor does not).
    auto fingerprint = [&](uint32_t (&fp)[8]) -> hipError_t {
        const size_t words = (size_t) height / 8 * width;
        DeviceGuard guard(device);
        hipError_t e = hipMemcpyAsync(fp, qweight, 16, hipMemcpyDeviceToHost, (hipStream_t) stream);
        if (e == hipSuccess) e = hipMemcpyAsync(fp + 4, qweight + (words >= 4 ? words - 4 : 0), 16, hipMemcpyDeviceToHost, (hipStream_t) stream);
        if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t) stream);
        return e;
    };
    {
        // (the previous handle's fingerprint and shape are COPIED while the registry lock is held: a concurrent exl_free_q4 may delete it)
        bool have_prev = false;
        uint32_t prev_fp[8];
        {
            std::lock_guard<std::mutex> lock(g_reg_mutex);
            auto it = g_by_qweight.find(qweight);
            if (it != g_by_qweight.end() && g_live.count(it->second)) {
                const Q4Matrix* ph = it->second;
                if (ph->fp_valid && ph->device == device && ph->height == height && ph->width == width) {
                    memcpy(prev_fp, ph->fp, sizeof(prev_fp));
                    have_prev = true;
                }
            }
        }
        if (have_prev) {
            uint32_t now[8];
            const hipError_t e = fingerprint(now);
            if (e != hipSuccess) { delete m; EXL_FAIL((int) e, "make_q4: %s", hipGetErrorString(e)); }
            if (memcmp(now, prev_fp, sizeof(now)) == 0) {
                delete m;
                EXL_FAIL(EXL_E_INVALID, "make_q4: this qweight tensor was already rewritten in place by a live handle (make_q4 consumes its "
                         "tensor: build a second handle from a fresh copy of the checkpoint tensor)");
            }
        }
    }
    // fingerprint of the tensor BEFORE the rewrite: words the rewrite leaves unchanged (uniform nibbles: all-zero or 0x88888888 synthetic
    // weights) cannot tell a rewritten tensor from a fresh one, so such a handle takes no part in the guard
    uint32_t fp_before[8];
    {
        const hipError_t e = fingerprint(fp_before);
        if (e != hipSuccess) { delete m; EXL_FAIL((int) e, "make_q4: %s", hipGetErrorString(e)); }
    }

    if (g_idx_host) {
        // stable counting sort of rows by group -> x_map (new row -> old row); integer-exact restatement of
        // the reference's host code (q4_matrix.cu:110-139) with 32-bit counters
        std::vector<uint32_t> start(groups + 1, 0), x_map(height);
        for (int i = 0; i < height; ++i) {
            if (g_idx_host[i] >= (uint32_t) groups) { delete m; EXL_FAIL(EXL_E_INVALID, "make_q4: g_idx[%d] = %u out of range", i, g_idx_host[i]); }
            start[g_idx_host[i] + 1]++;
        }
        for (int gidx = 0; gidx < groups; ++gidx) start[gidx + 1] += start[gidx];
        for (int row = 0; row < height; ++row) x_map[start[g_idx_host[row]]++] = (uint32_t) row;
        uint64_t hsh = 1469598103934665603ull;                        // FNV-1a over the map: equal maps <=> (with overwhelming odds) equal hashes
        for (int row = 0; row < height; ++row) { hsh ^= x_map[row]; hsh *= 1099511628211ull; }
        m->xmap_hash = hsh ? hsh : 1;
        m->xmap_host = x_map;
        int prev = 0;
        hipError_t e = hipGetDevice(&prev);
        if (e == hipSuccess) e = hipSetDevice(device);
        if (e != hipSuccess) { delete m; EXL_FAIL((int) e, "make_q4: hipSetDevice(%d) failed: %s", device, hipGetErrorString(e)); }
        const int r = launch_make_sequential(m, x_map.data(), (hipStream_t) stream);
        (void) hipSetDevice(prev);
        if (r) { free_matrix(m); return r; }
    }
    // Re-tile the packed weights in place into the streaming layout of the decode / prefill kernels (gemv_t16.h).
    // Needs whole 16-row blocks and 16-column tiles and groups that are whole 4-row pieces; every Llama shape qualifies.
    if (height % 128 == 0 && width % 16 == 0 && groupsize % 32 == 0) {
        DeviceGuard guard(device);
        if (!guard.ok) { free_matrix(m); EXL_FAIL(EXL_E_INVALID, "make_q4: cannot select device %d", device); }
        const int r = launch_retile_t16(m, (hipStream_t) stream);
        if (r) { free_matrix(m); return r; }
    }
    if (m->x_map || m->layout == EXL_LAYOUT_T16) {                    // the tensor was rewritten: remember what it looks like now
        const hipError_t e = fingerprint(m->fp);
        if (e != hipSuccess) { free_matrix(m); EXL_FAIL((int) e, "make_q4: %s", hipGetErrorString(e)); }
        m->fp_valid = memcmp(m->fp, fp_before, sizeof(fp_before)) != 0;
    }
    {
        std::lock_guard<std::mutex> lock(g_reg_mutex);
        g_matrices.push_back(m);
        g_live.insert(m);
        if (m->fp_valid) g_by_qweight[qweight] = m;
    }
    *out_handle = m;
    return 0;
}

extern "C" int exl_free_q4(void* handle)
{
    Q4Matrix* m = q4_from_handle(handle);
    EXL_REQUIRE(m, EXL_E_INVALID, "free_q4: invalid handle");
    {
        std::lock_guard<std::mutex> lock(g_reg_mutex);
        for (size_t i = 0; i < g_matrices.size(); ++i)
            if (g_matrices[i] == m) { g_matrices.erase(g_matrices.begin() + i); break; }
        g_live.erase(m);
        auto it = g_by_qweight.find(m->qweight);
        if (it != g_by_qweight.end() && it->second == m) g_by_qweight.erase(it);
    }
    free_matrix(m);
    return 0;
}

extern "C" int exl_q4_info(void* handle, int* device, int* height, int* width, int* groups, int* groupsize,
                           const uint32_t** x_map_dev)
{
    Q4Matrix* m = q4_from_handle(handle);
    EXL_REQUIRE(m, EXL_E_INVALID, "q4_info: invalid handle");
    if (device) *device = m->device;
    if (height) *height = m->height;
    if (width) *width = m->width;
    if (groups) *groups = m->groups;
    if (groupsize) *groupsize = m->groupsize;
    if (x_map_dev) *x_map_dev = m->x_map;
    return 0;
}

extern "C" int exl_q4_layout(void* handle, int* layout)
{
    Q4Matrix* m = q4_from_handle(handle);
    EXL_REQUIRE(m && layout, EXL_E_INVALID, "q4_layout: invalid handle");
    *layout = m->layout;
    return 0;
}

// ---- q4 matmul ------------------------------------------------------------------------------------------

static int q4_gemv(Q4Matrix* m, const void* x, int rows, void* out, int no_zero, hipStream_t s)
{
    if (!q4_gemv_covers(m)) {                                // K beyond the decode kernel's reach: the MFMA GEMM takes any K
        DeviceBuffers* b = exl_buffers(m->device);
        return launch_q4_gemm(m, (const f16*) x, rows, (f16*) out, no_zero, b->temp_state, b->temp_state_numel, s);
    }
    float* ws = nullptr;
    EXL_TRY(exl_workspace(m->device, 0, &ws));
    return launch_q4_gemv(m, (const f16*) x, rows, (f16*) out, no_zero, ws, exl_buffers(m->device)->workspace_floats, s);
}

static int q4_gemm(Q4Matrix* m, const void* x, int rows, void* out, int no_zero, f16* tmp, size_t tmp_numel,
                   hipStream_t s)
{
    return launch_q4_gemm(m, (const f16*) x, rows, (f16*) out, no_zero, tmp, tmp_numel, s);
}

// rows < thd (or thd == 0) -> GEMV, exactly the reference's test (exllama_ext.cpp:217); the GEMV handles at
// most 8 rows, so larger row counts fall through to the MFMA kernel even when thd says "never".
static int q4_dispatch(Q4Matrix* m, const void* x, int rows, void* out, int no_zero, f16* tmp, size_t tmp_numel,
                       hipStream_t s)
{
    const int thd = g_tuning.matmul_recons_thd;
    if ((thd == 0 || rows < thd) && rows <= 8) return q4_gemv(m, x, rows, out, no_zero, s);
    return q4_gemm(m, x, rows, out, no_zero, tmp, tmp_numel, s);
}

#define Q4_PROLOGUE(name)                                                                      \
    Q4Matrix* m = q4_from_handle(w);                                                           \
    EXL_REQUIRE(m, EXL_E_INVALID, name ": invalid q4 handle");                                 \
    EXL_REQUIRE(x_height >= 0, EXL_E_INVALID, name ": negative row count");                    \
    if (x_height == 0) return 0;                              /* empty activations: nothing to do */ \
    EXL_REQUIRE(x && out, EXL_E_INVALID, name ": null tensor pointer");                        \
    DeviceGuard guard(m->device);                                                              \
    EXL_REQUIRE(guard.ok, EXL_E_INVALID, name ": cannot select device %d", m->device);         \
    DeviceBuffers* bufs = exl_buffers(m->device);

extern "C" int exl_q4_matmul(void* w, const void* x, int x_height, void* out, int no_zero, void* stream)
{
    Q4_PROLOGUE("q4_matmul")
    return q4_dispatch(m, x, x_height, out, no_zero, bufs->temp_state, bufs->temp_state_numel, (hipStream_t) stream);
}

extern "C" int exl_q4_matmul_gemv(void* w, const void* x, int x_height, void* out, int no_zero, void* stream)
{
    Q4_PROLOGUE("q4_matmul_gemv")
    (void) bufs;
    return q4_gemv(m, x, x_height, out, no_zero, (hipStream_t) stream);
}

extern "C" int exl_q4_matmul_gemm(void* w, const void* x, int x_height, void* out, int no_zero, void* stream)
{
    Q4_PROLOGUE("q4_matmul_gemm")
    return q4_gemm(m, x, x_height, out, no_zero, bufs->temp_state, bufs->temp_state_numel, (hipStream_t) stream);
}

extern "C" int exl_q4_matmul_dual(void* w1, void* w2, const void* x, int x_height, void* out1, void* out2, int silu,
                                  void* stream, int* launched)
{
    EXL_REQUIRE(launched, EXL_E_INVALID, "q4_matmul_dual: launched is null");
    *launched = 0;
    Q4Matrix* m1 = q4_from_handle(w1);
    Q4Matrix* m2 = q4_from_handle(w2);
    EXL_REQUIRE(m1 && m2, EXL_E_INVALID, "q4_matmul_dual: invalid q4 handle");
    EXL_REQUIRE(x_height >= 0, EXL_E_INVALID, "q4_matmul_dual: negative row count");
    if (x_height == 0) { *launched = 1; return 0; }
    EXL_REQUIRE(x && out1 && (silu || out2), EXL_E_INVALID, "q4_matmul_dual: null tensor pointer");
    DeviceGuard guard(m1->device);
    EXL_REQUIRE(guard.ok, EXL_E_INVALID, "q4_matmul_dual: cannot select device %d", m1->device);
    DeviceBuffers* bufs = exl_buffers(m1->device);
    const PromptPrologue pro = {nullptr, 0.f, bufs->temp_state, bufs->temp_state_numel};     // act-order pairs that share a map: one gather
    const int r = launch_q4_gemm_dual(m1, m2, (const f16*) x, x_height, (f16*) out1, (f16*) out2, silu, pro, (hipStream_t) stream);
    if (r == 1) return 0;                                    // not eligible: the caller runs the two products itself
    if (r == 0) *launched = 1;
    return r;
}

extern "C" int exl_q4_attn_prompt(void* wq, void* wk, void* wv, const void* x, const void* norm_w, float eps, int bsz, int q_len,
                                 void* q_out, const void* sin, const void* cos, void* key_cache, void* value_cache, int heads, int kv_heads,
                                 int head_dim, int past_len, int max_seq_len, void* stream, int* launched)
{
    EXL_REQUIRE(launched, EXL_E_INVALID, "q4_qkv_rope_cache: launched is null");
    *launched = 0;
    Q4Matrix* mq = q4_from_handle(wq);
    Q4Matrix* mk = q4_from_handle(wk);
    Q4Matrix* mv = q4_from_handle(wv);
    EXL_REQUIRE(mq && mk && mv, EXL_E_INVALID, "q4_qkv_rope_cache: invalid q4 handle");
    EXL_REQUIRE(bsz >= 0 && q_len >= 0 && past_len >= 0, EXL_E_INVALID, "q4_qkv_rope_cache: negative size");
    if (bsz == 0 || q_len == 0) { *launched = 1; return 0; }
    EXL_REQUIRE(x && q_out && sin && cos && key_cache && value_cache, EXL_E_INVALID, "q4_qkv_rope_cache: null tensor pointer");
    EXL_REQUIRE(past_len + q_len <= max_seq_len, EXL_E_INVALID, "q4_qkv_rope_cache: past_len %d + q_len %d exceeds the cache length %d",
                past_len, q_len, max_seq_len);
    DeviceGuard guard(mq->device);
    EXL_REQUIRE(guard.ok, EXL_E_INVALID, "q4_qkv_rope_cache: cannot select device %d", mq->device);
    DeviceBuffers* bufs = exl_buffers(mq->device);
    const PromptPrologue pro = {(const f16*) norm_w, eps, bufs->temp_state, bufs->temp_state_numel};
    const int r = launch_q4_qkv_rope_cache(mq, mk, mv, (const f16*) x, bsz * q_len, (f16*) q_out, (const f16*) sin, (const f16*) cos,
                                           (f16*) key_cache, (f16*) value_cache, q_len, heads, kv_heads, head_dim, past_len, max_seq_len,
                                           pro, (hipStream_t) stream);
    if (r == 1) return 0;                                    // not eligible: the caller runs the separate ops
    if (r == 0) *launched = 1;
    return r;
}

extern "C" int exl_q4_qkv_rope_cache(void* wq, void* wk, void* wv, const void* x, int bsz, int q_len, void* q_out, const void* sin,
                                     const void* cos, void* key_cache, void* value_cache, int heads, int kv_heads, int head_dim,
                                     int past_len, int max_seq_len, void* stream, int* launched)
{
    return exl_q4_attn_prompt(wq, wk, wv, x, nullptr, 0.f, bsz, q_len, q_out, sin, cos, key_cache, value_cache, heads, kv_heads, head_dim,
                              past_len, max_seq_len, stream, launched);
}

// The MLP half of a decoder layer for a PROMPT (more than 512 rows): x += down( silu(gate(norm(x))) * up(norm(x)) ) -- what the
// reference runs as rms_norm, two q4_matmul, silu_mul and a third q4_matmul with the residual (model.py:266-273, :546-552), here:
// RMSNorm (writing the row order of gate / up when they are act-order matrices sharing a map) -> ONE kernel for both products and
// SiLU * mul -> down_proj GEMM accumulating into x.  `act`: caller's scratch of rows * intermediate halves.
extern "C" int exl_q4_mlp_prompt(void* x, const void* norm_w, float eps, void* gate, void* up, void* down, int rows, void* act,
                                 void* stream, int* launched)
{
    EXL_REQUIRE(launched, EXL_E_INVALID, "q4_mlp_prompt: launched is null");
    *launched = 0;
    Q4Matrix* gm = q4_from_handle(gate);
    Q4Matrix* um = q4_from_handle(up);
    Q4Matrix* dm = q4_from_handle(down);
    EXL_REQUIRE(gm && um && dm, EXL_E_INVALID, "q4_mlp_prompt: invalid q4 handle");
    EXL_REQUIRE(rows >= 0, EXL_E_INVALID, "q4_mlp_prompt: negative row count");
    if (rows == 0) { *launched = 1; return 0; }
    EXL_REQUIRE(x && norm_w && act, EXL_E_INVALID, "q4_mlp_prompt: null tensor pointer");
    EXL_REQUIRE(gm->height == dm->width && um->height == dm->width && dm->height == um->width && gm->width == um->width &&
                gm->device == dm->device && um->device == dm->device, EXL_E_INVALID, "q4_mlp_prompt: incompatible shapes");
    DeviceGuard guard(gm->device);
    EXL_REQUIRE(guard.ok, EXL_E_INVALID, "q4_mlp_prompt: cannot select device %d", gm->device);
    DeviceBuffers* bufs = exl_buffers(gm->device);
    EXL_REQUIRE(bufs->prepared, EXL_E_NO_BUFFERS, "q4_mlp_prompt: prepare_buffers was not called for device %d", gm->device);
    hipStream_t s = (hipStream_t) stream;
    const PromptPrologue pro = {(const f16*) norm_w, eps, bufs->temp_state, bufs->temp_state_numel};
    const int r = launch_q4_gemm_dual(gm, um, (const f16*) x, rows, (f16*) act, nullptr, 1, pro, s);
    if (r == 1) return 0;                                    // not eligible: the caller runs the separate ops
    if (r != 0) return r;
    *launched = 1;
    // (temp_state is free again: the dual kernel read it in stream order before down_proj's gather overwrites it)
    return q4_gemm(dm, act, rows, x, 1, bufs->temp_state, bufs->temp_state_numel, s);
}

// One decoder layer for a SHORT prompt (2 .. EXL_GEMM_SKINNY_MAX = 256 rows: BASELINE configs[0]'s 128-token prompt, a chat turn),
// in place on the residual stream x [bsz * q_len, hidden] -- what the reference runs per layer as rms_norm, three q4_matmul, two
// rope_, the cache copies, ATen attention, q4_matmul, rms_norm, two q4_matmul, silu_mul, q4_matmul from Python (model.py:421-552),
// here ten launches enqueued by ONE call: RMSNorm -> fragment order | q / k / v GEMM | RoPE + cache | attention | re-tile | o_proj
// (+ residual) | RMSNorm -> fragment order | gate / up GEMM + SiLU * mul (fragment-order output) | down_proj (+ residual)
// (q4_gemm_frag.hip: why fragment order).  *launched = 0 and nothing enqueued when the layer is not covered (row count, layout, a map on
// down_proj, maps of q / k / v or gate / up that differ): the caller runs the ops one by one.
extern "C" int exl_q4_layer_prompt(void* x, int bsz, int q_len, int past_len, const void* in_norm_w, const void* post_norm_w, float eps,
                                   void* wq, void* wk, void* wv, void* wo, void* wgate, void* wup, void* wdown, const void* sin, const void* cos,
                                   void* key_cache, void* value_cache, int heads, int kv_heads, int head_dim, int max_seq_len, void* stream,
                                   float* rowsq, size_t rowsq_floats, int rowsq_in_slots, int* rowsq_out_slots, int* launched)
{
    EXL_REQUIRE(launched, EXL_E_INVALID, "q4_layer_prompt: launched is null");
    *launched = 0;
    if (rowsq_out_slots) *rowsq_out_slots = 0;
    Q4Matrix* m[7] = {q4_from_handle(wq), q4_from_handle(wk), q4_from_handle(wv), q4_from_handle(wo), q4_from_handle(wgate),
                      q4_from_handle(wup), q4_from_handle(wdown)};
    for (int i = 0; i < 7; ++i) EXL_REQUIRE(m[i], EXL_E_INVALID, "q4_layer_prompt: invalid q4 handle");
    EXL_REQUIRE(bsz >= 0 && q_len >= 0 && past_len >= 0, EXL_E_INVALID, "q4_layer_prompt: negative size");
    const int rows = bsz * q_len;
    if (rows == 0) { *launched = 1; return 0; }
    EXL_REQUIRE(x && in_norm_w && post_norm_w && sin && cos && key_cache && value_cache, EXL_E_INVALID, "q4_layer_prompt: null tensor pointer");
    EXL_REQUIRE(past_len + q_len <= max_seq_len, EXL_E_INVALID, "q4_layer_prompt: past_len %d + q_len %d exceeds the cache length %d",
                past_len, q_len, max_seq_len);
    static const int skinny_max = getenv("EXL_GEMM_SKINNY_MAX") ? atoi(getenv("EXL_GEMM_SKINNY_MAX")) : 256;
    static const bool off = getenv("EXL_GEMM_NO_FRAG") != nullptr;      // A/B switch: the op-by-op short-prompt path
    if (off || rows < 2 || rows > skinny_max) return 0;
    const int h = m[0]->height, inter = m[4]->width, qd = heads * head_dim, kvd = kv_heads * head_dim;
    if (m[0]->width != qd || m[1]->width != kvd || m[2]->width != kvd || m[3]->height != qd || m[3]->width != h || m[5]->width != inter ||
        m[4]->height != h || m[5]->height != h || m[6]->height != inter || m[6]->width != h || head_dim % 16 != 0) return 0;
    for (int i = 0; i < 7; ++i) if (m[i]->layout != EXL_LAYOUT_T16 || m[i]->device != m[0]->device) return 0;
    if (!q4_same_map(m[1], m[0]) || !q4_same_map(m[2], m[0]) || !q4_same_map(m[5], m[4]) || m[6]->x_map) return 0;
    const Q4Matrix* qkv[3] = {m[0], m[1], m[2]};
    const Q4Matrix* om[1] = {m[3]};
    const Q4Matrix* gu[2] = {m[4], m[5]};
    const Q4Matrix* dm[1] = {m[6]};
    if (!gemm_t16r_covers(3, qkv, rows, 0) || !gemm_t16r_covers(1, om, rows, 0) || !gemm_t16r_covers(2, gu, rows, 1) ||
        !gemm_t16r_covers(1, dm, rows, 0)) return 0;                 // group sizes / widths the fragment-order kernel is not built for
    DeviceGuard guard(m[0]->device);
    EXL_REQUIRE(guard.ok, EXL_E_INVALID, "q4_layer_prompt: cannot select device %d", m[0]->device);
    // scratch (the library's own growing workspace: independent of what prepare_buffers was given)
    const size_t b_xf = frag_bytes(rows, h), b_q = (size_t) rows * qd * 2, b_kv = (size_t) rows * kvd * 2, b_af = frag_bytes(rows, qd),
                 b_act = frag_bytes(rows, inter);
    auto al = [](size_t v) { return (v + 255) & ~(size_t) 255; };
    const size_t b_sq = (size_t) rows * (size_t) (h / 16 + 4) * 4;    // o_proj's per-row partial sums of squares (launch_gemm_t16r: slots <= N / 16)
    const size_t b_ks = gemm_frag_ksplit_floats(rows, h) * 4;         // o_proj / down_proj with K cut over blocks: their fp32 slices
    const size_t total = al(b_xf) + 2 * al(b_q) + 2 * al(b_kv) + al(b_af) + al(b_act) + al(b_sq) + al(b_ks);
    float* wsf = nullptr;
    if (exl_gemm_workspace(m[0]->device, (total + 3) / 4, &wsf) != 0) return 0;
    unsigned char* p = (unsigned char*) wsf;
    void* xf = p; p += al(b_xf);
    f16* q = (f16*) p; p += al(b_q);
    f16* attn = (f16*) p; p += al(b_q);
    f16* k = (f16*) p; p += al(b_kv);
    f16* v = (f16*) p; p += al(b_kv);
    void* af = p; p += al(b_af);
    void* actf = p; p += al(b_act);
    float* osq = (float*) p; p += al(b_sq);
    float* kws = (float*) p;
    float* ws = nullptr;
    EXL_TRY(exl_workspace(m[0]->device, 0, &ws));
    hipStream_t s = (hipStream_t) stream;
    f16* xh = (f16*) x;
    f16* qkv_out[3] = {q, k, v};
    // The sums of squares of the two RMSNorms come from the GEMM that wrote the residual stream last: o_proj's epilogue for the second
    // norm (own scratch), the PREVIOUS layer's down_proj for the first -- through the caller's `rowsq` (rowsq_in_slots > 0: it holds
    // what that call left there for exactly this x; the caller's contract, include/exl_amd.h).  Without them the norm reads whole rows.
    const bool sq_in = rowsq && rowsq_in_slots > 0 && (size_t) rows * (size_t) rowsq_in_slots <= rowsq_floats;
    EXL_TRY(launch_to_frag(xh, (const f16*) in_norm_w, eps, m[0]->x_map, xf, rows, h, s, sq_in ? rowsq : nullptr, sq_in ? rowsq_in_slots : 0));
    int r = launch_gemm_t16r(3, qkv, xf, rows, qkv_out, 0, 0, nullptr, s);
    EXL_REQUIRE(r != 1, EXL_E_UNSUPPORTED, "q4_layer_prompt: q / k / v launch refused after the dry run accepted it");
    if (r) return r;
    EXL_TRY(launch_rope_qk_cache(q, k, v, (f16*) key_cache, (f16*) value_cache, (const f16*) sin, (const f16*) cos, bsz, q_len, heads, kv_heads,
                                 head_dim, max_seq_len, past_len, nullptr, s));
    // the MFMA attention kernels store their output in o_proj's fragment order themselves (no act-order map on o_proj: nothing to gather);
    // otherwise -- fewer than 16 query rows, another head_dim, a map -- row-major, then the re-tile launch
    static const bool no_attn_frag = getenv("EXL_ATTN_NO_FRAG") != nullptr;          // A/B switch
    if (q_len >= 16 && head_dim == 128 && !m[3]->x_map && !no_attn_frag) {
        EXL_TRY(launch_flash_prefill(q, (const f16*) key_cache, (const f16*) value_cache, (f16*) af, bsz, q_len, heads, kv_heads, head_dim, max_seq_len,
                                     past_len, s, 1));
    } else {
        EXL_TRY(launch_attention(q, (const f16*) key_cache, (const f16*) value_cache, attn, nullptr, bsz, q_len, heads, kv_heads, head_dim, max_seq_len,
                                 past_len, nullptr, ws, exl_buffers(m[0]->device)->workspace_floats, s));
        EXL_TRY(launch_to_frag(attn, nullptr, 0.f, m[3]->x_map, af, rows, qd, s));
    }
    f16* o_out[1] = {xh};
    int o_slots = 0;
    r = launch_gemm_t16r(1, om, af, rows, o_out, 1, 0, nullptr, s, 0, osq, &o_slots, kws, b_ks / 4);
    EXL_REQUIRE(r != 1, EXL_E_UNSUPPORTED, "q4_layer_prompt: o_proj not covered behind a covered q / k / v launch");
    if (r) return r;
    EXL_TRY(launch_to_frag(xh, (const f16*) post_norm_w, eps, m[4]->x_map, xf, rows, h, s, osq, o_slots));
    r = launch_gemm_t16r(2, gu, xf, rows, nullptr, 0, 1, actf, s);
    EXL_REQUIRE(r != 1, EXL_E_UNSUPPORTED, "q4_layer_prompt: gate / up not covered behind a covered attention half");
    if (r) return r;
    const bool sq_out = rowsq && rowsq_out_slots && (size_t) rows * (size_t) (h / 16 + 4) <= rowsq_floats;
    int d_slots = 0;
    r = launch_gemm_t16r(1, dm, actf, rows, o_out, 1, 0, nullptr, s, 0, sq_out ? rowsq : nullptr, &d_slots, kws, b_ks / 4);
    EXL_REQUIRE(r != 1, EXL_E_UNSUPPORTED, "q4_layer_prompt: down_proj not covered behind a covered attention half");
    if (r) return r;
    if (sq_out) *rowsq_out_slots = d_slots;
    *launched = 1;
    return 0;
}

// The short-prompt product by itself (what exl_q4_layer_prompt runs four times per layer): xf = fragment order of [RMSNorm(x) * norm_w |
// x], gathered through w[0]'s act-order map, then outs[i] (+)= xf @ W_i, or -- dual -- out_frag = silu(xf @ W_0) * (xf @ W_1) in the
// fragment order a consumer with K = width reads (exl_frag_bytes(rows, width) bytes).  `kernel`: 0 = the launcher's choice,
// 1 = q4_gemm_t16r, 2 = q4_gemm_t16g at any width, 3 / 4 / 5 = its <4, 4> / <4, 2> / <8, 4> block shapes.
extern "C" size_t exl_frag_bytes(int rows, int K) { return rows > 0 && K > 0 ? frag_bytes(rows, K) : 0; }

extern "C" int exl_q4_matmul_frag(void* const* w, int nmat, const void* x, int rows, const void* norm_w, float eps, void* const* outs,
                                  int no_zero, int dual, void* out_frag, int kernel, void* stream, const float* rowsq_in, int rowsq_in_slots,
                                  float* rowsq_out, int* rowsq_out_slots, int* launched)
{
    if (rowsq_out_slots) *rowsq_out_slots = 0;
    EXL_REQUIRE(launched, EXL_E_INVALID, "q4_matmul_frag: launched is null");
    *launched = 0;
    EXL_REQUIRE(w && nmat >= 1 && nmat <= 3 && (!dual || nmat == 2), EXL_E_INVALID, "q4_matmul_frag: 1 .. 3 matrices (dual: 2), got %d", nmat);
    const Q4Matrix* m[3] = {nullptr, nullptr, nullptr};
    for (int i = 0; i < nmat; ++i) {
        m[i] = q4_from_handle(w[i]);
        EXL_REQUIRE(m[i], EXL_E_INVALID, "q4_matmul_frag: invalid q4 handle");
        EXL_REQUIRE(m[i]->device == m[0]->device && m[i]->height == m[0]->height, EXL_E_INVALID, "q4_matmul_frag: matrices differ in device or height");
    }
    EXL_REQUIRE(rows >= 0, EXL_E_INVALID, "q4_matmul_frag: negative row count");
    if (rows == 0) { *launched = 1; return 0; }
    EXL_REQUIRE(x && (dual ? out_frag != nullptr : outs != nullptr), EXL_E_INVALID, "q4_matmul_frag: null tensor pointer");
    for (int i = 1; i < nmat; ++i) if (!q4_same_map(m[i], m[0])) return 0;
    if (rows > 256 || !gemm_t16r_covers(nmat, m, rows, dual)) return 0;
    DeviceGuard guard(m[0]->device);
    EXL_REQUIRE(guard.ok, EXL_E_INVALID, "q4_matmul_frag: cannot select device %d", m[0]->device);
    float* wsf = nullptr;
    const size_t xf_floats = ((frag_bytes(rows, m[0]->height) + 255) & ~(size_t) 255) / 4;
    const size_t ks_floats = nmat == 1 && !dual ? gemm_frag_ksplit_floats(rows, m[0]->width) : 0;
    if (exl_gemm_workspace(m[0]->device, xf_floats + ks_floats, &wsf) != 0) return 0;
    hipStream_t s = (hipStream_t) stream;
    EXL_TRY(launch_to_frag((const f16*) x, (const f16*) norm_w, eps, m[0]->x_map, wsf, rows, m[0]->height, s, rowsq_in, rowsq_in ? rowsq_in_slots : 0));
    f16* o[3] = {nullptr, nullptr, nullptr};
    if (!dual) for (int i = 0; i < nmat; ++i) { o[i] = (f16*) outs[i]; EXL_REQUIRE(o[i], EXL_E_INVALID, "q4_matmul_frag: null output pointer"); }
    const int r = launch_gemm_t16r(nmat, m, wsf, rows, o, no_zero, dual, out_frag, s, kernel, rowsq_out, rowsq_out_slots,
                                   ks_floats ? wsf + xf_floats : nullptr, ks_floats);
    if (r == 1) return 0;
    if (r) return r;
    *launched = 1;
    return 0;
}

extern "C" int exl_q4_matmul_lora(void* w, const void* x, int x_height, void* out, const void* lora_a,
                                  const void* lora_b, int rank, void* lora_temp, void* stream)
{
    Q4_PROLOGUE("q4_matmul_lora")
    EXL_REQUIRE(lora_a && lora_b && lora_temp && rank > 0, EXL_E_INVALID, "q4_matmul_lora: missing LoRA tensors");
    hipStream_t s = (hipStream_t) stream;
    EXL_TRY(launch_half_gemm((const f16*) x, (const f16*) lora_a, (f16*) lora_temp, x_height, m->height, rank, 0, s));
    EXL_TRY(launch_half_gemm((const f16*) lora_temp, (const f16*) lora_b, (f16*) out, x_height, rank, m->width, 0, s));
    return q4_dispatch(m, x, x_height, out, 1, bufs->temp_state, bufs->temp_state_numel, s);
}

extern "C" int exl_q4_reconstruct(void* w, void* out_w16, void* stream)
{
    Q4Matrix* m = q4_from_handle(w);
    EXL_REQUIRE(m && out_w16, EXL_E_INVALID, "q4_reconstruct: invalid argument");
    DeviceGuard guard(m->device);
    return launch_reconstruct(m, (f16*) out_w16, (hipStream_t) stream);
}

extern "C" int exl_column_remap(const void* x, void* x_new, int height, int width, const uint32_t* x_map_dev,
                                void* stream)
{
    EXL_REQUIRE(x && x_new && x_map_dev, EXL_E_INVALID, "column_remap: null pointer");
    EXL_REQUIRE(width % 8 == 0, EXL_E_UNSUPPORTED, "column_remap: width (%d) must be a multiple of 8", width);
    EXL_REQUIRE(height <= 65535, EXL_E_UNSUPPORTED, "column_remap: more than 65535 rows");
    return launch_column_remap((const f16*) x, (f16*) x_new, height, width, x_map_dev, (hipStream_t) stream);
}

extern "C" int exl_half_matmul(const void* x, const void* w, void* out, int height, int dim, int width, int no_zero,
                               void* stream)
{
    EXL_REQUIRE(x && w && out, EXL_E_INVALID, "half_matmul: null pointer");
    return launch_half_gemm((const f16*) x, (const f16*) w, (f16*) out, height, dim, width, no_zero, (hipStream_t) stream);
}

extern "C" int exl_rms_norm(const void* x, const void* w, void* out, float epsilon, int rows, int dim, void* stream)
{
    EXL_REQUIRE(x && w && out, EXL_E_INVALID, "rms_norm: null pointer");
    return launch_rms_norm((const f16*) x, (const f16*) w, (f16*) out, epsilon, rows, dim, (hipStream_t) stream);
}

extern "C" int exl_embedding(const int64_t* ids_dev, const void* table, void* out, int n_ids, int hidden, int vocab, void* stream)
{
    EXL_REQUIRE(ids_dev && table && out, EXL_E_INVALID, "embedding: null pointer");
    return launch_embedding(ids_dev, (const f16*) table, (f16*) out, n_ids, hidden, vocab, (hipStream_t) stream);
}

extern "C" int exl_head_matmul(const void* x, const void* w, float* out, int rows, int hidden, int vocab, void* stream)
{
    EXL_REQUIRE(x && w && out, EXL_E_INVALID, "head_matmul: null pointer");
    const int r = launch_head_rows((const f16*) x, (const f16*) w, out, rows, hidden, vocab, (hipStream_t) stream);   // <= 8 rows: the GEMV
    if (r != 1) return r;
    return launch_head_gemm((const f16*) x, (const f16*) w, out, rows, hidden, vocab, (hipStream_t) stream);            // whole sequences: MFMA GEMM
}

extern "C" int exl_rope(void* x, const void* sin, const void* cos, int bsz, int rows_per_batch, int head_dim,
                        int num_heads, int past_len, const int32_t* past_len_dev, void* stream)
{
    EXL_REQUIRE(x && sin && cos, EXL_E_INVALID, "rope_: null pointer");
    return launch_rope((f16*) x, (const f16*) sin, (const f16*) cos, bsz, rows_per_batch, head_dim, num_heads, past_len,
                       past_len_dev, (hipStream_t) stream);
}

extern "C" int exl_silu_mul(void* x, const void* y, int height, int width, void* stream)
{
    EXL_REQUIRE(x && y, EXL_E_INVALID, "silu_mul: null pointer");
    return launch_silu_mul((f16*) x, (const f16*) y, height, width, (hipStream_t) stream);
}

extern "C" int exl_update_cache(const void* key_states, const void* value_states, void* key_cache, void* value_cache,
                                int bsz, int q_len, int num_kv_heads, int head_dim, int max_seq_len, int past_len,
                                const int32_t* past_len_dev, void* stream)
{
    EXL_REQUIRE(key_states && value_states && key_cache && value_cache, EXL_E_INVALID, "update_cache: null pointer");
    EXL_REQUIRE(past_len >= 0 && past_len + q_len <= max_seq_len, EXL_E_INVALID,
                "update_cache: past_len + q_len (%d) exceeds max_seq_len (%d)", past_len + q_len, max_seq_len);
    return launch_update_cache((const f16*) key_states, (const f16*) value_states, (f16*) key_cache, (f16*) value_cache,
                               bsz, q_len, num_kv_heads, head_dim, max_seq_len, past_len, past_len_dev, (hipStream_t) stream);
}

extern "C" int exl_attention(const void* q, const void* key_cache, const void* value_cache, void* out,
                             const void* mask, int bsz, int q_len, int num_heads, int num_kv_heads, int head_dim,
                             int max_seq_len, int past_len, const int32_t* past_len_dev, void* stream)
{
    EXL_REQUIRE(q && key_cache && value_cache && out, EXL_E_INVALID, "attention: null pointer");
    EXL_REQUIRE((long) bsz * q_len <= 65535, EXL_E_UNSUPPORTED, "attention: bsz * q_len > 65535");
    int device = 0;
    EXL_HIP(hipGetDevice(&device));
    float* ws = nullptr;
    EXL_TRY(exl_workspace(device, 0, &ws));
    return launch_attention((const f16*) q, (const f16*) key_cache, (const f16*) value_cache, (f16*) out,
                            (const f16*) mask, bsz, q_len, num_heads, num_kv_heads, head_dim, max_seq_len, past_len,
                            past_len_dev, ws, exl_buffers(device)->workspace_floats, (hipStream_t) stream);
}

// ---- fused decode ops ---------------------------------------------------------------------------------------
extern "C" int exl_q4_attn(int device, const void* x, const void* rms_norm_weight, float epsilon, void* query_states,
                           void* key_states, void* value_states, void* q_proj, void* k_proj, void* v_proj,
                           const void* sin, const void* cos, int bsz, int q_len, int dim, int head_dim, int num_heads,
                           int num_kv_heads, int past_len, const int32_t* past_len_dev, void* key_cache,
                           void* value_cache, int max_seq_len, const void* q_a, const void* q_b, int q_rank,
                           const void* k_a, const void* k_b, int k_rank, const void* v_a, const void* v_b, int v_rank,
                           void* lora_temp, void* stream)
{
    EXL_REQUIRE(device >= 0, EXL_E_INVALID, "no device index");
    EXL_REQUIRE(device < EXL_MAX_DEVICES, EXL_E_INVALID, "invalid device index");
    Q4Matrix* qm = q4_from_handle(q_proj);
    Q4Matrix* km = q4_from_handle(k_proj);
    Q4Matrix* vm = q4_from_handle(v_proj);
    EXL_REQUIRE(qm && km && vm, EXL_E_INVALID, "q4_attn: invalid q4 handle");
    EXL_REQUIRE(x && rms_norm_weight && query_states && key_states && value_states && sin && cos && key_cache &&
                value_cache, EXL_E_INVALID, "q4_attn: null pointer");
    EXL_REQUIRE(qm->height == dim && km->height == dim && vm->height == dim, EXL_E_INVALID, "x and w have incompatible shapes");
    DeviceBuffers* bufs = exl_buffers(device);
    EXL_REQUIRE(bufs->prepared, EXL_E_NO_BUFFERS, "q4_attn: prepare_buffers was not called for device %d", device);
    const int rows = bsz * q_len;
    EXL_REQUIRE(bufs->temp_state_numel >= (size_t) 2 * rows * dim, EXL_E_TOO_SMALL, "temp_state buffer too small");
    if (!past_len_dev)
        EXL_REQUIRE(past_len >= 0 && past_len + q_len <= max_seq_len, EXL_E_INVALID, "q4_attn: cache overflow (%d + %d > %d)",
                    past_len, q_len, max_seq_len);
    DeviceGuard guard(device);
    hipStream_t s = (hipStream_t) stream;

    f16* temp_x = bufs->temp_state;                              // [rows, dim]
    f16* remap_tmp = bufs->temp_state + (size_t) rows * dim;     // act-order gather scratch for the GEMM path
    const size_t remap_numel = bufs->temp_state_numel - (size_t) rows * dim;
    bool fused = false;
    const int ranks[3] = {(q_a && q_b) ? q_rank : 0, (k_a && k_b) ? k_rank : 0, (v_a && v_b) ? v_rank : 0};
    const bool any_lora = ranks[0] > 0 || ranks[1] > 0 || ranks[2] > 0;
    if (rows == 1 && ranks[0] <= 64 && ranks[1] <= 64 && ranks[2] <= 64) {
        // one token: RMSNorm + q / k / v as ONE launch of the decode executor's kernel (decode_fused.hip: dec_op_gemv); adapters: the
        // executor's two adapter launches behind it (x A for the three matrices, then += (x A) B), 3 launches instead of 13
        Q4Matrix* mats[3] = {qm, km, vm};
        f16* outs[3] = {(f16*) query_states, (f16*) key_states, (f16*) value_states};
        const int r = dec_op_gemv(device, 0, 1, 0, (const f16*) x, (const f16*) rms_norm_weight, epsilon, 3, mats, outs, nullptr, s);
        if (r > 1) return r;
        fused = r == 0;
        if (fused && any_lora) {
            const f16* la[3] = {(const f16*) q_a, (const f16*) k_a, (const f16*) v_a};
            const f16* lb[3] = {(const f16*) q_b, (const f16*) k_b, (const f16*) v_b};
            const int widths[3] = {qm->width, km->width, vm->width};
            EXL_TRY(dec_op_lora(device, 3, (const f16*) x, (const f16*) rms_norm_weight, epsilon, dim, la, lb, ranks, outs, widths, 0, nullptr, s));
        }
    }
    if (!fused) EXL_TRY(launch_rms_norm((const f16*) x, (const f16*) rms_norm_weight, temp_x, epsilon, rows, dim, s));

    struct Proj { Q4Matrix* m; void* out; const void* a; const void* b; int rank; };
    const Proj projs[3] = {{qm, query_states, q_a, q_b, q_rank}, {km, key_states, k_a, k_b, k_rank},
                           {vm, value_states, v_a, v_b, v_rank}};
    for (const Proj& p : projs) {
        if (fused) break;
        int no_zero = 0;
        if (p.a && p.b && p.rank > 0) {
            EXL_REQUIRE(lora_temp, EXL_E_INVALID, "q4_attn: lora_temp missing");
            EXL_TRY(launch_half_gemm(temp_x, (const f16*) p.a, (f16*) lora_temp, rows, dim, p.rank, 0, s));
            EXL_TRY(launch_half_gemm((const f16*) lora_temp, (const f16*) p.b, (f16*) p.out, rows, p.rank, p.m->width, 0, s));
            no_zero = 1;
        }
        if (rows <= 8) EXL_TRY(q4_gemv(p.m, temp_x, rows, p.out, no_zero, s));
        else           EXL_TRY(q4_gemm(p.m, temp_x, rows, p.out, no_zero, remap_tmp, remap_numel, s));
    }
    // RoPE on q, RoPE on k, K / V rows into the cache: one launch (three in the reference, q4_attn.cu:160-204)
    return launch_rope_qk_cache((f16*) query_states, (f16*) key_states, (const f16*) value_states, (f16*) key_cache, (f16*) value_cache,
                                (const f16*) sin, (const f16*) cos, bsz, q_len, num_heads, num_kv_heads, head_dim, max_seq_len,
                                past_len, past_len_dev, s);
}

extern "C" int exl_q4_attn_2(void* x, const void* attn_output, void* o_proj, int height, const void* o_a,
                             const void* o_b, int o_rank, void* lora_temp, void* stream)
{
    Q4Matrix* om = q4_from_handle(o_proj);
    EXL_REQUIRE(om, EXL_E_INVALID, "q4_attn_2: invalid q4 handle");
    EXL_REQUIRE(x && attn_output, EXL_E_INVALID, "q4_attn_2: null pointer");
    DeviceGuard guard(om->device);
    DeviceBuffers* bufs = exl_buffers(om->device);
    hipStream_t s = (hipStream_t) stream;
    const int orank = (o_a && o_b) ? o_rank : 0;
    if (height == 1 && orank <= 64) {                            // one token: o_proj + residual through the executor's kernel, the adapter behind it
        Q4Matrix* mats[1] = {om};
        const int r = dec_op_gemv(om->device, 1, 0, 1, (const f16*) attn_output, nullptr, 0.f, 1, mats, nullptr, (f16*) x, s);
        if (r > 1) return r;
        if (r == 0) {
            if (orank <= 0) return 0;
            const f16* la[1] = {(const f16*) o_a};
            const f16* lb[1] = {(const f16*) o_b};
            f16* outs[1] = {(f16*) x};
            const int widths[1] = {om->width};
            return dec_op_lora(om->device, 1, (const f16*) attn_output, nullptr, 0.f, om->height, la, lb, &orank, outs, widths, 0, nullptr, s);
        }
    }
    if (orank > 0) {
        EXL_REQUIRE(lora_temp, EXL_E_INVALID, "q4_attn_2: lora_temp missing");
        EXL_TRY(launch_half_gemm((const f16*) attn_output, (const f16*) o_a, (f16*) lora_temp, height, om->height, o_rank, 0, s));
        EXL_TRY(launch_half_gemm((const f16*) lora_temp, (const f16*) o_b, (f16*) x, height, o_rank, om->width, 1, s));
    }
    if (height <= 8) return q4_gemv(om, attn_output, height, x, 1, s);
    return q4_gemm(om, attn_output, height, x, 1, bufs->temp_state, bufs->temp_state_numel, s);
}

extern "C" int exl_q4_mlp(int device, void* x, const void* rms_norm_weight, float epsilon, void* gate, void* up,
                          void* down, int height, int dim, const void* gate_a, const void* gate_b, int gate_rank,
                          const void* up_a, const void* up_b, int up_rank, const void* down_a, const void* down_b,
                          int down_rank, void* lora_temp, void* stream)
{
    EXL_REQUIRE(device >= 0, EXL_E_INVALID, "no device index");
    EXL_REQUIRE(device < EXL_MAX_DEVICES, EXL_E_INVALID, "invalid device index");
    Q4Matrix* gm = q4_from_handle(gate);
    Q4Matrix* um = q4_from_handle(up);
    Q4Matrix* dm = q4_from_handle(down);
    EXL_REQUIRE(gm && um && dm, EXL_E_INVALID, "q4_mlp: invalid q4 handle");
    EXL_REQUIRE(x && rms_norm_weight, EXL_E_INVALID, "q4_mlp: null pointer");
    EXL_REQUIRE(gm->height == dim && um->height == dim && dm->width == dim && dm->height == um->width &&
                gm->width == um->width, EXL_E_INVALID, "q4_mlp: incompatible shapes");
    DeviceBuffers* bufs = exl_buffers(device);
    EXL_REQUIRE(bufs->prepared, EXL_E_NO_BUFFERS, "q4_mlp: prepare_buffers was not called for device %d", device);
    EXL_REQUIRE(bufs->temp_state_numel >= (size_t) 2 * height * dim, EXL_E_TOO_SMALL, "temp_state buffer too small");
    const int inter = um->width;
    EXL_REQUIRE(bufs->temp_mlp_numel >= (size_t) 2 * height * inter, EXL_E_TOO_SMALL, "temp_mlp buffer too small");
    DeviceGuard guard(device);
    hipStream_t s = (hipStream_t) stream;

    f16* temp_x = bufs->temp_state;
    f16* remap_tmp = bufs->temp_state + (size_t) height * dim;
    const size_t remap_numel = bufs->temp_state_numel - (size_t) height * dim;
    f16* t0 = bufs->temp_mlp;
    f16* t1 = bufs->temp_mlp + (size_t) height * inter;
    const int gr = (gate_a && gate_b) ? gate_rank : 0, ur = (up_a && up_b) ? up_rank : 0, dr = (down_a && down_b) ? down_rank : 0;
    if (height == 1 && gr <= 64 && ur <= 64 && dr <= 64) {
        // one token: RMSNorm + gate / up + SiLU in one launch, down_proj + residual in the next (the executor's kernels).  Adapters on
        // gate / up: the two products go out un-fused and the executor's adapter launches add theirs and apply SiLU * mul; an adapter
        // on down_proj: the same pair of launches behind its GEMV.  6-8 launches where the separate products took ~20
        Q4Matrix* gu[2] = {gm, um};
        int r;
        if (gr > 0 || ur > 0) {
            f16* raw[2] = {t0, t1};
            r = dec_op_gemv(device, 2, 1, 0, (const f16*) x, (const f16*) rms_norm_weight, epsilon, 2, gu, raw, nullptr, s);
            if (r > 1) return r;
            if (r == 0) {
                const f16* la[2] = {(const f16*) gate_a, (const f16*) up_a};
                const f16* lb[2] = {(const f16*) gate_b, (const f16*) up_b};
                const int rk[2] = {gr, ur}, widths[2] = {inter, inter};
                EXL_TRY(dec_op_lora(device, 2, (const f16*) x, (const f16*) rms_norm_weight, epsilon, dim, la, lb, rk, raw, widths, 1, t0, s));
            }
        } else {
            f16* outs[1] = {t0};
            r = dec_op_gemv(device, 2, 1, 2, (const f16*) x, (const f16*) rms_norm_weight, epsilon, 2, gu, outs, nullptr, s);
            if (r > 1) return r;
        }
        if (r == 0) {
            Q4Matrix* dn[1] = {dm};
            const int r2 = dec_op_gemv(device, 3, 0, 1, t0, nullptr, 0.f, 1, dn, nullptr, (f16*) x, s);
            if (r2 > 1) return r2;
            if (r2 == 1) EXL_TRY(q4_gemv(dm, t0, height, x, 1, s));  // (down_proj alone not covered: its own GEMV on the fused activation)
            if (dr > 0) {
                const f16* la[1] = {(const f16*) down_a};
                const f16* lb[1] = {(const f16*) down_b};
                f16* outs[1] = {(f16*) x};
                const int widths[1] = {dim};
                return dec_op_lora(device, 1, t0, nullptr, 0.f, inter, la, lb, &dr, outs, widths, 0, nullptr, s);
            }
            return 0;
        }
    }
    EXL_TRY(launch_rms_norm((const f16*) x, (const f16*) rms_norm_weight, temp_x, epsilon, height, dim, s));

    int gz = 0, uz = 0;
    if (gate_a && gate_b && gate_rank > 0) {
        EXL_REQUIRE(lora_temp, EXL_E_INVALID, "q4_mlp: lora_temp missing");
        EXL_TRY(launch_half_gemm(temp_x, (const f16*) gate_a, (f16*) lora_temp, height, dim, gate_rank, 0, s));
        EXL_TRY(launch_half_gemm((const f16*) lora_temp, (const f16*) gate_b, t0, height, gate_rank, inter, 0, s));
        gz = 1;
    }
    if (up_a && up_b && up_rank > 0) {
        EXL_REQUIRE(lora_temp, EXL_E_INVALID, "q4_mlp: lora_temp missing");
        EXL_TRY(launch_half_gemm(temp_x, (const f16*) up_a, (f16*) lora_temp, height, dim, up_rank, 0, s));
        EXL_TRY(launch_half_gemm((const f16*) lora_temp, (const f16*) up_b, t1, height, up_rank, inter, 0, s));
        uz = 1;
    }
    if (height <= 8) {
        EXL_TRY(q4_gemv(gm, temp_x, height, t0, gz, s));
        EXL_TRY(q4_gemv(um, temp_x, height, t1, uz, s));
    } else {
        EXL_TRY(q4_gemm(gm, temp_x, height, t0, gz, remap_tmp, remap_numel, s));
        EXL_TRY(q4_gemm(um, temp_x, height, t1, uz, remap_tmp, remap_numel, s));
    }
    EXL_TRY(launch_silu_mul(t0, t1, height, inter, s));
    if (down_a && down_b && down_rank > 0) {
        EXL_REQUIRE(lora_temp, EXL_E_INVALID, "q4_mlp: lora_temp missing");
        EXL_TRY(launch_half_gemm(t0, (const f16*) down_a, (f16*) lora_temp, height, inter, down_rank, 0, s));
        EXL_TRY(launch_half_gemm((const f16*) lora_temp, (const f16*) down_b, (f16*) x, height, down_rank, dim, 1, s));
    }
    // the down projection's act-order gather may not reuse temp_x's region: use temp_state from the start
    if (height <= 8) return q4_gemv(dm, t0, height, x, 1, s);
    return q4_gemm(dm, t0, height, x, 1, bufs->temp_state, bufs->temp_state_numel, s);
}

// ---- repetition penalty on the host (fp32, deterministic) -----------------------------------------------------
// Semantics of /root/reference/exllama_ext/cpu_func/rep_penalty.cpp:5-31 and :36-74: walk the
// sequence backwards; the penalty stays at penalty_max for `sustain` tokens, then ramps linearly to 1 over `decay`.
struct PenaltyRamp {
    float v, dv; int s, beg;
    PenaltyRamp(float penalty_max, int sustain, int decay, int seq_len) {
        v = penalty_max;
        dv = decay ? (1.0f - penalty_max) / (float) decay : 0.0f;
        s = sustain == -1 ? seq_len : sustain;
        beg = seq_len - s - decay;
        if (beg < 0) beg = 0;
    }
    inline void step() { if (--s < 0) v += dv; }
};

extern "C" int exl_rep_penalty(int vocab_size, const uint64_t* sequence_host, float* rep_mask_host, float penalty_max,
                               int sustain, int decay, int seq_len)
{
    EXL_REQUIRE(rep_mask_host && (sequence_host || seq_len == 0), EXL_E_INVALID, "rep_penalty: null pointer");
    for (int i = 0; i < vocab_size; ++i) rep_mask_host[i] = 1.0f;
    PenaltyRamp ramp(penalty_max, sustain, decay, seq_len);
    for (int i = seq_len - 1; i >= ramp.beg; --i) {
        const uint64_t t = sequence_host[i];
        EXL_REQUIRE(t < (uint64_t) vocab_size, EXL_E_INVALID, "rep_penalty: token %llu outside the vocabulary", (unsigned long long) t);
        if (ramp.v > rep_mask_host[t]) rep_mask_host[t] = ramp.v;
        ramp.step();
    }
    return 0;
}

extern "C" int exl_apply_rep_penalty(int vocab_size, const uint64_t* sequence_host, float penalty_max, int sustain,
                                     int decay, int seq_len, int bsz, float* logits_host)
{
    EXL_REQUIRE(logits_host && (sequence_host || seq_len == 0), EXL_E_INVALID, "apply_rep_penalty: null pointer");
    std::vector<unsigned char> seen((size_t) vocab_size);
    for (int b = 0; b < bsz; ++b) {
        const uint64_t* seq = sequence_host + (size_t) b * seq_len;
        float* logits = logits_host + (size_t) b * vocab_size;
        std::fill(seen.begin(), seen.end(), 0);
        PenaltyRamp ramp(penalty_max, sustain, decay, seq_len);
        for (int i = seq_len - 1; i >= ramp.beg; --i) {
            const uint64_t t = seq[i];
            EXL_REQUIRE(t < (uint64_t) vocab_size, EXL_E_INVALID, "apply_rep_penalty: token %llu outside the vocabulary", (unsigned long long) t);
            if (!seen[t]) {
                if (logits[t] > 0.0f) logits[t] /= ramp.v; else logits[t] *= ramp.v;
                seen[t] = 1;
            }
            ramp.step();
        }
    }
    return 0;
}
