// Compiled CPython binding of the per-token entry points of include/exl_amd.h -- what the reference binds with pybind11
// (/root/reference/exllama_ext/exllama_ext.cpp:743-762: q4_matmul, q4_attn, q4_attn_2, q4_mlp, rms_norm, rope_), same argument
// lists, same checks, same RuntimeError behaviour as exllama_amd/cuda_ext.py's ctypes path (which stays as the A/B reference,
// EXL_NO_FAST_BINDING=1).  Why it exists: a ctypes call with ~35 arguments costs 12-18 us of host time
// (profiles/r03_dropin_host_profile.txt), three of them per layer made 1.4 ms of the drop-in path's 4.4 ms token.  Here a call is
// METH_FASTCALL argument unpacking, at::Tensor accessors (no Python attribute round trips) and the C-ABI call: < 1 us.
// No kernel code lives here; the file links libexl_amd.so and torch's Python library only to read tensor metadata and the
// current HIP stream.
#include <Python.h>
#include <torch/csrc/autograd/python_variable.h>
#include <ATen/core/Tensor.h>
#include <c10/hip/HIPStream.h>
#include <hip/hip_runtime_api.h>

#include "../../../include/exl_amd.h"

namespace {

struct Err { };                                     // a Python error has been set

[[noreturn]] void fail(const char* msg)
{
    PyErr_SetString(PyExc_RuntimeError, msg);
    throw Err();
}

[[noreturn]] void failf(const char* name, const char* what)
{
    PyErr_Format(PyExc_RuntimeError, "%s %s", name, what);
    throw Err();
}

const at::Tensor& tensor(PyObject* o, const char* name)
{
    if (!THPVariable_Check(o)) failf(name, "must be a torch.Tensor");
    return THPVariable_Unpack(o);
}

bool is_none(PyObject* o)                           // None or the meta-device sentinel of cuda_ext.none_tensor (reference: cuda_ext.py:80-82)
{
    if (o == Py_None) return true;
    if (!THPVariable_Check(o)) return false;
    return THPVariable_Unpack(o).device().is_meta();
}

// contiguous fp16 tensor on a HIP device -> data pointer
void* half_ptr(PyObject* o, const char* name, int* dev = nullptr)
{
    const at::Tensor& t = tensor(o, name);
    if (t.scalar_type() != at::kHalf) failf(name, "is incorrect datatype, must be torch.float16");
    if (!t.is_cuda()) failf(name, "must be on a HIP device (exllama_amd has no CPU path)");
    if (!t.is_contiguous()) failf(name, "must be contiguous");
    if (dev) *dev = (int) t.get_device();
    return t.data_ptr();
}

void* opt_half_ptr(PyObject* o, const char* name) { return is_none(o) ? nullptr : half_ptr(o, name); }

int lora_rank(PyObject* a) { return is_none(a) ? 0 : (int) THPVariable_Unpack(a).size(1); }

const int32_t* opt_i32_ptr(PyObject* o, const char* name)
{
    if (is_none(o)) return nullptr;
    const at::Tensor& t = tensor(o, name);
    if (t.scalar_type() != at::kInt || !t.is_cuda()) failf(name, "must be an int32 tensor on a HIP device");
    return (const int32_t*) t.data_ptr();
}

void* handle(PyObject* o, const char* name)
{
    void* p = PyLong_AsVoidPtr(o);
    if (!p || PyErr_Occurred()) { PyErr_Clear(); failf(name, "is not a q4 handle"); }
    return p;
}

long as_long(PyObject* o, const char* name)
{
    const long v = PyLong_AsLong(o);
    if (v == -1 && PyErr_Occurred()) { PyErr_Clear(); failf(name, "must be an integer"); }
    return v;
}

double as_double(PyObject* o, const char* name)
{
    const double v = PyFloat_AsDouble(o);
    if (v == -1.0 && PyErr_Occurred()) { PyErr_Clear(); failf(name, "must be a number"); }
    return v;
}

// Makes the tensors' device current for the native call (the reference: OptionalCUDAGuard, exllama_ext.cpp:213); restored on exit.
struct DevGuard {
    int prev = -1;
    explicit DevGuard(int dev) { int cur = 0; if (hipGetDevice(&cur) == hipSuccess && cur != dev && hipSetDevice(dev) == hipSuccess) prev = cur; }
    ~DevGuard() { if (prev >= 0) (void) hipSetDevice(prev); }
};

void* stream_of(int dev) { return (void*) c10::hip::getCurrentHIPStream((c10::DeviceIndex) dev).stream(); }

void check(int rc, const char* what)
{
    if (rc == 0) return;
    const char* msg = exl_last_error();
    PyErr_Format(PyExc_RuntimeError, "%s: %s", what, (msg && *msg) ? msg : "native call failed");
    throw Err();
}

// Positional + keyword arguments of a METH_FASTCALL | METH_KEYWORDS call -> v[0 .. nmax) in the order of `names` (missing trailing
// optionals: Py_None).  Same TypeErrors a Python def would raise.
constexpr int MAXARGS = 28;
void collect(const char* fn, PyObject* const* args, Py_ssize_t nargs, PyObject* kwnames, const char* const* names, int nmin, int nmax,
             PyObject** v)
{
    if (nargs > nmax) { PyErr_Format(PyExc_TypeError, "%s() takes at most %d arguments (%zd given)", fn, nmax, nargs); throw Err(); }
    for (int i = 0; i < nmax; ++i) v[i] = i < nargs ? args[i] : nullptr;
    const Py_ssize_t nkw = kwnames ? PyTuple_GET_SIZE(kwnames) : 0;
    for (Py_ssize_t k = 0; k < nkw; ++k) {
        PyObject* key = PyTuple_GET_ITEM(kwnames, k);
        int hit = -1;
        for (int i = 0; i < nmax && hit < 0; ++i)
            if (PyUnicode_CompareWithASCIIString(key, names[i]) == 0) hit = i;
        if (hit < 0) { PyErr_Format(PyExc_TypeError, "%s() got an unexpected keyword argument '%U'", fn, key); throw Err(); }
        if (v[hit]) { PyErr_Format(PyExc_TypeError, "%s() got multiple values for argument '%s'", fn, names[hit]); throw Err(); }
        v[hit] = args[nargs + k];
    }
    for (int i = 0; i < nmax; ++i) {
        if (v[i]) continue;
        if (i < nmin) { PyErr_Format(PyExc_TypeError, "%s() missing required argument '%s' (pos %d)", fn, names[i], i + 1); throw Err(); }
        v[i] = Py_None;
    }
}

#define FAST(fn, NMIN, NMAX, ...)                                                                                              \
    static PyObject* fn(PyObject*, PyObject* const* args_, Py_ssize_t nargs_, PyObject* kwnames_) { try {                    \
        static const char* const names_[] = {__VA_ARGS__};                                                                    \
        static_assert(sizeof(names_) / sizeof(names_[0]) == NMAX, "parameter name list");                                   \
        PyObject* a[MAXARGS];                                                                                                \
        collect(#fn, args_, nargs_, kwnames_, names_, NMIN, NMAX, a);
#define FAST_END Py_RETURN_NONE; } catch (const Err&) { return nullptr; } catch (const std::exception& e) { PyErr_SetString(PyExc_RuntimeError, e.what()); return nullptr; } }

// q4_matmul(x, w, out)                                                          exllama_ext.cpp:199-240
FAST(q4_matmul, 3, 3, "x", "w", "out")
    int dev = 0;
    const at::Tensor& x = tensor(a[0], "x");
    void* xp = half_ptr(a[0], "x", &dev);
    DevGuard guard(dev);
    const at::Tensor& out = tensor(a[2], "out");
    void* op = half_ptr(a[2], "out");
    if (x.dim() < 2 || out.dim() < 2 || x.size(0) != out.size(0)) fail("x and out have incompatible shapes");
    int height = 0, width = 0;
    void* w = handle(a[1], "w");
    check(exl_q4_info(w, nullptr, &height, &width, nullptr, nullptr, nullptr), "q4_matmul");
    if (height != x.size(-1)) fail("x and w have incompatible shapes");
    if (width != out.size(-1)) fail("out and w have incompatible shapes");
    check(exl_q4_matmul(w, xp, (int) x.size(0), op, 0, stream_of(dev)), "q4_matmul");
FAST_END

// rms_norm(x, w, out, epsilon)                                                  exllama_ext.cpp:606-643
FAST(rms_norm, 4, 4, "x", "w", "out", "epsilon")
    int dev = 0;
    const at::Tensor& x = tensor(a[0], "x");
    void* xp = half_ptr(a[0], "x", &dev);
    DevGuard guard(dev);
    const at::Tensor& w = tensor(a[1], "w");
    void* wp = half_ptr(a[1], "w");
    const at::Tensor& out = tensor(a[2], "out");
    void* op = half_ptr(a[2], "out");
    if (x.dim() != 2 || x.size(1) != w.size(0)) fail("x and w have incompatible shapes");
    if (out.dim() != 2 || x.size(0) != out.size(0) || x.size(1) != out.size(1)) fail("x and out have incompatible shapes");
    check(exl_rms_norm(xp, wp, op, (float) as_double(a[3], "epsilon"), (int) x.size(0), (int) x.size(1), stream_of(dev)), "rms_norm");
FAST_END

// rope_(x, sin, cos, past_len, num_heads, head_dim[, past_len_dev])             exllama_ext.cpp:647-680
FAST(rope_, 6, 7, "x", "sin", "cos", "past_len", "num_heads", "head_dim", "past_len_dev")
    int dev = 0;
    const at::Tensor& x = tensor(a[0], "x");
    void* xp = half_ptr(a[0], "x", &dev);
    DevGuard guard(dev);
    const at::Tensor& sn = tensor(a[1], "sin");
    void* sp = half_ptr(a[1], "sin");
    const at::Tensor& cs = tensor(a[2], "cos");
    void* cp = half_ptr(a[2], "cos");
    const long head_dim = as_long(a[5], "head_dim");
    if (head_dim != cs.size(-1)) fail("cos table does not match head_dim");
    if (head_dim != sn.size(-1)) fail("sin table does not match head_dim");
    const long bsz = x.size(0);
    const long rows_per_batch = bsz > 0 ? x.numel() / head_dim / bsz : 0;
    check(exl_rope(xp, sp, cp, (int) bsz, (int) rows_per_batch, (int) head_dim, (int) as_long(a[4], "num_heads"),
                   (int) as_long(a[3], "past_len"), opt_i32_ptr(a[6], "past_len_dev"), stream_of(dev)), "rope_");
FAST_END

// q4_attn(x, rms_norm_weight, epsilon, query_states, key_states, value_states, q_proj, k_proj, v_proj, sin, cos, q_len, past_len,
//         num_heads, num_kv_heads, head_dim, key_cache, value_cache, max_seq_len, q_a, q_b, k_a, k_b, v_a, v_b, lora_temp
//         [, past_len_dev])                                                      exllama_ext.cpp:424-528
FAST(q4_attn, 26, 27, "x", "rms_norm_weight", "epsilon", "query_states", "key_states", "value_states", "q_proj", "k_proj", "v_proj", "sin", "cos",
     "q_len", "past_len", "num_heads", "num_kv_heads", "head_dim", "key_cache", "value_cache", "max_seq_len", "q_a", "q_b", "k_a", "k_b",
     "v_a", "v_b", "lora_temp", "past_len_dev")
    for (int i = 3; i <= 4; ++i)                                       // the reference's first two TORCH_CHECKs (exllama_ext.cpp:455-456)
        if (tensor(a[i], names_[i]).scalar_type() != at::kHalf) failf(names_[i], "is incorrect datatype, must be torch.float16");
    int dev = 0;
    void* x = half_ptr(a[0], "x", &dev);
    DevGuard guard(dev);
    void* rw = half_ptr(a[1], "rms_norm_weight");
    const at::Tensor& q = tensor(a[3], "query_states");
    void* qp = half_ptr(a[3], "query_states");
    void* kp = half_ptr(a[4], "key_states");
    void* vp = half_ptr(a[5], "value_states");
    void* sp = half_ptr(a[9], "sin");
    void* cp = half_ptr(a[10], "cos");
    void* kc = half_ptr(a[16], "key_cache");
    void* vc = half_ptr(a[17], "value_cache");
    if (dev < 0) fail("no device index");
    if (q.dim() != 3) fail("query_states must be [bsz, q_len, dim]");
    check(exl_q4_attn(dev, x, rw, (float) as_double(a[2], "epsilon"), qp, kp, vp, handle(a[6], "q_proj"), handle(a[7], "k_proj"),
                      handle(a[8], "v_proj"), sp, cp, (int) q.size(0), (int) as_long(a[11], "q_len"), (int) q.size(2),
                      (int) as_long(a[15], "head_dim"), (int) as_long(a[13], "num_heads"), (int) as_long(a[14], "num_kv_heads"),
                      (int) as_long(a[12], "past_len"), opt_i32_ptr(a[26], "past_len_dev"), kc, vc,
                      (int) as_long(a[18], "max_seq_len"), opt_half_ptr(a[19], "q_a"), opt_half_ptr(a[20], "q_b"), lora_rank(a[19]),
                      opt_half_ptr(a[21], "k_a"), opt_half_ptr(a[22], "k_b"), lora_rank(a[21]), opt_half_ptr(a[23], "v_a"),
                      opt_half_ptr(a[24], "v_b"), lora_rank(a[23]), opt_half_ptr(a[25], "lora_temp"), stream_of(dev)), "q4_attn");
FAST_END

// q4_attn_2(x, attn_output, o_proj, o_a, o_b, lora_temp)                        exllama_ext.cpp:530-563
FAST(q4_attn_2, 6, 6, "x", "attn_output", "o_proj", "o_a", "o_b", "lora_temp")
    int dev = 0;
    void* x = half_ptr(a[0], "x", &dev);
    DevGuard guard(dev);
    const at::Tensor& at_ = tensor(a[1], "attn_output");
    void* ap = half_ptr(a[1], "attn_output");
    // the reference passes x [bsz, q_len, dim] and uses x.size(0) as the row count (rows == 1 there)
    const long height = at_.size(-1) > 0 ? at_.numel() / at_.size(-1) : 0;
    check(exl_q4_attn_2(x, ap, handle(a[2], "o_proj"), (int) height, opt_half_ptr(a[3], "o_a"), opt_half_ptr(a[4], "o_b"), lora_rank(a[3]),
                        opt_half_ptr(a[5], "lora_temp"), stream_of(dev)), "q4_attn_2");
FAST_END

// q4_mlp(x, rms_norm_weight, epsilon, gate, up, down, gate_a, gate_b, up_a, up_b, down_a, down_b, lora_temp)     exllama_ext.cpp:567-602
FAST(q4_mlp, 13, 13, "x", "rms_norm_weight", "epsilon", "gate", "up", "down", "gate_a", "gate_b", "up_a", "up_b", "down_a", "down_b", "lora_temp")
    int dev = 0;
    const at::Tensor& x = tensor(a[0], "x");
    void* xp = half_ptr(a[0], "x", &dev);
    DevGuard guard(dev);
    void* rw = half_ptr(a[1], "rms_norm_weight");
    if (dev < 0) fail("no device index");
    if (x.dim() != 2) fail("x must be [rows, dim]");
    check(exl_q4_mlp(dev, xp, rw, (float) as_double(a[2], "epsilon"), handle(a[3], "gate"), handle(a[4], "up"), handle(a[5], "down"),
                     (int) x.size(0), (int) x.size(1), opt_half_ptr(a[6], "gate_a"), opt_half_ptr(a[7], "gate_b"), lora_rank(a[6]),
                     opt_half_ptr(a[8], "up_a"), opt_half_ptr(a[9], "up_b"), lora_rank(a[8]), opt_half_ptr(a[10], "down_a"),
                     opt_half_ptr(a[11], "down_b"), lora_rank(a[10]), opt_half_ptr(a[12], "lora_temp"), stream_of(dev)), "q4_mlp");
FAST_END

// attention(q, key_cache, value_cache, out, past_len, num_heads[, mask[, past_len_dev]])     (exl_attention; no pybind counterpart)
FAST(attention, 6, 8, "q", "key_cache", "value_cache", "out", "past_len", "num_heads", "mask", "past_len_dev")
    int dev = 0;
    const at::Tensor& q = tensor(a[0], "q");
    void* qp = half_ptr(a[0], "q", &dev);
    DevGuard guard(dev);
    const at::Tensor& kc = tensor(a[1], "key_cache");
    void* kp = half_ptr(a[1], "key_cache");
    void* vp = half_ptr(a[2], "value_cache");
    void* op = half_ptr(a[3], "out");
    if (q.dim() != 3 || kc.dim() != 4) fail("q must be [bsz, q_len, heads * head_dim], caches [bsz, kv_heads, max_seq, head_dim]");
    const long bsz = q.size(0), q_len = q.size(1), kvh = kc.size(1), max_seq = kc.size(2), hd = kc.size(3);
    const long heads = as_long(a[5], "num_heads"), past = as_long(a[4], "past_len");
    if (q.size(2) != heads * hd) fail("q does not match num_heads * head_dim");
    if (bsz > 1 && kc.size(0) != bsz) fail("cache batch size must equal bsz when bsz > 1");
    void* mp = nullptr;
    if (!is_none(a[6])) {
        mp = half_ptr(a[6], "mask");
        if (THPVariable_Unpack(a[6]).numel() != bsz * q_len * (past + q_len)) fail("mask has the wrong shape");
    }
    check(exl_attention(qp, kp, vp, op, mp, (int) bsz, (int) q_len, (int) heads, (int) kvh, (int) hd, (int) max_seq, (int) past,
                        opt_i32_ptr(a[7], "past_len_dev"), stream_of(dev)), "attention");
FAST_END

#define KW (METH_FASTCALL | METH_KEYWORDS)
#define FN(f) (PyCFunction) (void (*)(void)) f
PyMethodDef methods[] = {
    {"q4_matmul", FN(q4_matmul), KW, "q4_matmul($module, /, x, w, out)\n--\n\nexllama_ext.cpp:199-240"},
    {"rms_norm", FN(rms_norm), KW, "rms_norm($module, /, x, w, out, epsilon)\n--\n\nexllama_ext.cpp:606-643"},
    {"rope_", FN(rope_), KW, "rope_($module, /, x, sin, cos, past_len, num_heads, head_dim, past_len_dev=None)\n--\n\nexllama_ext.cpp:647-680"},
    {"q4_attn", FN(q4_attn), KW,
     "q4_attn($module, /, x, rms_norm_weight, epsilon, query_states, key_states, value_states, q_proj, k_proj, v_proj, sin, cos, q_len, past_len, "
     "num_heads, num_kv_heads, head_dim, key_cache, value_cache, max_seq_len, q_a, q_b, k_a, k_b, v_a, v_b, lora_temp, past_len_dev=None)\n--\n\n"
     "exllama_ext.cpp:424-528"},
    {"q4_attn_2", FN(q4_attn_2), KW, "q4_attn_2($module, /, x, attn_output, o_proj, o_a, o_b, lora_temp)\n--\n\nexllama_ext.cpp:530-563"},
    {"q4_mlp", FN(q4_mlp), KW,
     "q4_mlp($module, /, x, rms_norm_weight, epsilon, gate, up, down, gate_a, gate_b, up_a, up_b, down_a, down_b, lora_temp)\n--\n\n"
     "exllama_ext.cpp:567-602"},
    {"attention", FN(attention), KW,
     "attention($module, /, q, key_cache, value_cache, out, past_len, num_heads, mask=None, past_len_dev=None)\n--\n\nexl_attention"},
    {nullptr, nullptr, 0, nullptr}};

PyModuleDef moddef = {PyModuleDef_HEAD_INIT, "_exl_fast", "compiled binding of the per-token entry points of libexl_amd.so", -1, methods};

}  // namespace

PyMODINIT_FUNC PyInit__exl_fast(void) { return PyModule_Create(&moddef); }
