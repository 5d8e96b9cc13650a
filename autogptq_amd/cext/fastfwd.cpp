/* fastfwd.cpp -- the eager per-call path of QuantLinear.forward / forward_multi in C++ (CPython API + ATen; no device code, no compute).
 *
 * The reference's callers are eager: generate() under inference_mode calls every QuantLinear once per token (auto_gptq/modeling/_base.py:415-418).  A decode
 * launch of this library runs for 4.5 - 11 us; round 4 measured 8.8 us of HOST time per eager call (attribute reads, torch.empty through the Python
 * argument parser, data_ptr(), the current-stream query, the trampoline), i.e. the eager stack was host-bound at 3.0 of the 3.5 TB/s the same launches
 * reach under hipGraph replay.  This module does the per-call part -- checks on x, the output allocation (at::empty on x's options), the current HIP
 * stream, the C-ABI call -- behind ONE METH_FASTCALL entry per kind of call; everything that depends only on the layer is resolved once by post_init.
 * It calls the SAME gptq_forward_ex / gptq_forward_multi_ex of the SAME loaded libgptq_mi355x.so (addresses handed over by _lib.py), and answers None
 * whenever a call is not the plain case (other device, dtype cast, non-contiguous x, a tuning struct): the Python path then does what it always did.
 * Optional: without it (not built, or a torch it was not built for) the Python path is the only path.                                                        */
#define PY_SSIZE_T_CLEAN
#include <Python.h>

#include <ATen/ATen.h>
#include <c10/hip/HIPStream.h>
#include <torch/csrc/autograd/python_variable.h>

typedef int (*fwd_ex_t)(const void*, const void*, void*, int, void*, size_t, void*, const void*);
typedef int (*fwd_multi_t)(const void* const*, int, const void*, void* const*, int, void*, size_t, void*, const void*);

static fwd_ex_t g_forward_ex = nullptr;
static fwd_multi_t g_forward_multi_ex = nullptr;

static bool as_u64(PyObject* o, unsigned long long* v) {
    *v = PyLong_AsUnsignedLongLongMask(o);
    return !(*v == (unsigned long long)-1 && PyErr_Occurred());
}

/* bind(forward_ex_addr, forward_multi_ex_addr) */
static PyObject* bind(PyObject*, PyObject* const* args, Py_ssize_t nargs) {
    unsigned long long a = 0, b = 0;
    if (nargs != 2 || !as_u64(args[0], &a) || !as_u64(args[1], &b) || !a || !b) {
        if (!PyErr_Occurred()) PyErr_SetString(PyExc_TypeError, "bind(forward_ex_addr, forward_multi_ex_addr)");
        return nullptr;
    }
    g_forward_ex = (fwd_ex_t)(size_t)a;
    g_forward_multi_ex = (fwd_multi_t)(size_t)b;
    Py_RETURN_NONE;
}

/* the plain case: a CUDA tensor of the layer's dtype on the layer's device, which is the current device; [.., K] contiguous */
static inline bool plain_x(const at::Tensor& x, long K, long dtype_code, long dev_index, int64_t* M) {
    if (!x.is_cuda() || (long)x.scalar_type() != dtype_code || x.dim() < 1 || x.size(-1) != K || !x.is_contiguous()) return false;
    if (x.get_device() != dev_index || c10::hip::current_device() != dev_index) return false;
    *M = K ? x.numel() / K : 0;
    return *M > 0 && *M <= 2147483647;
}

/* dtype_code(tensor) -> the at::ScalarType of a tensor as an int (what forward() compares x against) */
static PyObject* dtype_code(PyObject*, PyObject* const* args, Py_ssize_t nargs) {
    if (nargs != 1 || !THPVariable_Check(args[0])) { PyErr_SetString(PyExc_TypeError, "dtype_code(tensor)"); return nullptr; }
    return PyLong_FromLong((long)THPVariable_Unpack(args[0]).scalar_type());
}

/* Row counts 1..63 of a layer / group that are KNOWN to need no workspace arrive as a bit mask (maintained by the Python path, which computes
 * gptq_workspace_bytes once per row count): the fast path serves exactly those -- decode rows on full-width layers -- and never touches the
 * per-(device, stream) workspace registry. */
static inline bool no_workspace(unsigned long long mask, int64_t M) { return M < 64 && ((mask >> M) & 1ull); }

/* forward(layer_addr, x, K, n_out, dtype_code, dev_index, ws0_mask) -> Tensor [.., n_out] | None (not the plain case) | int (a non-zero status) */
static PyObject* forward(PyObject*, PyObject* const* args, Py_ssize_t nargs) {
    if (nargs != 7 || !g_forward_ex) { PyErr_SetString(PyExc_TypeError, "forward: 7 arguments after bind()"); return nullptr; }
    if (!THPVariable_Check(args[1])) Py_RETURN_NONE;
    unsigned long long layer, mask;
    const unsigned long long ws_ptr = 0, ws_bytes = 0;
    if (!as_u64(args[6], &mask)) return nullptr;
    if (!mask) Py_RETURN_NONE;
    const long K = PyLong_AsLong(args[2]), n_out = PyLong_AsLong(args[3]), dtype_code = PyLong_AsLong(args[4]), dev_index = PyLong_AsLong(args[5]);
    if (!as_u64(args[0], &layer) || PyErr_Occurred()) return nullptr;
    const at::Tensor& x = THPVariable_Unpack(args[1]);
    int64_t M = 0;
    if (!plain_x(x, K, dtype_code, dev_index, &M) || !no_workspace(mask, M)) Py_RETURN_NONE;
    try {
        std::vector<int64_t> shape(x.sizes().begin(), x.sizes().end());
        shape.back() = n_out;
        at::Tensor out = at::empty(shape, x.options());
        void* st = (void*)c10::hip::getCurrentHIPStream((c10::DeviceIndex)dev_index).stream();
        const int rc = g_forward_ex((const void*)(size_t)layer, x.data_ptr(), out.data_ptr(), (int)M, (void*)(size_t)ws_ptr, (size_t)ws_bytes, st, nullptr);
        if (rc) return PyLong_FromLong(rc);
        return THPVariable_Wrap(std::move(out));
    } catch (const std::exception& e) {
        PyErr_SetString(PyExc_RuntimeError, e.what());
        return nullptr;
    }
}

/* forward_multi(layers_array_addr, n, x, K, n_outs (tuple of n ints), dtype_code, dev_index, outs_array_addr, ws0_mask)
 *   -> tuple of n tensors | None | int.  One decode row: ONE allocation, the outputs are its column slices. */
static PyObject* forward_multi(PyObject*, PyObject* const* args, Py_ssize_t nargs) {
    if (nargs != 9 || !g_forward_multi_ex) { PyErr_SetString(PyExc_TypeError, "forward_multi: 9 arguments after bind()"); return nullptr; }
    if (!THPVariable_Check(args[2]) || !PyTuple_Check(args[4])) Py_RETURN_NONE;
    unsigned long long arr, optrs, mask;
    const unsigned long long ws_ptr = 0, ws_bytes = 0;
    if (!as_u64(args[8], &mask)) return nullptr;
    if (!mask) Py_RETURN_NONE;
    const long n = PyLong_AsLong(args[1]), K = PyLong_AsLong(args[3]), dtype_code = PyLong_AsLong(args[5]), dev_index = PyLong_AsLong(args[6]);
    if (!as_u64(args[0], &arr) || !as_u64(args[7], &optrs) || PyErr_Occurred()) return nullptr;
    if (n < 1 || n > 16 || PyTuple_GET_SIZE(args[4]) != n) Py_RETURN_NONE;
    const at::Tensor& x = THPVariable_Unpack(args[2]);
    int64_t M = 0;
    if (!plain_x(x, K, dtype_code, dev_index, &M) || !no_workspace(mask, M)) Py_RETURN_NONE;
    try {
        int64_t widths[16], total = 0;
        for (long i = 0; i < n; ++i) { widths[i] = PyLong_AsLong(PyTuple_GET_ITEM(args[4], i)); total += widths[i]; }
        if (PyErr_Occurred()) return nullptr;
        std::vector<int64_t> shape(x.sizes().begin(), x.sizes().end());
        at::Tensor outs[16];
        void** const op = (void**)(size_t)optrs;
        if (M == 1) {
            shape.back() = total;
            at::Tensor all = at::empty(shape, x.options());
            int64_t off = 0;
            for (long i = 0; i < n; ++i) { outs[i] = all.narrow(-1, off, widths[i]); off += widths[i]; }
        } else {
            for (long i = 0; i < n; ++i) { shape.back() = widths[i]; outs[i] = at::empty(shape, x.options()); }
        }
        for (long i = 0; i < n; ++i) op[i] = outs[i].data_ptr();
        void* st = (void*)c10::hip::getCurrentHIPStream((c10::DeviceIndex)dev_index).stream();
        const int rc = g_forward_multi_ex((const void* const*)(size_t)arr, (int)n, x.data_ptr(), (void* const*)op, (int)M, (void*)(size_t)ws_ptr,
                                          (size_t)ws_bytes, st, nullptr);
        if (rc) return PyLong_FromLong(rc);
        PyObject* tup = PyTuple_New(n);
        if (!tup) return nullptr;
        for (long i = 0; i < n; ++i) PyTuple_SET_ITEM(tup, i, THPVariable_Wrap(std::move(outs[i])));
        return tup;
    } catch (const std::exception& e) {
        PyErr_SetString(PyExc_RuntimeError, e.what());
        return nullptr;
    }
}

static PyMethodDef methods[] = {
    {"bind", (PyCFunction)(void (*)(void))bind, METH_FASTCALL, "hand over the addresses of gptq_forward_ex / gptq_forward_multi_ex"},
    {"dtype_code", (PyCFunction)(void (*)(void))dtype_code, METH_FASTCALL, "at::ScalarType of a tensor as an int"},
    {"forward", (PyCFunction)(void (*)(void))forward, METH_FASTCALL, "QuantLinear.forward's plain case: checks, output allocation, current stream, gptq_forward_ex"},
    {"forward_multi", (PyCFunction)(void (*)(void))forward_multi, METH_FASTCALL, "forward_multi's plain case through gptq_forward_multi_ex"},
    {nullptr, nullptr, 0, nullptr}};

static struct PyModuleDef moddef = {PyModuleDef_HEAD_INIT, "_fastfwd", "eager per-call path of the mi355x QuantLinear (ATen + the C ABI of libgptq_mi355x.so)", -1, methods};

PyMODINIT_FUNC PyInit__fastfwd(void) { return PyModule_Create(&moddef); }
