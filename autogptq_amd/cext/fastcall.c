/* fastcall.c -- CPython-side trampoline for the two per-token entry points of libgptq_mi355x.so.
 *
 * The reference's callers are eager (generate() under inference_mode, auto_gptq/modeling/_base.py:415-418): one Python call
 * per layer and token.  A decode kernel here runs for 5-14 us, and a ctypes call with eight converted arguments costs 2-3 us
 * of host time on its own.  This module is the same call through METH_FASTCALL: integers in, status out, nothing converted
 * twice.  It holds no compute and no device code; the function addresses are handed over once by autogptq_amd/_lib.py
 * (taken from the ctypes handle of the already loaded library), so there is exactly one copy of the C ABI in the process.
 * Without it (not built) the Python side calls the same functions through ctypes.                                        */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <stddef.h>

typedef int (*fwd_ex_t)(const void *, const void *, void *, int, void *, size_t, void *, const void *);
typedef int (*fwd_multi_t)(const void *const *, int, const void *, void *const *, int, void *, size_t, void *, const void *);
typedef int (*fwd_mlp_t)(const void *, const void *, const void *, const void *, void *, int, void *, size_t, void *);

static fwd_ex_t g_forward_ex = NULL;
static fwd_multi_t g_forward_multi_ex = NULL;
static fwd_mlp_t g_mlp_forward = NULL;

/* args[0..n) -> v[0..n) as unsigned 64-bit; 0 with a Python exception set if one of them is not an int (None, a float, ...) */
static int as_u64s(PyObject *const *args, Py_ssize_t n, unsigned long long *v) {
    for (Py_ssize_t i = 0; i < n; ++i) {
        v[i] = PyLong_AsUnsignedLongLongMask(args[i]);
        if (v[i] == (unsigned long long)-1 && PyErr_Occurred()) return 0;
    }
    return 1;
}
/* a row count: a Python int that fits a C int (PyLong_AsLong reports overflow through the exception state) */
static int as_int(PyObject *o, int *out) {
    const long v = PyLong_AsLong(o);
    if (v == -1 && PyErr_Occurred()) return 0;
    if (v < -2147483647L - 1 || v > 2147483647L) { PyErr_SetString(PyExc_OverflowError, "row / layer count does not fit a C int"); return 0; }
    *out = (int)v;
    return 1;
}

/* bind(forward_ex_addr, forward_multi_ex_addr[, mlp_forward_addr]) */
static PyObject *bind(PyObject *self, PyObject *const *args, Py_ssize_t nargs) {
    unsigned long long v[3] = {0, 0, 0};
    if (nargs != 2 && nargs != 3) { PyErr_SetString(PyExc_TypeError, "bind(forward_ex_addr, forward_multi_ex_addr[, mlp_forward_addr])"); return NULL; }
    if (!as_u64s(args, nargs, v)) return NULL;
    if (!v[0] || !v[1]) { PyErr_SetString(PyExc_ValueError, "bind: NULL function address"); return NULL; }
    g_forward_ex = (fwd_ex_t)(size_t)v[0];
    g_forward_multi_ex = (fwd_multi_t)(size_t)v[1];
    g_mlp_forward = (fwd_mlp_t)(size_t)v[2];
    Py_RETURN_NONE;
}

/* forward(layer_addr, x_ptr, out_ptr, M, ws_ptr, ws_bytes, stream, tuning_addr) -> status */
static PyObject *forward(PyObject *self, PyObject *const *args, Py_ssize_t nargs) {
    unsigned long long v[8];
    int M;
    if (nargs != 8 || !g_forward_ex) { PyErr_SetString(PyExc_TypeError, "forward: 8 integer arguments after bind()"); return NULL; }
    if (!as_u64s(args, 8, v) || !as_int(args[3], &M)) return NULL;
    const int rc = g_forward_ex((const void *)(size_t)v[0], (const void *)(size_t)v[1], (void *)(size_t)v[2], M, (void *)(size_t)v[4], (size_t)v[5],
                                (void *)(size_t)v[6], (const void *)(size_t)v[7]);
    return PyLong_FromLong(rc);
}

/* forward_multi(layers_array_addr, n, x_ptr, outs_array_addr, M, ws_ptr, ws_bytes, stream, tuning_addr) -> status */
static PyObject *forward_multi(PyObject *self, PyObject *const *args, Py_ssize_t nargs) {
    unsigned long long v[9];
    int n, M;
    if (nargs != 9 || !g_forward_multi_ex) { PyErr_SetString(PyExc_TypeError, "forward_multi: 9 integer arguments after bind()"); return NULL; }
    if (!as_u64s(args, 9, v) || !as_int(args[1], &n) || !as_int(args[4], &M)) return NULL;
    const int rc = g_forward_multi_ex((const void *const *)(size_t)v[0], n, (const void *)(size_t)v[2], (void *const *)(size_t)v[3], M,
                                      (void *)(size_t)v[5], (size_t)v[6], (void *)(size_t)v[7], (const void *)(size_t)v[8]);
    return PyLong_FromLong(rc);
}

/* mlp_forward(gate_addr, up_addr, down_addr, x_ptr, out_ptr, M, ws_ptr, ws_bytes, stream) -> status */
static PyObject *mlp_forward(PyObject *self, PyObject *const *args, Py_ssize_t nargs) {
    unsigned long long v[9];
    int M;
    if (nargs != 9 || !g_mlp_forward) { PyErr_SetString(PyExc_TypeError, "mlp_forward: 9 integer arguments after bind() with its address"); return NULL; }
    if (!as_u64s(args, 9, v) || !as_int(args[5], &M)) return NULL;
    const int rc = g_mlp_forward((const void *)(size_t)v[0], (const void *)(size_t)v[1], (const void *)(size_t)v[2], (const void *)(size_t)v[3],
                                 (void *)(size_t)v[4], M, (void *)(size_t)v[6], (size_t)v[7], (void *)(size_t)v[8]);
    return PyLong_FromLong(rc);
}

static PyMethodDef methods[] = {
    {"bind", (PyCFunction)(void (*)(void))bind, METH_FASTCALL, "hand over the addresses of gptq_forward_ex / gptq_forward_multi_ex"},
    {"forward", (PyCFunction)(void (*)(void))forward, METH_FASTCALL, "gptq_forward_ex with integer arguments"},
    {"forward_multi", (PyCFunction)(void (*)(void))forward_multi, METH_FASTCALL, "gptq_forward_multi_ex with integer arguments"},
    {"mlp_forward", (PyCFunction)(void (*)(void))mlp_forward, METH_FASTCALL, "gptq_mlp_forward with integer arguments"},
    {NULL, NULL, 0, NULL}};

static struct PyModuleDef moddef = {PyModuleDef_HEAD_INIT, "_fastcall", "trampoline into libgptq_mi355x.so", -1, methods};

PyMODINIT_FUNC PyInit__fastcall(void) { return PyModule_Create(&moddef); }
