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

static fwd_ex_t g_forward_ex = NULL;
static fwd_multi_t g_forward_multi_ex = NULL;

static unsigned long long as_u64(PyObject *o) { return PyLong_AsUnsignedLongLongMask(o); }

static PyObject *bind(PyObject *self, PyObject *const *args, Py_ssize_t nargs) {
    if (nargs != 2) { PyErr_SetString(PyExc_TypeError, "bind(forward_ex_addr, forward_multi_ex_addr)"); return NULL; }
    g_forward_ex = (fwd_ex_t)(size_t)as_u64(args[0]);
    g_forward_multi_ex = (fwd_multi_t)(size_t)as_u64(args[1]);
    Py_RETURN_NONE;
}

/* forward(layer_addr, x_ptr, out_ptr, M, ws_ptr, ws_bytes, stream, tuning_addr) -> status */
static PyObject *forward(PyObject *self, PyObject *const *args, Py_ssize_t nargs) {
    if (nargs != 8 || !g_forward_ex) { PyErr_SetString(PyExc_TypeError, "forward: 8 integer arguments after bind()"); return NULL; }
    const int rc = g_forward_ex((const void *)(size_t)as_u64(args[0]), (const void *)(size_t)as_u64(args[1]), (void *)(size_t)as_u64(args[2]),
                                (int)PyLong_AsLong(args[3]), (void *)(size_t)as_u64(args[4]), (size_t)as_u64(args[5]),
                                (void *)(size_t)as_u64(args[6]), (const void *)(size_t)as_u64(args[7]));
    return PyLong_FromLong(rc);
}

/* forward_multi(layers_array_addr, n, x_ptr, outs_array_addr, M, ws_ptr, ws_bytes, stream, tuning_addr) -> status */
static PyObject *forward_multi(PyObject *self, PyObject *const *args, Py_ssize_t nargs) {
    if (nargs != 9 || !g_forward_multi_ex) { PyErr_SetString(PyExc_TypeError, "forward_multi: 9 integer arguments after bind()"); return NULL; }
    const int rc = g_forward_multi_ex((const void *const *)(size_t)as_u64(args[0]), (int)PyLong_AsLong(args[1]), (const void *)(size_t)as_u64(args[2]),
                                      (void *const *)(size_t)as_u64(args[3]), (int)PyLong_AsLong(args[4]), (void *)(size_t)as_u64(args[5]),
                                      (size_t)as_u64(args[6]), (void *)(size_t)as_u64(args[7]), (const void *)(size_t)as_u64(args[8]));
    return PyLong_FromLong(rc);
}

static PyMethodDef methods[] = {
    {"bind", (PyCFunction)(void (*)(void))bind, METH_FASTCALL, "hand over the addresses of gptq_forward_ex / gptq_forward_multi_ex"},
    {"forward", (PyCFunction)(void (*)(void))forward, METH_FASTCALL, "gptq_forward_ex with integer arguments"},
    {"forward_multi", (PyCFunction)(void (*)(void))forward_multi, METH_FASTCALL, "gptq_forward_multi_ex with integer arguments"},
    {NULL, NULL, 0, NULL}};

static struct PyModuleDef moddef = {PyModuleDef_HEAD_INIT, "_fastcall", "trampoline into libgptq_mi355x.so", -1, methods};

PyMODINIT_FUNC PyInit__fastcall(void) { return PyModule_Create(&moddef); }
