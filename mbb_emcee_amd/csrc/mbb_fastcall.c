/* mbb_fastcall.c -- the boundary call of likelihood.__call__ as one C-level callable (CPython extension `_mbbfast`).
 *
 * What a sampler hands the likelihood once per half-step (emcee: mbb_fit.py:80-81) -- a C-contiguous float64 array of rows
 * [n, 5], or one row [5] -- goes: rows into the block the kernel reads (device memory behind the PCIe BAR, or pinned
 * memory: mbb_boundary_buffers), mbb_lnlike_call(ctx, n), the results out of the pinned block into a fresh array.  In
 * Python that is a numpy slice assignment, a ctypes call and a numpy copy: ~1.5 us of a ~10 us call; here ~0.4.
 * Anything else -- another dtype or layout, more rows than the blocks hold, a row the reference raises for (a non-zero
 * return of the native call) -- gives None, and likelihood.__call__ goes on as it did without this module (which is also
 * what it does when the module is not built).  No arithmetic here: the evaluation is the HIP library's.
 */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#define NPY_NO_DEPRECATED_API NPY_1_7_API_VERSION
#include <numpy/arrayobject.h>
#include <stddef.h>
#include <stdint.h>
#include <string.h>

typedef int (*call_fn)(void *ctx, int n);

typedef struct {
    PyObject_HEAD
    vectorcallfunc vectorcall;
    call_fn fn;            /* mbb_lnlike_call */
    void *ctx;
    double *rows;          /* [cap][5]: where the kernel reads its parameter rows from */
    const double *lnl;     /* [cap]: where it writes lnprob (pinned host memory) */
    Py_ssize_t cap;
    const volatile unsigned long long *gen;   /* mbb_boundary_generation: changes when the two blocks are freed ... */
    unsigned long long gen0;                  /* ... and what it held when their addresses were taken */
} FastCall;

static PyObject *fastcall_vectorcall(PyObject *self, PyObject *const *args, size_t nargsf, PyObject *kwnames)
{
    FastCall *f = (FastCall *)self;
    if (PyVectorcall_NARGS(nargsf) != 1 || (kwnames && PyTuple_GET_SIZE(kwnames) != 0)) {
        PyErr_SetString(PyExc_TypeError, "FastCall takes one positional argument");
        return NULL;
    }
    PyObject *o = args[0];
    if (!PyArray_CheckExact(o)) Py_RETURN_NONE;
    PyArrayObject *a = (PyArrayObject *)o;
    if (PyArray_TYPE(a) != NPY_DOUBLE || !PyArray_ISCARRAY_RO(a)) Py_RETURN_NONE;
    const int nd = PyArray_NDIM(a);
    npy_intp n;
    if (nd == 2 && PyArray_DIM(a, 1) == 5 && PyArray_DIM(a, 0) > 0) n = PyArray_DIM(a, 0);
    else if (nd == 1 && PyArray_DIM(a, 0) == 5) n = 1;
    else Py_RETURN_NONE;
    if (n > f->cap) Py_RETURN_NONE;
    /* another entry point of the context may have made the blocks anew since (more rows than they held): never write
     * through stale addresses -- None sends likelihood.__call__ to ask for the buffers again */
    if (*f->gen != f->gen0) Py_RETURN_NONE;
    memcpy(f->rows, PyArray_DATA(a), (size_t)n * 5 * sizeof(double));
    int rc;
    Py_BEGIN_ALLOW_THREADS
    rc = f->fn(f->ctx, (int)n);
    Py_END_ALLOW_THREADS
    if (rc != 0 || *f->gen != f->gen0) Py_RETURN_NONE;
    if (nd == 1) return PyFloat_FromDouble(f->lnl[0]);
    PyObject *out = PyArray_SimpleNew(1, &n, NPY_DOUBLE);
    if (!out) return NULL;
    memcpy(PyArray_DATA((PyArrayObject *)out), f->lnl, (size_t)n * sizeof(double));
    return out;
}

static int fastcall_init(PyObject *self, PyObject *args, PyObject *kwds)
{
    FastCall *f = (FastCall *)self;
    unsigned long long fn = 0, ctx = 0, rows = 0, lnl = 0, gen = 0, gen0 = 0;
    Py_ssize_t cap = 0;
    static char *names[] = {"fn", "ctx", "rows", "lnl", "cap", "gen", "gen0", NULL};
    if (!PyArg_ParseTupleAndKeywords(args, kwds, "KKKKnKK", names, &fn, &ctx, &rows, &lnl, &cap, &gen, &gen0)) return -1;
    if (!fn || !ctx || !rows || !lnl || cap <= 0 || !gen) {
        PyErr_SetString(PyExc_ValueError, "FastCall needs the addresses of mbb_lnlike_call, the context, the two blocks and their generation word");
        return -1;
    }
    f->gen = (const volatile unsigned long long *)(uintptr_t)gen; f->gen0 = gen0;
    f->fn = (call_fn)(uintptr_t)fn; f->ctx = (void *)(uintptr_t)ctx;
    f->rows = (double *)(uintptr_t)rows; f->lnl = (const double *)(uintptr_t)lnl; f->cap = cap;
    f->vectorcall = fastcall_vectorcall;
    return 0;
}

static PyTypeObject FastCallType = {
    PyVarObject_HEAD_INIT(NULL, 0)
    .tp_name = "_mbbfast.FastCall",
    .tp_basicsize = sizeof(FastCall),
    .tp_flags = Py_TPFLAGS_DEFAULT | Py_TPFLAGS_HAVE_VECTORCALL,
    .tp_doc = "FastCall(fn, ctx, rows, lnl, cap, gen, gen0)(pars) -> float64[n] | float | None (None: take the general path)",
    .tp_new = PyType_GenericNew,
    .tp_init = fastcall_init,
    .tp_call = PyVectorcall_Call,
    .tp_vectorcall_offset = offsetof(FastCall, vectorcall),
};

static struct PyModuleDef moddef = {PyModuleDef_HEAD_INIT, "_mbbfast", "the boundary call of mbb_emcee_amd.likelihood in C", -1, NULL};

PyMODINIT_FUNC PyInit__mbbfast(void)
{
    import_array();
    if (PyType_Ready(&FastCallType) < 0) return NULL;
    PyObject *m = PyModule_Create(&moddef);
    if (!m) return NULL;
    Py_INCREF(&FastCallType);
    if (PyModule_AddObject(m, "FastCall", (PyObject *)&FastCallType) < 0) { Py_DECREF(&FastCallType); Py_DECREF(m); return NULL; }
    return m;
}
