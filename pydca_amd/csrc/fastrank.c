/* The ranked list both classes return -- [((i, j), score), ...] -- built with the CPython / NumPy C API.
 *
 * The order comes from the device (csrc/rank.hip); what was left on the host is making L (L - 1) / 2 Python tuples, which at
 * L = 500 cost 16 - 19 ms in pure Python (tolist + itemgetter + a list of NumPy scalars + zip) against 21 ms for the whole GPU
 * chain of `mfdca compute_fn`.  Here it is one loop: the (i, j) tuples are shared objects picked from a per-L cache the caller
 * keeps, the scores become numpy.float64 scalars as in the reference (meanfield_dca.py:940, plmdca.py:479), and each outer
 * tuple is filled in place.  Host logic only -- nothing of the device path lives here; without this module pydca_amd/_ranking.py
 * builds the same list in Python. */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#define NPY_NO_DEPRECATED_API NPY_1_7_API_VERSION
#include <numpy/arrayobject.h>
#include <numpy/arrayscalars.h>

/* ranked(pairs: list of (i, j) tuples in pair order -- or None with L given: the tuples are then made on the fly, which is what a
 * process that ranks ONCE wants --, order: int32[n] contiguous, scores: float64[m] contiguous, L = 0) -> list of n tuples */
static PyObject* fastrank_ranked(PyObject* self, PyObject* args)
{
    PyObject *pairs, *order_o, *scores_o;
    int L = 0;
    if (!PyArg_ParseTuple(args, "OOO|i", &pairs, &order_o, &scores_o, &L)) return NULL;
    if (pairs == Py_None) pairs = NULL;
    else if (!PyList_Check(pairs)) { PyErr_SetString(PyExc_TypeError, "pairs: list or None"); return NULL; }
    if (!pairs && L < 2) return PyList_New(0);
    PyArrayObject* order = (PyArrayObject*)PyArray_FROM_OTF(order_o, NPY_INT32, NPY_ARRAY_IN_ARRAY);
    if (!order) return NULL;
    PyArrayObject* scores = (PyArrayObject*)PyArray_FROM_OTF(scores_o, NPY_FLOAT64, NPY_ARRAY_IN_ARRAY);
    if (!scores) { Py_DECREF(order); return NULL; }
    const npy_intp n = PyArray_SIZE(order), m = PyArray_SIZE(scores);
    const Py_ssize_t np_ = pairs ? PyList_GET_SIZE(pairs) : (Py_ssize_t)L * (L - 1) / 2;
    const npy_int32* ord = (const npy_int32*)PyArray_DATA(order);
    const double* sc = (const double*)PyArray_DATA(scores);
    PyObject** ints = NULL;          /* on-the-fly mode: one int object per site index, the first pair index of every i */
    Py_ssize_t* first = NULL;
    PyObject* out = PyList_New(n);
    if (!out) goto fail;
    if (!pairs) {
        ints = (PyObject**)PyMem_Calloc((size_t)L, sizeof(PyObject*));
        first = (Py_ssize_t*)PyMem_Malloc(sizeof(Py_ssize_t) * (size_t)L);
        if (!ints || !first) { PyErr_NoMemory(); goto fail_out; }
        for (int i = 0; i < L; ++i) {
            ints[i] = PyLong_FromLong(i);
            if (!ints[i]) goto fail_out;
            first[i] = (Py_ssize_t)i * (2 * (Py_ssize_t)L - i - 1) / 2;      /* pairs (0, .) ... (i - 1, .) come before (i, i + 1) */
        }
    }
    for (npy_intp k = 0; k < n; ++k) {
        const npy_int32 idx = ord[k];
        if (idx < 0 || idx >= m || idx >= np_) { PyErr_SetString(PyExc_IndexError, "rank order points outside the score vector"); goto fail_out; }
        PyObject* s = PyArrayScalar_New(Double);
        if (!s) goto fail_out;
        PyArrayScalar_ASSIGN(s, Double, sc[idx]);
        PyObject* t = PyTuple_New(2);
        if (!t) { Py_DECREF(s); goto fail_out; }
        PyObject* pr;
        if (pairs) {
            /* the cached tuples are visited in RANK order: 7 MB of objects touched at random, from DRAM whenever the reader and the
             * device chain of the next alignment have been through the caches since (10.5 ms instead of 4 for L = 500) -- the
             * reference count of the tuple sixteen places ahead is fetched now */
            if (k + 16 < n) {
                const npy_int32 ahead = ord[k + 16];
                if (ahead >= 0 && ahead < np_) __builtin_prefetch(PyList_GET_ITEM(pairs, ahead), 1, 1);
            }
            pr = PyList_GET_ITEM(pairs, idx); Py_INCREF(pr);
        }
        else {
            int lo = 0, hi = L - 2;                  /* the i with first[i] <= idx < first[i + 1] */
            while (lo < hi) { const int mid = (lo + hi + 1) / 2; if (first[mid] <= idx) lo = mid; else hi = mid - 1; }
            const int i = lo, j = i + 1 + (int)(idx - first[i]);
            pr = PyTuple_New(2);
            if (!pr) { Py_DECREF(s); Py_DECREF(t); goto fail_out; }
            Py_INCREF(ints[i]); Py_INCREF(ints[j]);
            PyTuple_SET_ITEM(pr, 0, ints[i]);
            PyTuple_SET_ITEM(pr, 1, ints[j]);
            PyObject_GC_UnTrack(pr);
        }
        PyTuple_SET_ITEM(t, 0, pr);
        PyTuple_SET_ITEM(t, 1, s);
        /* a tuple of an (int, int) tuple and a NumPy scalar cannot be part of a reference cycle: taken out of the collector's
         * lists at once (what the collector itself does with such tuples on its first pass over them) -- otherwise the first
         * allocation after the caller re-enables the collector walks a quarter of a million new objects (4 ms at L = 500) */
        PyObject_GC_UnTrack(t);
        PyList_SET_ITEM(out, k, t);
    }
    if (ints) { for (int i = 0; i < L; ++i) Py_XDECREF(ints[i]); PyMem_Free(ints); }
    PyMem_Free(first);
    Py_DECREF(order); Py_DECREF(scores);
    return out;
fail_out:
    Py_XDECREF(out);
    if (ints) { for (int i = 0; i < L; ++i) Py_XDECREF(ints[i]); PyMem_Free(ints); }
    PyMem_Free(first);
fail:
    Py_DECREF(order); Py_DECREF(scores);
    return NULL;
}

/* pair_tuples(L) -> [(0, 1), (0, 2), ..., (L - 2, L - 1)] with ONE int object per site index */
static PyObject* fastrank_pair_tuples(PyObject* self, PyObject* args)
{
    int L;
    if (!PyArg_ParseTuple(args, "i", &L)) return NULL;
    if (L < 0) { PyErr_SetString(PyExc_ValueError, "L < 0"); return NULL; }
    PyObject** ints = (PyObject**)PyMem_Malloc(sizeof(PyObject*) * (size_t)(L > 0 ? L : 1));
    if (!ints) return PyErr_NoMemory();
    for (int i = 0; i < L; ++i) {
        ints[i] = PyLong_FromLong(i);
        if (!ints[i]) { for (int j = 0; j < i; ++j) Py_DECREF(ints[j]); PyMem_Free(ints); return NULL; }
    }
    const Py_ssize_t n = (Py_ssize_t)L * (L - 1) / 2;
    PyObject* out = PyList_New(n > 0 ? n : 0);
    Py_ssize_t k = 0;
    for (int i = 0; out && i < L; ++i)
        for (int j = i + 1; j < L; ++j) {
            PyObject* t = PyTuple_New(2);
            if (!t) { Py_CLEAR(out); break; }
            Py_INCREF(ints[i]); Py_INCREF(ints[j]);
            PyTuple_SET_ITEM(t, 0, ints[i]);
            PyTuple_SET_ITEM(t, 1, ints[j]);
            PyObject_GC_UnTrack(t);
            PyList_SET_ITEM(out, k++, t);
        }
    for (int i = 0; i < L; ++i) Py_DECREF(ints[i]);
    PyMem_Free(ints);
    return out;
}

static PyMethodDef methods[] = {
    {"ranked", fastrank_ranked, METH_VARARGS, "ranked(pairs, order, scores) -> [((i, j), numpy.float64), ...]"},
    {"pair_tuples", fastrank_pair_tuples, METH_VARARGS, "pair_tuples(L) -> [(0, 1), ..., (L-2, L-1)]"},
    {NULL, NULL, 0, NULL}};
static struct PyModuleDef moduledef = {PyModuleDef_HEAD_INIT, "_fastrank", NULL, -1, methods};
PyMODINIT_FUNC PyInit__fastrank(void)
{
    import_array();
    return PyModule_Create(&moduledef);
}
