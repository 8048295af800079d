/*
 * pymarshal.c -- CPython helpers that move interval lists between the reference's Python representation
 * (List[List[Tuple[int,int]]], NeuralSemiCRFInterval.py:56-102 builds it tuple by tuple on the host) and the packed
 * int32 buffers of the C ABI (pairs [K][2], offsets [B+1]).  Host-side only; no device code.
 *
 *   unpack(pairs_addr, offsets_addr, B, T) -> list of B lists of (begin, end) tuples
 *   pack_into(intervals, pairs_addr, cap, offsets_addr, T) -> K   (IndexError on an index outside [0, T))
 *   count(intervals) -> total number of intervals
 *
 * The int objects 0..T-1 are created once per call and shared by all tuples (one allocation per interval instead of
 * three): 655 566 intervals (T=2048, NBatch=352, randn) in ~10 ms instead of ~48 ms for numpy's structured tolist().
 */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <stdint.h>

static PyObject* m_unpack(PyObject* self, PyObject* args)
{
    unsigned long long pa, oa;
    long B, T;
    if (!PyArg_ParseTuple(args, "KKll", &pa, &oa, &B, &T)) return NULL;
    const int32_t* pairs = (const int32_t*)(uintptr_t)pa;
    const int32_t* off = (const int32_t*)(uintptr_t)oa;
    if (B < 0 || T < 1) { PyErr_SetString(PyExc_ValueError, "bad B/T"); return NULL; }
    PyObject** ints = (PyObject**)PyMem_Malloc(sizeof(PyObject*) * (size_t)T);
    if (!ints) return PyErr_NoMemory();
    for (long i = 0; i < T; ++i) {
        ints[i] = PyLong_FromLong(i);
        if (!ints[i]) { for (long j = 0; j < i; ++j) Py_DECREF(ints[j]); PyMem_Free(ints); return NULL; }
    }
    PyObject* out = PyList_New(B);
    int ok = out != NULL;
    for (long c = 0; ok && c < B; ++c) {
        const long n0 = off[c], n1 = off[c + 1];
        if (n1 < n0) { PyErr_SetString(PyExc_ValueError, "offsets not ascending"); ok = 0; break; }
        PyObject* lst = PyList_New(n1 - n0);
        if (!lst) { ok = 0; break; }
        PyList_SET_ITEM(out, c, lst);
        for (long i = n0; i < n1; ++i) {
            const long b = pairs[2 * i], e = pairs[2 * i + 1];
            if (b < 0 || b >= T || e < 0 || e >= T) { PyErr_SetString(PyExc_ValueError, "interval index out of range"); ok = 0; break; }
            PyObject* t = PyTuple_New(2);
            if (!t) { ok = 0; break; }
            Py_INCREF(ints[b]); Py_INCREF(ints[e]);
            PyTuple_SET_ITEM(t, 0, ints[b]);
            PyTuple_SET_ITEM(t, 1, ints[e]);
            PyList_SET_ITEM(lst, i - n0, t);
        }
    }
    for (long i = 0; i < T; ++i) Py_DECREF(ints[i]);
    PyMem_Free(ints);
    if (!ok) {
        if (out) {
            /* unfilled slots are NULL: fill them so that the list can be released */
            for (long c = 0; c < B; ++c) {
                PyObject* lst = PyList_GET_ITEM(out, c);
                if (!lst) { Py_INCREF(Py_None); PyList_SET_ITEM(out, c, Py_None); continue; }
                if (PyList_Check(lst))
                    for (Py_ssize_t i = 0; i < PyList_GET_SIZE(lst); ++i)
                        if (!PyList_GET_ITEM(lst, i)) { Py_INCREF(Py_None); PyList_SET_ITEM(lst, i, Py_None); }
            }
            Py_DECREF(out);
        }
        return NULL;
    }
    return out;
}

static PyObject* m_count(PyObject* self, PyObject* arg)
{
    PyObject* seq = PySequence_Fast(arg, "intervals must be a sequence of sequences");
    if (!seq) return NULL;
    Py_ssize_t total = 0;
    const Py_ssize_t B = PySequence_Fast_GET_SIZE(seq);
    for (Py_ssize_t c = 0; c < B; ++c) {
        const Py_ssize_t n = PyObject_Length(PySequence_Fast_GET_ITEM(seq, c));
        if (n < 0) { Py_DECREF(seq); return NULL; }
        total += n;
    }
    Py_DECREF(seq);
    return PyLong_FromSsize_t(total);
}

static PyObject* m_pack_into(PyObject* self, PyObject* args)
{
    PyObject* iv;
    unsigned long long pa, oa;
    long long cap;
    long T;
    if (!PyArg_ParseTuple(args, "OKLKl", &iv, &pa, &cap, &oa, &T)) return NULL;
    int32_t* pairs = (int32_t*)(uintptr_t)pa;
    int32_t* off = (int32_t*)(uintptr_t)oa;
    PyObject* seq = PySequence_Fast(iv, "intervals must be a sequence of sequences");
    if (!seq) return NULL;
    const Py_ssize_t B = PySequence_Fast_GET_SIZE(seq);
    long long k = 0;
    off[0] = 0;
    for (Py_ssize_t c = 0; c < B; ++c) {
        PyObject* lst = PySequence_Fast(PySequence_Fast_GET_ITEM(seq, c), "each chain must hold a sequence of (begin, end)");
        if (!lst) { Py_DECREF(seq); return NULL; }
        const Py_ssize_t n = PySequence_Fast_GET_SIZE(lst);
        for (Py_ssize_t i = 0; i < n; ++i) {
            PyObject* p = PySequence_Fast_GET_ITEM(lst, i);
            long b, e;
            if (PyTuple_CheckExact(p) && PyTuple_GET_SIZE(p) == 2) {
                b = PyLong_AsLong(PyTuple_GET_ITEM(p, 0));
                e = PyLong_AsLong(PyTuple_GET_ITEM(p, 1));
            } else {
                PyObject* ps = PySequence_Fast(p, "an interval must be a (begin, end) pair");
                if (!ps) { Py_DECREF(lst); Py_DECREF(seq); return NULL; }
                if (PySequence_Fast_GET_SIZE(ps) != 2) {
                    Py_DECREF(ps); Py_DECREF(lst); Py_DECREF(seq);
                    PyErr_SetString(PyExc_ValueError, "an interval must be a (begin, end) pair");
                    return NULL;
                }
                b = PyLong_AsLong(PySequence_Fast_GET_ITEM(ps, 0));
                e = PyLong_AsLong(PySequence_Fast_GET_ITEM(ps, 1));
                Py_DECREF(ps);
            }
            if ((b == -1 || e == -1) && PyErr_Occurred()) { Py_DECREF(lst); Py_DECREF(seq); return NULL; }
            if (b < 0 || b >= T || e < 0 || e >= T) {
                Py_DECREF(lst); Py_DECREF(seq);
                PyErr_Format(PyExc_IndexError, "interval index out of range for T=%ld", T);
                return NULL;
            }
            if (k >= cap) { Py_DECREF(lst); Py_DECREF(seq); PyErr_SetString(PyExc_ValueError, "pairs buffer too small"); return NULL; }
            pairs[2 * k] = (int32_t)b; pairs[2 * k + 1] = (int32_t)e;
            ++k;
        }
        Py_DECREF(lst);
        off[c + 1] = (int32_t)k;
    }
    Py_DECREF(seq);
    return PyLong_FromLongLong(k);
}

static PyMethodDef methods[] = {
    {"unpack", m_unpack, METH_VARARGS, "packed int32 pairs/offsets (host addresses) -> list of lists of (begin, end)"},
    {"pack_into", m_pack_into, METH_VARARGS, "list of lists of (begin, end) -> packed int32 buffers; returns K"},
    {"count", m_count, METH_O, "total number of intervals"},
    {NULL, NULL, 0, NULL}};

static struct PyModuleDef moddef = {PyModuleDef_HEAD_INIT, "_semicrf_marshal", "interval-list marshalling", -1, methods};

PyMODINIT_FUNC PyInit__semicrf_marshal(void) { return PyModule_Create(&moddef); }
