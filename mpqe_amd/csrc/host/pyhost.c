/* _pyhost -- the interpreter-facing part of the drop-in entry points' host path (mpqe_amd/dropin.py), as a CPython
 * extension: what would otherwise be ~10 interpreter round trips per margin_loss call.
 *
 *   choice(getrandbits, lens, len_all, base, cand, nq, out) -> words consumed
 *       random.choice per query with python's OWN generator (reference model.py:470-476): raw Mersenne-Twister outputs
 *       are taken from `getrandbits` (the bound method of the interpreter's random.Random) in rounds of exactly as many
 *       as the open queries need at least, and handed to the C-ABI library's mpqe_host_random_choice, which replays
 *       CPython's rejection loop over them -- the generator ends where the reference's list comprehension leaves it.
 *       lens / base / cand / out are ADDRESSES of int64 arrays (0 = absent), as mpqe_host_random_choice takes them.
 *   step_call(fn, args) -> status
 *       mpqe_step_forward_backward_ex(...) through a function pointer with its 24 arguments read from a packed block of
 *       host memory (StepCall below; the caller keeps one per packed step and rewrites the few fields that change):
 *       the ctypes marshalling of 24 arguments is most of a forward-only call's host time.
 *
 * No arithmetic of the data path lives here; the library (include/mpqe_amd.h) stays free of any Python dependency. */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <stdint.h>
#include <stdlib.h>

#include "mpqe_amd.h"

typedef int (*choice_fn)(const uint32_t *, int64_t, const int64_t *, int64_t, const int64_t *, const int64_t *, int64_t,
                         int64_t *, int64_t *);
static choice_fn g_choice = NULL;

static PyObject *py_bind(PyObject *self, PyObject *args) {
    unsigned long long addr;
    if (!PyArg_ParseTuple(args, "K", &addr)) return NULL;
    g_choice = (choice_fn)(uintptr_t)addr;
    Py_RETURN_NONE;
}

static PyObject *py_choice(PyObject *self, PyObject *args) {
    PyObject *getrandbits;
    unsigned long long lens, base, cand, out;
    long long len_all, nq;
    if (!PyArg_ParseTuple(args, "OKLKKLK", &getrandbits, &lens, &len_all, &base, &cand, &nq, &out)) return NULL;
    if (!g_choice) {
        PyErr_SetString(PyExc_RuntimeError, "_pyhost.bind(address of mpqe_host_random_choice) first");
        return NULL;
    }
    int64_t cursor[2] = {0, 0};
    long long consumed = 0;
    uint32_t stack_words[1024];
    uint32_t *words = stack_words;
    size_t cap = 1024;
    while (cursor[0] < nq) {
        const long long n = nq - cursor[0];
        if ((size_t)n > cap) {
            if (words != stack_words) free(words);
            words = (uint32_t *)malloc((size_t)n * 4);
            cap = (size_t)n;
            if (!words) return PyErr_NoMemory();
        }
        PyObject *bits = PyLong_FromLongLong(32 * n);
        if (!bits) goto fail;
        PyObject *r = PyObject_CallOneArg(getrandbits, bits);
        Py_DECREF(bits);
        if (!r) goto fail;
        /* n outputs, the first in the low 32 bits (CPython _randommodule.c: getrandbits fills words low to high) */
        if (!PyLong_Check(r) || _PyLong_AsByteArray((PyLongObject *)r, (unsigned char *)words, (size_t)n * 4, 1, 0) < 0) {
            Py_DECREF(r);
            if (!PyErr_Occurred()) PyErr_SetString(PyExc_TypeError, "getrandbits did not return an int");
            goto fail;
        }
        Py_DECREF(r);
        const int st = g_choice(words, n, (const int64_t *)(uintptr_t)lens, len_all, (const int64_t *)(uintptr_t)base,
                                (const int64_t *)(uintptr_t)cand, nq, cursor, (int64_t *)(uintptr_t)out);
        consumed += cursor[1];
        if (st != 0) {
            PyErr_SetString(PyExc_IndexError, "Cannot choose from an empty sequence");
            goto fail;
        }
    }
    if (words != stack_words) free(words);
    return PyLong_FromLongLong(consumed);
fail:
    if (words != stack_words) free(words);
    return NULL;
}

/* the arguments of mpqe_step_forward_backward_ex, in order, as one block (mpqe_amd/dropin.py mirrors it with ctypes) */
typedef struct {
    const mpqe_step_params_t *params;
    const mpqe_step_batch_t *batches;
    int64_t num_batches;
    const int64_t *anchor_ids, *targets, *negs;
    double margin;
    const mpqe_step_grads_t *grads;
    int64_t backward;
    float *loss, *scores_pos, *scores_neg;
    void *desc;
    uint64_t desc_bytes;
    int64_t upload_desc;
    void *workspace;
    uint64_t workspace_bytes;
    int32_t *err;
    const mpqe_step_lanes_t *lanes;
    void *const *events;
    int64_t num_events;
    void *touch;
    void *stream;
    const mpqe_step_extra_t *extra;
} StepCall;

typedef int (*step_fn)(const mpqe_step_params_t *, const mpqe_step_batch_t *, int, const int64_t *, const int64_t *,
                       const int64_t *, float, const mpqe_step_grads_t *, int, float *, float *, float *, void *, size_t, int,
                       void *, size_t, int32_t *, const mpqe_step_lanes_t *, void *const *, int, void *, void *,
                       const mpqe_step_extra_t *);

static PyObject *py_step_call(PyObject *self, PyObject *args) {
    unsigned long long fn, block;
    if (!PyArg_ParseTuple(args, "KK", &fn, &block)) return NULL;
    const StepCall *c = (const StepCall *)(uintptr_t)block;
    if (!fn || !c) {
        PyErr_SetString(PyExc_ValueError, "step_call(function address, argument block address)");
        return NULL;
    }
    const int st = ((step_fn)(uintptr_t)fn)(c->params, c->batches, (int)c->num_batches, c->anchor_ids, c->targets, c->negs,
                                            (float)c->margin, c->grads, (int)c->backward, c->loss, c->scores_pos,
                                            c->scores_neg, c->desc, (size_t)c->desc_bytes, (int)c->upload_desc, c->workspace,
                                            (size_t)c->workspace_bytes, c->err, c->lanes, c->events, (int)c->num_events,
                                            c->touch, c->stream, c->extra);
    return PyLong_FromLong(st);
}

static PyMethodDef methods[] = {
    {"bind", py_bind, METH_VARARGS, "bind(address of mpqe_host_random_choice)"},
    {"choice", py_choice, METH_VARARGS, "choice(getrandbits, lens, len_all, base, cand, nq, out) -> words consumed"},
    {"step_call", py_step_call, METH_VARARGS, "step_call(address of mpqe_step_forward_backward_ex, address of a StepCall block) -> status"},
    {NULL, NULL, 0, NULL}};

static struct PyModuleDef moduledef = {PyModuleDef_HEAD_INIT, "_pyhost", NULL, -1, methods};

PyMODINIT_FUNC PyInit__pyhost(void) { return PyModule_Create(&moduledef); }
