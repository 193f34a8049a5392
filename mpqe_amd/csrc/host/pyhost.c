/* _pyhost -- the interpreter-facing part of the drop-in entry points' host path (mpqe_amd/dropin.py), as a CPython
 * extension: what would otherwise be ~10 interpreter round trips per margin_loss call.
 *
 *   choice(getrandbits, lens, len_all, base, cand, nq, out) -> words consumed
 *       random.choice per query with python's OWN generator (reference model.py:470-476): raw Mersenne-Twister outputs
 *       are taken from `getrandbits` (the bound method of the interpreter's random.Random) in rounds of exactly as many
 *       as the open queries need at least, and handed to the C-ABI library's mpqe_host_random_choice, which replays
 *       CPython's rejection loop over them -- the generator ends where the reference's list comprehension leaves it.
 *       lens / base / cand / out are ADDRESSES of int64 arrays (0 = absent), as mpqe_host_random_choice takes them.
 *   choice_mt(rng, lens, len_all, base, cand, nq, out) -> words consumed
 *       the same with the outputs taken straight from the Mersenne-Twister state of `rng` (a random.Random: CPython's
 *       _random.Random keeps {int index; uint32_t state[624]} right behind the object header) instead of through
 *       getrandbits(32 n) and a big-integer round trip per round. Only after mt_selftest() has passed in this
 *       interpreter (mpqe_amd/_lib.py runs it at load: outputs AND the state left behind equal getrandbits' across a
 *       regeneration boundary); otherwise the callers keep to choice().
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
#include <string.h>

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

/* ---- python's generator, read in place. MT19937 (Matsumoto & Nishimura 1998) exactly as CPython's _randommodule.c runs
 * it: 624 words, regenerated in one sweep when the index reaches 624, tempered on the way out. */
#define MT_N 624
#define MT_M 397
typedef struct {
    PyObject_HEAD
    int index;
    uint32_t state[MT_N];
} MtObject;

static void mt_words(MtObject *o, uint32_t *out, long long n) {
    uint32_t *mt = o->state;
    for (long long i = 0; i < n; ++i) {
        if (o->index >= MT_N) {
            int kk;
            uint32_t y;
            for (kk = 0; kk < MT_N - MT_M; ++kk) {
                y = (mt[kk] & 0x80000000u) | (mt[kk + 1] & 0x7fffffffu);
                mt[kk] = mt[kk + MT_M] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
            }
            for (; kk < MT_N - 1; ++kk) {
                y = (mt[kk] & 0x80000000u) | (mt[kk + 1] & 0x7fffffffu);
                mt[kk] = mt[kk + (MT_M - MT_N)] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
            }
            y = (mt[MT_N - 1] & 0x80000000u) | (mt[0] & 0x7fffffffu);
            mt[MT_N - 1] = mt[MT_M - 1] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
            o->index = 0;
        }
        uint32_t y = mt[o->index++];
        y ^= y >> 11;
        y ^= (y << 7) & 0x9d2c5680u;
        y ^= (y << 15) & 0xefc60000u;
        y ^= y >> 18;
        out[i] = y;
    }
}

static int g_mt_ok = 0;
static PyTypeObject *g_mt_type = NULL;        /* _random.Random */

/* mt_words(rng, n) -> bytes: n raw outputs (little-endian words), advancing rng. For the self test only. */
static PyObject *py_mt_words(PyObject *self, PyObject *args) {
    PyObject *rng;
    long long n;
    if (!PyArg_ParseTuple(args, "OL", &rng, &n)) return NULL;
    if (!g_mt_type || !PyObject_TypeCheck(rng, g_mt_type) || n < 0 || n > (1 << 20)) {
        PyErr_SetString(PyExc_TypeError, "mt_words(random.Random, n) after mt_bind(_random.Random)");
        return NULL;
    }
    PyObject *b = PyBytes_FromStringAndSize(NULL, (Py_ssize_t)n * 4);
    if (!b) return NULL;
    mt_words((MtObject *)rng, (uint32_t *)PyBytes_AS_STRING(b), n);
    return b;
}

/* mt_bind(_random.Random type, ok): the C base type of random.Random; ok != 0 once the caller's self test has passed */
static PyObject *py_mt_bind(PyObject *self, PyObject *args) {
    PyObject *tp;
    int ok;
    if (!PyArg_ParseTuple(args, "Oi", &tp, &ok)) return NULL;
    if (!PyType_Check(tp) || ((PyTypeObject *)tp)->tp_basicsize < (Py_ssize_t)sizeof(MtObject)) {
        PyErr_SetString(PyExc_TypeError, "mt_bind(_random.Random, ok)");
        return NULL;
    }
    Py_INCREF(tp);
    Py_XDECREF((PyObject *)g_mt_type);
    g_mt_type = (PyTypeObject *)tp;
    g_mt_ok = ok;
    Py_RETURN_NONE;
}

static PyObject *py_choice_mt(PyObject *self, PyObject *args) {
    PyObject *rng;
    unsigned long long lens, base, cand, out;
    long long len_all, nq;
    if (!PyArg_ParseTuple(args, "OKLKKLK", &rng, &lens, &len_all, &base, &cand, &nq, &out)) return NULL;
    if (!g_choice || !g_mt_ok || !g_mt_type || !PyObject_TypeCheck(rng, g_mt_type)) {
        PyErr_SetString(PyExc_RuntimeError, "_pyhost.choice_mt: not bound, self test not passed, or not a random.Random");
        return NULL;
    }
    int64_t cursor[2] = {0, 0};
    long long consumed = 0;
    uint32_t words[1024];
    while (cursor[0] < nq) {
        long long n = nq - cursor[0];          /* as many outputs as the open queries need at least */
        if (n > 1024) n = 1024;
        mt_words((MtObject *)rng, words, n);
        const int st = g_choice(words, n, (const int64_t *)(uintptr_t)lens, len_all, (const int64_t *)(uintptr_t)base,
                                (const int64_t *)(uintptr_t)cand, nq, cursor, (int64_t *)(uintptr_t)out);
        consumed += cursor[1];
        if (st != 0) {
            PyErr_SetString(PyExc_IndexError, "Cannot choose from an empty sequence");
            return NULL;
        }
    }
    return PyLong_FromLongLong(consumed);
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

/* margin_call(rng, fn, block, A, B, anchors, anchors_stride, targets, a_dst, t_dst, n_dst, lens, len_all, base, cand, loss,
 *             stream, margin, notify_value, join_stream) -> status
 * One forward-only margin_loss call of the drop-in (mpqe_amd/dropin.py) from a batch that is a window of its formula's id
 * arrays: the window's anchors (slot-major rows `anchors + i * anchors_stride`, B ids each) and targets copied into the
 * pinned arena the step reads, the negatives drawn (choice_mt), the changing fields of the call's StepCall block written,
 * the library called (block->extra->join_event set: the call runs on a side stream and the consumer's stream `join_stream`
 * -- 0: the null stream -- waits for it). Everything else in the block -- and block->extra->notify -- is as the caller left it. */
static PyObject *py_margin_call(PyObject *self, PyObject *const *args, Py_ssize_t nargs) {
    if (nargs != 20) {
        PyErr_SetString(PyExc_TypeError, "margin_call takes 20 arguments");
        return NULL;
    }
    PyObject *rng = args[0];
    unsigned long long v[16];
    for (int i = 0; i < 16; ++i) {
        v[i] = PyLong_AsUnsignedLongLongMask(args[1 + i]);
        if (v[i] == (unsigned long long)-1 && PyErr_Occurred()) return NULL;
    }
    const double margin = PyFloat_AsDouble(args[17]);
    if (margin == -1.0 && PyErr_Occurred()) return NULL;
    const unsigned long long nv = PyLong_AsUnsignedLongLongMask(args[18]);
    if (nv == (unsigned long long)-1 && PyErr_Occurred()) return NULL;
    const unsigned long long js = PyLong_AsUnsignedLongLongMask(args[19]);
    if (js == (unsigned long long)-1 && PyErr_Occurred()) return NULL;
    const step_fn fn = (step_fn)(uintptr_t)v[0];
    StepCall *c = (StepCall *)(uintptr_t)v[1];
    const long long A = (long long)v[2], B = (long long)v[3];
    const char *anchors = (const char *)(uintptr_t)v[4];
    const long long stride = (long long)v[5];
    const int64_t *targets = (const int64_t *)(uintptr_t)v[6];
    int64_t *a_dst = (int64_t *)(uintptr_t)v[7], *t_dst = (int64_t *)(uintptr_t)v[8], *n_dst = (int64_t *)(uintptr_t)v[9];
    const int64_t *lens = (const int64_t *)(uintptr_t)v[10];
    const long long len_all = (long long)v[11];
    const int64_t *base = (const int64_t *)(uintptr_t)v[12], *cand = (const int64_t *)(uintptr_t)v[13];
    if (!fn || !c || !c->extra || A < 0 || B < 1 || !anchors || !targets || !a_dst || !t_dst || !n_dst || !cand ||
        !g_choice || !g_mt_ok || !g_mt_type || !PyObject_TypeCheck(rng, g_mt_type)) {
        PyErr_SetString(PyExc_RuntimeError, "_pyhost.margin_call: bad arguments, or choice_mt is not available");
        return NULL;
    }
    for (long long i = 0; i < A; ++i) memcpy(a_dst + i * B, anchors + i * stride, (size_t)B * 8);
    memcpy(t_dst, targets, (size_t)B * 8);
    int64_t cursor[2] = {0, 0};
    uint32_t words[1024];
    while (cursor[0] < B) {
        long long n = B - cursor[0];
        if (n > 1024) n = 1024;
        mt_words((MtObject *)rng, words, n);
        if (g_choice(words, n, lens, len_all, base, cand, B, cursor, n_dst) != 0) {
            PyErr_SetString(PyExc_IndexError, "Cannot choose from an empty sequence");
            return NULL;
        }
    }
    c->anchor_ids = a_dst;
    c->targets = t_dst;
    c->negs = n_dst;
    c->margin = margin;
    c->backward = 0;
    c->loss = (float *)(uintptr_t)v[14];
    c->scores_pos = c->scores_neg = NULL;
    c->stream = (void *)(uintptr_t)v[15];
    ((mpqe_step_extra_t *)(uintptr_t)c->extra)->notify_value = (uint32_t)nv;
    ((mpqe_step_extra_t *)(uintptr_t)c->extra)->join_stream = (void *)(uintptr_t)js;
    const int st = fn(c->params, c->batches, (int)c->num_batches, c->anchor_ids, c->targets, c->negs, (float)c->margin,
                      c->grads, 0, c->loss, NULL, NULL, c->desc, (size_t)c->desc_bytes, (int)c->upload_desc, c->workspace,
                      (size_t)c->workspace_bytes, c->err, c->lanes, c->events, (int)c->num_events, c->touch, c->stream, c->extra);
    return PyLong_FromLong(st);
}

static PyMethodDef methods[] = {
    {"margin_call", (PyCFunction)(void (*)(void))py_margin_call, METH_FASTCALL,
     "margin_call(rng, fn, block, A, B, anchors, anchors_stride, targets, a_dst, t_dst, n_dst, lens, len_all, base, cand, loss, "
     "stream, margin, notify_value, join_stream) -> status"},
    {"bind", py_bind, METH_VARARGS, "bind(address of mpqe_host_random_choice)"},
    {"choice", py_choice, METH_VARARGS, "choice(getrandbits, lens, len_all, base, cand, nq, out) -> words consumed"},
    {"choice_mt", py_choice_mt, METH_VARARGS, "choice_mt(rng, lens, len_all, base, cand, nq, out) -> words consumed"},
    {"mt_bind", py_mt_bind, METH_VARARGS, "mt_bind(_random.Random, ok)"},
    {"mt_words", py_mt_words, METH_VARARGS, "mt_words(rng, n) -> bytes (self test)"},
    {"step_call", py_step_call, METH_VARARGS, "step_call(address of mpqe_step_forward_backward_ex, address of a StepCall block) -> status"},
    {NULL, NULL, 0, NULL}};

static struct PyModuleDef moduledef = {PyModuleDef_HEAD_INIT, "_pyhost", NULL, -1, methods};

PyMODINIT_FUNC PyInit__pyhost(void) { return PyModule_Create(&moduledef); }
