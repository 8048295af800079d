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


/* ------------------------------------------------------------------------------------------------------------------------
 * The cross-segment event merge of TransKun.transcribe (ModelTransformer.py:803-843) and Data.resolveOverlapping (Data.py:170-214)
 * for the packed per-step results of transkun_amd.transcribe (rows of 7 doubles: start, end, hasOnset, hasOffset, velocity,
 * symbol index, chain index).  The reference walks Python Note objects event by event; at the event density of a batched
 * transcription (thousands of events per step) that walk, not the device, set the pace.  Here the merge state is a C array per
 * (recording, symbol) and Note objects are only made for the events that survive, once, at the end.
 *
 *   tm_new(n_files, P, merge) -> capsule
 *   tm_add(capsule, step, rows_addr, K, active_files)        rows in chain order (ascending time within a chain)
 *   tm_finish(capsule, file, pitches, NoteClass, vel_is_float, resolve) -> list of Note
 * ---------------------------------------------------------------------------------------------------------------------- */
#include <descrobject.h>
#include <structmember.h>
#include <stdlib.h>
#include <string.h>

typedef struct { double start, end, vel; char on, off, settled; } TmEv;
typedef struct { double start, end; PyObject* obj; } TmFin;           /* eager mode: a finished Note and its sort key */
typedef struct { TmEv* v; long n, cap; int seen; long first_step; double first_start, first_end;
                 long nfin; TmFin* f; long nf, fcap; } TmTrack;        /* eager: events [0, nfin) are settled, their Notes in f[0 .. nf) */
typedef struct { long n_files, P; int merge; TmTrack* tracks;
                 int eager;                                           /* 0: Notes at tm_finish; 1: made as events settle, resolveOverlapping applied; 2: without */
                 PyObject* cls; PyObject* pseq; Py_ssize_t so[6]; int vel_float; } TmMerger;

static void tm_free(PyObject* cap)
{
    TmMerger* m = (TmMerger*)PyCapsule_GetPointer(cap, "semicrf.tm");
    if (!m) return;
    if (m->tracks) {
        for (long i = 0; i < m->n_files * m->P; ++i) {
            free(m->tracks[i].v);
            for (long k = 0; k < m->tracks[i].nf; ++k) Py_XDECREF(m->tracks[i].f[k].obj);
            free(m->tracks[i].f);
        }
        free(m->tracks);
    }
    Py_XDECREF(m->cls); Py_XDECREF(m->pseq);
    free(m);
}

static PyObject* m_tm_new(PyObject* self, PyObject* args)
{
    long n_files, P; int merge;
    if (!PyArg_ParseTuple(args, "llp", &n_files, &P, &merge)) return NULL;
    if (n_files < 1 || P < 1) { PyErr_SetString(PyExc_ValueError, "bad sizes"); return NULL; }
    TmMerger* m = (TmMerger*)calloc(1, sizeof(TmMerger));
    if (!m) return PyErr_NoMemory();
    m->n_files = n_files; m->P = P; m->merge = merge;
    m->tracks = (TmTrack*)calloc((size_t)(n_files * P), sizeof(TmTrack));
    if (!m->tracks) { free(m); return PyErr_NoMemory(); }
    return PyCapsule_New(m, "semicrf.tm", tm_free);
}

static int tm_push(TmTrack* t, const TmEv* e)
{
    if (t->n == t->cap) {
        const long nc = t->cap ? 2 * t->cap : 16;
        TmEv* nv = (TmEv*)realloc(t->v, (size_t)nc * sizeof(TmEv));
        if (!nv) return -1;
        t->v = nv; t->cap = nc;
    }
    t->v[t->n++] = *e;
    return 0;
}

static Py_ssize_t slot_offset(PyObject* cls, const char* name)
{
    PyObject* d = PyObject_GetAttrString(cls, name);
    if (!d) return -1;
    Py_ssize_t off = -1;
    if (Py_TYPE(d) == &PyMemberDescr_Type) off = ((PyMemberDescrObject*)d)->d_member->offset;
    Py_DECREF(d);
    if (off < 0) PyErr_Format(PyExc_TypeError, "the Note class must define __slots__ with `%s`", name);
    return off;
}


/* A Note with its slots filled directly (no __init__ call). */
static PyObject* tm_make_note(TmMerger* m, double start, double end, long sym, double vel, int on, int off, int vel_float, PyObject* pseq,
                              PyTypeObject* tp, const Py_ssize_t* so)
{
    PyObject* o = tp->tp_alloc(tp, 0);
    if (!o) return NULL;
    PyObject* vals[6];
    vals[0] = PyFloat_FromDouble(start);
    vals[1] = PyFloat_FromDouble(end);
    vals[2] = PySequence_Fast_GET_ITEM(pseq, sym); Py_INCREF(vals[2]);
    vals[3] = vel_float ? PyFloat_FromDouble(vel) : PyLong_FromLong((long)vel);
    vals[4] = on ? Py_True : Py_False; Py_INCREF(vals[4]);
    vals[5] = off ? Py_True : Py_False; Py_INCREF(vals[5]);
    int ok = 1;
    for (int k = 0; k < 6; ++k) {
        if (!vals[k]) { ok = 0; continue; }
        *(PyObject**)((char*)o + so[k]) = vals[k];
    }
    (void)m;
    if (!ok) { Py_DECREF(o); return NULL; }
    return o;
}

/* Eager mode.  An event is SETTLED once nothing can change what becomes of it; settled events become Notes at once, while the device
 * works on the next step.
 *   without resolveOverlapping (eager == 2): the output is in track order and only the track's LAST event can still change (the merge
 *     rule, ModelTransformer.py:806-822, touches nothing else): every other event settles.
 *   with it (eager == 1; Data.py:170-214: in (start, end, pitch) order an event that starts before the previous event of its pitch has
 *     ended cuts that one short): the unsettled event i that sorts first settles when the surviving event j that sorts next is known
 *     for good -- j is not the track's last event (whose fields and survival are open) and j.start < bound, the time below which no
 *     event of a LATER step can start (the caller's statement: the next segment's begin time): then nothing can come to lie between
 *     i and j or in front of i.
 * final != 0 (tm_finish): the last event counts as carrying an offset (:831-834) and everything settles.  -1: allocation failure. */
static int tm_emit(TmMerger* m, TmTrack* t, long sym, const TmEv* e, double end)
{
    if (t->nf == t->fcap) {
        const long nc = t->fcap ? 2 * t->fcap : 16;
        TmFin* nv = (TmFin*)realloc(t->f, (size_t)nc * sizeof(TmFin));
        if (!nv) return -1;
        t->f = nv; t->fcap = nc;
    }
    PyObject* o = tm_make_note(m, e->start, end, sym, e->vel, e->on, 1, m->vel_float, m->pseq, (PyTypeObject*)m->cls, m->so);
    if (!o) return -1;
    t->f[t->nf].start = e->start; t->f[t->nf].end = end; t->f[t->nf].obj = o; ++t->nf;
    return 0;
}
static int tm_before(const TmEv* v, long a, long b)      /* (start, end, index) order */
{
    if (v[a].start != v[b].start) return v[a].start < v[b].start;
    if (v[a].end != v[b].end) return v[a].end < v[b].end;
    return a < b;
}
static int tm_settle(TmMerger* m, TmTrack* t, long sym, int final, double bound)
{
    if (m->eager == 2) {
        while (t->nfin < t->n) {
            const long i = t->nfin;
            const int is_last = i == t->n - 1;
            if (is_last && !final) break;
            if ((t->v[i].off || is_last) && tm_emit(m, t, sym, &t->v[i], t->v[i].end)) return -1;
            t->v[i].settled = 1;
            ++t->nfin;
        }
        return 0;
    }
    /* the usual case in one pass: the unsettled tail is in sort order already (the merge rule appends behind the last event's end;
     * only a REPLACED last event can come to lie in front of its predecessors) */
    {
        while (t->nfin < t->n && t->v[t->nfin].settled) ++t->nfin;
        const long last = t->n - 1;
        int sorted = 1;
        for (long k = t->nfin + 1; k < t->n && sorted; ++k)
            if (t->v[k].settled || tm_before(t->v, k, k - 1)) sorted = 0;
        if (sorted && t->nfin < t->n && !t->v[t->nfin].settled) {
            long i = t->nfin;
            while (i < t->n) {
                if (!(t->v[i].off || i == last)) { t->v[i].settled = 1; ++i; continue; }        /* dropped */
                long j = i + 1;
                while (j < t->n && !(t->v[j].off || j == last)) ++j;                           /* the next survivor */
                if (!final && (i == last || j >= t->n || j == last || !(t->v[j].start < bound))) break;
                double end = t->v[i].end;
                if (j < t->n && end > t->v[j].start) end = t->v[j].start;
                if (t->v[i].start < end && tm_emit(m, t, sym, &t->v[i], end)) return -1;
                t->v[i].settled = 1;
                for (long k = i + 1; k < j; ++k) t->v[k].settled = 1;                         /* the dropped ones in between */
                i = j;
            }
            while (t->nfin < t->n && t->v[t->nfin].settled) ++t->nfin;
            return 0;
        }
    }
    while (1) {
        while (t->nfin < t->n && t->v[t->nfin].settled) ++t->nfin;
        if (t->nfin >= t->n) return 0;
        const long last = t->n - 1;
        long i = -1, j = -1;                              /* first and second unsettled SURVIVOR-or-open events in sort order */
        for (long k = t->nfin; k < t->n; ++k) {
            if (t->v[k].settled) continue;
            if (!(t->v[k].off || k == last)) { t->v[k].settled = 1; continue; }      /* no offset, not the last: dropped (:837-841) */
            if (i < 0 || tm_before(t->v, k, i)) { j = i; i = k; }
            else if (j < 0 || tm_before(t->v, k, j)) j = k;
        }
        if (i < 0) continue;                              /* (only dropped ones were left) */
        if (!final && (i == last || (j >= 0 && (j == last || !(t->v[j].start < bound))))) return 0;
        if (!final && j < 0) return 0;                    /* (i is not the last, yet no other candidate: cannot happen, stay safe) */
        double end = t->v[i].end;
        if (j >= 0 && end > t->v[j].start) end = t->v[j].start;
        if (t->v[i].start < end && tm_emit(m, t, sym, &t->v[i], end)) return -1;      /* (else: left without duration) */
        t->v[i].settled = 1;
    }
}

/* tm_eager(capsule, NoteClass, pitches, resolve): switch the merger to eager mode (before the first tm_add; needs merge on) */
static PyObject* m_tm_eager(PyObject* self, PyObject* args)
{
    PyObject *cap, *cls, *pitches; int resolve;
    if (!PyArg_ParseTuple(args, "OOOp", &cap, &cls, &pitches, &resolve)) return NULL;
    TmMerger* m = (TmMerger*)PyCapsule_GetPointer(cap, "semicrf.tm");
    if (!m) return NULL;
    if (!m->merge) { PyErr_SetString(PyExc_ValueError, "eager mode needs the incomplete-event merge (a track's events are then in time order)"); return NULL; }
    if (!PyType_Check(cls)) { PyErr_SetString(PyExc_TypeError, "NoteClass must be a class"); return NULL; }
    PyObject* pseq = PySequence_Fast(pitches, "pitches must be a sequence");
    if (!pseq) return NULL;
    if (PySequence_Fast_GET_SIZE(pseq) != m->P) { Py_DECREF(pseq); PyErr_SetString(PyExc_ValueError, "bad pitches"); return NULL; }
    const char* names[6] = {"start", "end", "pitch", "velocity", "hasOnset", "hasOffset"};
    for (int i = 0; i < 6; ++i) if ((m->so[i] = slot_offset(cls, names[i])) < 0) { Py_DECREF(pseq); return NULL; }
    Py_XDECREF(m->cls); Py_XDECREF(m->pseq);
    Py_INCREF(cls); m->cls = cls; m->pseq = pseq;
    m->eager = resolve ? 1 : 2;
    Py_RETURN_NONE;
}

static PyObject* m_tm_add(PyObject* self, PyObject* args)
{
    PyObject *cap, *active;
    long step; unsigned long long addr; long long K; int vel_float = 0;
    PyObject* bounds = NULL;          /* eager mode with resolveOverlapping: per active recording, the time below which no later step's event starts */
    if (!PyArg_ParseTuple(args, "OlKLO|pO", &cap, &step, &addr, &K, &active, &vel_float, &bounds)) return NULL;
    TmMerger* m = (TmMerger*)PyCapsule_GetPointer(cap, "semicrf.tm");
    if (!m) return NULL;
    PyObject* seq = PySequence_Fast(active, "active must be a sequence of recording indices");
    if (!seq) return NULL;
    const Py_ssize_t na = PySequence_Fast_GET_SIZE(seq);
    long files[256];
    if (na > 256) { Py_DECREF(seq); PyErr_SetString(PyExc_ValueError, "too many recordings in one step"); return NULL; }
    for (Py_ssize_t i = 0; i < na; ++i) {
        files[i] = PyLong_AsLong(PySequence_Fast_GET_ITEM(seq, i));
        if (files[i] < 0 || files[i] >= m->n_files) { Py_DECREF(seq); PyErr_SetString(PyExc_IndexError, "recording index out of range"); return NULL; }
    }
    Py_DECREF(seq);
    const double* rows = (const double*)(uintptr_t)addr;
    for (long long i = 0; i < K; ++i) {
        const double* r = rows + 7 * i;
        const long sym = (long)r[5], chain = (long)r[6];
        const long sg = chain / m->P;
        if (sym < 0 || sym >= m->P || sg < 0 || sg >= na) { PyErr_SetString(PyExc_IndexError, "event row out of range"); return NULL; }
        TmTrack* t = &m->tracks[files[sg] * m->P + sym];
        TmEv e; e.start = r[0]; e.end = r[1]; e.on = r[2] != 0.0; e.off = r[3] != 0.0; e.vel = r[4]; e.settled = 0;
        if (!t->seen) { t->seen = 1; t->first_step = step; t->first_start = e.start; t->first_end = e.end; }
        if (m->merge && t->n > 0) {                                   /* ModelTransformer.py:806-822 */
            TmEv* last = &t->v[t->n - 1];
            if (e.start < last->end) {
                if (e.on) *last = e;
                else { last->off = e.off; if (e.end > last->end) last->end = e.end; }
                continue;
            }
        }
        if (e.on && tm_push(t, &e)) return PyErr_NoMemory();            /* :824-825 */
    }
    if (m->eager) {                                                     /* Notes for what has settled: the device is busy with the next step */
        m->vel_float = vel_float;
        PyObject* bseq = (bounds && bounds != Py_None) ? PySequence_Fast(bounds, "bounds must be a sequence of floats") : NULL;
        if (bounds && bounds != Py_None && !bseq) return NULL;
        if (bseq && PySequence_Fast_GET_SIZE(bseq) != na) { Py_DECREF(bseq); PyErr_SetString(PyExc_ValueError, "one bound per active recording"); return NULL; }
        for (Py_ssize_t a = 0; a < na; ++a) {
            double bound = -1e300;                                      /* no statement: with resolveOverlapping nothing settles before tm_finish */
            if (bseq) {
                bound = PyFloat_AsDouble(PySequence_Fast_GET_ITEM(bseq, a));
                if (bound == -1.0 && PyErr_Occurred()) { Py_DECREF(bseq); return NULL; }
            }
            for (long sym = 0; sym < m->P; ++sym)
                if (tm_settle(m, &m->tracks[files[a] * m->P + sym], sym, 0, bound)) { Py_XDECREF(bseq); return PyErr_Occurred() ? NULL : PyErr_NoMemory(); }
        }
        Py_XDECREF(bseq);
    }
    Py_RETURN_NONE;
}

typedef struct { double start, end, vel; long sym, pitch, idx; char on, off; } TmOut;
static int tm_cmp_time(const void* a, const void* b)
{
    const TmOut* x = (const TmOut*)a; const TmOut* y = (const TmOut*)b;
    if (x->start != y->start) return x->start < y->start ? -1 : 1;
    if (x->end != y->end) return x->end < y->end ? -1 : 1;
    if (x->pitch != y->pitch) return x->pitch < y->pitch ? -1 : 1;
    return x->idx < y->idx ? -1 : (x->idx > y->idx);
}
typedef struct { long step; double start, end; long pitch, sym; } TmKey;
static int tm_cmp_key(const void* a, const void* b)
{
    const TmKey* x = (const TmKey*)a; const TmKey* y = (const TmKey*)b;
    if (x->step != y->step) return x->step < y->step ? -1 : 1;
    if (x->start != y->start) return x->start < y->start ? -1 : 1;
    if (x->end != y->end) return x->end < y->end ? -1 : 1;
    return x->pitch < y->pitch ? -1 : (x->pitch > y->pitch);
}

/* eager mode: settle what is left, then hand out the Notes -- with resolveOverlapping in (start, end, pitch) order (a heap merge of the
 * tracks, each already in that order), else in the reference's byType order (symbols as first seen, :837-841) */
typedef struct { long trk; long pos; double start, end; long pitch; } TmHead;
static int tm_head_less(const TmHead* x, const TmHead* y)
{
    if (x->start != y->start) return x->start < y->start;
    if (x->end != y->end) return x->end < y->end;
    if (x->pitch != y->pitch) return x->pitch < y->pitch;
    return x->trk < y->trk;
}
static void tm_sift(TmHead* h, long n, long i)
{
    while (1) {
        long l = 2 * i + 1, r = l + 1, b = i;
        if (l < n && tm_head_less(&h[l], &h[b])) b = l;
        if (r < n && tm_head_less(&h[r], &h[b])) b = r;
        if (b == i) return;
        TmHead t = h[i]; h[i] = h[b]; h[b] = t;
        i = b;
    }
}
static PyObject* tm_finish_eager(TmMerger* m, long file, int vel_float, int resolve)
{
    if ((resolve ? 1 : 2) != m->eager) { PyErr_SetString(PyExc_ValueError, "tm_finish: `resolve` differs from what tm_eager was given"); return NULL; }
    m->vel_float = vel_float;
    TmTrack* tr = m->tracks + file * m->P;
    long total = 0;
    for (long s = 0; s < m->P; ++s) {
        if (tm_settle(m, &tr[s], s, 1, 1e300)) return PyErr_Occurred() ? NULL : PyErr_NoMemory();
        total += tr[s].nf;
    }
    PyObject* out = PyList_New(total);
    if (!out) return NULL;
    long n = 0;
    if (m->eager == 1) {
        TmHead* h = (TmHead*)malloc((size_t)(m->P > 0 ? m->P : 1) * sizeof(TmHead));
        if (!h) { Py_DECREF(out); return PyErr_NoMemory(); }
        long nh = 0;
        for (long s = 0; s < m->P; ++s)
            if (tr[s].nf > 0) {
                h[nh].trk = s; h[nh].pos = 0; h[nh].start = tr[s].f[0].start; h[nh].end = tr[s].f[0].end;
                h[nh].pitch = PyLong_AsLong(PySequence_Fast_GET_ITEM(m->pseq, s)); ++nh;
            }
        for (long i = nh / 2 - 1; i >= 0; --i) tm_sift(h, nh, i);
        while (nh > 0) {
            TmTrack* t = &tr[h[0].trk];
            PyList_SET_ITEM(out, n++, t->f[h[0].pos].obj);
            t->f[h[0].pos].obj = NULL;
            if (++h[0].pos < t->nf) { h[0].start = t->f[h[0].pos].start; h[0].end = t->f[h[0].pos].end; }
            else { h[0] = h[nh - 1]; --nh; }
            if (nh > 0) tm_sift(h, nh, 0);
        }
        free(h);
    } else {
        TmKey* keys = (TmKey*)malloc((size_t)(m->P > 0 ? m->P : 1) * sizeof(TmKey));
        if (!keys) { Py_DECREF(out); return PyErr_NoMemory(); }
        long nk = 0;
        for (long s = 0; s < m->P; ++s)
            if (tr[s].seen) {
                keys[nk].step = tr[s].first_step; keys[nk].start = tr[s].first_start; keys[nk].end = tr[s].first_end;
                keys[nk].pitch = PyLong_AsLong(PySequence_Fast_GET_ITEM(m->pseq, s)); keys[nk].sym = s; ++nk;
            }
        qsort(keys, (size_t)nk, sizeof(TmKey), tm_cmp_key);
        for (long k = 0; k < nk; ++k) {
            TmTrack* t = &tr[keys[k].sym];
            for (long i = 0; i < t->nf; ++i) { PyList_SET_ITEM(out, n++, t->f[i].obj); t->f[i].obj = NULL; }
        }
        free(keys);
    }
    for (long s = 0; s < m->P; ++s) tr[s].nf = 0;                       /* (the list owns the Notes now) */
    return out;
}

static PyObject* m_tm_finish(PyObject* self, PyObject* args)
{
    PyObject *cap, *pitches, *cls;
    long file; int vel_float, resolve;
    if (!PyArg_ParseTuple(args, "OlOOpp", &cap, &file, &pitches, &cls, &vel_float, &resolve)) return NULL;
    TmMerger* m = (TmMerger*)PyCapsule_GetPointer(cap, "semicrf.tm");
    if (!m) return NULL;
    if (file < 0 || file >= m->n_files) { PyErr_SetString(PyExc_IndexError, "recording index out of range"); return NULL; }
    if (m->eager) return tm_finish_eager(m, file, vel_float, resolve);
    PyObject* pseq = PySequence_Fast(pitches, "pitches must be a sequence");
    if (!pseq) return NULL;
    if (PySequence_Fast_GET_SIZE(pseq) != m->P || !PyType_Check(cls)) { Py_DECREF(pseq); PyErr_SetString(PyExc_ValueError, "bad pitches / class"); return NULL; }
    const char* names[6] = {"start", "end", "pitch", "velocity", "hasOnset", "hasOffset"};
    Py_ssize_t so[6];
    for (int i = 0; i < 6; ++i) if ((so[i] = slot_offset(cls, names[i])) < 0) { Py_DECREF(pseq); return NULL; }
    TmTrack* tr = m->tracks + file * m->P;
    long total = 0;
    for (long s = 0; s < m->P; ++s) {
        if (tr[s].n > 0) tr[s].v[tr[s].n - 1].off = 1;               /* :831-834 */
        total += tr[s].n;
    }
    TmOut* ev = (TmOut*)malloc((size_t)(total > 0 ? total : 1) * sizeof(TmOut));
    TmKey* keys = (TmKey*)malloc((size_t)m->P * sizeof(TmKey));
    long* lastp = (long*)malloc((size_t)m->P * sizeof(long));
    if (!ev || !keys || !lastp) { free(ev); free(keys); free(lastp); Py_DECREF(pseq); return PyErr_NoMemory(); }
    /* the reference flattens byType.values(): symbols in the order their first event was SEEN (:837-841) */
    long nk = 0;
    for (long s = 0; s < m->P; ++s)
        if (tr[s].seen) {
            keys[nk].step = tr[s].first_step; keys[nk].start = tr[s].first_start; keys[nk].end = tr[s].first_end;
            keys[nk].pitch = PyLong_AsLong(PySequence_Fast_GET_ITEM(pseq, s)); keys[nk].sym = s; ++nk;
        }
    qsort(keys, (size_t)nk, sizeof(TmKey), tm_cmp_key);
    long n = 0;
    for (long k = 0; k < nk; ++k) {
        const long s = keys[k].sym;
        for (long i = 0; i < tr[s].n; ++i)
            if (tr[s].v[i].off) {
                TmOut* o = &ev[n];
                o->start = tr[s].v[i].start; o->end = tr[s].v[i].end; o->vel = tr[s].v[i].vel; o->sym = s; o->pitch = keys[k].pitch;
                o->on = tr[s].v[i].on; o->off = 1; o->idx = n; ++n;
            }
    }
    if (resolve) {                                                      /* Data.py:170-214 */
        qsort(ev, (size_t)n, sizeof(TmOut), tm_cmp_time);
        for (long s = 0; s < m->P; ++s) lastp[s] = -1;
        for (long i = 0; i < n; ++i) {
            const long j = lastp[ev[i].sym];
            if (j >= 0 && ev[j].end > ev[i].start) ev[j].end = ev[i].start;
            lastp[ev[i].sym] = i;
        }
        long w = 0;
        for (long i = 0; i < n; ++i) if (ev[i].start < ev[i].end) { ev[w] = ev[i]; ev[w].idx = w; ++w; }
        n = w;
        qsort(ev, (size_t)n, sizeof(TmOut), tm_cmp_time);
    }
    PyObject* out = PyList_New(n);
    PyTypeObject* tp = (PyTypeObject*)cls;
    int ok = out != NULL;
    for (long i = 0; ok && i < n; ++i) {
        PyObject* o = tp->tp_alloc(tp, 0);                             /* slots start as NULL; filled below, no __init__ call */
        if (!o) { ok = 0; break; }
        PyObject* vals[6];
        vals[0] = PyFloat_FromDouble(ev[i].start);
        vals[1] = PyFloat_FromDouble(ev[i].end);
        vals[2] = PySequence_Fast_GET_ITEM(pseq, ev[i].sym); Py_INCREF(vals[2]);
        vals[3] = vel_float ? PyFloat_FromDouble(ev[i].vel) : PyLong_FromLong((long)ev[i].vel);
        vals[4] = ev[i].on ? Py_True : Py_False; Py_INCREF(vals[4]);
        vals[5] = ev[i].off ? Py_True : Py_False; Py_INCREF(vals[5]);
        for (int k = 0; k < 6; ++k) {
            if (!vals[k]) { ok = 0; continue; }
            *(PyObject**)((char*)o + so[k]) = vals[k];
        }
        PyList_SET_ITEM(out, i, o);
    }
    free(ev); free(keys); free(lastp); Py_DECREF(pseq);
    if (!ok) {
        if (out) { for (long i = 0; i < n; ++i) if (!PyList_GET_ITEM(out, i)) { Py_INCREF(Py_None); PyList_SET_ITEM(out, i, Py_None); } Py_DECREF(out); }
        return PyErr_Occurred() ? NULL : PyErr_NoMemory();
    }
    return out;
}

static PyMethodDef methods[] = {
    {"unpack", m_unpack, METH_VARARGS, "packed int32 pairs/offsets (host addresses) -> list of lists of (begin, end)"},
    {"pack_into", m_pack_into, METH_VARARGS, "list of lists of (begin, end) -> packed int32 buffers; returns K"},
    {"count", m_count, METH_O, "total number of intervals"},
    {"tm_new", m_tm_new, METH_VARARGS, "event merger for n_files recordings of P symbols"},
    {"tm_add", m_tm_add, METH_VARARGS, "merge the packed events of one step (rows of 7 doubles at a host address)"},
    {"tm_finish", m_tm_finish, METH_VARARGS, "the final Note list of one recording"},
    {"tm_eager", m_tm_eager, METH_VARARGS, "make the Notes as events settle (merge on): tm_eager(capsule, NoteClass, pitches, resolve)"},
    {NULL, NULL, 0, NULL}};

static struct PyModuleDef moddef = {PyModuleDef_HEAD_INIT, "_semicrf_marshal", "interval-list marshalling", -1, methods};

PyMODINIT_FUNC PyInit__semicrf_marshal(void) { return PyModule_Create(&moddef); }
