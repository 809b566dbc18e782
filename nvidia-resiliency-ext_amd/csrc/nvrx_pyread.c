/*
 * nvrx_pyread.c -- CPython helper of the straggler package: builds the nested dicts of a Report straight from the f32
 * blocks of the result block (scores [ranks][2+2S], statistics [rows][8]).
 *
 * The reference returns its reports as populated dicts (reporting.py:535-545: section -> {rank -> score},
 * name -> {Statistic -> value}); here the numbers arrive as two f32 arrays in pinned memory and the dicts are built on
 * first read.  Doing that with numpy's tolist() + dict(zip(...)) costs 54-57 us for the 8 ranks x 64 sections report
 * (1 400 dict inserts, as many float objects, plus the intermediate lists); this module does the same inserts without
 * the intermediates.  Pure host-side formatting: no arithmetic happens here, the package falls back to the Python
 * implementation if the module was not built (same results).
 *
 *   sections(names, ranks, buf, offset, n_rows, width, first_col, cols[, second_col]) -> {name: {rank: float}} (or a pair of them)
 *   ranks(ranks, buf, offset, n_rows, width, col) -> {rank: float}
 *   summaries(names, stat_keys, stats, rows[, name_tmpl[, recycle[, offset]]]) -> {name: {stat_key: float, ..., stat_keys[5]: int}}
 *   inplace([on]) -> bool          is the in-place fill of cloned dicts active (CPython 3.10 only; see below)
 *   copy_sets(d) -> {key: set(value) for key, value in d.items()}      (fresh sets for every caller of identify_stragglers)
 *   flagged(buf, offset, rows, width, S, has_rel, has_indiv, ids, names, cols, memo) -> (gpu_rel, gpu_indiv, sec_rel, sec_indiv)
 *       the sets identify_stragglers returns, straight from the score kernel's flag bytes ([rows][2+2S] u8 at buf+offset):
 *       no numpy call, no Python frame per column -- what a report read once a minute pays for is cold code and cold
 *       objects, so the fewer of both the better (Report.identify_stragglers: 75 -> ... us cold on the build host).
 */
#define PY_SSIZE_T_CLEAN
#include <Python.h>

/* Dicts are created at their final size (a dict grown insert by insert reallocates its table at 6, 11, 22, 43 ...
 * entries: every inner dict of 8 ranks once, the outer dict of 64 names three times) and keys are inserted with the
 * hash they already carry.  Both are CPython-private but exported entry points (cpython/dictobject.h). */
#if PY_VERSION_HEX >= 0x03080000 && !defined(PYPY_VERSION)
#define NEW_DICT(n) _PyDict_NewPresized(n)
#define SET_KNOWN(d, k, v, h) _PyDict_SetItem_KnownHash(d, k, v, h)
#else
#define NEW_DICT(n) PyDict_New()
#define SET_KNOWN(d, k, v, h) PyDict_SetItem(d, k, v)
#endif

#define MAX_STACK_HASHES 256

/* ---- filling a cloned dict in place (CPython 3.10 only, verified before every use) ---------------------------------
 * Every inner dict of a report has the same keys: the ranks of the table, or the six Statistic members.  The fastest way
 * the public API offers to make one is a presized dict + one insert per entry (~20 ns each: hash, probe, entry write,
 * bookkeeping) -- 1 424 of them per report.  A clone of a TEMPLATE dict (PyDict_Copy of a dict without deletions copies
 * the key table with one memcpy) already has the keys in place, in insertion order, in the first n entry slots; what is
 * left is to put the values there.  That store goes through CPython 3.10's dict layout (Objects/dict-common.h,
 * Objects/dictobject.c: struct _dictkeysobject / PyDictKeyEntry), which is private, so it is fenced three ways:
 *   - compiled only for 3.10 (the layout changed in 3.11);
 *   - a self-test at import builds, clones, fills and reads back a dict through the public API -- any disagreement
 *     turns the path off for the process (NVRX_DEBUG_PYREAD_INPLACE=0 does the same by hand);
 *   - before EVERY fill the clone is checked: combined table, n entries used, and the key pointer found in each of the
 *     first n entry slots is the very key object expected there.  A layout that differs cannot pass that by accident;
 *     a clone that fails it is filled through PyDict_SetItem instead.
 * Values stored are floats / ints (never containers), the same as the template's None: the clone's GC state (untracked)
 * stays right.  54 % of the time of a section mapping goes away (tools/report_read_bench.py). */
#if PY_VERSION_HEX >= 0x030A0000 && PY_VERSION_HEX < 0x030B0000 && !defined(PYPY_VERSION)
#define NVRX_INPLACE_POSSIBLE 1
typedef struct {
    Py_hash_t me_hash;
    PyObject *me_key;
    PyObject *me_value;
} nvrx_dict_entry310;
typedef struct {
    Py_ssize_t dk_refcnt;
    Py_ssize_t dk_size;
    void *dk_lookup;
    Py_ssize_t dk_usable;
    Py_ssize_t dk_nentries;
    char dk_indices[];
} nvrx_dict_keys310;

/* the first entry slot of a combined-table dict, or NULL when d is not one with exactly n live entries in slots 0..n-1 */
static nvrx_dict_entry310 *inplace_entries(PyObject *d, Py_ssize_t n) {
    PyDictObject *mp = (PyDictObject *)d;
    if (!PyDict_CheckExact(d) || mp->ma_values != NULL || mp->ma_used != n || mp->ma_keys == NULL) return NULL;
    nvrx_dict_keys310 *k = (nvrx_dict_keys310 *)mp->ma_keys;
    if (k->dk_nentries != n || k->dk_size < 8 || (k->dk_size & (k->dk_size - 1)) != 0) return NULL;
    const Py_ssize_t ix = k->dk_size <= 0xff ? 1 : k->dk_size <= 0xffff ? 2 : k->dk_size <= 0xffffffffLL ? 4 : 8;
    return (nvrx_dict_entry310 *)(&k->dk_indices[k->dk_size * ix]);
}
#else
#define NVRX_INPLACE_POSSIBLE 0
#endif

static int g_inplace = 0; /* set by the self-test at import */

/* {key: None for key in keys}: the template an inner dict is cloned from */
static PyObject *template_dict(PyObject *keys) {
    const Py_ssize_t n = PyTuple_GET_SIZE(keys);
    PyObject *t = NEW_DICT(n);
    if (!t) return NULL;
    for (Py_ssize_t i = 0; i < n; i++)
        if (PyDict_SetItem(t, PyTuple_GET_ITEM(keys, i), Py_None) < 0) {
            Py_DECREF(t);
            return NULL;
        }
    if (PyDict_GET_SIZE(t) != n) {  /* (duplicate keys: no template) */
        Py_DECREF(t);
        PyErr_SetString(PyExc_ValueError, "nvrx_pyread: duplicate keys");
        return NULL;
    }
    return t;
}

/* A new dict {keys[i]: values[i]} -- values are NEW references, all of them consumed whatever happens.  tmpl (may be NULL)
 * is template_dict(keys); hashes (may be NULL) the keys' hashes. */
static PyObject *dict_from(PyObject *keys, PyObject **values, Py_ssize_t n, PyObject *tmpl, const Py_hash_t *hashes, int values_are_containers) {
    PyObject *d = NULL;
    Py_ssize_t i = 0;
    for (Py_ssize_t j = 0; j < n; j++)
        if (!values[j]) goto fail;  /* an allocation failed upstream */
#if NVRX_INPLACE_POSSIBLE
    if (g_inplace && tmpl) {
        d = PyDict_Copy(tmpl);
        if (!d) goto fail;
        nvrx_dict_entry310 *e = inplace_entries(d, n);
        int ok = e != NULL;
        for (Py_ssize_t j = 0; ok && j < n; j++) ok = e[j].me_key == PyTuple_GET_ITEM(keys, j) && e[j].me_value == Py_None;
        if (ok) {
            for (Py_ssize_t j = 0; j < n; j++) {
                e[j].me_value = values[j];  /* the reference moves into the dict */
                Py_DECREF(Py_None);
            }
            /* a dict that holds containers must be known to the collector (PyDict_SetItem does this when it stores one;
             * the clone of a template of Nones starts out untracked) */
            if (values_are_containers && !PyObject_GC_IsTracked(d)) PyObject_GC_Track(d);
            return d;
        }
        Py_CLEAR(d);  /* not the layout this was written for: the public way */
    }
#endif
    d = NEW_DICT(n);
    if (!d) goto fail;
    for (; i < n; i++) {
        const int rc = hashes ? SET_KNOWN(d, PyTuple_GET_ITEM(keys, i), values[i], hashes[i]) : PyDict_SetItem(d, PyTuple_GET_ITEM(keys, i), values[i]);
        Py_DECREF(values[i]);
        if (rc < 0) {
            i++;
            goto fail;
        }
    }
    return d;
fail:
    for (; i < n; i++) Py_XDECREF(values[i]);
    Py_XDECREF(d);
    if (!PyErr_Occurred()) PyErr_NoMemory();
    return NULL;
}

/* ---- recycling a mapping nobody holds any more --------------------------------------------------------------------------
 * A report's mappings have the same keys as the previous report's.  When the previous mapping is still around -- the caller
 * keeps the last one it built per plan in a list -- and NOTHING else refers to it (reference count 1: the report it was built
 * for is gone, nobody kept the dict), building the next one needs no dict at all: the values of the inner dicts are swapped
 * in place and the very same objects are handed out again (a cloned dict costs ~60 ns, a report has 192 of them).  Any
 * doubt -- somebody else holds the outer dict or an inner one, a key was added / removed / replaced, a value is not what
 * this module stores -- and that dict is built the usual way; a held object is never touched.  Same fences as the
 * in-place fill (CPython 3.10 layout, self-test, keys checked by identity before every refill). */
#if NVRX_INPLACE_POSSIBLE
/* entry slots of `d` if it is an unshared dict whose keys are exactly keys[0..n) in order, else NULL */
static nvrx_dict_entry310 *recyclable_entries(PyObject *d, PyObject *keys, Py_ssize_t n) {
    if (!g_inplace || !d || !PyDict_CheckExact(d) || Py_REFCNT(d) != 1) return NULL;
    nvrx_dict_entry310 *e = inplace_entries(d, n);
    if (!e) return NULL;
    for (Py_ssize_t j = 0; j < n; j++)
        if (e[j].me_key != PyTuple_GET_ITEM(keys, j) || e[j].me_value == NULL) return NULL;
    return e;
}

/* values[0..n) (new references) moved into the unshared inner dict d; 0 = d does not qualify, nothing was touched */
static int refill_inner(PyObject *d, PyObject *keys, PyObject **values, Py_ssize_t n) {
    nvrx_dict_entry310 *e = recyclable_entries(d, keys, n);
    if (!e) return 0;
    for (Py_ssize_t j = 0; j < n; j++)
        if (!values[j] || !(PyFloat_CheckExact(e[j].me_value) || PyLong_CheckExact(e[j].me_value))) return 0;
    for (Py_ssize_t j = 0; j < n; j++) {
        PyObject *old = e[j].me_value;
        e[j].me_value = values[j];
        Py_DECREF(old);  /* a float or an int: no code runs */
    }
    return 1;
}
#endif

/* the list slot's object, now owned by the caller (the slot holds None), or NULL when the slot is empty / there is no list */
static PyObject *take_recycled(PyObject *list, Py_ssize_t i) {
    if (!list || i >= PyList_GET_SIZE(list)) return NULL;
    PyObject *o = PyList_GET_ITEM(list, i);
    if (o == Py_None) return NULL;
    Py_INCREF(Py_None);
    PyList_SET_ITEM(list, i, Py_None);
    return o;
}

static void keep_recycled(PyObject *list, Py_ssize_t i, PyObject *o) {
    if (!list || i >= PyList_GET_SIZE(list) || !o) return;
    PyObject *old = PyList_GET_ITEM(list, i);
    Py_INCREF(o);
    PyList_SET_ITEM(list, i, o);
    Py_DECREF(old);
}

/* hashes of a tuple's items into out (heap-allocated beyond MAX_STACK_HASHES); NULL + exception on an unhashable key */
static Py_hash_t *tuple_hashes(PyObject *t, Py_hash_t *stack) {
    const Py_ssize_t n = PyTuple_GET_SIZE(t);
    Py_hash_t *h = n <= MAX_STACK_HASHES ? stack : PyMem_Malloc((size_t)n * sizeof(Py_hash_t));
    if (!h) {
        PyErr_NoMemory();
        return NULL;
    }
    for (Py_ssize_t i = 0; i < n; i++) {
        h[i] = PyObject_Hash(PyTuple_GET_ITEM(t, i));
        if (h[i] == -1 && PyErr_Occurred()) {
            if (h != stack) PyMem_Free(h);
            return NULL;
        }
    }
    return h;
}

/* one {name: {rank: float}} mapping over columns first_col + c of the [n_rows][width] f32 block p; name_tmpl (may be NULL):
 * {name: None for name in names}, kept by the caller across reports -- the outer dict is then a filled clone as well */
static PyObject *section_mapping(const float *p, int n_rows, int width, int first_col, PyObject *names, PyObject *ranks,
                                 const long *col, const Py_hash_t *name_hash, const Py_hash_t *rank_hash, PyObject *rank_tmpl,
                                 PyObject *name_tmpl, PyObject *prev /* owned, may be NULL: an earlier result to recycle */) {
    const Py_ssize_t n_names = PyTuple_GET_SIZE(names);
    PyObject *stack_vals[64], *stack_inner[MAX_STACK_HASHES];
    PyObject **vals = n_rows <= 64 ? stack_vals : PyMem_Malloc((size_t)n_rows * sizeof(PyObject *));
    PyObject **inner = n_names <= MAX_STACK_HASHES ? stack_inner : PyMem_Malloc((size_t)n_names * sizeof(PyObject *));
    PyObject *out = NULL;
    if (!vals || !inner) {
        PyErr_NoMemory();
        goto done;
    }
#if NVRX_INPLACE_POSSIBLE
    {
        nvrx_dict_entry310 *eo = recyclable_entries(prev, names, n_names);
        if (eo) {
            for (Py_ssize_t i = 0; i < n_names; i++) {
                const float *q = p + first_col + (col ? col[i] : i);
                for (int r = 0; r < n_rows; r++) vals[r] = PyFloat_FromDouble((double)q[(Py_ssize_t)r * width]);
                PyObject *old = eo[i].me_value;
                if (refill_inner(old, ranks, vals, n_rows)) continue;
                PyObject *fresh = dict_from(ranks, vals, n_rows, rank_tmpl, rank_hash, 0);  /* somebody holds the old one */
                if (!fresh) goto done;  /* (prev is a consistent dict at every point; it is released below) */
                eo[i].me_value = fresh;
                Py_DECREF(old);
            }
            out = prev;
            prev = NULL;
            goto done;
        }
    }
#endif
    for (Py_ssize_t i = 0; i < n_names; i++) {
        const float *q = p + first_col + (col ? col[i] : i);
        for (int r = 0; r < n_rows; r++) vals[r] = PyFloat_FromDouble((double)q[(Py_ssize_t)r * width]);
        inner[i] = dict_from(ranks, vals, n_rows, rank_tmpl, rank_hash, 0);
        if (!inner[i]) {
            for (Py_ssize_t j = 0; j < i; j++) Py_DECREF(inner[j]);
            goto done;
        }
    }
    out = dict_from(names, inner, n_names, name_tmpl, name_hash, 1);  /* consumes the inner dicts */
done:
    Py_XDECREF(prev);
    if (vals && vals != stack_vals) PyMem_Free(vals);
    if (inner && inner != stack_inner) PyMem_Free(inner);
    return out;
}

/* sections(names, ranks, buf, offset, n_rows, width, first_col, cols[, second_col]) -> mapping, or a pair of mappings when
 * second_col >= 0 (both score families of a report in one call: the name / rank hashes and the column table are shared).
 * buf + offset: the first of n_rows rows of width f32 (the report's rows inside the result block, or an ndarray). */
static PyObject *pyread_sections(PyObject *self, PyObject *args) {
    PyObject *names, *ranks, *cols;
    Py_buffer view;
    Py_ssize_t offset;
    int n_rows, width, first_col, second_col = -1;
    PyObject *name_tmpl = Py_None, *recycle = Py_None;
    if (!PyArg_ParseTuple(args, "O!O!y*niiiO|iOO", &PyTuple_Type, &names, &PyTuple_Type, &ranks, &view, &offset, &n_rows, &width, &first_col,
                          &cols, &second_col, &name_tmpl, &recycle))
        return NULL;
    if (name_tmpl == Py_None || !PyDict_CheckExact(name_tmpl)) name_tmpl = NULL;
    if (recycle == Py_None || !PyList_CheckExact(recycle)) recycle = NULL;
    PyObject *out = NULL, *a = NULL, *b = NULL, *rank_tmpl = NULL;
    Py_hash_t nh_stack[MAX_STACK_HASHES], rh_stack[MAX_STACK_HASHES], *nh = NULL, *rh = NULL;
    long col_stack[MAX_STACK_HASHES], *col = NULL;
    const Py_ssize_t n_names = PyTuple_GET_SIZE(names);
    if (n_rows < 0 || width <= 0 || first_col < 0 || offset < 0 || (offset % (Py_ssize_t)sizeof(float)) != 0 || PyTuple_GET_SIZE(ranks) != n_rows ||
        view.len < offset + (Py_ssize_t)n_rows * width * (Py_ssize_t)sizeof(float) ||
        (cols != Py_None && (!PyTuple_Check(cols) || PyTuple_GET_SIZE(cols) != n_names))) {
        PyErr_SetString(PyExc_ValueError, "nvrx_pyread.sections: inconsistent shapes");
        goto done;
    }
    const int last_first = second_col > first_col ? second_col : first_col;
    if (cols != Py_None) {
        col = n_names <= MAX_STACK_HASHES ? col_stack : PyMem_Malloc((size_t)n_names * sizeof(long));
        if (!col) {
            PyErr_NoMemory();
            goto done;
        }
        for (Py_ssize_t i = 0; i < n_names; i++) {
            col[i] = PyLong_AsLong(PyTuple_GET_ITEM(cols, i));
            if (col[i] == -1 && PyErr_Occurred()) goto done;
            if (col[i] < 0 || last_first + col[i] >= width) {
                PyErr_SetString(PyExc_ValueError, "nvrx_pyread.sections: column out of range");
                goto done;
            }
        }
    } else if (n_names && last_first + n_names - 1 >= width) {
        PyErr_SetString(PyExc_ValueError, "nvrx_pyread.sections: column out of range");
        goto done;
    }
    if (!(nh = tuple_hashes(names, nh_stack)) || !(rh = tuple_hashes(ranks, rh_stack))) goto done;
    const float *p = (const float *)((const char *)view.buf + offset);
    if (g_inplace && n_rows > 0) {
        rank_tmpl = template_dict(ranks);
        if (!rank_tmpl) PyErr_Clear();  /* (e.g. duplicate ranks: the dicts are built entry by entry) */
    }
    a = section_mapping(p, n_rows, width, first_col, names, ranks, col, nh, rh, rank_tmpl, name_tmpl, take_recycled(recycle, 0));
    if (!a) goto done;
    keep_recycled(recycle, 0, a);
    if (second_col >= 0) {
        b = section_mapping(p, n_rows, width, second_col, names, ranks, col, nh, rh, rank_tmpl, name_tmpl, take_recycled(recycle, 1));
        if (!b) goto done;
        keep_recycled(recycle, 1, b);
        out = PyTuple_Pack(2, a, b);
    } else {
        out = a;
        a = NULL;
    }
done:
    Py_XDECREF(a);
    Py_XDECREF(b);
    Py_XDECREF(rank_tmpl);
    if (nh && nh != nh_stack) PyMem_Free(nh);
    if (rh && rh != rh_stack) PyMem_Free(rh);
    if (col && col != col_stack) PyMem_Free(col);
    PyBuffer_Release(&view);
    return out;
}

/* ranks(ranks, buf, offset, n_rows, width, col) -> {rank: float}: one column of the score block (the GPU scores) */
static PyObject *pyread_ranks(PyObject *self, PyObject *args) {
    PyObject *ranks;
    Py_buffer view;
    Py_ssize_t offset;
    int n_rows, width, col;
    if (!PyArg_ParseTuple(args, "O!y*niii", &PyTuple_Type, &ranks, &view, &offset, &n_rows, &width, &col)) return NULL;
    PyObject *out = NULL;
    if (n_rows < 0 || width <= 0 || col < 0 || col >= width || offset < 0 || (offset % (Py_ssize_t)sizeof(float)) != 0 ||
        PyTuple_GET_SIZE(ranks) != n_rows || view.len < offset + (Py_ssize_t)n_rows * width * (Py_ssize_t)sizeof(float)) {
        PyErr_SetString(PyExc_ValueError, "nvrx_pyread.ranks: inconsistent shapes");
        goto done;
    }
    out = NEW_DICT(n_rows);
    if (!out) goto done;
    const float *p = (const float *)((const char *)view.buf + offset) + col;
    for (int r = 0; r < n_rows; r++) {
        PyObject *f = PyFloat_FromDouble((double)p[(Py_ssize_t)r * width]);
        if (!f || PyDict_SetItem(out, PyTuple_GET_ITEM(ranks, r), f) < 0) {
            Py_XDECREF(f);
            Py_CLEAR(out);
            goto done;
        }
        Py_DECREF(f);
    }
done:
    PyBuffer_Release(&view);
    return out;
}

static PyObject *pyread_summaries(PyObject *self, PyObject *args) {
    PyObject *names, *keys, *rows;
    Py_buffer view;
    PyObject *name_tmpl = Py_None, *recycle = Py_None;
    Py_ssize_t offset = 0;  /* of the first statistics row inside buf (the result block itself, read in place) */
    if (!PyArg_ParseTuple(args, "O!O!y*O!|OOn", &PyTuple_Type, &names, &PyTuple_Type, &keys, &view, &PyTuple_Type, &rows, &name_tmpl, &recycle,
                          &offset))
        return NULL;
    if (name_tmpl == Py_None || !PyDict_CheckExact(name_tmpl)) name_tmpl = NULL;
    if (recycle == Py_None || !PyList_CheckExact(recycle)) recycle = NULL;
    PyObject *out = NULL, *key_tmpl = NULL, *stack_inner[MAX_STACK_HASHES], **inner = NULL;
    PyObject *prev = take_recycled(recycle, 0);
    Py_hash_t kh[6];
    const Py_ssize_t n_names = PyTuple_GET_SIZE(names);
    if (offset < 0 || offset > view.len || (offset % (Py_ssize_t)sizeof(float)) != 0) {
        PyErr_SetString(PyExc_ValueError, "nvrx_pyread.summaries: bad offset");
        goto done;
    }
    const Py_ssize_t total_rows = (view.len - offset) / (Py_ssize_t)(8 * sizeof(float));
    if (PyTuple_GET_SIZE(keys) != 6 || PyTuple_GET_SIZE(rows) != n_names) {
        PyErr_SetString(PyExc_ValueError, "nvrx_pyread.summaries: six statistic keys and one row per name expected");
        goto done;
    }
    for (int k = 0; k < 6; k++) {
        kh[k] = PyObject_Hash(PyTuple_GET_ITEM(keys, k));
        if (kh[k] == -1 && PyErr_Occurred()) goto done;
    }
    if (g_inplace) {
        key_tmpl = template_dict(keys);
        if (!key_tmpl) PyErr_Clear();
    }
    inner = n_names <= MAX_STACK_HASHES ? stack_inner : PyMem_Malloc((size_t)n_names * sizeof(PyObject *));
    if (!inner) {
        PyErr_NoMemory();
        goto done;
    }
    const float *p = (const float *)((const char *)view.buf + offset);
    Py_ssize_t built = 0;
#if NVRX_INPLACE_POSSIBLE
    nvrx_dict_entry310 *eo = recyclable_entries(prev, names, n_names);
#endif
    for (; built < n_names; built++) {
        const long row = PyLong_AsLong(PyTuple_GET_ITEM(rows, built));
        if (row == -1 && PyErr_Occurred()) goto fail;
        if (row < 0 || row >= total_rows) {
            PyErr_SetString(PyExc_ValueError, "nvrx_pyread.summaries: row out of range");
            goto fail;
        }
        const float *v = p + row * 8;
        PyObject *vals[6];
        /* NUM (column 5) is an integer in the reference's summaries (straggler.py:194); a NUM that is not a finite number
         * cannot come out of the statistics kernel: ValueError / OverflowError from PyLong_FromDouble, the same exceptions
         * the Python builder raises */
        for (int k = 0; k < 6; k++) vals[k] = k == 5 ? PyLong_FromDouble((double)v[k]) : PyFloat_FromDouble((double)v[k]);
#if NVRX_INPLACE_POSSIBLE
        if (eo) {  /* the previous mapping is being refilled: nothing is collected in inner[] */
            PyObject *old = eo[built].me_value;
            if (refill_inner(old, keys, vals, 6)) continue;
            PyObject *fresh = dict_from(keys, vals, 6, key_tmpl, kh, 0);  /* (also the way a non-finite NUM raises) */
            if (!fresh) goto done;
            eo[built].me_value = fresh;
            Py_DECREF(old);
            continue;
        }
#endif
        inner[built] = dict_from(keys, vals, 6, key_tmpl, kh, 0);
        if (!inner[built]) goto fail;
    }
#if NVRX_INPLACE_POSSIBLE
    if (eo) {
        out = prev;
        prev = NULL;
        keep_recycled(recycle, 0, out);
        goto done;
    }
#endif
    out = dict_from(names, inner, n_names, name_tmpl, NULL, 1);  /* consumes the inner dicts */
    keep_recycled(recycle, 0, out);
    goto done;
fail:
#if NVRX_INPLACE_POSSIBLE
    if (!eo)
#endif
        for (Py_ssize_t j = 0; j < built; j++) Py_DECREF(inner[j]);
done:
    Py_XDECREF(prev);
    if (inner && inner != stack_inner) PyMem_Free(inner);
    Py_XDECREF(key_tmpl);
    PyBuffer_Release(&view);
    return out;
}

static PyObject *copy_set_dict(PyObject *d);

static PyObject *pyread_copy_sets(PyObject *self, PyObject *arg) {
    if (!PyDict_Check(arg)) {
        PyErr_SetString(PyExc_TypeError, "nvrx_pyread.copy_sets: a dict of sets expected");
        return NULL;
    }
    return copy_set_dict(arg);  /* a set built from a set keeps the stored hashes: nothing is re-hashed */
}

/* one level deep copy of {name: set}; NULL on error */
static PyObject *copy_set_dict(PyObject *d) {
    PyObject *out = PyDict_New();
    if (!out) return NULL;
    PyObject *key, *value;
    Py_ssize_t pos = 0;
    while (PyDict_Next(d, &pos, &key, &value)) {
        PyObject *c = PySet_New(value);
        if (!c || PyDict_SetItem(out, key, c) < 0) {
            Py_XDECREF(c);
            Py_DECREF(out);
            return NULL;
        }
        Py_DECREF(c);
    }
    return out;
}

/* the flagged rows of one column as a set of ids[r]; *out stays NULL when the column holds no flag */
static int column_members(const unsigned char *f, int rows, int width, int col, PyObject **ids, PyObject **out) {
    for (int r = 0; r < rows; r++) {
        if (!f[(Py_ssize_t)r * width + col]) continue;
        if (!*out && !(*out = PySet_New(NULL))) return -1;
        if (PySet_Add(*out, ids[r]) < 0) return -1;
    }
    return 0;
}

static PyObject *pyread_flagged(PyObject *self, PyObject *args) {
    Py_buffer view;
    Py_ssize_t offset;
    int rows, width, S, has_rel, has_indiv;
    PyObject *ids_obj, *names, *cols, *memo;
    if (!PyArg_ParseTuple(args, "y*niiippOO!OO", &view, &offset, &rows, &width, &S, &has_rel, &has_indiv, &ids_obj, &PyTuple_Type,
                          &names, &cols, &memo))
        return NULL;
    PyObject *ids_fast = NULL, *gr = NULL, *gi = NULL, *sr = NULL, *si = NULL, *result = NULL;
    const Py_ssize_t n_names = PyTuple_GET_SIZE(names);
    const Py_ssize_t n = (Py_ssize_t)rows * width;
    if (rows < 0 || S < 0 || width != 2 + 2 * S || offset < 0 || view.len < offset + n ||
        (cols != Py_None && (!PyTuple_Check(cols) || PyTuple_GET_SIZE(cols) != n_names)) ||
        (memo != Py_None && (!PyList_Check(memo) || PyList_GET_SIZE(memo) != 2))) {
        PyErr_SetString(PyExc_ValueError, "nvrx_pyread.flagged: inconsistent shapes");
        goto done;
    }
    ids_fast = PySequence_Fast(ids_obj, "nvrx_pyread.flagged: ids must be a sequence");
    if (!ids_fast) goto done;
    if (PySequence_Fast_GET_SIZE(ids_fast) != rows) {
        PyErr_SetString(PyExc_ValueError, "nvrx_pyread.flagged: one id per row expected");
        goto done;
    }
    PyObject **ids = PySequence_Fast_ITEMS(ids_fast);
    const unsigned char *f = (const unsigned char *)view.buf + offset;

    unsigned char any = 0;
    for (Py_ssize_t i = 0; i < n; i++) any |= f[i];
    if (any && memo != Py_None) {
        /* a straggler usually stays one for many reports: the sets of an unchanged flag table are handed out as copies
         * (a set built from a set keeps the stored hashes, StragglerId.__hash__ is a Python function) */
        PyObject *key = PyList_GET_ITEM(memo, 0), *hit = PyList_GET_ITEM(memo, 1);
        /* the hit belongs to one (ids, names, cols, families) combination: a memo list shared across views must miss */
        if (PyBytes_Check(key) && PyBytes_GET_SIZE(key) == n && PyTuple_Check(hit) && PyTuple_GET_SIZE(hit) == 9 &&
            PyTuple_GET_ITEM(hit, 0) == ids_obj && PyTuple_GET_ITEM(hit, 5) == names && PyTuple_GET_ITEM(hit, 6) == cols &&
            PyTuple_GET_ITEM(hit, 7) == (has_rel ? Py_True : Py_False) && PyTuple_GET_ITEM(hit, 8) == (has_indiv ? Py_True : Py_False) &&
            memcmp(PyBytes_AS_STRING(key), f, (size_t)n) == 0) {
            gr = PySet_New(PyTuple_GET_ITEM(hit, 1));
            gi = PySet_New(PyTuple_GET_ITEM(hit, 2));
            sr = copy_set_dict(PyTuple_GET_ITEM(hit, 3));
            si = copy_set_dict(PyTuple_GET_ITEM(hit, 4));
            if (gr && gi && sr && si) result = PyTuple_Pack(4, gr, gi, sr, si);
            goto done;
        }
    }
    gr = PySet_New(NULL);
    gi = PySet_New(NULL);
    sr = PyDict_New();
    si = PyDict_New();
    if (!gr || !gi || !sr || !si) goto done;
    if (any) {
        PyObject *tmp = NULL;
        if (has_indiv) {
            if (column_members(f, rows, width, 0, ids, &tmp) < 0) { Py_XDECREF(tmp); goto done; }
            if (tmp) { Py_SETREF(gi, tmp); tmp = NULL; }
        }
        if (has_rel) {
            if (column_members(f, rows, width, 1, ids, &tmp) < 0) { Py_XDECREF(tmp); goto done; }
            if (tmp) { Py_SETREF(gr, tmp); tmp = NULL; }
        }
        for (int family = 0; family < 2; family++) {  /* 0: individual (columns 2..), 1: relative (columns 2+S..) */
            if (!(family ? has_rel : has_indiv)) continue;
            PyObject *dst = family ? sr : si;
            for (Py_ssize_t i = 0; i < n_names; i++) {  /* the report's own section order */
                long c = i;
                if (cols != Py_None) {
                    c = PyLong_AsLong(PyTuple_GET_ITEM(cols, i));
                    if (c == -1 && PyErr_Occurred()) goto done;
                }
                if (c < 0 || c >= S) {
                    PyErr_SetString(PyExc_ValueError, "nvrx_pyread.flagged: column out of range");
                    goto done;
                }
                tmp = NULL;
                if (column_members(f, rows, width, 2 + (family ? S : 0) + (int)c, ids, &tmp) < 0 ||
                    (tmp && PyDict_SetItem(dst, PyTuple_GET_ITEM(names, i), tmp) < 0)) {
                    Py_XDECREF(tmp);
                    goto done;
                }
                Py_XDECREF(tmp);
            }
        }
        if (memo != Py_None) {
            PyObject *key = PyBytes_FromStringAndSize((const char *)f, n);
            PyObject *c_gr = PySet_New(gr), *c_gi = PySet_New(gi), *c_sr = copy_set_dict(sr), *c_si = copy_set_dict(si);
            PyObject *hit = (key && c_gr && c_gi && c_sr && c_si)
                                ? PyTuple_Pack(9, ids_obj, c_gr, c_gi, c_sr, c_si, names, cols, has_rel ? Py_True : Py_False,
                                               has_indiv ? Py_True : Py_False)
                                : NULL;
            Py_XDECREF(c_gr);
            Py_XDECREF(c_gi);
            Py_XDECREF(c_sr);
            Py_XDECREF(c_si);
            if (!key || !hit) {  /* both objects exist before either slot changes: the memo is never half updated */
                Py_XDECREF(key);
                Py_XDECREF(hit);
                goto done;
            }
            /* (memo was checked to be a list of two: these cannot fail; SetItem steals the references) */
            PyList_SetItem(memo, 0, key);
            PyList_SetItem(memo, 1, hit);
        }
    }
    result = PyTuple_Pack(4, gr, gi, sr, si);
done:
    Py_XDECREF(gr);
    Py_XDECREF(gi);
    Py_XDECREF(sr);
    Py_XDECREF(si);
    Py_XDECREF(ids_fast);
    PyBuffer_Release(&view);
    return result;
}

/* Self-test of the in-place fill (import time): a dict of eight int keys and one of six str keys, cloned, filled and read
 * back through the public API, then used as a normal dict (insert, delete, compare).  1 = every check agreed. */
static int inplace_self_test(void) {
#if NVRX_INPLACE_POSSIBLE
    int ok = 0;
    PyObject *keys = NULL, *tmpl = NULL, *d = NULL, *ref = NULL, *vals[8];
    const int saved = g_inplace;
    for (int pass = 0; pass < 2; pass++) {
        const Py_ssize_t n = pass ? 6 : 8;
        keys = PyTuple_New(n);
        if (!keys) goto out;
        for (Py_ssize_t i = 0; i < n; i++) {
            PyObject *k = pass ? PyUnicode_FromFormat("key%zd", i) : PyLong_FromSsize_t(i * 1000003 + 7);
            if (!k) goto out;
            PyTuple_SET_ITEM(keys, i, k);
        }
        tmpl = template_dict(keys);
        ref = PyDict_New();
        if (!tmpl || !ref) goto out;
        for (Py_ssize_t i = 0; i < n; i++) {
            vals[i] = PyFloat_FromDouble(0.5 + (double)i);
            if (!vals[i] || PyDict_SetItem(ref, PyTuple_GET_ITEM(keys, i), vals[i]) < 0) goto out;
        }
        g_inplace = 1;
        if (inplace_entries(tmpl, n) == NULL) goto out;  /* the template itself must look as expected */
        d = dict_from(keys, vals, n, tmpl, NULL, 0);     /* consumes vals */
        if (!d || PyDict_GET_SIZE(d) != n || PyObject_RichCompareBool(d, ref, Py_EQ) != 1) goto out;
        Py_ssize_t pos = 0, seen = 0;
        PyObject *k, *v;
        while (PyDict_Next(d, &pos, &k, &v)) {           /* insertion order, the very objects */
            if (k != PyTuple_GET_ITEM(keys, seen) || v != PyDict_GetItem(ref, k)) goto out;
            seen++;
        }
        if (seen != n) goto out;
        for (Py_ssize_t i = 0; i < n; i++)                /* the template is untouched */
            if (PyDict_GetItem(tmpl, PyTuple_GET_ITEM(keys, i)) != Py_None) goto out;
        /* ... and the filled clone behaves: grows, shrinks, compares */
        if (PyDict_SetItemString(d, "extra", Py_True) < 0 || PyDict_GET_SIZE(d) != n + 1) goto out;
        if (PyDict_DelItemString(d, "extra") < 0 || PyObject_RichCompareBool(d, ref, Py_EQ) != 1) goto out;
        if (PyDict_DelItem(d, PyTuple_GET_ITEM(keys, 0)) < 0 || PyDict_GET_SIZE(d) != n - 1) goto out;
        Py_CLEAR(keys);
        Py_CLEAR(tmpl);
        Py_CLEAR(d);
        Py_CLEAR(ref);
    }
    ok = 1;
out:
    if (PyErr_Occurred()) PyErr_Clear();
    Py_XDECREF(keys);
    Py_XDECREF(tmpl);
    Py_XDECREF(d);
    Py_XDECREF(ref);
    g_inplace = saved;
    return ok;
#else
    return 0;
#endif
}

static PyObject *pyread_inplace(PyObject *self, PyObject *args) {
    int want = -1;
    if (!PyArg_ParseTuple(args, "|i", &want)) return NULL;
    if (want == 0) g_inplace = 0;
    if (want > 0) g_inplace = inplace_self_test();
    return PyBool_FromLong(g_inplace);
}

static PyMethodDef pyread_methods[] = {
    {"inplace", pyread_inplace, METH_VARARGS, "inplace([on]) -> bool: is the in-place fill of cloned dicts active (0 turns it off, 1 re-runs the self-test)"},
    {"flagged", pyread_flagged, METH_VARARGS, "the sets of identify_stragglers from the score kernel's flag bytes"},
    {"sections", pyread_sections, METH_VARARGS, "section -> {rank -> score} from an f32 score block (one or both score families)"},
    {"ranks", pyread_ranks, METH_VARARGS, "rank -> score: one column of an f32 score block"},
    {"summaries", pyread_summaries, METH_VARARGS, "name -> {Statistic -> value} from f32 statistics rows"},
    {"copy_sets", pyread_copy_sets, METH_O, "a dict of sets, copied one level deep"},
    {NULL, NULL, 0, NULL},
};

static struct PyModuleDef pyread_module = {PyModuleDef_HEAD_INIT, "_nvrx_pyread", "dict builders of the straggler Report", -1,
                                           pyread_methods};

PyMODINIT_FUNC PyInit__nvrx_pyread(void) {
    PyObject *m = PyModule_Create(&pyread_module);
    if (!m) return NULL;
    const char *e = getenv("NVRX_DEBUG_PYREAD_INPLACE");
    g_inplace = (e && e[0] == '0') ? 0 : inplace_self_test();
    return m;
}
