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
 *   summaries(names, stat_keys, stats, rows) -> {name: {stat_key: float, ..., stat_keys[5]: int}}
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

/* one {name: {rank: float}} mapping over columns first_col + c of the [n_rows][width] f32 block p */
static PyObject *section_mapping(const float *p, int n_rows, int width, int first_col, PyObject *names, PyObject *ranks,
                                 const long *col, const Py_hash_t *name_hash, const Py_hash_t *rank_hash) {
    const Py_ssize_t n_names = PyTuple_GET_SIZE(names);
    PyObject *out = NEW_DICT(n_names);
    if (!out) return NULL;
    for (Py_ssize_t i = 0; i < n_names; i++) {
        const float *q = p + first_col + (col ? col[i] : i);
        PyObject *d = NEW_DICT(n_rows);
        if (!d) goto fail;
        for (int r = 0; r < n_rows; r++) {
            PyObject *f = PyFloat_FromDouble((double)q[(Py_ssize_t)r * width]);
            if (!f || SET_KNOWN(d, PyTuple_GET_ITEM(ranks, r), f, rank_hash[r]) < 0) {
                Py_XDECREF(f);
                Py_DECREF(d);
                goto fail;
            }
            Py_DECREF(f);
        }
        if (SET_KNOWN(out, PyTuple_GET_ITEM(names, i), d, name_hash[i]) < 0) {
            Py_DECREF(d);
            goto fail;
        }
        Py_DECREF(d);
    }
    return out;
fail:
    Py_DECREF(out);
    return NULL;
}

/* sections(names, ranks, buf, offset, n_rows, width, first_col, cols[, second_col]) -> mapping, or a pair of mappings when
 * second_col >= 0 (both score families of a report in one call: the name / rank hashes and the column table are shared).
 * buf + offset: the first of n_rows rows of width f32 (the report's rows inside the result block, or an ndarray). */
static PyObject *pyread_sections(PyObject *self, PyObject *args) {
    PyObject *names, *ranks, *cols;
    Py_buffer view;
    Py_ssize_t offset;
    int n_rows, width, first_col, second_col = -1;
    if (!PyArg_ParseTuple(args, "O!O!y*niiiO|i", &PyTuple_Type, &names, &PyTuple_Type, &ranks, &view, &offset, &n_rows, &width, &first_col,
                          &cols, &second_col))
        return NULL;
    PyObject *out = NULL, *a = NULL, *b = NULL;
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
    a = section_mapping(p, n_rows, width, first_col, names, ranks, col, nh, rh);
    if (!a) goto done;
    if (second_col >= 0) {
        b = section_mapping(p, n_rows, width, second_col, names, ranks, col, nh, rh);
        if (!b) goto done;
        out = PyTuple_Pack(2, a, b);
    } else {
        out = a;
        a = NULL;
    }
done:
    Py_XDECREF(a);
    Py_XDECREF(b);
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
    if (!PyArg_ParseTuple(args, "O!O!y*O!", &PyTuple_Type, &names, &PyTuple_Type, &keys, &view, &PyTuple_Type, &rows)) return NULL;
    PyObject *out = NULL;
    Py_hash_t kh[6];
    const Py_ssize_t n_names = PyTuple_GET_SIZE(names);
    const Py_ssize_t total_rows = view.len / (Py_ssize_t)(8 * sizeof(float));
    if (PyTuple_GET_SIZE(keys) != 6 || PyTuple_GET_SIZE(rows) != n_names) {
        PyErr_SetString(PyExc_ValueError, "nvrx_pyread.summaries: six statistic keys and one row per name expected");
        goto done;
    }
    for (int k = 0; k < 6; k++) {
        kh[k] = PyObject_Hash(PyTuple_GET_ITEM(keys, k));
        if (kh[k] == -1 && PyErr_Occurred()) goto done;
    }
    out = NEW_DICT(n_names);
    if (!out) goto done;
    const float *p = (const float *)view.buf;
    for (Py_ssize_t i = 0; i < n_names; i++) {
        const long row = PyLong_AsLong(PyTuple_GET_ITEM(rows, i));
        if (row == -1 && PyErr_Occurred()) goto fail;
        if (row < 0 || row >= total_rows) {
            PyErr_SetString(PyExc_ValueError, "nvrx_pyread.summaries: row out of range");
            goto fail;
        }
        const float *v = p + row * 8;
        PyObject *d = NEW_DICT(6);
        if (!d) goto fail;
        for (int k = 0; k < 6; k++) {
            /* NUM (column 5) is an integer in the reference's summaries (straggler.py:194); a NUM that is not a finite
             * number cannot come out of the statistics kernel: ValueError / OverflowError from PyLong_FromDouble, the
             * same exceptions the Python builder raises */
            PyObject *x = k == 5 ? PyLong_FromDouble((double)v[k]) : PyFloat_FromDouble((double)v[k]);
            if (!x || SET_KNOWN(d, PyTuple_GET_ITEM(keys, k), x, kh[k]) < 0) {
                Py_XDECREF(x);
                Py_DECREF(d);
                goto fail;
            }
            Py_DECREF(x);
        }
        if (PyDict_SetItem(out, PyTuple_GET_ITEM(names, i), d) < 0) {
            Py_DECREF(d);
            goto fail;
        }
        Py_DECREF(d);
    }
    goto done;
fail:
    Py_CLEAR(out);
done:
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

static PyMethodDef pyread_methods[] = {
    {"flagged", pyread_flagged, METH_VARARGS, "the sets of identify_stragglers from the score kernel's flag bytes"},
    {"sections", pyread_sections, METH_VARARGS, "section -> {rank -> score} from an f32 score block (one or both score families)"},
    {"ranks", pyread_ranks, METH_VARARGS, "rank -> score: one column of an f32 score block"},
    {"summaries", pyread_summaries, METH_VARARGS, "name -> {Statistic -> value} from f32 statistics rows"},
    {"copy_sets", pyread_copy_sets, METH_O, "a dict of sets, copied one level deep"},
    {NULL, NULL, 0, NULL},
};

static struct PyModuleDef pyread_module = {PyModuleDef_HEAD_INIT, "_nvrx_pyread", "dict builders of the straggler Report", -1,
                                           pyread_methods};

PyMODINIT_FUNC PyInit__nvrx_pyread(void) { return PyModule_Create(&pyread_module); }
