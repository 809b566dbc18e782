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
 *   sections(names, ranks, scores, n_rows, width, first_col, cols) -> {name: {rank: float}}
 *   summaries(names, stat_keys, stats, rows) -> {name: {stat_key: float, ..., stat_keys[5]: int}}
 *   copy_sets(d) -> {key: set(value) for key, value in d.items()}      (fresh sets for every caller of identify_stragglers)
 */
#define PY_SSIZE_T_CLEAN
#include <Python.h>

static PyObject *pyread_sections(PyObject *self, PyObject *args) {
    PyObject *names, *ranks, *cols;
    Py_buffer view;
    int n_rows, width, first_col;
    if (!PyArg_ParseTuple(args, "O!O!y*iiiO", &PyTuple_Type, &names, &PyTuple_Type, &ranks, &view, &n_rows, &width, &first_col, &cols))
        return NULL;
    PyObject *out = NULL;
    const Py_ssize_t n_names = PyTuple_GET_SIZE(names);
    if (n_rows < 0 || width <= 0 || first_col < 0 || PyTuple_GET_SIZE(ranks) != n_rows ||
        view.len < (Py_ssize_t)n_rows * width * (Py_ssize_t)sizeof(float) ||
        (cols != Py_None && (!PyTuple_Check(cols) || PyTuple_GET_SIZE(cols) != n_names))) {
        PyErr_SetString(PyExc_ValueError, "nvrx_pyread.sections: inconsistent shapes");
        goto done;
    }
    out = PyDict_New();
    if (!out) goto done;
    const float *p = (const float *)view.buf;
    for (Py_ssize_t i = 0; i < n_names; i++) {
        long c = i;
        if (cols != Py_None) {
            c = PyLong_AsLong(PyTuple_GET_ITEM(cols, i));
            if (c == -1 && PyErr_Occurred()) goto fail;
        }
        if (c < 0 || first_col + c >= width) {
            PyErr_SetString(PyExc_ValueError, "nvrx_pyread.sections: column out of range");
            goto fail;
        }
        PyObject *d = PyDict_New();
        if (!d) goto fail;
        for (int r = 0; r < n_rows; r++) {
            PyObject *f = PyFloat_FromDouble((double)p[(Py_ssize_t)r * width + first_col + c]);
            if (!f || PyDict_SetItem(d, PyTuple_GET_ITEM(ranks, r), f) < 0) {
                Py_XDECREF(f);
                Py_DECREF(d);
                goto fail;
            }
            Py_DECREF(f);
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

static PyObject *pyread_summaries(PyObject *self, PyObject *args) {
    PyObject *names, *keys, *rows;
    Py_buffer view;
    if (!PyArg_ParseTuple(args, "O!O!y*O!", &PyTuple_Type, &names, &PyTuple_Type, &keys, &view, &PyTuple_Type, &rows)) return NULL;
    PyObject *out = NULL;
    const Py_ssize_t n_names = PyTuple_GET_SIZE(names);
    const Py_ssize_t total_rows = view.len / (Py_ssize_t)(8 * sizeof(float));
    if (PyTuple_GET_SIZE(keys) != 6 || PyTuple_GET_SIZE(rows) != n_names) {
        PyErr_SetString(PyExc_ValueError, "nvrx_pyread.summaries: six statistic keys and one row per name expected");
        goto done;
    }
    out = PyDict_New();
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
        PyObject *d = PyDict_New();
        if (!d) goto fail;
        for (int k = 0; k < 6; k++) {
            /* NUM (column 5) is an integer in the reference's summaries (straggler.py:194) */
            PyObject *x = k == 5 ? PyLong_FromDouble((double)v[k]) : PyFloat_FromDouble((double)v[k]);
            if (!x || PyDict_SetItem(d, PyTuple_GET_ITEM(keys, k), x) < 0) {
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

static PyObject *pyread_copy_sets(PyObject *self, PyObject *arg) {
    if (!PyDict_Check(arg)) {
        PyErr_SetString(PyExc_TypeError, "nvrx_pyread.copy_sets: a dict of sets expected");
        return NULL;
    }
    PyObject *out = PyDict_New();
    if (!out) return NULL;
    PyObject *key, *value;
    Py_ssize_t pos = 0;
    while (PyDict_Next(arg, &pos, &key, &value)) {
        PyObject *c = PySet_New(value);  /* a set built from a set keeps the stored hashes: nothing is re-hashed */
        if (!c || PyDict_SetItem(out, key, c) < 0) {
            Py_XDECREF(c);
            Py_DECREF(out);
            return NULL;
        }
        Py_DECREF(c);
    }
    return out;
}

static PyMethodDef pyread_methods[] = {
    {"sections", pyread_sections, METH_VARARGS, "section -> {rank -> score} from an f32 score block"},
    {"summaries", pyread_summaries, METH_VARARGS, "name -> {Statistic -> value} from f32 statistics rows"},
    {"copy_sets", pyread_copy_sets, METH_O, "a dict of sets, copied one level deep"},
    {NULL, NULL, 0, NULL},
};

static struct PyModuleDef pyread_module = {PyModuleDef_HEAD_INIT, "_nvrx_pyread", "dict builders of the straggler Report", -1,
                                           pyread_methods};

PyMODINIT_FUNC PyInit__nvrx_pyread(void) { return PyModule_Create(&pyread_module); }
