/* wfk_pyflatten.c -- the hot case of waveforms_amd/_flatten.py:flatten() as a CPython extension.
 *
 * Host-side serialisation only (no arithmetic on samples): walks the tuple IR of the reference
 * (expr = (terms, amps), term = (factors, powers), factor = (Type, *args, shift); waveforms/_waveform.pyx:15-48)
 * in the order of Waveform._tolist (waveforms/waveform.py:259-276) and appends to the struct-of-arrays of
 * include/wfk.h.  A 2 GS/s channel of 1668 pulses is 1668 pieces / 4600 terms / 10900 factors: the pure-Python
 * walk costs ~11 us per piece (19-28 ms per channel), this one ~0.3.  Anything outside the hot case -- a factor
 * that is not a native fixed-arity primitive, a complex power, a bounds list that does not end in +inf -- makes
 * flatten_members() return None and the Python walk takes the whole call.
 */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <math.h>
#include <stdint.h>
#include <string.h>

typedef struct {
  char* p;
  size_t len, cap;     /* bytes */
} Buf;

static int buf_push(Buf* b, const void* src, size_t n) {
  if (b->len + n > b->cap) {
    size_t cap = b->cap ? b->cap * 2 : 4096;
    while (cap < b->len + n) cap *= 2;
    char* q = (char*)realloc(b->p, cap);
    if (!q) return -1;
    b->p = q;
    b->cap = cap;
  }
  memcpy(b->p + b->len, src, n);
  b->len += n;
  return 0;
}
/* (the common case -- room left -- stays inline: the walk appends ~6 values per factor, 11 000 factors per channel) */
#define PUSH(buf, type, value)                                      \
  do {                                                              \
    type v_ = (type)(value);                                        \
    if ((buf).len + sizeof v_ <= (buf).cap) {                       \
      memcpy((buf).p + (buf).len, &v_, sizeof v_);                  \
      (buf).len += sizeof v_;                                       \
    } else if (buf_push(&(buf), &v_, sizeof v_)) goto nomem;        \
  } while (0)

/* a Python float (np.float64 is one) or int -> double; 0 on success, -1: anything else (error cleared) -- other
 * numeric types go to the Python walk: np.complex64 would answer __float__ by dropping its imaginary part */
static inline int as_double(PyObject* o, double* out) {
  if (PyFloat_CheckExact(o)) { *out = PyFloat_AS_DOUBLE(o); return 0; }      /* (no subtype walk for the usual case) */
#if PY_VERSION_HEX < 0x030C0000
  if (PyLong_CheckExact(o)) {                                                /* powers: small ints, one digit read in place */
    const Py_ssize_t sz = Py_SIZE(o);
    if (sz == 0) { *out = 0.0; return 0; }
    if (sz == 1) { *out = (double)((PyLongObject*)o)->ob_digit[0]; return 0; }
    if (sz == -1) { *out = -(double)((PyLongObject*)o)->ob_digit[0]; return 0; }
  }
#endif
  if (PyFloat_Check(o)) { *out = PyFloat_AS_DOUBLE(o); return 0; }
  if (!PyLong_Check(o) || PyBool_Check(o)) return -1;
  double v = PyLong_AsDouble(o);
  if (v == -1.0 && PyErr_Occurred()) { PyErr_Clear(); return -1; }
  *out = v;
  return 0;
}

/* flatten_members(members, argc) -> None | (mb_piece_off, pc_bound, pc_term_off, tm_amp_re, tm_amp_im, tm_factor_off,
 *                                            fc_type, fc_power, fc_shift, fc_arg_off, pool, any_complex)
 * members: sequence of (bounds, seq); argc: dict {type id: number of scalar args} of the primitives that may be
 * appended as they are.  The offset arrays are local to the call (they start at 0, one entry more than items). */
static PyObject* flatten_members(PyObject* self, PyObject* args) {
  PyObject *members, *argc;
  if (!PyArg_ParseTuple(args, "OO!", &members, &PyDict_Type, &argc)) return NULL;
  Buf mb = {0}, bound = {0}, pt = {0}, are = {0}, aim = {0}, tf = {0}, ft = {0}, fp = {0}, fs = {0}, fa = {0}, pool = {0};
  int any_complex = 0, unsupported = 0;
  int32_t n_pieces = 0, n_terms = 0, n_factors = 0;
  int64_t n_pool = 0;
  /* the dict as a table: primitive ids are small non-negative ints (include/wfk.h) */
  int8_t argc_tab[64];
  memset(argc_tab, -1, sizeof argc_tab);
  {
    PyObject *key, *val;
    Py_ssize_t pos = 0;
    while (PyDict_Next(argc, &pos, &key, &val)) {
      const long id = PyLong_AsLong(key), na = PyLong_AsLong(val);
      if ((id == -1 || na == -1) && PyErr_Occurred()) { PyErr_Clear(); continue; }
      if (id >= 0 && id < 64 && na >= 0 && na < 100) argc_tab[id] = (int8_t)na;
    }
  }
  PyObject* mseq = PySequence_Fast(members, "members must be a sequence");
  if (!mseq) return NULL;
  PUSH(mb, int32_t, 0);
  PUSH(pt, int32_t, 0);
  PUSH(tf, int32_t, 0);
  PUSH(fa, int64_t, 0);
  for (Py_ssize_t im = 0; im < PySequence_Fast_GET_SIZE(mseq) && !unsupported; ++im) {
    PyObject* member = PySequence_Fast_GET_ITEM(mseq, im);
    if (!PyTuple_Check(member) || PyTuple_GET_SIZE(member) != 2) { unsupported = 1; break; }
    PyObject* bounds = PyTuple_GET_ITEM(member, 0);
    PyObject* seq = PyTuple_GET_ITEM(member, 1);
    if (!PyTuple_Check(bounds) || !PyTuple_Check(seq) || PyTuple_GET_SIZE(bounds) != PyTuple_GET_SIZE(seq) ||
        PyTuple_GET_SIZE(bounds) == 0) { unsupported = 1; break; }
    const Py_ssize_t np_ = PyTuple_GET_SIZE(bounds);
    double last;
    if (as_double(PyTuple_GET_ITEM(bounds, np_ - 1), &last) || !(isinf(last) && last > 0)) { unsupported = 1; break; }
    for (Py_ssize_t ip = 0; ip < np_ && !unsupported; ++ip) {
      /* (the walk is pointer chasing over objects touched for the first time: ask for the next piece's tuples early) */
      if (ip + 1 < np_) {
        PyObject* nx = PyTuple_GET_ITEM(seq, ip + 1);
        __builtin_prefetch(nx);
        __builtin_prefetch(PyTuple_GET_ITEM(bounds, ip + 1));
      }
      double b;
      if (as_double(PyTuple_GET_ITEM(bounds, ip), &b)) { unsupported = 1; break; }
      PUSH(bound, double, b);
      ++n_pieces;
      PyObject* expr = PyTuple_GET_ITEM(seq, ip);
      if (!PyTuple_Check(expr) || PyTuple_GET_SIZE(expr) != 2) { unsupported = 1; break; }
      PyObject* terms = PyTuple_GET_ITEM(expr, 0);
      PyObject* amps = PyTuple_GET_ITEM(expr, 1);
      if (!PyTuple_Check(terms) || !PyTuple_Check(amps)) { unsupported = 1; break; }
      Py_ssize_t nt = PyTuple_GET_SIZE(terms);
      if (PyTuple_GET_SIZE(amps) < nt) nt = PyTuple_GET_SIZE(amps);       /* zip() */
      for (Py_ssize_t it = 0; it < nt && !unsupported; ++it) {
        PyObject* term = PyTuple_GET_ITEM(terms, it);
        PyObject* amp = PyTuple_GET_ITEM(amps, it);
        if (it + 1 < nt) {
          __builtin_prefetch(PyTuple_GET_ITEM(terms, it + 1));
          __builtin_prefetch(PyTuple_GET_ITEM(amps, it + 1));
        }
        if (!PyTuple_Check(term) || PyTuple_GET_SIZE(term) != 2) { unsupported = 1; break; }
        PyObject* factors = PyTuple_GET_ITEM(term, 0);
        PyObject* powers = PyTuple_GET_ITEM(term, 1);
        if (!PyTuple_Check(factors) || !PyTuple_Check(powers)) { unsupported = 1; break; }
        double re, im = 0.0;
        if (PyFloat_CheckExact(amp)) {
          re = PyFloat_AS_DOUBLE(amp);
        } else if (PyComplex_Check(amp)) {
          /* (isinstance(amp, complex): Python complex and np.complex128; np.complex64 is not one -> Python walk) */
          re = PyComplex_RealAsDouble(amp);
          im = PyComplex_ImagAsDouble(amp);
          any_complex = 1;
        } else if (as_double(amp, &re)) { unsupported = 1; break; }
        Py_ssize_t nf = PyTuple_GET_SIZE(factors);
        if (PyTuple_GET_SIZE(powers) < nf) nf = PyTuple_GET_SIZE(powers);
        for (Py_ssize_t k = 0; k < nf; ++k) {
          PyObject* f = PyTuple_GET_ITEM(factors, k);
          if (k + 1 < nf) {
            __builtin_prefetch(PyTuple_GET_ITEM(factors, k + 1));
            __builtin_prefetch(PyTuple_GET_ITEM(powers, k + 1));
          }
          if (!PyTuple_Check(f) || PyTuple_GET_SIZE(f) < 2) { unsupported = 1; break; }
          PyObject* tid = PyTuple_GET_ITEM(f, 0);
          if (!PyLong_CheckExact(tid)) { unsupported = 1; break; }
          long id;
#if PY_VERSION_HEX < 0x030C0000
          /* (type ids are small non-negative ints: one digit -- read it in place; CPython < 3.12 layout) */
          if (Py_SIZE(tid) == 1) id = (long)((PyLongObject*)tid)->ob_digit[0];
          else if (Py_SIZE(tid) == 0) id = 0;
          else
#endif
          {
            id = PyLong_AsLong(tid);
            if (id == -1 && PyErr_Occurred()) { PyErr_Clear(); unsupported = 1; break; }
          }
          if (id < 0 || id >= 64 || argc_tab[id] < 0) { unsupported = 1; break; }   /* not a plain native primitive */
          const long na = argc_tab[id];
          if (PyTuple_GET_SIZE(f) != na + 2) { unsupported = 1; break; }
          double pw, sh;
          if (as_double(PyTuple_GET_ITEM(powers, k), &pw) || as_double(PyTuple_GET_ITEM(f, na + 1), &sh)) { unsupported = 1; break; }
          for (long a = 0; a < na; ++a) {
            double v;
            if (as_double(PyTuple_GET_ITEM(f, 1 + a), &v)) { unsupported = 1; break; }
            PUSH(pool, double, v);
            ++n_pool;
          }
          if (unsupported) break;
          PUSH(ft, int32_t, id);
          PUSH(fp, double, pw);
          PUSH(fs, double, sh);
          PUSH(fa, int64_t, n_pool);
          ++n_factors;
        }
        if (unsupported) break;
        PUSH(are, double, re);
        PUSH(aim, double, im);
        PUSH(tf, int32_t, n_factors);
        ++n_terms;
      }
      PUSH(pt, int32_t, n_terms);
    }
    PUSH(mb, int32_t, n_pieces);
  }
  Py_DECREF(mseq);
  PyObject* out = NULL;
  if (unsupported) {
    out = Py_None;
    Py_INCREF(out);
  } else {
    /* bytearrays: NumPy wraps them without a copy and the arrays stay writable */
    Buf* all[11] = {&mb, &bound, &pt, &are, &aim, &tf, &ft, &fp, &fs, &fa, &pool};
    out = PyTuple_New(12);
    for (int i = 0; out && i < 11; ++i) {
      PyObject* ba = PyByteArray_FromStringAndSize(all[i]->p ? all[i]->p : "", (Py_ssize_t)all[i]->len);
      if (!ba) { Py_CLEAR(out); break; }
      PyTuple_SET_ITEM(out, i, ba);
    }
    if (out) PyTuple_SET_ITEM(out, 11, PyLong_FromLong(any_complex));
  }
  free(mb.p); free(bound.p); free(pt.p); free(are.p); free(aim.p); free(tf.p); free(ft.p); free(fp.p); free(fs.p); free(fa.p); free(pool.p);
  return out;
nomem:
  Py_DECREF(mseq);
  free(mb.p); free(bound.p); free(pt.p); free(are.p); free(aim.p); free(tf.p); free(ft.p); free(fp.p); free(fs.p); free(fa.p); free(pool.p);
  return PyErr_NoMemory();
}

static PyMethodDef methods[] = {
    {"flatten_members", flatten_members, METH_VARARGS,
     "flatten_members(members, argc) -> None | tuple of bytearrays (struct-of-arrays of the members) + any_complex"},
    {NULL, NULL, 0, NULL}};

static struct PyModuleDef module = {PyModuleDef_HEAD_INIT, "_cflatten", "native walk of the tuple IR (host-side serialisation)", -1, methods};

PyMODINIT_FUNC PyInit__cflatten(void) { return PyModule_Create(&module); }
