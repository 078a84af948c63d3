/*
 * advntr_pyhost.h -- OPTIONAL helper for a CPython host, exported by libadvntr_hip.so beside the C ABI of advntr_hip.h.
 *
 * Not part of the drop-in boundary: advntr_hip.h is plain C (pointers and sizes) and bindable from any host language; the one
 * entry point below knows what a Python list of str is.  It exists because the reference's host language IS Python and a
 * million read strings cost more interpreter time to join and measure than their scoring takes on the device (DESIGN.md
 * section 8); a host in another language hands its reads to advntr_encode_texts / advntr_encode_ascii directly.  The library
 * does not link against libpython: the handful of C-API functions are looked up at run time (dlsym), and the entry point fails
 * with -1 in a process that is not a Python interpreter.
 */
#ifndef ADVNTR_PYHOST_H
#define ADVNTR_PYHOST_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* The buffers of the strings of a Python list, without a call back into the interpreter per string.  list = a PyObject* that is
 * a list of str, handed over WITH the interpreter lock held (ctypes.PyDLL); texts[i] / lengths[i] receive the UTF-8 buffer and
 * length of item i -- for an ASCII str (every read file) that is the str's own buffer, nothing is allocated or cached -- and
 * feed advntr_encode_texts.  CONTRACT: the buffers are borrowed; the caller keeps the list and its strings unchanged until the
 * call that consumes the pointers has returned (advntr_amd/_lib.py holds a reference to the list across both calls).  Returns the
 * number of items, or -1 when the interpreter's C API is not reachable from this process or `list` is not a list, or -(i + 2)
 * when item i is not a str or not ASCII -- the walk stops at that item, so at most ONE non-ASCII str has had its UTF-8 form
 * made (and cached by the interpreter) when the caller falls back to its general route.                                    */
int64_t advntr_pylist_texts(void *list, const char **texts, int64_t *lengths, int64_t capacity);

#ifdef __cplusplus
}
#endif
#endif
