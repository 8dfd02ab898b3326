"""Criteo display-advertising text as input of the CTR models: the step before the feeder.

The reference's loaders read its own small CSV with pandas (DeepFM/main.py:13-43); a Criteo-shape run reads
tab-separated text - label, 13 integer counts, 26 hexadecimal tokens per line - at rates pandas cannot deliver
(the GPU path consumes 10^7 - 10^8 samples per second).  The parsing and the hashing of the tokens into table
rows is native code on the host's cores (csrc/criteo_tsv.cpp, ctr_criteo_tsv_parse); this module maps files,
cuts them into blocks at line boundaries and hands out arrays in the layout CTRModel.run / the Feeder take:
ids int32 [n, 26], dense float32 [n, 13], label float32 [n].
"""
from __future__ import annotations

import ctypes as C
import mmap
import os

import numpy as np

from . import _lib

N_DENSE, N_CATE = 13, 26


def _threads(n):
    if n:
        return int(n)
    env = os.environ.get('OMP_NUM_THREADS')
    return max(1, min(int(env) if env else (os.cpu_count() or 1), 64))


def parse_text(text, cate_list, *, log_dense=True, threads=None, max_rows=None, out=None):
    """text: bytes-like (bytes, bytearray, memoryview, mmap, uint8 array) holding whole lines.
    cate_list: 26 table sizes; a token's row is its hexadecimal value modulo the size.
    out (optional): (ids, dense, label) arrays to fill (e.g. pinned staging buffers), at least max_rows long.
    Returns (ids [n, 26] int32, dense [n, 13] float32, label [n] float32) with n = the number of samples read."""
    lib = _lib.load()
    cards = np.ascontiguousarray(np.asarray(cate_list, dtype=np.int64))
    if cards.shape != (N_CATE,):
        raise ValueError('cate_list must hold %d table sizes' % N_CATE)
    buf = np.frombuffer(text, dtype=np.uint8)
    ptr, n_bytes = C.c_void_p(buf.ctypes.data), int(buf.size)
    if out is None:
        rows = int(lib.ctr_criteo_tsv_count(ptr, n_bytes)) if max_rows is None else int(max_rows)
        out = (np.empty((rows, N_CATE), dtype=np.int32), np.empty((rows, N_DENSE), dtype=np.float32),
               np.empty(rows, dtype=np.float32))
    ids, dense, label = out
    rows = min(len(label), len(ids), len(dense)) if max_rows is None else int(max_rows)
    for a, dt, w in ((ids, np.int32, N_CATE), (dense, np.float32, N_DENSE), (label, np.float32, None)):
        if a.dtype != dt or not a.flags['C_CONTIGUOUS'] or (w is not None and (a.ndim != 2 or a.shape[1] != w)) or len(a) < rows:
            raise ValueError('output arrays must be C-contiguous ids int32 [n, 26], dense float32 [n, 13], label float32 [n]')
    got = C.c_int64(0)
    rc = lib.ctr_criteo_tsv_parse(ptr, n_bytes, C.c_void_p(cards.ctypes.data), int(bool(log_dense)), _threads(threads),
                                  rows, C.c_void_p(label.ctypes.data), C.c_void_p(dense.ctypes.data),
                                  C.c_void_p(ids.ctypes.data), C.byref(got))
    if rc != 0:
        raise ValueError(lib.ctr_last_error().decode())
    n = int(got.value)
    return ids[:n], dense[:n], label[:n]


def iter_file(path, cate_list, *, block_bytes=256 << 20, log_dense=True, threads=None):
    """Yields (ids, dense, label) per block of about block_bytes of the file (cut at line boundaries; the file is
    mapped, never read whole)."""
    size = os.path.getsize(path)
    if size == 0:
        return
    with open(path, 'rb') as fh, mmap.mmap(fh.fileno(), 0, access=mmap.ACCESS_READ) as mm:
        view = memoryview(mm)
        try:
            lo = 0
            while lo < size:
                hi = min(size, lo + int(block_bytes))
                if hi < size:
                    nl = mm.find(b'\n', hi)
                    hi = size if nl < 0 else nl + 1
                yield parse_text(view[lo:hi], cate_list, log_dense=log_dense, threads=threads)
                lo = hi
        finally:
            view.release()


def read_file(path, cate_list, **kw):
    """The whole file as (ids, dense, label)."""
    parts = list(iter_file(path, cate_list, **kw))
    if not parts:
        return (np.empty((0, N_CATE), np.int32), np.empty((0, N_DENSE), np.float32), np.empty(0, np.float32))
    return tuple(np.concatenate([p[i] for p in parts]) for i in range(3))


def load_dataset(file_path, cate_list, *, log_dense=True, threads=None):
    """train.txt / test.txt of a directory in the 7-tuple of the reference's load_dataset (DeepFM/main.py:13-43):
    X_train_cate, X_train_cont, y_train, X_test_cate, X_test_cont, y_test, cate_list - continuous columns
    standardised with statistics over train and test together, as the reference's StandardScaler is fitted
    (DeepFM/main.py:19-24)."""
    tr = read_file(os.path.join(file_path, 'train.txt'), cate_list, log_dense=log_dense, threads=threads)
    te = read_file(os.path.join(file_path, 'test.txt'), cate_list, log_dense=log_dense, threads=threads)
    both = np.concatenate([tr[1], te[1]]).astype(np.float64)
    mean = both.mean(0) if len(both) else np.zeros(N_DENSE)
    std = both.std(0) if len(both) else np.ones(N_DENSE)
    std[std == 0] = 1.0                                   # StandardScaler leaves constant columns at zero
    scale = lambda x: ((x - mean) / std).astype(np.float32)
    return (tr[0], scale(tr[1]), tr[2], te[0], scale(te[1]), te[2], [int(v) for v in cate_list])
