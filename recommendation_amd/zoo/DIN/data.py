"""Batch iterators of the DIN entry point; same contract as the reference's DIN/data.py:4-76.

A train record is (uid, hist_list, item, label); a test record is (uid, hist_list, (pos, neg)).
Iteration yields (batch_number, (u, i, y_or_j, hist_i, sl)) with hist_i an int64 [B, max_len] array
zero-padded behind each history, batches in file order, the last one possibly short."""
import numpy as np


class _Batches:
    def __init__(self, data, batch_size):
        self.data = data
        self.batch_size = int(batch_size)
        self.epoch_size = -(-len(data) // self.batch_size)
        self.i = 0

    def __iter__(self):
        return self

    def _unpack(self, rec):
        raise NotImplementedError

    def __next__(self):
        if self.i >= self.epoch_size:
            raise StopIteration
        chunk = self.data[self.i * self.batch_size:(self.i + 1) * self.batch_size]
        self.i += 1
        u, i, third = [], [], []
        for rec in chunk:
            a, b, c = self._unpack(rec)
            u.append(a)
            i.append(b)
            third.append(c)
        sl = [len(rec[1]) for rec in chunk]
        hist_i = np.zeros((len(chunk), max(sl)), dtype=np.int64)
        for row, rec in enumerate(chunk):
            hist_i[row, :sl[row]] = rec[1]
        return self.i, (u, i, third, hist_i, sl)


class DataInput(_Batches):
    def _unpack(self, rec):
        return rec[0], rec[2], rec[3]


class DataInputTest(_Batches):
    def _unpack(self, rec):
        return rec[0], rec[2][0], rec[2][1]
