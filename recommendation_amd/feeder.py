"""Host -> device feed of the training loop: the step before the hot path.

The reference hands numpy slices to session.run, which narrows them (int64 -> int32,
float64 -> float32; DeepFM/DeepFM.py:37-39) and copies them synchronously.  Here a worker thread
narrows the next mini-batches into pinned staging buffers and a copy stream moves them to the GPU
while the current step computes; the consumer receives device tensors ready for
CTRModel.train_step / predict_step.  Batch boundaries are the reference loop's
(train_loop.batches: `while end < total`).
"""
from __future__ import annotations

import queue
import threading

import numpy as np
import torch


class Feeder:
    """Iterates (idx int32 [b,F], x_cont float32 [b,C], y float32 [b]) device tensors.

    depth staging slots (pinned host + device); the worker runs up to depth batches ahead.  A
    pinned buffer is refilled only after the copy that read it has finished; a device buffer is
    overwritten only after the compute stream has passed everything the consumer enqueued for the
    batch it held (an event recorded when the consumer asks for the next batch)."""

    def __init__(self, session, X_cate, X_cont, y, ranges, depth=3):
        self.session = session
        self.dev = session.device
        self.Xc, self.Xf, self.y = X_cate, X_cont, y
        self.ranges = list(ranges)
        self.depth = max(2, int(depth))
        bmax = max([e - s for s, e in self.ranges] + [1])
        F, C = X_cate.shape[1], X_cont.shape[1]
        pin = lambda shape, dt: torch.empty(shape, dtype=dt).pin_memory()
        dv = lambda shape, dt: torch.empty(shape, dtype=dt, device=self.dev)
        self.host = [(pin((bmax, F), torch.int32), pin((bmax, C), torch.float32), pin((bmax,), torch.float32))
                     for _ in range(self.depth)]
        self.devb = [(dv((bmax, F), torch.int32), dv((bmax, C), torch.float32), dv((bmax,), torch.float32))
                     for _ in range(self.depth)]
        self.copy_stream = torch.cuda.Stream(self.dev)
        self.copied = [torch.cuda.Event() for _ in range(self.depth)]     # H2D of the slot finished
        self.released = [None] * self.depth                              # compute passed the slot's last use

    @staticmethod
    def _narrow(dst, src):
        """dst (pinned torch) <- src (numpy, any int/float dtype), value-preserving cast
        (torch's copy converts on the intra-op thread pool)."""
        src = np.ascontiguousarray(src)
        dst.copy_(torch.from_numpy(src))

    def _fill(self, q):
        for k, (s, e) in enumerate(self.ranges):
            slot = k % self.depth
            if k >= self.depth:
                self._issued[slot].acquire()         # the copy that read this slot's staging buffer was issued
                self.copied[slot].synchronize()      # ... and has finished
            n = e - s
            h = self.host[slot]
            self._narrow(h[0][:n], self.Xc[s:e])
            self._narrow(h[1][:n], self.Xf[s:e])
            self._narrow(h[2][:n], self.y[s:e])
            q.put((slot, n))
        q.put(None)

    def __iter__(self):
        q = queue.Queue()
        self._issued = [threading.Semaphore(0) for _ in range(self.depth)]
        worker = threading.Thread(target=self._fill, args=(q,), daemon=True)
        worker.start()
        main = torch.cuda.current_stream(self.dev)
        while True:
            msg = q.get()
            if msg is None:
                break
            slot, n = msg
            h, d = self.host[slot], self.devb[slot]
            with torch.cuda.stream(self.copy_stream):
                if self.released[slot] is not None:
                    self.copy_stream.wait_event(self.released[slot])
                for a, b in zip(d, h):
                    a[:n].copy_(b[:n], non_blocking=True)
                self.copied[slot].record(self.copy_stream)
            self._issued[slot].release()
            main.wait_event(self.copied[slot])
            yield d[0][:n], d[1][:n], d[2][:n]
            # everything the consumer enqueued for this batch is on `main` now
            ev = torch.cuda.Event()
            ev.record(main)
            self.released[slot] = ev
        worker.join()
