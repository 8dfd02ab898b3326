"""Session: the opaque handle the model classes receive in run()/saver calls.

Stands where tf.Session stands in the reference's main.py files (DeepFM/main.py:69): it is
only ever passed back into model.run(...) and model.saver.save/restore(...).  Here it carries
the GPU device, the RNG seed of the dropout streams and the scratch workspace.
"""
from __future__ import annotations

import contextlib
import os

import torch

from . import _lib
from .ops import Workspace


class Session:
    def __init__(self, device='cuda:0', seed=0):
        _lib.load()  # fail loudly when the HIP library is not built
        if not torch.cuda.is_available():
            raise RuntimeError('recommendation_amd needs a GPU: the hot path is HIP-only (no CPU fallback)')
        self.device = torch.device(device)
        torch.cuda.set_device(self.device)
        self.seed = int(seed)
        self.ws = Workspace(self.device)
        # second stream + workspace for work that does not feed the rest of the step (the towers' weight
        # gradients run there beside the embedding backward); joined before the dense optimizer
        self.ws_side = Workspace(self.device)
        self._side = None
        self._fork_ev = None
        self._join_ev = None
        self._side_open = False
        self._mark_ev = None
        # False: side_stream() is a no-op context (per-kernel timing passes, graph captures; CTR_NO_SIDE=1: experiments)
        self.side_enabled = os.environ.get('CTR_NO_SIDE', '0') != '1'
        # captured graphs (with the static buffers, streams and events they refer to) whose model was closed or
        # collected: destroyed only at a quiescent point (drain_graphs), never from the garbage collector
        self._retired = []

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def close(self):
        """Waits for the device and destroys the graphs of models that are gone."""
        torch.cuda.synchronize(self.device)
        self.drain_graphs()

    def retire_graphs(self, entries):
        """Takes over captured-graph entries (sequences whose first item is the torch.cuda.CUDAGraph) from a model
        that no longer needs them.  Nothing is destroyed here: this may run inside the garbage collector."""
        self._retired.extend(entries)

    def drain_graphs(self):
        """Destroys retired graphs.  Graph executables are released only here, with the device idle before and
        after: first the graph, then the buffers, streams and events its nodes point at."""
        if not self._retired:
            return
        torch.cuda.synchronize(self.device)
        while self._retired:
            items = list(self._retired.pop())
            while items:
                items.pop(0)                 # the graph goes first; its last reference dies here
        torch.cuda.synchronize(self.device)

    def side_stream(self):
        """Context manager: the enclosed launches go to the side stream, ordered after everything issued so far
        on the current stream.  Kernels launched inside must take their scratch from `ws_side`."""
        if not self.side_enabled:
            return contextlib.nullcontext()
        if self._side is None:
            self._side = torch.cuda.Stream(self.device)
            self._fork_ev = torch.cuda.Event()
            self._join_ev = torch.cuda.Event()
        self._fork_ev.record()
        self._side.wait_event(self._fork_ev)
        self._side_open = True
        return torch.cuda.stream(self._side)

    def side_join(self):
        """The current stream waits for everything issued to the side stream."""
        if self._side_open:
            self._join_ev.record(self._side)
            torch.cuda.current_stream().wait_event(self._join_ev)
            self._side_open = False

    def side_mark(self):
        """Records a mark on the side stream after everything issued to it so far; wait_mark() makes the current
        stream wait for it (for side work whose result the main stream needs before the join, e.g. the pair sort
        that runs beside the forward pass)."""
        if self._mark_ev is None:
            self._mark_ev = torch.cuda.Event()
        self._mark_ev.record(self._side)

    def wait_mark(self):
        torch.cuda.current_stream().wait_event(self._mark_ev)

    def synchronize(self):
        torch.cuda.synchronize(self.device)
