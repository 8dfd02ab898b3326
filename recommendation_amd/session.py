"""Session: the opaque handle the model classes receive in run()/saver calls.

Stands where tf.Session stands in the reference's main.py files (DeepFM/main.py:69): it is
only ever passed back into model.run(...) and model.saver.save/restore(...).  Here it carries
the GPU device, the RNG seed of the dropout streams and the scratch workspace.
"""
from __future__ import annotations

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

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        torch.cuda.synchronize(self.device)
        return False

    def synchronize(self):
        torch.cuda.synchronize(self.device)
