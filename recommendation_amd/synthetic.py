"""Synthetic Criteo-shape inputs, generated on the device (torch is used as plumbing only).

Cardinalities are the public Criteo-Kaggle / Criteo-1TB per-field counts (external public
figures, not from the reference); ids follow a bounded Zipf law per field, with ranks spread
over the table by a multiplicative hash so hot rows are not neighbours in memory.
"""
from __future__ import annotations

import numpy as np
import torch

CRITEO_KAGGLE = [1460, 583, 10131227, 2202608, 305, 24, 12517, 633, 3, 93145, 5683, 8351593, 3194, 27, 14992,
                 5461306, 10, 5652, 2173, 4, 7046547, 18, 15, 286181, 105, 142572]
CRITEO_1TB = [227605432, 39060, 17295, 7424, 20265, 3, 7122, 1543, 63, 130229467, 3067956, 405282, 10, 2209,
              11938, 155, 4, 976, 14, 292775614, 40790948, 187188510, 590152, 12973, 108, 36]
N_DENSE = 13


def scaled(cards, divisor):
    """Cardinalities divided by `divisor` (tables that would drop below 3 rows keep 3)."""
    return [max(3, int(v // divisor)) for v in cards]


def zipf_ids(cards, B, alpha, gen, device):
    """[B, F] int32 ids: rank r in [1, V] drawn with P(r) ~ r^-alpha (continuous inverse-CDF
    approximation), then id = (r * 2654435761) mod V.  alpha == 0 gives uniform ids."""
    cols = []
    for V in cards:
        u = torch.rand(B, generator=gen, device=device, dtype=torch.float64)
        if alpha == 0.0:
            r = torch.floor(u * V)
        elif abs(alpha - 1.0) < 1e-9:
            r = torch.floor(torch.exp(u * np.log(V)))-1
        else:
            a = 1.0 - alpha
            r = torch.floor(((V ** a - 1.0) * u + 1.0) ** (1.0 / a)) - 1
        r = r.clamp_(0, V - 1).to(torch.int64)
        cols.append(((r * 2654435761) % V).to(torch.int32))
    return torch.stack(cols, dim=1).contiguous()


def make_batches(cards, n_dense, B, n_batches, alpha, seed, device):
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    out = []
    for _ in range(n_batches):
        idx = zipf_ids(cards, B, alpha, gen, device)
        xc = torch.randn(B, n_dense, generator=gen, device=device, dtype=torch.float32)
        y = (torch.rand(B, generator=gen, device=device) < 0.25).to(torch.float32)
        out.append((idx, xc, y))
    return out
