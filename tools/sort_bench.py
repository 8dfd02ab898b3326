"""Times ctr_emb_sort alone (HIP events, median of 30 calls) on the benchmark shapes: DeepFM c2 (B = 65536, Criteo-Kaggle
tables, Zipf and uniform ids) and the DCN share (B = 8192, Criteo-1TB / 8).  CTR_SORT_FIELDS=0 times the global sort."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from recommendation_amd import ops                                    # noqa: E402
from recommendation_amd.session import Session                        # noqa: E402
from recommendation_amd.synthetic import CRITEO_1TB, CRITEO_KAGGLE, make_batches, scaled   # noqa: E402


def main():
    sess = Session()
    dev = sess.device
    out = {}
    for name, cards, B, alpha in (('deepfm_c2_zipf', CRITEO_KAGGLE, 65536, 1.05), ('deepfm_c2_uniform', CRITEO_KAGGLE, 65536, 0.0),
                                  ('dcn_share_zipf', scaled(CRITEO_1TB, 8), 8192, 1.05),
                                  ('deepfm_B16384', CRITEO_KAGGLE, 16384, 1.05), ('deepfm_B4096', CRITEO_KAGGLE, 4096, 1.05)):
        F = len(cards)
        offs = torch.as_tensor(np.concatenate([[0], np.cumsum(cards)]).astype(np.int64), device=dev)
        total = int(np.sum(cards))
        idx = make_batches(cards, 13, B, 1, alpha, 7, dev)[0][0]
        keys = torch.zeros(B * F, dtype=torch.int32, device=dev)
        src = torch.zeros(B * F, dtype=torch.int32, device=dev)
        for _ in range(5):
            ops.emb_sort(idx, offs, N=F, total_rows=total, rmode=False, keys_out=keys, src_out=src, ws=sess.ws)
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(30)]
        for a, b in evs:
            a.record()
            ops.emb_sort(idx, offs, N=F, total_rows=total, rmode=False, keys_out=keys, src_out=src, ws=sess.ws)
            b.record()
        torch.cuda.synchronize()
        us = sorted(1e3 * a.elapsed_time(b) for a, b in evs)
        k = keys.to(torch.int64)
        rows = (offs[:-1][None] + idx.to(torch.int64)).reshape(-1)
        okay = bool((k[1:] >= k[:-1]).all()) and bool(torch.equal(k, rows[src.to(torch.int64)]))
        out[name] = dict(median_us=round(us[len(us) // 2], 1), min_us=round(us[0], 1), sorted_ok=okay)
    print(json.dumps(dict(fields_path=os.environ.get('CTR_SORT_FIELDS', '1') != '0', **out)))


if __name__ == '__main__':
    main()
