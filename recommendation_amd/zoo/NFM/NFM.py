"""Neural FM on the HIP hot path; same class surface as the reference's NFM/NFM.py:4-124."""
import torch

from recommendation_amd import ops
from recommendation_amd.engine import Tower, round_up, unit_backward
from recommendation_amd.model_base import CTRModel
from recommendation_amd._lib import LOSS_SIGMOID_LOGLOSS


class NFM(CTRModel):
    MODEL = 'NFM'
    OPTIMIZER = 'adagrad'                   # NFM/NFM.py:30
    LOSS_KIND = LOSS_SIGMOID_LOGLOSS        # NFM/NFM.py:81-84

    def __init__(self, args, cate_num, cont_num, cate_list):
        super().__init__(args, cate_num, cont_num, cate_list)
        self.deep_layers = [int(v) for v in args.deep_layers]
        self.dropout_bi = float(args.dropout_bi)
        self.dropout_deep = float(args.dropout_deep)

    def _dense_shapes(self, sh):
        n = len(self.deep_layers)
        d = self.embed_size + self.cont_num                                  # NFM/NFM.py:60-61
        self.d_in = d
        for i, L in enumerate(self.deep_layers):
            sh['deep/deep_w_%d' % i] = (d, L)
            sh['deep/deep_b_%d' % i] = (1, L)
            d = L
        sh['output/output_w_%d' % (n - 1)] = (d, 1)
        sh['output/output_b_%d' % (n - 1)] = (1,)

    def _post_init(self):
        self.tower = Tower(self.dense, 'deep/deep', len(self.deep_layers))

    def _alloc(self, B, bufs, dev):
        E, Ls = self.embed_size, self.deep_layers
        z = lambda *s: torch.zeros(*s, dtype=torch.float32, device=dev)
        bufs['z'] = z(B, self.N * E)
        bufs['x0'] = z(B, round_up(self.d_in, 4))
        bufs['dx0'] = z(B, round_up(self.d_in, 4))
        bufs['s'] = z(B, E)
        bufs['acts'] = [z(B, L) for L in Ls]
        bufs['gbufs'] = [z(B, L) for L in Ls[:-1]]
        bufs['glast'] = z(B, Ls[-1])

    def _forward(self, idx, xc, bufs, train):
        E, n, ws = self.embed_size, len(self.deep_layers), self.session.ws
        self.front.forward(idx, xc, z_out=bufs['z'], s_out=bufs['s'], second_out=bufs['x0'][:, :E],
                           drop_second=self._drop(self.dropout_bi, 2, train))
        ops.copy2d(xc, bufs['x0'][:, E:E + self.cont_num])
        drops = [self._drop(self.dropout_deep, 10 + i, train) for i in range(n)]
        self.tower.forward(bufs['x0'], bufs['acts'], drops, ws)
        ops.rowdot_fwd(bufs['acts'][-1], self.dense.p('output/output_w_%d' % (n - 1)).reshape(-1),
                       self.dense.p('output/output_b_%d' % (n - 1)), bufs['logit'])

    def _backward(self, idx, xc, bufs):
        E, n, ws = self.embed_size, len(self.deep_layers), self.session.ws
        B = idx.shape[0]
        dl, last = bufs['dlogit'], bufs['acts'][-1]
        wn, bn = 'output/output_w_%d' % (n - 1), 'output/output_b_%d' % (n - 1)
        fused = unit_backward(self.dense, wn, bn, last, bufs['glast'], dl, ws, mask_from=0, rate=self.dropout_deep)
        self.tower.backward(bufs['x0'], bufs['acts'], bufs['glast'], bufs['gbufs'], [self.dropout_deep] * n,
                            bufs['dx0'], ws, g_last_masked=fused, side=self.session)
        ops.embed_bwd_dz(bufs['z'], B=B, N=self.N, E=E, g_second=bufs['dx0'][:, :E], s=bufs['s'], z=bufs['z'],
                         drop_second=self._drop(self.dropout_bi, 2, True))
        self._tables_backward(idx, bufs['z'], bufs)
