"""ESMM (CTR x CVR towers) on the HIP hot path; class surface of the reference's ESMM/ESMM.py:4-138."""
import torch

from recommendation_amd import ops
from recommendation_amd.engine import Tower, round_up, unit_backward
from recommendation_amd.model_base import CTRModel


class ESMM(CTRModel):
    MODEL = 'ESMM'
    OPTIMIZER = 'adagrad'                   # ESMM/ESMM.py:29

    def __init__(self, args, cate_num, cont_num, cate_list):
        super().__init__(args, cate_num, cont_num, cate_list)
        self.deep_layers = [int(v) for v in args.deep_layers]
        self.dropout = float(args.dropout)

    def _dense_shapes(self, sh):
        self.d_in = self.N + self.cont_num                                   # ESMM/ESMM.py:50-53
        for t in ('ctr', 'cvr'):
            d = self.d_in
            for i, L in enumerate(self.deep_layers):
                sh['%s_deep_part/%s_deep_w_%d' % (t, t, i)] = (d, L)
                sh['%s_deep_part/%s_deep_b_%d' % (t, t, i)] = (1, L)
                d = L
            sh['%s_deep_part/%s_output_w' % (t, t)] = (d, 1)
            sh['%s_deep_part/%s_output_b' % (t, t)] = (1,)

    def _post_init(self):
        n = len(self.deep_layers)
        self.towers = {t: Tower(self.dense, '%s_deep_part/%s_deep' % (t, t), n) for t in ('ctr', 'cvr')}

    def _alloc(self, B, bufs, dev):
        Ls = self.deep_layers
        z = lambda *s: torch.zeros(*s, dtype=torch.float32, device=dev)
        if self.mode == 'R' or self.world > 1:
            bufs['z'] = z(B, self.N * self.embed_size)
        bufs['inp'] = z(B, round_up(self.d_in, 4))
        bufs['dinp'] = z(B, round_up(self.d_in, 4))
        bufs['dz'] = z(B, self.N * self.embed_size)
        for t in ('ctr', 'cvr'):
            bufs[t + '_acts'] = [z(B, L) for L in Ls]
            bufs[t + '_gbufs'] = [z(B, L) for L in Ls[:-1]]
            bufs[t + '_glast'] = z(B, Ls[-1])
            bufs[t + '_logit'] = z(B)
            bufs[t + '_dlogit'] = z(B)

    def _forward(self, idx, xc, bufs, train):
        n, ws = len(self.deep_layers), self.session.ws
        self.front.forward(idx, xc, z_out=bufs.get('z'), first_out=bufs['inp'][:, :self.N])
        ops.copy2d(xc, bufs['inp'][:, self.N:self.N + self.cont_num])
        for k, t in enumerate(('ctr', 'cvr')):
            drops = [self._drop(self.dropout, 10 + 10 * k + i, train) for i in range(n)]
            self.towers[t].forward(bufs['inp'], bufs[t + '_acts'], drops, ws)
            ops.rowdot_fwd(bufs[t + '_acts'][-1], self.dense.p('%s_deep_part/%s_output_w' % (t, t)).reshape(-1),
                           self.dense.p('%s_deep_part/%s_output_b' % (t, t)), bufs[t + '_logit'])

    def _loss(self, bufs, yy, train):
        if train:
            ops.esmm_head(bufs['ctr_logit'], bufs['cvr_logit'], yy, y_out=bufs['y_out'], loss_out=self._loss_dev,
                          dctr=bufs['ctr_dlogit'], dcvr=bufs['cvr_dlogit'], ws=self.session.ws,
                          grad_scale=1.0 / self.world)
        else:
            ops.esmm_head(bufs['ctr_logit'], bufs['cvr_logit'], None, y_out=bufs['y_out'])

    def _backward(self, idx, xc, bufs):
        n, ws = len(self.deep_layers), self.session.ws
        B = idx.shape[0]
        for k, t in enumerate(('ctr', 'cvr')):
            dl, last = bufs[t + '_dlogit'], bufs[t + '_acts'][-1]
            wn, bn = '%s_deep_part/%s_output_w' % (t, t), '%s_deep_part/%s_output_b' % (t, t)
            fused = unit_backward(self.dense, wn, bn, last, bufs[t + '_glast'], dl, ws, mask_from=0, rate=self.dropout)
            self.towers[t].backward(bufs['inp'], bufs[t + '_acts'], bufs[t + '_glast'], bufs[t + '_gbufs'],
                                    [self.dropout] * n, bufs['dinp'], ws, accumulate_dx=(k == 1), g_last_masked=fused,
                                    side=self.session)
        ops.embed_bwd_dz(bufs['dz'], B=B, N=self.N, E=self.embed_size, g_first=bufs['dinp'][:, :self.N])
        self._tables_backward(idx, bufs['dz'], bufs)
