"""Deep & Cross on the HIP hot path; same class surface as the reference's DCN/DCN.py:5-132."""
import os

import torch

from recommendation_amd import ops
from recommendation_amd.engine import Tower, ones_padded, round_up, unit_backward
from recommendation_amd.model_base import CTRModel
from recommendation_amd._lib import LOSS_SIGMOID_LOGLOSS


class DCN(CTRModel):
    MODEL = 'DCN'
    OPTIMIZER = 'adam'                      # DCN/DCN.py:33
    LOSS_KIND = LOSS_SIGMOID_LOGLOSS        # DCN/DCN.py:89-92

    def __init__(self, args, cate_num, cont_num, cate_list):
        super().__init__(args, cate_num, cont_num, cate_list)
        self.cross_layer_num = int(args.cross_layer_num)
        self.deep_layers = [int(v) for v in args.deep_layers]
        self.dropout = float(args.dropout)

    def _dense_shapes(self, sh):
        n = len(self.deep_layers)
        self.total_size = D = self.N * self.embed_size + self.cont_num     # DCN/DCN.py:19
        for l in range(self.cross_layer_num):
            sh['cross_part/cross_w_%d' % l] = (D, 1)
        for l in range(self.cross_layer_num):
            sh['cross_part/cross_b_%d' % l] = (D, 1)
        d = D
        for i, L in enumerate(self.deep_layers):
            sh['deep_part/deep_w_%d' % i] = (d, L)
            sh['deep_part/deep_b_%d' % i] = (1, L)
            d = L
        self.stack_width = D + self.deep_layers[-1]
        sh['stack_layer/stack_w_%d' % (n - 1)] = (self.stack_width, 1)     # DCN/DCN.py:85 (leaked loop index)
        sh['stack_layer/stack_b_%d' % (n - 1)] = (1,)

    def _post_init(self):
        self.tower = Tower(self.dense, 'deep_part/deep', len(self.deep_layers))
        if os.environ.get('CTR_GEMM_F16', '1') != '0':
            # the tower's wide products as two-term fp16 splits: every operand's largest magnitude is tracked
            self.tower.enable_amax(self.session.device)
        self._wn = ['cross_part/cross_w_%d' % l for l in range(self.cross_layer_num)]
        self._bn = ['cross_part/cross_b_%d' % l for l in range(self.cross_layer_num)]

    def _alloc(self, B, bufs, dev):
        D, Ls = self.total_size, self.deep_layers
        z = lambda *s: torch.zeros(*s, dtype=torch.float32, device=dev)
        bufs['x0'] = z(B, round_up(D + 1, 4))          # one spare column of ones: the bias gradient rides the wgrad
        bufs['x0'][:, D] = 1.0
        bufs['dx0'] = z(B, round_up(D, 4))
        # rows 16 bytes apart in multiples: the cross kernels move x_L and its gradient sixteen bytes per lane
        bufs['stack'] = z(B, round_up(self.stack_width, 4))[:, :self.stack_width]
        bufs['dstack'] = z(B, round_up(self.stack_width, 4))[:, :self.stack_width]
        bufs['aux'] = z(B, 2 * self.cross_layer_num)
        padded = [ones_padded(B, L, dev) for L in Ls[:-1]]
        bufs['acts'] = [a for a, _ in padded] + [bufs['stack'][:, D:]]
        bufs['ext'] = [bufs['x0'][:, :D + 1]] + [e for _, e in padded]      # layer i's input with its ones column
        bufs['gbufs'] = [z(B, L) for L in Ls[:-1]]

    def _forward(self, idx, xc, bufs, train):
        D, n, ws = self.total_size, len(self.deep_layers), self.session.ws
        self.tower.begin_step()
        self.front.forward(idx, xc, z_out=bufs['x0'], cont_tail=True, **self._x_amax_kw(bufs['x0']))
        self._x_amax_after(bufs['x0'], D)
        ops.cross_fwd(bufs['x0'], self.dense.stacked(self._wn), self.dense.stacked(self._bn),
                      bufs['stack'][:, :D], bufs['aux'], ws)
        drops = [self._drop(self.dropout, 10 + i, train) for i in range(n)]
        self.tower.forward(bufs['x0'], bufs['acts'], drops, ws, x_amax=self.tower.x_amax)
        ops.rowdot_fwd(bufs['stack'], self.dense.p('stack_layer/stack_w_%d' % (n - 1)).reshape(-1),
                       self.dense.p('stack_layer/stack_b_%d' % (n - 1)), bufs['logit'])

    def _backward(self, idx, xc, bufs):
        D, n, ws = self.total_size, len(self.deep_layers), self.session.ws
        B = idx.shape[0]
        stack, dstack, dl = bufs['stack'], bufs['dstack'], bufs['dlogit']
        wn, bn = 'stack_layer/stack_w_%d' % (n - 1), 'stack_layer/stack_b_%d' % (n - 1)
        # the unit's backward also raises the record of the gradient it hands to the tower (no pass of its own)
        g_amax = self.tower.g_last_amax()
        fused = unit_backward(self.dense, wn, bn, stack, dstack, dl, ws, mask_from=D, rate=self.dropout, g_amax=g_amax)
        self.tower.backward(bufs['x0'], bufs['acts'], dstack[:, D:], bufs['gbufs'], [self.dropout] * n,
                            bufs['dx0'], ws, g_last_masked=fused, side=self.session, ext=bufs['ext'],
                            g_last_has_amax=fused and g_amax is not None)
        ops.cross_bwd(bufs['x0'], self.dense.stacked(self._wn), self.dense.stacked(self._bn), bufs['aux'],
                      dstack[:, :D], bufs['dx0'], self.dense.stacked(self._wn, self.dense.grad),
                      self.dense.stacked(self._bn, self.dense.grad), ws, accumulate=True)
        self._tables_backward(idx, bufs['dx0'], bufs)
