"""Product-based NN on the HIP hot path; same class surface as the reference's PNN/PNN.py:5-162.

The inner-product term is reproduced as the reference computes it: tf.norm without an axis is a
Frobenius norm over the WHOLE batch (PNN/PNN.py:81), so lp is one scalar per product unit and is
broadcast over the samples.  args.inner_mode = 'per_sample' selects the per-sample norm instead."""
import torch

from recommendation_amd import ops
from recommendation_amd.engine import Tower, round_up, unit_backward
from recommendation_amd.model_base import CTRModel
from recommendation_amd._lib import LOSS_SIGMOID_SCE_PROB


class PNN(CTRModel):
    MODEL = 'PNN'
    OPTIMIZER = 'adam'                      # PNN/PNN.py:32
    CONT_EMBED = True                       # PNN/PNN.py:55-59
    LOSS_KIND = LOSS_SIGMOID_SCE_PROB       # PNN/PNN.py:117-122 (cross-entropy applied to a probability)

    def __init__(self, args, cate_num, cont_num, cate_list):
        super().__init__(args, cate_num, cont_num, cate_list)
        self.inner_product = bool(args.inner_product)
        self.product_size = int(args.product_size)
        self.hidden_layers = [int(v) for v in args.hidden_layers]
        self.dropout = float(args.dropout)
        self.inner_mode = getattr(args, 'inner_mode', 'batch_global')

    def _dense_shapes(self, sh):
        P, N, E = self.product_size, self.N, self.embed_size
        sh['product/lz_w'] = (P, N, E)
        if self.inner_product:
            sh['product/lp_inner_w'] = (P, N)
        else:
            sh['product/lp_outer_w'] = (P, E, E)
        sh['product/product_bias'] = (P,)
        d = P
        for i, L in enumerate(self.hidden_layers):
            sh['hidden/hidden_w_%d' % i] = (d, L)
            sh['hidden/hidden_b_%d' % i] = (1, L)
            d = L
        sh['hidden/output_w'] = (d, 1)
        sh['hidden/output_b'] = (1,)

    def _post_init(self):
        self.tower = Tower(self.dense, 'hidden/hidden', len(self.hidden_layers))
        self._cw = 'embedding/embed_cont_w_%d' % (self.cate_num - 1)

    def _alloc(self, B, bufs, dev):
        P, N, E, Ls = self.product_size, self.N, self.embed_size, self.hidden_layers
        z = lambda *s: torch.zeros(*s, dtype=torch.float32, device=dev)
        bufs['z'] = z(B, N * E)
        bufs['dz'] = z(B, N * E)
        bufs['s'] = z(B, E)
        bufs['prod'] = z(B, P)
        bufs['dprod'] = z(B, P)
        bufs['acts'] = [z(B, L) for L in Ls]
        bufs['gbufs'] = [z(B, L) for L in Ls[:-1]]
        bufs['glast'] = z(B, Ls[-1])
        if self.inner_product:
            bufs['usq'] = z(B, P)
            bufs['usq_sum'] = z(P)
            bufs['lp'] = z(P)
            bufs['gsum'] = z(P)
            bufs['coef'] = z(P)
            bufs['lp_b'] = z(B, P)
            bufs['coef_b'] = z(B, P)
        else:
            bufs['ss'] = z(B, E * E)
            bufs['dss'] = z(B, E * E)
            bufs['ds'] = z(B, E)

    def _forward(self, idx, xc, bufs, train):
        P, N, E, ws = self.product_size, self.N, self.embed_size, self.session.ws
        d = self.dense
        self.front.forward(idx, xc, z_out=bufs['z'], cont_w=d.p(self._cw), s_out=bufs['s'])
        prod = bufs['prod']
        drop = self._drop(self.dropout, 3, train)
        ops.gemm(bufs['z'], d.p('product/lz_w').view(P, N * E), prod, trans_b=True, ws=ws)          # lz
        if self.inner_product:
            ops.pnn_inner_fwd(bufs['z'], d.p('product/lp_inner_w'), bufs['usq'], N=N, E=E)
            if self.inner_mode == 'batch_global':
                ops.colsum(bufs['usq'], bufs['usq_sum'], ws=ws)
                if self.world > 1:                       # the norm runs over the whole (global) batch
                    self.comm.all_reduce_sum(bufs['usq_sum'])
                ops.vec_op(0, bufs['usq_sum'], None, bufs['lp'])                                  # lp[p] = ||u_p||_F
                ops.bias_act(prod, d.p('product/product_bias'), bufs['lp'], relu=True, dropout=drop)
            else:
                ops.vec_op(0, bufs['usq'].view(-1), None, bufs['lp_b'].view(-1))
                ops.copy2d(bufs['lp_b'], prod, accumulate=True)
                ops.bias_act(prod, d.p('product/product_bias'), None, relu=True, dropout=drop)
        else:
            ops.outer_fwd(bufs['s'], bufs['ss'])
            ops.gemm(bufs['ss'], d.p('product/lp_outer_w').view(P, E * E), prod, trans_b=True, accumulate=True,
                     bias=d.p('product/product_bias'), relu=True, dropout=drop, ws=ws)
        n = len(self.hidden_layers)
        drops = [self._drop(self.dropout, 10 + i, train) for i in range(n)]
        self.tower.forward(prod, bufs['acts'], drops, ws)
        ops.rowdot_fwd(bufs['acts'][-1], d.p('hidden/output_w').reshape(-1), d.p('hidden/output_b'), bufs['logit'])

    def _backward(self, idx, xc, bufs):
        P, N, E, ws = self.product_size, self.N, self.embed_size, self.session.ws
        d = self.dense
        B = idx.shape[0]
        n = len(self.hidden_layers)
        dl, last = bufs['dlogit'], bufs['acts'][-1]
        fused = unit_backward(d, 'hidden/output_w', 'hidden/output_b', last, bufs['glast'], dl, ws, mask_from=0,
                              rate=self.dropout)
        self.tower.backward(bufs['prod'], bufs['acts'], bufs['glast'], bufs['gbufs'], [self.dropout] * n,
                            bufs['dprod'], ws, g_last_masked=fused, side=self.session)
        gpre = bufs['dprod']
        ops.relu_mask_apply(gpre, bufs['prod'], 1.0 / (1.0 - self.dropout) if self.dropout else 1.0)
        ops.colsum(gpre, d.g('product/product_bias'), ws=ws)
        ops.gemm(gpre, bufs['z'], d.g('product/lz_w').view(P, N * E), trans_a=True, ws=ws)             # d lz_w
        ops.gemm(gpre, d.p('product/lz_w').view(P, N * E), bufs['dz'], ws=ws)                          # dZ from lz
        g_sum = None
        if self.inner_product:
            if self.inner_mode == 'batch_global':
                ops.colsum(gpre, bufs['gsum'], ws=ws)
                if self.world > 1:
                    self.comm.all_reduce_sum(bufs['gsum'])
                ops.vec_op(1, bufs['gsum'], bufs['lp'], bufs['coef'])          # d lp / d u = u / lp
                coef = bufs['coef']
            else:
                ops.vec_op(1, gpre.view(-1), bufs['lp_b'].view(-1), bufs['coef_b'].view(-1))
                coef = bufs['coef_b']
            ops.pnn_inner_bwd(bufs['z'], d.p('product/lp_inner_w'), coef, bufs['dz'], d.g('product/lp_inner_w'),
                              N=N, E=E, ws=ws)
        else:
            ops.gemm(gpre, d.p('product/lp_outer_w').view(P, E * E), bufs['dss'], ws=ws)
            ops.gemm(gpre, bufs['ss'], d.g('product/lp_outer_w').view(P, E * E), trans_a=True, ws=ws)
            ops.outer_bwd(bufs['dss'], bufs['s'], bufs['ds'])
            g_sum = bufs['ds']
        if g_sum is not None:
            ops.embed_bwd_dz(bufs['dz'], B=B, N=N, E=E, gz=bufs['dz'], g_sum=g_sum)
        self.front.backward_cont_w(bufs['dz'], xc, d.g(self._cw), ws)
        self._tables_backward(idx, bufs['dz'], bufs)
