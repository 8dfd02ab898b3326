"""Attentional FM on the HIP hot path; same class surface as the reference's AFM/AFM.py:4-128.

The reference takes the attention softmax over a size-1 axis (AFM/AFM.py:79), so every attention
weight is exactly 1 and attention_w / attention_b / attention_h never receive a gradient.  That
behaviour is what this class computes: the variables exist (they are checkpointed and the
optimizer sees a zero gradient) but the attention network does not influence the output.

`args.attention = 'pairs'` (not a reference flag) moves the softmax to the pair axis, which is the
network the code at AFM/AFM.py:60-85 sets up: the attention variables then train."""
import torch

from recommendation_amd import ops
from recommendation_amd.model_base import CTRModel
from recommendation_amd._lib import LOSS_IDENTITY_LOGLOSS


class AFM(CTRModel):
    MODEL = 'AFM'
    OPTIMIZER = 'adagrad'                   # AFM/AFM.py:29
    CONT_EMBED = True                       # AFM/AFM.py:52-57
    LOSS_KIND = LOSS_IDENTITY_LOGLOSS       # AFM/AFM.py:85-88: no sigmoid before log_loss

    def __init__(self, args, cate_num, cont_num, cate_list):
        super().__init__(args, cate_num, cont_num, cate_list)
        self.attention_size = int(args.attention_size)
        self.dropout = float(args.dropout)
        self.attention = getattr(args, 'attention', 'reference')
        if self.attention not in ('reference', 'pairs'):
            raise ValueError("attention must be 'reference' (softmax over the size-1 axis) or 'pairs'")

    def _dense_shapes(self, sh):
        E, A = self.embed_size, self.attention_size
        sh['attention_based_pooling/attention_w'] = (E, A)
        sh['attention_based_pooling/attention_b'] = (1, A)
        sh['attention_based_pooling/attention_h'] = (1, A)

    def _post_init(self):
        self._cw = 'embedding/embed_cont_w_%d' % (self.cate_num - 1)
        self._ones = torch.ones(self.embed_size, dtype=torch.float32, device=self.session.device)

    def _att(self, get):
        return [get('attention_based_pooling/attention_' + k) for k in ('w', 'b', 'h')]

    def _alloc(self, B, bufs, dev):
        N, E = self.N, self.embed_size
        z = lambda *s: torch.zeros(*s, dtype=torch.float32, device=dev)
        bufs['z'] = z(B, N * E)
        bufs['dz'] = z(B, N * E)
        bufs['x'] = z(B, E)
        bufs['dx'] = z(B, E)

    def _forward(self, idx, xc, bufs, train):
        N, E = self.N, self.embed_size
        self.front.forward(idx, xc, z_out=bufs['z'], cont_w=self.dense.p(self._cw))
        if self.attention == 'pairs':
            ops.afm_attn_fwd(bufs['z'], *self._att(self.dense.p), bufs['x'], N=N, E=E,
                             pair_drop=self._drop(self.dropout, 4, train))
        else:
            ops.afm_pair_fwd(bufs['z'], bufs['x'], N=N, E=E, pair_drop=self._drop(self.dropout, 4, train))
        ops.dropout_apply(bufs['x'], self._drop(self.dropout, 5, train))
        ops.rowdot_fwd(bufs['x'], self._ones, None, bufs['logit'])               # y_out = sum_e x

    def _backward(self, idx, xc, bufs):
        N, E, ws = self.N, self.embed_size, self.session.ws
        ops.rowdot_bwd_x(bufs['dlogit'], self._ones, bufs['dx'])
        ops.dropout_apply(bufs['dx'], self._drop(self.dropout, 5, True))
        if self.attention == 'pairs':
            ops.afm_attn_bwd(bufs['z'], bufs['dx'], *self._att(self.dense.p), bufs['dz'], *self._att(self.dense.g),
                             N=N, E=E, ws=ws, pair_drop=self._drop(self.dropout, 4, True))
        else:
            ops.afm_pair_bwd(bufs['z'], bufs['dx'], bufs['dz'], N=N, E=E,
                             pair_drop=self._drop(self.dropout, 4, True))
        self.front.backward_cont_w(bufs['dz'], xc, self.dense.g(self._cw), ws)
        self._tables_backward(idx, bufs['dz'], bufs)
