"""Deep Interest Network on the HIP hot path; same class surface as the reference's DIN/DIN.py:4-146:
DIN(args, user_count, item_count, cate_count, cate_list), train(sess, uij, l), eval(sess, uij),
save(sess, path), restore(sess, path).

Graph (DIN/DIN.py:17-110): item/category lookups with the cate_list double indirection, target
attention over the padded history (MLP 4H -> 80 -> 40 -> 1 with sigmoids, length mask, 1/sqrt(H),
softmax, weighted sum), inference-mode batch norm, dense H, then the FCN 2H -> 80 -> 40 -> 1 with
Dice, logits = item_b[i] + d3, sigmoid cross-entropy, plain SGD with the learning rate fed per call.
`user_emb_w` exists in the reference but is never used (DIN/DIN.py:18,48) and is not allocated here;
`logits_all` / test() (DIN/DIN.py:73-89,133-138) runs the same FCN over every item, in item tiles."""
import os
from collections import OrderedDict

import numpy as np
import torch

from recommendation_amd import ops
from recommendation_amd.engine import BufferCache, DenseParams, EmbeddingTables
from recommendation_amd._lib import LOSS_SCE_LOGIT, OPT_SGD
from recommendation_amd.model_base import GraphOwner, _capture

BN_EPS = 1e-3


class DIN(GraphOwner):
    GRAPH_SIZES = 4                        # captured (B, T) shapes kept per model

    def __init__(self, args, user_count, item_count, cate_count, cate_list):
        self.H = H = int(args.hidden_units)
        if H % 2 or (H // 2) & (H // 2 - 1) or H < 8:
            raise ValueError('hidden_units must be twice a power of two >= 4 (rows are moved 16 bytes per lane)')
        self.user_count, self.item_count, self.cate_count = int(user_count), int(item_count), int(cate_count)
        self.cate_list_host = np.asarray(cate_list).astype(np.int32)
        if self.cate_list_host.shape[0] != self.item_count:
            raise ValueError('cate_list must have item_count entries')
        sh = OrderedDict()
        for name, (i, o) in (('f1_att', (4 * H, 80)), ('f2_att', (80, 40)), ('f3_att', (40, 1))):
            sh['embedding/%s/kernel' % name] = (i, o)
            sh['embedding/%s/bias' % name] = (o,)
        for k in ('gamma', 'beta', 'moving_mean', 'moving_variance'):
            sh['embedding/batch_normalization/%s' % k] = (H,)
        sh['embedding/dense/kernel'] = (H, H)
        sh['embedding/dense/bias'] = (H,)
        for k in ('gamma', 'beta', 'moving_mean', 'moving_variance'):
            sh['fcn/b1/%s' % k] = (2 * H,)
        for name, (i, o) in (('f1', (2 * H, 80)), ('f2', (80, 40)), ('f3', (40, 1))):
            sh['fcn/%s/kernel' % name] = (i, o)
            sh['fcn/%s/bias' % name] = (o,)
        for tag in ('i', 'j', 'all'):          # one pair of Dice alphas per branch (DIN/DIN.py:55-57,63-65,83-85)
            sh['fcn/alphadice_1_%s' % tag] = (80,)
            sh['fcn/alphadice_2_%s' % tag] = (40,)
        self.dense_shapes = sh
        self.global_step = 0
        self.global_epoch_step = 0
        self.strict_ids = bool(getattr(args, 'strict_ids', True))
        # the attention scoring network as three fused launches (H = 16 or 32, the reference's default); any other
        # width, or fused_attention=False, goes layer by layer through the materialised [B*T, 4H]
        self.fused_attention = bool(getattr(args, 'fused_attention', True)) and ops.din_attn_mlp_supported(H)
        self._train_graphs = {}            # (B, T) -> captured training step
        self._lr_dev = None                # the device word a captured step reads its learning rate from
        self._ready = False
        self._bufs = BufferCache()

    # ---- variables ------------------------------------------------------------------------
    def variable_shapes(self):
        H = self.H
        out = OrderedDict()
        out['embedding/item_emb_w'] = (self.item_count, H // 2)
        out['embedding/item_b'] = (self.item_count,)
        out['embedding/cate_emb_w'] = (self.cate_count, H // 2)
        out.update(self.dense_shapes)
        return out

    def initialize(self, session, params=None, seed=0):
        """tf defaults (DIN/DIN.py:18-21,47,54,150): Glorot-uniform kernels and embeddings, zero biases,
        BN gamma=1 / beta=0 / moving stats 0,1, Dice alphas 0, item_b 0."""
        dev = session.device
        if getattr(self, 'session', None) is not None:
            self.close()                   # captured steps point at the tables, variables and buffers replaced below
        self.session = session
        self._bufs.clear()
        H = self.H
        self.tables = EmbeddingTables([self.item_count, self.cate_count], H // 2, dev)
        self.bias_table = EmbeddingTables([self.item_count], 4, dev)     # item_b lives in column 0
        self.dense = DenseParams(self.dense_shapes, dev)
        self.cate_list = torch.as_tensor(self.cate_list_host, device=dev)
        # loss + out-of-range-id counter, read back together (tf.nn.embedding_lookup raises on an id outside the
        # table on the CPU, DIN/DIN.py:26-27,40-41; here the step runs and train()/eval() raise afterwards)
        self._scalars = torch.zeros(2, dtype=torch.float32, device=dev)
        self._loss_dev = self._scalars[:1]
        self._oor = self._scalars[1:].view(torch.int32)
        self._auc_dev = torch.zeros(1, dtype=torch.float32, device=dev)
        rng = np.random.Generator(np.random.PCG64(seed))
        init = OrderedDict()
        for k, s in self.variable_shapes().items():
            if k.endswith('gamma') or k.endswith('moving_variance'):
                init[k] = np.ones(s, dtype=np.float32)
            elif (k.endswith('bias') or k.endswith('beta') or k.endswith('moving_mean') or 'alphadice' in k
                  or k.endswith('item_b')):
                init[k] = np.zeros(s, dtype=np.float32)
            else:
                lim = np.sqrt(6.0 / (s[0] + s[1]))
                init[k] = rng.uniform(-lim, lim, size=s).astype(np.float32)
        if params:
            for k, v in params.items():
                if k not in init:
                    raise KeyError('unknown variable %s' % k)
                init[k] = np.asarray(v, dtype=np.float32).reshape(init[k].shape)
        self.set_params(init)
        self._ready = True

    def _ensure(self, session):
        if not self._ready:
            self.initialize(session)

    def _graph_tables(self):
        return (self._train_graphs,)

    def _retire_graphs(self):
        super()._retire_graphs()
        self._bufs.pinned.clear()          # no graph points at the per-shape buffers any more

    def set_params(self, params):
        for k, v in params.items():
            t = torch.as_tensor(np.ascontiguousarray(v, dtype=np.float32))
            if k == 'embedding/item_emb_w':
                self.tables.table(0).copy_(t)
            elif k == 'embedding/cate_emb_w':
                self.tables.table(1).copy_(t)
            elif k == 'embedding/item_b':
                self.bias_table.weight.zero_()
                self.bias_table.weight[:, 0].copy_(t)
            else:
                self.dense.p(k).copy_(t.reshape(self.dense.shapes[k]))

    def get_params(self):
        out = OrderedDict()
        out['embedding/item_emb_w'] = self.tables.table(0).cpu().numpy()
        out['embedding/item_b'] = self.bias_table.weight[:, 0].cpu().numpy()
        out['embedding/cate_emb_w'] = self.tables.table(1).cpu().numpy()
        for k in self.dense.shapes:
            out[k] = self.dense.p(k).cpu().numpy()
        return out

    def get_grads(self):
        return OrderedDict((k, self.dense.g(k).cpu().numpy()) for k in self.dense.shapes)

    def save(self, sess, path):
        from safetensors.torch import save_file
        sd = OrderedDict((k, torch.as_tensor(v)) for k, v in self.get_params().items())
        sd['optimization/global_step'] = torch.tensor([self.global_step], dtype=torch.int64)
        sd['optimization/global_epoch_step'] = torch.tensor([self.global_epoch_step], dtype=torch.int64)
        d = os.path.dirname(os.path.abspath(path))
        os.makedirs(d, exist_ok=True)
        save_file(sd, path + '.safetensors')

    def restore(self, sess, path):
        from safetensors.torch import load_file
        self._ensure(sess)
        sd = load_file(path + '.safetensors')
        names = set(self.variable_shapes())
        self.set_params({k: v.numpy() for k, v in sd.items() if k in names})
        self.global_step = int(sd['optimization/global_step'][0])
        self.global_epoch_step = int(sd['optimization/global_epoch_step'][0])

    # ---- buffers ---------------------------------------------------------------------------
    def _buffers(self, B, T):
        return self._bufs.lookup((B, T), lambda: self._make_buffers(B, T))

    def _make_buffers(self, B, T):
        dev, H = self.session.device, self.H
        z = lambda *s: torch.zeros(*s, dtype=torch.float32, device=dev)
        n = B + B * T
        b = dict(ids=torch.zeros(n, dtype=torch.int32, device=dev),
                 pair=torch.zeros(n, 2, dtype=torch.int32, device=dev), emb=z(n, H), demb=z(n, H),
                 ib4=z(B, 4), dib4=z(B, 4), din_all=z(0 if self.fused_attention else B * T, 4 * H), d1=z(B * T, 80), d2=z(B * T, 40),
                 g1=z(B * T, 80), g2=z(B * T, 40), score=z(B * T), dscore=z(B * T), att=z(B * T),
                 pooled=z(B, H), dpooled=z(B, H), xhat1=z(B, H), hist_bn=z(B, H), dhist_bn=z(B, H),
                 din=z(B, 2 * H), ddin=z(B, 2 * H), xhat2=z(B, 2 * H), din_bn=z(B, 2 * H),
                 ddin_bn=z(B, 2 * H), f1=z(B, 80), a1=z(B, 80), f2=z(B, 40), a2=z(B, 40), df1=z(B, 80),
                 da1=z(B, 80), df2=z(B, 40), da2=z(B, 40), t80=z(B, 80), t40=z(B, 40), d3=z(B), logit=z(B),
                 dlogit=z(B), y_out=z(B),
                 keys=torch.zeros(2 * n, dtype=torch.int32, device=dev),
                 src=torch.zeros(2 * n, dtype=torch.int32, device=dev),
                 bkeys=torch.zeros(B, dtype=torch.int32, device=dev),
                 bsrc=torch.zeros(B, dtype=torch.int32, device=dev))
        b['u_emb'] = b['din'][:, :H]        # the dense layer writes the user vector straight into the FCN's input
        return b

    def _dev_i32(self, a):
        if isinstance(a, torch.Tensor) and a.is_cuda:
            return a.to(torch.int32).contiguous()
        return torch.as_tensor(np.ascontiguousarray(np.asarray(a).astype(np.int32))).to(self.session.device)

    # ---- graph pieces ----------------------------------------------------------------------
    def _lookup(self, ids, b):
        """emb[k] = [item_emb_w[id], cate_emb_w[cate_list[id]]] for all ids (DIN/DIN.py:24-42)."""
        ops.din_build_idx(ids, self.cate_list, self.item_count, b['pair'])
        ops.embed_fwd(self.tables.weight, self.tables.row_offset, b['pair'][:ids.numel()], E=self.H // 2,
                      z_out=b['emb'][:ids.numel()], oor_count=self._oor)

    def _user_vector(self, b, sl, B, T):
        """attention() + batch_normalization + dense (DIN/DIN.py:44-49); returns u_emb."""
        d, H, ws = self.dense, self.H, self.session.ws
        i_emb, h_emb = b['emb'][:B], b['emb'][B:B + B * T]
        if self.fused_attention:
            # the scoring network in one launch; [B*T, 4H] is never formed
            ops.din_attn_mlp_fwd(i_emb, h_emb, d.p('embedding/f1_att/kernel'), d.p('embedding/f1_att/bias'),
                                 d.p('embedding/f2_att/kernel'), d.p('embedding/f2_att/bias'),
                                 d.p('embedding/f3_att/kernel').reshape(-1), d.p('embedding/f3_att/bias'), b['score'],
                                 B=B, T=T, H=H, d1=b['d1'], d2=b['d2'], sl=sl)
        else:
            ops.din_concat_fwd(i_emb, h_emb, b['din_all'], B=B, T=T, H=H)
            ops.gemm(b['din_all'], d.p('embedding/f1_att/kernel'), b['d1'], bias=d.p('embedding/f1_att/bias'),
                     sigmoid=True, ws=ws)
            ops.gemm(b['d1'], d.p('embedding/f2_att/kernel'), b['d2'], bias=d.p('embedding/f2_att/bias'),
                     sigmoid=True, ws=ws)
            ops.rowdot_fwd(b['d2'], d.p('embedding/f3_att/kernel').reshape(-1), d.p('embedding/f3_att/bias'),
                           b['score'])
        ops.din_attn_pool_fwd(b['score'], sl, h_emb, b['att'], b['pooled'], B=B, T=T, H=H)
        p = 'embedding/batch_normalization/'
        ops.bn_infer(b['pooled'], b['hist_bn'], gamma=d.p(p + 'gamma'), beta=d.p(p + 'beta'),
                     mean=d.p(p + 'moving_mean'), var=d.p(p + 'moving_variance'), eps=BN_EPS, xhat=b['xhat1'])
        ops.gemm(b['hist_bn'], d.p('embedding/dense/kernel'), b['u_emb'], bias=d.p('embedding/dense/bias'), ws=ws)

    def _fcn(self, b, x_emb, tag, B, keep, d3_out=None):
        """BN(concat(u_emb, x_emb)) -> f1 -> dice -> f2 -> dice -> f3 (DIN/DIN.py:52-58); d3 goes to d3_out, else
        to b['d3']."""
        d, H, ws = self.dense, self.H, self.session.ws
        ops.copy2d(x_emb, b['din'][:, H:])           # b['u_emb'] is din[:, :H]
        p = 'fcn/b1/'
        ops.bn_infer(b['din'], b['din_bn'], gamma=d.p(p + 'gamma'), beta=d.p(p + 'beta'), mean=d.p(p + 'moving_mean'),
                     var=d.p(p + 'moving_variance'), eps=BN_EPS, xhat=b['xhat2'] if keep else None)
        ops.gemm(b['din_bn'], d.p('fcn/f1/kernel'), b['f1'], bias=d.p('fcn/f1/bias'), ws=ws)
        ops.dice_fwd(b['f1'], d.p('fcn/alphadice_1_%s' % tag), b['a1'], BN_EPS)
        ops.gemm(b['a1'], d.p('fcn/f2/kernel'), b['f2'], bias=d.p('fcn/f2/bias'), ws=ws)
        ops.dice_fwd(b['f2'], d.p('fcn/alphadice_2_%s' % tag), b['a2'], BN_EPS)
        ops.rowdot_fwd(b['a2'], d.p('fcn/f3/kernel').reshape(-1), d.p('fcn/f3/bias'), b['d3'] if d3_out is None else d3_out)

    def _item_bias(self, ids, out4):
        ops.embed_fwd(self.bias_table.weight, self.bias_table.row_offset, ids.view(-1, 1), E=4, z_out=out4)

    SMALL_ROWS = 16384                   # up to here the per-column layers' backward runs as one launch each

    def _bn_backward(self, dy, xhat, dx, prefix):
        """Backward of an inference-mode batch norm: dx, and the gradients of gamma and beta."""
        d, ws = self.dense, self.session.ws
        if dy.shape[0] <= self.SMALL_ROWS:
            ops.bn_infer_bwd(dy, xhat, dx, gamma=d.p(prefix + 'gamma'), var=d.p(prefix + 'moving_variance'), eps=BN_EPS,
                             dgamma=d.g(prefix + 'gamma'), dbeta=d.g(prefix + 'beta'))
            return
        ops.colsum(dy, d.g(prefix + 'gamma'), w=xhat, w_group=1, ws=ws)
        ops.colsum(dy, d.g(prefix + 'beta'), ws=ws)
        ops.bn_infer(dy, dx, gamma=d.p(prefix + 'gamma'), var=d.p(prefix + 'moving_variance'), eps=BN_EPS)

    def _dice_backward(self, x, alpha_name, dy, dx, scratch):
        """Backward of Dice: dx, and the gradient of its alphas."""
        d, ws = self.dense, self.session.ws
        if x.shape[0] <= self.SMALL_ROWS:
            ops.dice_bwd_sum(x, d.p(alpha_name), dy, dx, d.g(alpha_name), BN_EPS)
            return
        ops.dice_bwd(x, d.p(alpha_name), dy, dx, scratch, BN_EPS)
        ops.colsum(scratch, d.g(alpha_name), ws=ws)

    def _attention_mlp_backward_unfused(self, b, ds, i_emb, h_emb, dq, dk, B, T):
        """The scoring network's backward layer by layer over the materialised [B*T, 4H] (any H)."""
        d, H, ws = self.dense, self.H, self.session.ws
        ops.rowdot_bwd_x(ds, d.p('embedding/f3_att/kernel').reshape(-1), b['g2'])
        ops.dense_wgrad(b['d2'], ds.view(B * T, 1), d.g('embedding/f3_att/kernel'), d.g('embedding/f3_att/bias'), ws=ws)
        ops.dsigmoid_apply(b['g2'], b['d2'])
        ops.dense_wgrad(b['d1'], b['g2'], d.g('embedding/f2_att/kernel'), d.g('embedding/f2_att/bias'), ws=ws)
        ops.gemm(b['g2'], d.p('embedding/f2_att/kernel'), b['g1'], trans_b=True, ws=ws)
        ops.dsigmoid_apply(b['g1'], b['d1'])
        ops.dense_wgrad(b['din_all'], b['g1'], d.g('embedding/f1_att/kernel'), d.g('embedding/f1_att/bias'), ws=ws)
        ops.gemm(b['g1'], d.p('embedding/f1_att/kernel'), b['din_all'], trans_b=True, ws=ws)   # d(din_all), in place
        ops.din_concat_bwd(b['din_all'], i_emb, h_emb, dq, dk, B=B, T=T, H=H, acc_q=False, acc_k=True)

    # ---- public API ------------------------------------------------------------------------
    def train(self, sess, uij, l):
        """uij = (u, i, y, hist_i[B,T], sl); one SGD step with learning rate l; returns the loss."""
        self._ensure(sess)
        i = self._dev_i32(uij[1])
        hist = self._dev_i32(uij[3])
        sl = self._dev_i32(uij[4])
        y = uij[2]
        if not (isinstance(y, torch.Tensor) and y.is_cuda):
            y = torch.as_tensor(np.asarray(y, dtype=np.float32)).to(sess.device)
        self.train_step(i, y.to(torch.float32), hist, sl, l)
        host = self._scalars.cpu()
        self._raise_on_bad_ids(int(host[1:].view(torch.int32)[0]))
        return np.float32(host[0].item())

    def _raise_on_bad_ids(self, n):
        if n and self.strict_ids:
            self._oor.zero_()
            raise IndexError('%d lookup(s) with an item id outside [0, item_count) in this batch (read as zero rows)'
                             % (n // 2))      # an item id is two lookups: item table and category table

    def train_step(self, i, y, hist, sl, l):
        """The same step on device tensors (int32 ids / lengths, float32 labels); returns the device
        scalar loss without a host sync."""
        sess = self.session
        B, T = hist.shape
        if i.numel() != B or sl.numel() != B or y.numel() != B:
            raise ValueError('DIN.train: batch sizes disagree')
        b = self._buffers(B, T)
        d, H, ws = self.dense, self.H, sess.ws
        # forward
        ids = b['ids']                             # [i | hist]: one lookup for the target and the history
        ids[:B].copy_(i.reshape(-1))
        ids[B:].copy_(hist.reshape(-1))
        self._lookup(ids, b)
        self._item_bias(i, b['ib4'])
        self._user_vector(b, sl, B, T)
        self._fcn(b, b['emb'][:B], 'i', B, keep=True, d3_out=b['logit'])
        ops.copy2d(b['ib4'][:, :1], b['logit'].view(B, 1), accumulate=True)          # logits = i_b + d3
        ops.loss_head(LOSS_SCE_LOGIT, b['logit'], y, y_out=b['y_out'], loss_out=self._loss_dev, dx=b['dlogit'], ws=ws)
        # backward: FCN
        dl = b['dlogit']
        ops.rowdot_bwd_x(dl, d.p('fcn/f3/kernel').reshape(-1), b['da2'])
        ops.dense_wgrad(b['a2'], dl.view(B, 1), d.g('fcn/f3/kernel'), d.g('fcn/f3/bias'), ws=ws)
        self._dice_backward(b['f2'], 'fcn/alphadice_2_i', b['da2'], b['df2'], b['t40'])
        ops.dense_wgrad(b['a1'], b['df2'], d.g('fcn/f2/kernel'), d.g('fcn/f2/bias'), ws=ws)
        ops.gemm(b['df2'], d.p('fcn/f2/kernel'), b['da1'], trans_b=True, ws=ws)
        self._dice_backward(b['f1'], 'fcn/alphadice_1_i', b['da1'], b['df1'], b['t80'])
        ops.dense_wgrad(b['din_bn'], b['df1'], d.g('fcn/f1/kernel'), d.g('fcn/f1/bias'), ws=ws)
        ops.gemm(b['df1'], d.p('fcn/f1/kernel'), b['ddin_bn'], trans_b=True, ws=ws)
        self._bn_backward(b['ddin_bn'], b['xhat2'], b['ddin'], 'fcn/b1/')
        # backward: dense H, batch norm, attention pooling
        du = b['ddin'][:, :H]
        ops.dense_wgrad(b['hist_bn'], du, d.g('embedding/dense/kernel'), d.g('embedding/dense/bias'), ws=ws)
        ops.gemm(du, d.p('embedding/dense/kernel'), b['dhist_bn'], trans_b=True, ws=ws)
        self._bn_backward(b['dhist_bn'], b['xhat1'], b['dpooled'], 'embedding/batch_normalization/')
        i_emb, h_emb = b['emb'][:B], b['emb'][B:B + B * T]
        dq, dk = b['demb'][:B], b['demb'][B:B + B * T]
        ops.din_attn_pool_bwd(b['dpooled'], b['att'], sl, h_emb, b['dscore'], dk, B=B, T=T, H=H)
        # backward: attention MLP
        ds = b['dscore']
        if self.fused_attention:
            W1, W2 = d.p('embedding/f1_att/kernel'), d.p('embedding/f2_att/kernel')
            w3 = d.p('embedding/f3_att/kernel').reshape(-1)
            ops.din_attn_mlp_bwd(i_emb, h_emb, W1, W2, w3, b['d1'], b['d2'], ds, b['g1'], b['g2'], dq, dk, B=B, T=T, H=H,
                                 acc_k=True, sl=sl)
            ops.din_attn_mlp_wgrad(i_emb, h_emb, b['d1'], b['d2'], ds, b['g1'], b['g2'],
                                   d.g('embedding/f1_att/kernel'), d.g('embedding/f1_att/bias'),
                                   d.g('embedding/f2_att/kernel'), d.g('embedding/f2_att/bias'),
                                   d.g('embedding/f3_att/kernel').reshape(-1), d.g('embedding/f3_att/bias'),
                                   B=B, T=T, H=H, ws=ws, sl=sl)
        else:
            self._attention_mlp_backward_unfused(b, ds, i_emb, h_emb, dq, dk, B, T)
        ops.copy2d(b['ddin'][:, H:], dq, accumulate=True)                                # i_emb also feeds the FCN
        # sparse updates (SGD) on item/category tables and on item_b
        sgd = ops.make_opt(OPT_SGD, lr=float(l), lr_dev=self._lr_dev)      # a captured step reads l from a device word
        n = B + B * T
        t = self.tables
        ops.emb_sort(b['pair'][:n], t.row_offset, N=2, total_rows=t.total_rows, rmode=False, keys_out=b['keys'],
                     src_out=b['src'], ws=ws)
        ops.emb_sparse_apply(b['keys'], b['src'], 2 * n, N=2, E=H // 2, total_rows=t.total_rows, dz=b['demb'][:n],
                             opt=sgd, tables=t.weight, ws=ws)
        ops.copy2d(dl.view(B, 1), b['dib4'][:, :1])
        bt = self.bias_table
        ops.emb_sort(i.view(B, 1), bt.row_offset, N=1, total_rows=bt.total_rows, rmode=False, keys_out=b['bkeys'],
                     src_out=b['bsrc'], ws=ws)
        ops.emb_sparse_apply(b['bkeys'], b['bsrc'], B, N=1, E=4, total_rows=bt.total_rows, dz=b['dib4'], opt=sgd,
                             tables=bt.weight, ws=ws)
        # dense variables (the reference never passes global_step to minimize(): it stays 0, DIN/DIN.py:110)
        self.dense.apply(sgd)
        return self._loss_dev

    def train_step_graphed(self, i, y, hist, sl, l):
        """train_step replayed from a captured HIP graph, one per (B, T): at B = 2048 the step is some seventy
        launches on small tensors and a third of its time is the host's launch cost.  The inputs are copied into the
        graph's own buffers and the learning rate into a device word the update kernels read; same arithmetic as
        train_step.  Callers whose T changes from batch to batch (the reference loop pads to each batch's longest
        history) fall back to train_step once GRAPH_SIZES graphs exist."""
        B, T = hist.shape
        entry = self._train_graphs.get((B, T))
        if entry is None and len(self._train_graphs) >= self.GRAPH_SIZES:
            return self.train_step(i, y, hist, sl, l)
        if entry is None:
            loss = self.train_step(i, y, hist, sl, l)           # buffers and the workspace reach their final size
            dev = self.session.device
            lr_w = torch.zeros(1, dtype=torch.float32, device=dev)
            s_in = (i.clone(), y.clone(), hist.clone(), sl.clone())
            torch.cuda.synchronize(dev)
            graph = torch.cuda.CUDAGraph()
            self._lr_dev = lr_w
            try:
                _capture(graph, lambda: self.train_step(*s_in, 0.0), self.session)
            finally:
                self._lr_dev = None
            # the graph's nodes point at this shape's buffers: the entry holds them and the cache may not evict them
            self._bufs.pin((B, T))
            self._train_graphs[(B, T)] = (graph, s_in, lr_w, self._bufs[(B, T)])
            return loss
        graph, s_in, lr_w, _ = entry
        for dst, src in zip(s_in, (i, y, hist, sl)):
            dst.copy_(src)
        ops.set_step_words(None, 0, lr_w, float(l))
        graph.replay()
        return self._loss_dev

    def eval(self, sess, uij):
        """uij = (u, i, j, hist_i, sl) -> (mf_auc, p_and_n[B,2]) (DIN/DIN.py:91-97,123-131)."""
        self._ensure(sess)
        i, j = self._dev_i32(uij[1]), self._dev_i32(uij[2])
        hist, sl = self._dev_i32(uij[3]), self._dev_i32(uij[4])
        B, T = hist.shape
        b = self._buffers(B, T)
        H = self.H
        self._lookup(torch.cat([i, hist.reshape(-1)]), b)
        self._item_bias(i, b['ib4'])
        self._user_vector(b, sl, B, T)
        self._fcn(b, b['emb'][:B], 'i', B, keep=False)
        xi = torch.zeros(B, 1, dtype=torch.float32, device=sess.device)
        ops.copy2d(b['d3'].view(B, 1), xi)
        ops.copy2d(b['ib4'][:, :1], xi, accumulate=True)
        # the negative item through the same FCN weights but its own Dice alphas (DIN/DIN.py:60-66)
        ops.din_build_idx(j, self.cate_list, self.item_count, b['pair'])
        jemb = torch.zeros(B, H, dtype=torch.float32, device=sess.device)
        ops.embed_fwd(self.tables.weight, self.tables.row_offset, b['pair'][:B], E=H // 2, z_out=jemb,
                      oor_count=self._oor)
        self._item_bias(j, b['dib4'])
        self._fcn(b, jemb, 'j', B, keep=False)
        xj = torch.zeros(B, 1, dtype=torch.float32, device=sess.device)
        ops.copy2d(b['d3'].view(B, 1), xj)
        ops.copy2d(b['dib4'][:, :1], xj, accumulate=True)
        ops.mean_positive_diff(xi.view(-1), xj.view(-1), self._auc_dev)
        pn = torch.zeros(B, 2, dtype=torch.float32, device=sess.device)
        ops.loss_head(LOSS_SCE_LOGIT, xi.view(-1), None, y_out=b['y_out'])
        ops.copy2d(b['y_out'].view(B, 1), pn[:, :1])
        ops.loss_head(LOSS_SCE_LOGIT, xj.view(-1), None, y_out=b['y_out'])
        ops.copy2d(b['y_out'].view(B, 1), pn[:, 1:])
        pn_host = pn.cpu().numpy()
        self._raise_on_bad_ids(int(self._oor.item()))
        return np.float32(self._auc_dev.item()), pn_host

    ITEM_TILE_ROWS = 1 << 17               # rows of [B * items-per-tile, 2H] one pass of test() materialises

    def test(self, sess, uid, hist_i, sl, i=None):
        """logits_all (DIN/DIN.py:73-89,133-138): sigmoid(item_b[n] + FCN(BN([u_emb[b], all_emb[n]]))) for every
        user row b and every item n -> float32 [B, item_count], through the FCN's weights with the `_all` Dice alphas.

        The reference's call feeds u, hist_i and sl only, but u_emb comes out of the attention whose query is the
        target item's embedding (DIN/DIN.py:24-27,44): its session run stops on the unfed placeholder i.  Here the
        target items are the explicit argument `i`, and leaving it out raises as that run does.  The reference also
        tiles all_emb 512 times (DIN/DIN.py:79), which fixes its batch at 512; any B works here.  The [B, items, 2H]
        input is formed ITEM_TILE_ROWS rows at a time."""
        if i is None:
            raise ValueError("DIN.test: the user vector depends on the target items (placeholder 'i' of the reference "
                             "graph, which its own test() leaves unfed); pass i=")
        self._ensure(sess)
        i, hist, sl = self._dev_i32(i), self._dev_i32(hist_i), self._dev_i32(sl)
        B, T = hist.shape
        if i.numel() != B or sl.numel() != B:
            raise ValueError('DIN.test: batch sizes disagree')
        dev, H, d, ws, I = sess.device, self.H, self.dense, sess.ws, self.item_count
        b = self._buffers(B, T)
        self._lookup(torch.cat([i.reshape(-1), hist.reshape(-1)]), b)
        self._user_vector(b, sl, B, T)
        t = max(1, min(I, self.ITEM_TILE_ROWS // B))
        R = B * t
        z = lambda *s: torch.zeros(*s, dtype=torch.float32, device=dev)
        half, x, xbn, f1, a1, f2, a2 = z(R, H), z(R, 2 * H), z(R, 2 * H), z(R, 80), z(R, 80), z(R, 40), z(R, 40)
        d3, y, e, ib4, ib = z(R), z(R), z(t, H), z(t, 4), z(R, 4)
        pair = torch.zeros(t, 2, dtype=torch.int32, device=dev)
        out = z(B, I)
        p = 'fcn/b1/'
        for n0 in range(0, I, t):
            tt = min(t, I - n0)
            r = B * tt
            rows = torch.arange(r, dtype=torch.int32, device=dev)
            perm_u = torch.div(rows, tt, rounding_mode='floor').to(torch.int32)       # row -> user row b
            perm_e = (rows - perm_u * tt).to(torch.int32)                             # row -> item n0 + n
            ids = torch.arange(n0, n0 + tt, dtype=torch.int32, device=dev)
            # all_emb rows of this tile: [item_emb_w[n], cate_emb_w[cate_list[n]]] (DIN/DIN.py:77)
            ops.din_build_idx(ids, self.cate_list, I, pair)
            ops.embed_fwd(self.tables.weight, self.tables.row_offset, pair[:tt], E=H // 2, z_out=e[:tt],
                          oor_count=self._oor)
            self._item_bias(ids, ib4[:tt])
            ops.rows_gather(b['u_emb'], perm_u, r, half, F=1, E=H)
            ops.copy2d(half[:r], x[:r, :H])
            ops.rows_gather(e, perm_e, r, half, F=1, E=H)
            ops.copy2d(half[:r], x[:r, H:])
            ops.bn_infer(x[:r], xbn[:r], gamma=d.p(p + 'gamma'), beta=d.p(p + 'beta'), mean=d.p(p + 'moving_mean'),
                         var=d.p(p + 'moving_variance'), eps=BN_EPS)
            ops.gemm(xbn[:r], d.p('fcn/f1/kernel'), f1[:r], bias=d.p('fcn/f1/bias'), ws=ws)
            ops.dice_fwd(f1[:r], d.p('fcn/alphadice_1_all'), a1[:r], BN_EPS)
            ops.gemm(a1[:r], d.p('fcn/f2/kernel'), f2[:r], bias=d.p('fcn/f2/bias'), ws=ws)
            ops.dice_fwd(f2[:r], d.p('fcn/alphadice_2_all'), a2[:r], BN_EPS)
            ops.rowdot_fwd(a2[:r], d.p('fcn/f3/kernel').reshape(-1), d.p('fcn/f3/bias'), d3)
            ops.rows_gather(ib4, perm_e, r, ib, F=1, E=4)
            ops.copy2d(ib[:r, :1], d3[:r].view(r, 1), accumulate=True)                # item_b + d_layer_3_all
            ops.loss_head(LOSS_SCE_LOGIT, d3[:r], None, y_out=y[:r])
            ops.copy2d(y[:r].view(B, tt), out[:, n0:n0 + tt])
        host = out.cpu().numpy()
        self._raise_on_bad_ids(int(self._oor.item()))
        return host
