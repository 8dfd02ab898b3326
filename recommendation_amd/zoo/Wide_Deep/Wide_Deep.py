"""Wide & Deep on the HIP hot path; same class surface as the reference's Wide_Deep/Wide_Deep.py:5-150.

Two parameter groups with their own optimizers, as the reference builds them
(Wide_Deep/Wide_Deep.py:98-118): wide_net (E-wide lookups of the cross and category ids, concatenated,
one linear unit) under FTRL with l1 = 0.1, deep_net (category embeddings + dense features, relu
tower) under Adagrad.  Reproduced as written: the loss is sigmoid cross-entropy applied to y_out,
which is already a probability; wide_cross_b_* exist but never enter the graph; both minimize()
calls advance the one global_step, so it grows by two per training step."""
from collections import OrderedDict

import numpy as np
import torch

from recommendation_amd import ops
from recommendation_amd.engine import BufferCache, DenseParams, EmbedFront, EmbeddingTables, Tower, round_up
from recommendation_amd.model_base import Saver, truncated_normal
from recommendation_amd._lib import LOSS_SIGMOID_SCE_PROB, OPT_ADAGRAD, OPT_FTRL

FTRL_L1 = 0.1                # Wide_Deep/Wide_Deep.py:108


class Wide_Deep:
    def __init__(self, args, cross_num, cate_num, cont_num, cross_list, cate_list):
        self.cross_num, self.cate_num, self.cont_num = int(cross_num), int(cate_num), int(cont_num)
        self.embed_size = int(args.embed_size)
        self.hidden_layers = [int(v) for v in args.hidden_layers]
        self.regular_rate = args.regular_rate          # declared, never added to the loss
        self.dropout = float(args.dropout)
        self.learning_rate = float(args.learning_rate)
        self.decay_steps = int(args.decay_steps)
        self.decay_rate = float(args.decay_rate)
        self.cross_list = [int(v) for v in cross_list]
        self.cate_list = [int(v) for v in cate_list]
        if len(self.cross_list) != self.cross_num or len(self.cate_list) != self.cate_num:
            raise ValueError('cross_list / cate_list do not match cross_num / cate_num')
        self.init_std = float(getattr(args, 'init_std', 0.01))
        self.strict_ids = bool(getattr(args, 'strict_ids', True))
        self._built = self._ready = False
        self._bufs = BufferCache()

    # ---- graph definition ------------------------------------------------------------------
    def build(self):
        E = self.embed_size
        self.Fw = self.cross_num + self.cate_num
        w = OrderedDict()
        for i in range(self.cross_num):
            w['wide_net/wide_cross_b_%d' % i] = (E,)
        w['wide_net/wide_output_w'] = (self.Fw * E, 1)
        w['wide_net/wide_output_b'] = (1,)
        d = OrderedDict()
        self.deep_in = self.cate_num * E + self.cont_num
        d_in = self.deep_in
        for i, L in enumerate(self.hidden_layers):
            d['deep_net/deep_hidden_w_%d' % i] = (d_in, L)
            d['deep_net/deep_hidden_b_%d' % i] = (1, L)
            d_in = L
        d['deep_net/deep_output_w'] = (d_in, 1)
        d['deep_net/deep_output_b'] = (1,)
        self.wide_shapes, self.deep_shapes = w, d
        self.saver = Saver(self)
        self._built = True

    def variable_shapes(self):
        E = self.embed_size
        out = OrderedDict()
        for i, v in enumerate(self.cross_list):
            out['wide_net/wide_cross_w_%d' % i] = (v, E)
        for i, v in enumerate(self.cate_list):
            out['wide_net/wide_cate_w_%d' % i] = (v, E)
        out.update(self.wide_shapes)
        for i, v in enumerate(self.cate_list):
            out['deep_net/deep_cate_w_%d' % i] = (v, E)
        out.update(self.deep_shapes)
        return out

    # ---- variables -------------------------------------------------------------------------
    def initialize(self, session, params=None, seed=0):
        if not self._built:
            raise RuntimeError('call build() first')
        dev = session.device
        self.session = session
        E = self.embed_size
        self.wide_tables = EmbeddingTables(self.cross_list + self.cate_list, E, dev)
        self.deep_tables = EmbeddingTables(self.cate_list, E, dev)
        self.wide_front = EmbedFront(self.wide_tables, 'S', 0, False)
        self.deep_front = EmbedFront(self.deep_tables, 'S', self.cont_num, False)
        self.wide_dense = DenseParams(self.wide_shapes, dev)
        self.deep_dense = DenseParams(self.deep_shapes, dev)
        self.wide_tables.init_slots('ftrl', dense_adam=False)
        self.wide_dense.init_slots('ftrl')
        self.deep_tables.init_slots('adagrad', dense_adam=False)
        self.deep_dense.init_slots('adagrad')
        self.tower = Tower(self.deep_dense, 'deep_net/deep_hidden', len(self.hidden_layers))
        # loss + out-of-range-id counter, read back together (tf.nn.embedding_lookup raises on such an id on the
        # CPU, Wide_Deep/Wide_Deep.py:51,57,74; here the step runs and run() raises afterwards)
        self._scalars = torch.zeros(2, dtype=torch.float32, device=dev)
        self._loss_dev = self._scalars[:1]
        self._oor = self._scalars[1:].view(torch.int32)
        self.wide_front.oor_count = self.deep_front.oor_count = self._oor
        rng = np.random.Generator(np.random.PCG64(seed))
        init = OrderedDict((k, truncated_normal(rng, s, self.init_std)) for k, s in self.variable_shapes().items())
        if params:
            for k, v in params.items():
                if k not in init:
                    raise KeyError('unknown variable %s' % k)
                v = np.asarray(v, dtype=np.float32)
                if tuple(v.shape) != tuple(init[k].shape):
                    raise ValueError('%s: shape %s != %s' % (k, v.shape, init[k].shape))
                init[k] = v
        self.set_params(init)
        self.global_step = 0
        self._ready = True

    def _ensure_initialized(self, session):
        if not self._ready:
            self.initialize(session)

    def _locate(self, k):
        """(tensor view) of variable k inside the table / flat buffers."""
        f = k.rsplit('_', 1)[1]
        if k.startswith('wide_net/wide_cross_w_'):
            return self.wide_tables.table(int(f))
        if k.startswith('wide_net/wide_cate_w_'):
            return self.wide_tables.table(self.cross_num + int(f))
        if k.startswith('deep_net/deep_cate_w_'):
            return self.deep_tables.table(int(f))
        return (self.wide_dense if k.startswith('wide_net/') else self.deep_dense).p(k)

    def set_params(self, params):
        for k, v in params.items():
            t = torch.as_tensor(np.ascontiguousarray(v, dtype=np.float32))
            dst = self._locate(k)
            dst.copy_(t.reshape(dst.shape))

    def get_params(self):
        return OrderedDict((k, self._locate(k).cpu().numpy()) for k in self.variable_shapes())

    def get_grads(self):
        """Gradients of the dense variables of the last training step."""
        out = OrderedDict((k, self.wide_dense.g(k).cpu().numpy()) for k in self.wide_shapes)
        out.update((k, self.deep_dense.g(k).cpu().numpy()) for k in self.deep_shapes)
        return out

    def _state_dict(self):
        sd = OrderedDict((k, torch.as_tensor(v)) for k, v in self.get_params().items())
        sd['optimization/Variable'] = torch.tensor([self.global_step], dtype=torch.int64)
        for name, t in (('wide_tables/n', self.wide_tables.slot0), ('wide_tables/z', self.wide_tables.slot1),
                        ('wide_dense/n', self.wide_dense.slot0), ('wide_dense/z', self.wide_dense.slot1),
                        ('deep_tables/acc', self.deep_tables.slot0), ('deep_dense/acc', self.deep_dense.slot0)):
            sd[name] = t.detach().cpu()
        return sd

    def _load_state_dict(self, sd):
        names = set(self.variable_shapes())
        self.set_params({k: v.numpy() for k, v in sd.items() if k in names})
        self.global_step = int(sd['optimization/Variable'][0])
        for name, t in (('wide_tables/n', self.wide_tables.slot0), ('wide_tables/z', self.wide_tables.slot1),
                        ('wide_dense/n', self.wide_dense.slot0), ('wide_dense/z', self.wide_dense.slot1),
                        ('deep_tables/acc', self.deep_tables.slot0), ('deep_dense/acc', self.deep_dense.slot0)):
            if name in sd:
                t.copy_(sd[name])

    # ---- step ------------------------------------------------------------------------------
    def _lr(self):
        return self.learning_rate * (self.decay_rate ** (self.global_step // self.decay_steps))

    def _buffers(self, B):
        return self._bufs.lookup(B, lambda: self._make_buffers(B))

    def _make_buffers(self, B):
        dev, E, Ls = self.session.device, self.embed_size, self.hidden_layers
        z = lambda *s: torch.zeros(*s, dtype=torch.float32, device=dev)
        i32 = lambda n: torch.zeros(n, dtype=torch.int32, device=dev)
        return dict(
            zw=z(B, self.Fw * E), dzw=z(B, self.Fw * E), x0=z(B, round_up(self.deep_in, 4)),
            dx0=z(B, round_up(self.deep_in, 4)), acts=[z(B, L) for L in Ls], gbufs=[z(B, L) for L in Ls[:-1]],
            dlast=z(B, Ls[-1]), wide_out=z(B), logit=z(B), dlogit=z(B), y_out=z(B),
            wkeys=i32(B * self.Fw), wsrc=i32(B * self.Fw), dkeys=i32(B * self.cate_num),
            dsrc=i32(B * self.cate_num))

    def _drop(self, stream_id, train):
        if not train:
            return None
        return ops.make_dropout(self.dropout, self.session.seed * 1000003 + self.global_step, stream_id)

    def _forward(self, widx, idx, xc, b, train):
        wd, dd, ws = self.wide_dense, self.deep_dense, self.session.ws
        n = len(self.hidden_layers)
        self.wide_front.forward(widx, None, z_out=b['zw'])
        ops.rowdot_fwd(b['zw'], wd.p('wide_net/wide_output_w').reshape(-1), wd.p('wide_net/wide_output_b'), b['wide_out'])
        self.deep_front.forward(idx, xc, z_out=b['x0'], cont_tail=True)
        self.tower.forward(b['x0'], b['acts'], [self._drop(10 + i, train) for i in range(n)], ws)
        ops.rowdot_fwd(b['acts'][-1], dd.p('deep_net/deep_output_w').reshape(-1), dd.p('deep_net/deep_output_b'),
                       b['logit'])
        ops.copy2d(b['wide_out'].view(-1, 1), b['logit'].view(-1, 1), accumulate=True)     # wide_output + deep_output

    def train_step(self, widx, idx, xc, yy):
        """One step on device tensors (wide ids = [X_cross | X_cate] int32); returns the device scalar loss."""
        B = idx.shape[0]
        b = self._buffers(B)
        wd, dd, ws, E = self.wide_dense, self.deep_dense, self.session.ws, self.embed_size
        n = len(self.hidden_layers)
        self._forward(widx, idx, xc, b, True)
        ops.loss_head(LOSS_SIGMOID_SCE_PROB, b['logit'], yy, y_out=b['y_out'], loss_out=self._loss_dev, dx=b['dlogit'], ws=ws)
        dl = b['dlogit']
        # wide: d wide_output_w / b, dZw
        ops.dense_wgrad(b['zw'], dl.view(B, 1), wd.g('wide_net/wide_output_w'), wd.g('wide_net/wide_output_b'), ws=ws)
        ops.rowdot_bwd_x(dl, wd.p('wide_net/wide_output_w').reshape(-1), b['dzw'])
        # deep: output unit, tower, dX0
        ops.dense_wgrad(b['acts'][-1], dl.view(B, 1), dd.g('deep_net/deep_output_w'), dd.g('deep_net/deep_output_b'),
                        ws=ws)
        ops.rowdot_bwd_x(dl, dd.p('deep_net/deep_output_w').reshape(-1), b['dlast'])
        self.tower.backward(b['x0'], b['acts'], b['dlast'], b['gbufs'], [self.dropout] * n, b['dx0'], ws)
        lr = self._lr()
        ftrl = ops.make_opt(OPT_FTRL, lr=lr, l1=FTRL_L1)
        adagrad = ops.make_opt(OPT_ADAGRAD, lr=lr)
        self.wide_front.backward_tables(widx, b['dzw'], b['wkeys'], b['wsrc'], ws, ftrl, lazy=True)
        self.deep_front.backward_tables(idx, b['dx0'], b['dkeys'], b['dsrc'], ws, adagrad, lazy=True)
        self._apply_wide_dense(ftrl)
        dd.apply(adagrad)
        self.global_step += 2                              # two minimize() calls share the counter
        return self._loss_dev

    def _apply_wide_dense(self, ftrl):
        """FTRL on wide_output_w / wide_output_b only: wide_cross_b_* receive no gradient, and TF skips
        variables without one (a dense FTRL step with g = 0 would still rewrite w from z)."""
        wd = self.wide_dense
        for k in ('wide_net/wide_output_w', 'wide_net/wide_output_b'):
            p, g = wd.p(k).reshape(-1), wd.g(k).reshape(-1)
            ops.dense_update(p, g, ftrl, wd._view(wd.slot0, k).reshape(-1), wd._view(wd.slot1, k).reshape(-1))

    def predict_step(self, widx, idx, xc):
        b = self._buffers(idx.shape[0])
        self._forward(widx, idx, xc, b, False)
        ops.loss_head(LOSS_SIGMOID_SCE_PROB, b['logit'], None, y_out=b['y_out'])
        return b['y_out']

    def _upload(self, X_cross, X_cate, X_cont, y):
        dev = self.session.device
        i32 = lambda a: torch.as_tensor(np.ascontiguousarray(np.asarray(a).astype(np.int32)))
        xr, xc = i32(X_cross), i32(X_cate)
        if xr.dim() != 2 or xc.dim() != 2 or xr.shape[1] != self.cross_num or xc.shape[1] != self.cate_num:
            raise ValueError('X_cross / X_cate must be [B, cross_num] / [B, cate_num]')
        xf = torch.as_tensor(np.ascontiguousarray(np.asarray(X_cont, dtype=np.float32)))
        yy = torch.as_tensor(np.ascontiguousarray(np.asarray(y, dtype=np.float32)))
        if not (xr.shape[0] == xc.shape[0] == xf.shape[0] == yy.shape[0]) or xf.shape[1] != self.cont_num:
            raise ValueError('batch sizes / cont_num disagree')
        widx = torch.cat([xr, xc], dim=1).contiguous()
        return widx.to(dev), xc.to(dev), xf.to(dev), yy.to(dev)

    def run(self, session, X_cross, X_cate, X_cont, y, is_train=False, step=None):
        """Wide_Deep/Wide_Deep.py:130-150: training returns the loss (pre-update weights), evaluation
        returns (actual, round(y_out))."""
        self._ensure_initialized(session)
        widx, idx, xc, yy = self._upload(X_cross, X_cate, X_cont, y)
        if idx.shape[0] == 0:
            raise ValueError('empty batch')
        if is_train:
            self.train_step(widx, idx, xc, yy)
            host = self._scalars.cpu()
            self._raise_on_bad_ids(int(host[1:].view(torch.int32)[0]))
            return np.float32(host[0].item())
        pred = torch.round(self.predict_step(widx, idx, xc)).cpu().numpy()
        self._raise_on_bad_ids(int(self._oor.item()))
        return np.asarray(yy.cpu().numpy(), dtype=np.float32), pred

    def _raise_on_bad_ids(self, n):
        if n and self.strict_ids:
            self._oor.zero_()
            raise IndexError('%d lookup(s) with an id outside its table in this batch (read as zero rows, no update)' % n)

    def predict_proba(self, session, X_cross, X_cate, X_cont):
        self._ensure_initialized(session)
        widx, idx, xc, _ = self._upload(X_cross, X_cate, X_cont, np.zeros(len(X_cate), dtype=np.float32))
        return self.predict_step(widx, idx, xc).cpu().numpy()
