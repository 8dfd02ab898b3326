"""Neural collaborative filtering on the HIP hot path; same class surface as the reference's
NCF/NCF.py:4-151.

tf.layers.dense over a one-hot id (NCF/NCF.py:38-62) is a row lookup plus a bias, followed by relu:
the four kernels live in one table with two fields (user, item) and rows [GMF | MLP] of width 2E, so
one gather serves all four.  The kernels are dense TF variables under Adam: m and v of every row
decay each step, which is the table path with adam_mode 'tf1'.  Reproduced as written: the output
unit has a relu before the sigmoid, and the loss is sigmoid cross-entropy applied to y_out."""
from collections import OrderedDict

import numpy as np
import torch

from recommendation_amd import ops
from recommendation_amd.engine import BufferCache, DenseParams, EmbedFront, EmbeddingTables, adam_lr_t
from recommendation_amd.model_base import Saver, truncated_normal
from recommendation_amd._lib import LOSS_SIGMOID_SCE_PROB, OPT_ADAM

_EMB = ('user_embed_GMF', 'item_embed_GMF', 'user_embed_MLP', 'item_embed_MLP')


class NCF:
    def __init__(self, args, user_size, item_size):
        self.user_size, self.item_size = int(user_size), int(item_size)
        self.embed_size = int(args.embed_size)
        if self.embed_size < 2 or self.embed_size & (self.embed_size - 1):
            raise ValueError('embed_size must be a power of two (rows of 2*embed_size floats are gathered)')
        self.regular_rate = args.regular_rate          # declared, never added to the loss
        self.dropout = float(args.dropout)
        self.learning_rate = float(args.learning_rate)
        self.decay_steps = int(args.decay_steps)
        self.decay_rate = float(args.decay_rate)
        self.init_std = float(getattr(args, 'init_std', 0.01))
        self._built = self._ready = False
        self._bufs = BufferCache()

    def build(self):
        E = self.embed_size
        sh = OrderedDict()
        for i, (a, b) in enumerate(((2 * E, 2 * E), (2 * E, E), (E, E // 2))):
            sh['MLP/MLP_layer%d/kernel' % (i + 1)] = (a, b)
            sh['MLP/MLP_layer%d/bias' % (i + 1)] = (b,)
        sh['concat/logits/kernel'] = (E + E // 2, 1)
        sh['concat/logits/bias'] = (1,)
        sh['_embed_bias'] = (4 * E,)                  # [user GMF | user MLP | item GMF | item MLP], the gather's column order
        self.dense_shapes = sh
        self.saver = Saver(self)
        self._built = True

    def variable_shapes(self):
        E = self.embed_size
        out = OrderedDict()
        for n, v in zip(_EMB, (self.user_size, self.item_size, self.user_size, self.item_size)):
            out['embedding/%s/kernel' % n] = (v, E)
            out['embedding/%s/bias' % n] = (E,)
        out.update((k, s) for k, s in self.dense_shapes.items() if not k.startswith('embedding/') and k != '_embed_bias')
        return out

    # column block of each embedding inside a gathered row / inside _embed_bias
    def _col(self, name):
        E = self.embed_size
        return {'user_embed_GMF': 0, 'user_embed_MLP': E, 'item_embed_GMF': 2 * E, 'item_embed_MLP': 3 * E}[name]

    def initialize(self, session, params=None, seed=0):
        if not self._built:
            raise RuntimeError('call build() first')
        dev, E = session.device, self.embed_size
        self.session = session
        self.tables = EmbeddingTables([self.user_size, self.item_size], 2 * E, dev)
        self.front = EmbedFront(self.tables, 'S', 0, False)
        self.dense = DenseParams(self.dense_shapes, dev)
        self.tables.init_slots('adam', dense_adam=True)
        self.dense.init_slots('adam')
        self._loss_dev = torch.zeros(1, dtype=torch.float32, device=dev)
        rng = np.random.Generator(np.random.PCG64(seed))
        init = OrderedDict((k, truncated_normal(rng, s, self.init_std) if k.endswith('kernel') else np.zeros(s))
                           for k, s in self.variable_shapes().items())
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

    def _locate(self, k, tab=None, flat=None):
        E = self.embed_size
        if k.startswith('embedding/'):
            name, kind = k.split('/')[1:]
            c = self._col(name)
            if kind == 'kernel':
                t = self.tables.table(0 if name.startswith('user') else 1, tab)
                return t[:, (c % (2 * E)):(c % (2 * E)) + E]
            return self.dense._view(self.dense.flat if flat is None else flat, '_embed_bias')[c:c + E]
        return self.dense._view(self.dense.flat if flat is None else flat, k)

    def set_params(self, params):
        for k, v in params.items():
            t = torch.as_tensor(np.ascontiguousarray(v, dtype=np.float32))
            dst = self._locate(k)
            dst.copy_(t.reshape(dst.shape))

    def get_params(self):
        return OrderedDict((k, self._locate(k).cpu().numpy()) for k in self.variable_shapes())

    def get_grads(self):
        return OrderedDict((k, self._locate(k, self.tables.grad, self.dense.grad).cpu().numpy())
                           for k in self.variable_shapes())

    def _state_dict(self):
        sd = OrderedDict((k, torch.as_tensor(v)) for k, v in self.get_params().items())
        sd['optim_op/Variable'] = torch.tensor([self.global_step], dtype=torch.int64)
        for name, t in (('tables/m', self.tables.slot0), ('tables/v', self.tables.slot1),
                        ('dense/m', self.dense.slot0), ('dense/v', self.dense.slot1)):
            sd[name] = t.detach().cpu()
        return sd

    def _load_state_dict(self, sd):
        names = set(self.variable_shapes())
        self.set_params({k: v.numpy() for k, v in sd.items() if k in names})
        self.global_step = int(sd['optim_op/Variable'][0])
        for name, t in (('tables/m', self.tables.slot0), ('tables/v', self.tables.slot1),
                        ('dense/m', self.dense.slot0), ('dense/v', self.dense.slot1)):
            if name in sd:
                t.copy_(sd[name])

    # ---- step ------------------------------------------------------------------------------
    def _buffers(self, B):
        return self._bufs.lookup(B, lambda: self._make_buffers(B))

    def _make_buffers(self, B):
        dev, E = self.session.device, self.embed_size
        z = lambda *s: torch.zeros(*s, dtype=torch.float32, device=dev)
        return dict(a=z(B, 4 * E), da=z(B, 4 * E), mlp_in=z(B, 2 * E), dmlp_in=z(B, 2 * E),
                             h1=z(B, 2 * E), h2=z(B, E), concat=z(B, E + E // 2), dconcat=z(B, E + E // 2),
                             g1=z(B, 2 * E), g2=z(B, E), logit=z(B), dlogit=z(B), y_out=z(B),
                             keys=torch.zeros(2 * B, dtype=torch.int32, device=dev),
                             src=torch.zeros(2 * B, dtype=torch.int32, device=dev))

    def _drop(self, stream_id, train):
        if not train:
            return None
        return ops.make_dropout(self.dropout, self.session.seed * 1000003 + self.global_step, stream_id)

    def _forward(self, idx, b, train):
        d, E, ws = self.dense, self.embed_size, self.session.ws
        a = b['a']
        self.front.forward(idx, None, z_out=a)
        ops.bias_act(a, bias1=d.p('_embed_bias'), relu=True)                         # relu(W[x] + b), all four
        ops.copy2d(a[:, E:2 * E], b['mlp_in'][:, :E])
        ops.copy2d(a[:, 3 * E:], b['mlp_in'][:, E:])
        h3 = b['concat'][:, E:]
        ops.gemm(b['mlp_in'], d.p('MLP/MLP_layer1/kernel'), b['h1'], bias=d.p('MLP/MLP_layer1/bias'), relu=True,
                 dropout=self._drop(11, train), ws=ws)
        ops.gemm(b['h1'], d.p('MLP/MLP_layer2/kernel'), b['h2'], bias=d.p('MLP/MLP_layer2/bias'), relu=True,
                 dropout=self._drop(12, train), ws=ws)
        ops.gemm(b['h2'], d.p('MLP/MLP_layer3/kernel'), h3, bias=d.p('MLP/MLP_layer3/bias'), relu=True,
                 dropout=self._drop(13, train), ws=ws)
        ops.mul2d(a[:, :E], a[:, 2 * E:3 * E], b['concat'][:, :E])                    # GMF
        ops.rowdot_fwd(b['concat'], d.p('concat/logits/kernel').reshape(-1), d.p('concat/logits/bias'), b['logit'])
        ops.bias_act(b['logit'].view(-1, 1), relu=True)                              # dense(1) has a relu too

    def train_step(self, idx, yy):
        """idx: int32 [B, 2] = (user, item) on the device; returns the device scalar loss."""
        B = idx.shape[0]
        b = self._buffers(B)
        d, E, ws = self.dense, self.embed_size, self.session.ws
        keep = 1.0 / (1.0 - self.dropout) if self.dropout else 1.0
        self._forward(idx, b, True)
        ops.loss_head(LOSS_SIGMOID_SCE_PROB, b['logit'], yy, y_out=b['y_out'], loss_out=self._loss_dev, dx=b['dlogit'],
                      ws=ws)
        dl = b['dlogit'].view(B, 1)
        ops.relu_mask_apply(dl, b['logit'].view(B, 1), 1.0)
        ops.dense_wgrad(b['concat'], dl, d.g('concat/logits/kernel'), d.g('concat/logits/bias'), ws=ws)
        ops.rowdot_bwd_x(dl.view(-1), d.p('concat/logits/kernel').reshape(-1), b['dconcat'])
        # MLP backward (relu + dropout masks come from the stored activations)
        g3 = b['dconcat'][:, E:]
        ops.relu_mask_apply(g3, b['concat'][:, E:], keep)
        ops.dense_wgrad(b['h2'], g3, d.g('MLP/MLP_layer3/kernel'), d.g('MLP/MLP_layer3/bias'), ws=ws)
        ops.gemm(g3, d.p('MLP/MLP_layer3/kernel'), b['g2'], trans_b=True, mask_src=b['h2'], mask_scale=keep, ws=ws)
        ops.dense_wgrad(b['h1'], b['g2'], d.g('MLP/MLP_layer2/kernel'), d.g('MLP/MLP_layer2/bias'), ws=ws)
        ops.gemm(b['g2'], d.p('MLP/MLP_layer2/kernel'), b['g1'], trans_b=True, mask_src=b['h1'], mask_scale=keep, ws=ws)
        ops.dense_wgrad(b['mlp_in'], b['g1'], d.g('MLP/MLP_layer1/kernel'), d.g('MLP/MLP_layer1/bias'), ws=ws)
        ops.gemm(b['g1'], d.p('MLP/MLP_layer1/kernel'), b['dmlp_in'], trans_b=True, ws=ws)
        # gradient of the four activated embeddings, in the gather's column order
        a, da, dg = b['a'], b['da'], b['dconcat'][:, :E]
        ops.mul2d(dg, a[:, 2 * E:3 * E], da[:, :E])                                   # d user GMF = g * item GMF
        ops.mul2d(dg, a[:, :E], da[:, 2 * E:3 * E])                                   # d item GMF = g * user GMF
        ops.copy2d(b['dmlp_in'][:, :E], da[:, E:2 * E])
        ops.copy2d(b['dmlp_in'][:, E:], da[:, 3 * E:])
        ops.relu_mask_apply(da, a, 1.0)
        ops.colsum(da, d.g('_embed_bias'), ws=ws)
        lr = adam_lr_t(self.learning_rate * (self.decay_rate ** (self.global_step // self.decay_steps)),
                       self.global_step + 1)
        adam = ops.make_opt(OPT_ADAM, lr=lr)
        self.front.backward_tables(idx, da, b['keys'], b['src'], ws, adam, lazy=False)
        d.apply(adam)
        self.global_step += 1
        return self._loss_dev

    def predict_step(self, idx):
        b = self._buffers(idx.shape[0])
        self._forward(idx, b, False)
        ops.loss_head(LOSS_SIGMOID_SCE_PROB, b['logit'], None, y_out=b['y_out'])
        return b['y_out']

    def _upload(self, X, y):
        X = np.asarray(X)
        if X.ndim != 2 or X.shape[1] != 2:
            raise ValueError('X must be [B, 2] = (userid, itemid)')
        idx = torch.as_tensor(np.ascontiguousarray(X.astype(np.int32))).to(self.session.device)
        yy = torch.as_tensor(np.ascontiguousarray(np.asarray(y, dtype=np.float32))).to(self.session.device)
        if yy.shape[0] != idx.shape[0]:
            raise ValueError('batch sizes disagree')
        return idx, yy

    def run(self, session, X, y, is_train=False, step=None):
        """NCF/NCF.py:134-151: X[:, 0] = user ids, X[:, 1] = item ids."""
        self._ensure_initialized(session)
        idx, yy = self._upload(X, y)
        if idx.shape[0] == 0:
            raise ValueError('empty batch')
        if is_train:
            return np.float32(self.train_step(idx, yy).item())
        y_out = self.predict_step(idx)
        return np.asarray(yy.cpu().numpy(), dtype=np.float32), torch.round(y_out).cpu().numpy()

    def predict_proba(self, session, X):
        self._ensure_initialized(session)
        idx, _ = self._upload(X, np.zeros(len(X), dtype=np.float32))
        return self.predict_step(idx).cpu().numpy()
