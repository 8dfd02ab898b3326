"""Common host logic of the six CSV-family model classes (DeepFM, DCN, NFM, PNN, AFM, ESMM).

Mirrors the reference's model-class surface (DeepFM/DeepFM.py:5-131): __init__(args, cate_num,
cont_num, cate_list), build(), run(session, X_cate, X_cont, y, is_train, step), .saver.
Subclasses describe their dense variables and implement _forward/_backward with ops.py kernels.

Extra, optional attributes read from `args` (absent in the reference's argparse namespaces):
  mode        'R' (default): the embedding block exactly as the reference writes it;
              'S': standard per-field lookup (the only form that exists at Criteo scale)
  adam_mode   'tf1' : TF-1.x AdamOptimizer on tables (m, v decay for every row; dense pass)
              'lazy': rows without a gradient are untouched.  Default: 'tf1' in R-mode, 'lazy' in S-mode
  init_std    stddev of the truncated-normal initialiser (0.01)
"""
from __future__ import annotations

import gc
import os
from collections import OrderedDict

import numpy as np
import torch

from . import ops
from .engine import OPT_KIND, DenseParams, EmbedFront, EmbeddingTables, adam_lr_t, round_up, truncated_normal
from ._lib import OPT_ADAM


class Saver:
    """Stand-in for tf.train.Saver (DeepFM/DeepFM.py:25, DeepFM/main.py:74-80,129):
    save(sess, path) / restore(sess, path).  Variables are stored by their reference names
    (plus optimizer slots and global_step) in one safetensors file at `path`."""

    def __init__(self, model):
        self.model = model

    def save(self, sess, path):
        from safetensors.torch import save_file
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        save_file(self.model._state_dict(), path + '.safetensors')
        with open(os.path.join(os.path.dirname(os.path.abspath(path)), 'checkpoint'), 'w') as f:
            f.write('model_checkpoint_path: "%s"\n' % os.path.basename(path))
        return path

    def restore(self, sess, path):
        from safetensors.torch import load_file
        self.model._ensure_initialized(sess)
        self.model._load_state_dict(load_file(path + '.safetensors'))


def get_checkpoint_state(directory):
    """tf.train.get_checkpoint_state counterpart: latest checkpoint path recorded in
    `directory`/checkpoint, or None."""
    p = os.path.join(directory, 'checkpoint')
    if not os.path.exists(p):
        return None
    with open(p) as f:
        line = f.readline()
    name = line.split('"')[1]
    full = os.path.join(directory, name)
    return full if os.path.exists(full + '.safetensors') else None


# Captured graphs are kept until the process ends.  What is known: a graph owns no memory here (_capture asserts that
# nothing is allocated while capturing; every buffer it points at belongs to its model or the session), and a
# torch-free HIP program with the same fork/join capture survives capture -> destroy -> launch in every order
# (tools/repro/graph_lifetime.cpp).  Still, with the graphs of collected models destroyed, 1 run in 11 of the GPU test
# suite died inside hipGraphLaunch at the first replay of a newly captured graph; with the graphs kept, none did.
# So graph executables are not destroyed, and their number is bounded instead: a model captures at most
# GRAPH_SIZES_PER_MODEL batch sizes per direction and runs other sizes eagerly.
_LIVE_GRAPHS = []
GRAPH_SIZES_PER_MODEL = 4
BUFFER_SIZES_PER_MODEL = 8      # per-batch-size activation buffers kept besides the captured sizes (LRU)


def _capture(graph, fn):
    """Record fn's launches into graph.  The cyclic garbage collector is held off meanwhile: were it to run inside
    the capture and free an older model's graph, the runtime call that destroys it would be illegal there and the
    process would abort (torch collects once on entering the capture, before it starts)."""
    was_on = gc.isenabled()
    gc.disable()
    try:
        with torch.cuda.graph(graph):
            # A graph must own no memory: a tensor allocated while capturing would live in the graph's private pool
            # and could be handed out again once the graph is gone.  Every buffer a step touches is allocated by
            # the eager step that precedes the capture; this check keeps it that way.  (Counted from inside the
            # context: entering it allocates torch's own two generator-state words.)
            allocs0 = torch.cuda.memory_stats()['allocation.all.allocated']
            out = fn()
            grew = torch.cuda.memory_stats()['allocation.all.allocated'] - allocs0
        if grew:
            raise RuntimeError('%d device allocation(s) happened inside a graph capture' % grew)
        if os.environ.get('CTR_GRAPH_KEEP', '1') != '0':     # '0': experiments on graph destruction only
            _LIVE_GRAPHS.append(graph)
        return out
    finally:
        if was_on:
            gc.enable()


GRAPH_MAX_BATCH = 16384     # above this a step is GPU-bound and a captured graph gains nothing


class CTRModel:
    MODEL = None
    OPTIMIZER = 'adam'
    CONT_EMBED = False        # AFM / PNN append value-scaled continuous rows to Z
    NEEDS_DENSE_Z = True      # model consumes Z itself (not only its reductions)

    def __init__(self, args, cate_num, cont_num, cate_list):
        self.args = args
        self.cate_num = int(cate_num)
        self.cont_num = int(cont_num)
        self.embed_size = int(args.embed_size)
        self.regular_rate = getattr(args, 'regular_rate', 0.0)     # declared, never added to the loss
        self.learning_rate = float(args.learning_rate)
        self.decay_steps = int(args.decay_steps)
        self.decay_rate = float(args.decay_rate)
        self.cate_list = [int(v) for v in cate_list]
        self.mode = getattr(args, 'mode', 'R')
        if self.mode not in ('R', 'S'):
            raise ValueError("mode must be 'R' or 'S'")
        self.adam_mode = getattr(args, 'adam_mode', 'tf1' if self.mode == 'R' else 'lazy')
        self.init_std = float(getattr(args, 'init_std', 0.01))
        # multi-GPU: world_size/rank (+ optional comm object); tables are then sharded table-wise
        self.world = int(getattr(args, 'world_size', 1))
        self.rank = int(getattr(args, 'rank', 0))
        self.comm = getattr(args, 'comm', None)
        # 'table': a field's whole table on one rank; 'row': row r of every field on rank r % world
        self.shard = getattr(args, 'shard', 'table')
        if self.shard not in ('table', 'row'):
            raise ValueError("shard must be 'table' or 'row'")
        if self.world > 1 and self.mode != 'S':
            raise ValueError("multi-GPU runs use mode 'S' (the dense R-mode block exists only at parity sizes)")
        if self.mode == 'R' and any(v < 2 for v in self.cate_list):
            # the block reads and trains row 1 of every table (DeepFM/DeepFM.py:49-52); the reference's loader
            # guarantees V = n_unique + 1 >= 2 (DeepFM/main.py:40) and TF cannot look up row 1 of a 1-row table
            raise ValueError("mode 'R' needs at least 2 rows in every table (cate_list = %s)" % (self.cate_list,))
        # True: an id outside [0, V_f) raises IndexError from run() (what tf.nn.embedding_lookup does on the CPU);
        # False (default for these six models, whose one-hot block is silent in the reference): zero row, counted
        self.strict_ids = bool(getattr(args, 'strict_ids', False))
        self.global_step = 0
        self._built = False
        self._ready = False
        self._bufs = OrderedDict()       # per-batch-size buffers, least recently used first
        self._graphs = {}
        self._train_graphs = {}
        self._stage = OrderedDict()      # pinned staging buffers of _upload, by (feed, shape), LRU
        self._indirect = None            # (step word, lr word) on the device while a training graph is captured

    # ---- graph definition -----------------------------------------------------------------
    def build(self):
        E = self.embed_size
        self.n_cat_rows = int(np.sum(self.cate_list)) if self.mode == 'R' else self.cate_num
        self.N = self.n_cat_rows + (self.cont_num if self.CONT_EMBED else 0)
        shapes = OrderedDict()
        if self.CONT_EMBED:
            shapes['embedding/embed_cont_w_%d' % (self.cate_num - 1)] = (self.cont_num, E)
        self._dense_shapes(shapes)
        self.dense_shapes = shapes
        self.saver = Saver(self)
        self._built = True

    def _dense_shapes(self, shapes):
        raise NotImplementedError

    def variable_shapes(self):
        """name -> shape of every trainable variable, reference naming."""
        out = OrderedDict(('embedding/embed_cate_w_%d' % i, (v, self.embed_size))
                          for i, v in enumerate(self.cate_list))
        out.update(self.dense_shapes)
        return out

    # ---- variables ------------------------------------------------------------------------
    def initialize(self, session, params=None, seed=0):
        """Allocate device state and initialise variables (the counterpart of
        sess.run(tf.global_variables_initializer())).  `params` injects name -> array."""
        if not self._built:
            raise RuntimeError('call build() first')
        dev = session.device
        self.session = session
        self._graphs, self._train_graphs = {}, {}      # captured graphs point at the storage allocated below
        self.dense = DenseParams(self.dense_shapes, dev)
        if self.world > 1:
            from .distributed import RowShardedFront, ShardedFront, TorchComm
            if self.comm is None:
                self.comm = TorchComm()
            cls = RowShardedFront if self.shard == 'row' else ShardedFront
            self.front = cls(self.cate_list, self.embed_size, self.cont_num, self.CONT_EMBED, self.comm, dev)
            self.tables = self.front.t                      # this rank's shard
            self._own = (0, self.cate_num) if self.shard == 'row' else (self.front.a, self.front.b)
        else:
            self.tables = EmbeddingTables(self.cate_list, self.embed_size, dev)
            self.front = EmbedFront(self.tables, self.mode, self.cont_num, self.CONT_EMBED)
            self._own = (0, self.cate_num)
        self.tables.init_slots(self.OPTIMIZER, dense_adam=(self.adam_mode == 'tf1'))
        self.dense.init_slots(self.OPTIMIZER)
        # the two words the host reads back after a step share one buffer (one copy): the loss and the
        # out-of-range-id counter of the standard lookup
        self._scalars = torch.zeros(2, dtype=torch.float32, device=dev)
        self._loss_dev = self._scalars[:1]
        self._oor = self._scalars[1:].view(torch.int32)
        if self.mode == 'S' and self.world == 1:
            self.front.oor_count = self._oor
        rng = np.random.Generator(np.random.PCG64(seed))
        init = OrderedDict()
        for k, s in self.variable_shapes().items():
            init[k] = truncated_normal(rng, s, self.init_std)
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
        self._post_init()

    def _post_init(self):
        pass

    def _ensure_initialized(self, session):
        if not self._ready:
            self.initialize(session)

    def set_params(self, params):
        for k, v in params.items():
            t = torch.as_tensor(np.ascontiguousarray(v, dtype=np.float32))
            if k.startswith('embedding/embed_cate_w_'):
                f = int(k.rsplit('_', 1)[1])
                if self.world > 1 and self.shard == 'row':
                    if t.shape[0] == self.cate_list[f]:     # a full table: keep this rank's rows
                        t = t[self.rank::self.world]
                    self.tables.table(f).copy_(t)
                elif self._own[0] <= f < self._own[1]:
                    self.tables.table(f - self._own[0]).copy_(t)
            else:
                self.dense.p(k).copy_(t.reshape(self.dense.shapes[k]))

    def get_params(self):
        out = OrderedDict()
        for f in range(*self._own):                       # sharded: only the fields this rank owns
            out['embedding/embed_cate_w_%d' % f] = self.tables.table(f - self._own[0]).cpu().numpy()
        for k in self.dense.shapes:
            out[k] = self.dense.p(k).cpu().numpy()
        return out

    def get_grads(self):
        """Dense-variable gradients of the last training step (tables: only with adam_mode 'tf1')."""
        out = OrderedDict((k, self.dense.g(k).cpu().numpy()) for k in self.dense.shapes)
        if self.tables.grad is not None:
            for f in range(*self._own):
                out['embedding/embed_cate_w_%d' % f] = self.tables.table(f - self._own[0], self.tables.grad).cpu().numpy()
        return out

    def _state_dict(self):
        sd = OrderedDict((k, torch.as_tensor(v)) for k, v in self.get_params().items())
        sd['optimization/Variable'] = torch.tensor([self.global_step], dtype=torch.int64)
        for name, t in (('tables/slot0', self.tables.slot0), ('tables/slot1', self.tables.slot1),
                        ('dense/slot0', self.dense.slot0), ('dense/slot1', self.dense.slot1)):
            if t is not None:
                sd[name] = t.detach().cpu()
        return sd

    def _load_state_dict(self, sd):
        names = set(self.variable_shapes())
        self.set_params({k: v.numpy() for k, v in sd.items() if k in names})
        self.global_step = int(sd['optimization/Variable'][0])
        for name, t in (('tables/slot0', self.tables.slot0), ('tables/slot1', self.tables.slot1),
                        ('dense/slot0', self.dense.slot0), ('dense/slot1', self.dense.slot1)):
            if t is not None and name in sd:
                t.copy_(sd[name])

    # ---- step ------------------------------------------------------------------------------
    def _lr(self):
        """tf.train.exponential_decay(staircase=True) at the pre-update global_step."""
        return self.learning_rate * (self.decay_rate ** (self.global_step // self.decay_steps))

    def _step_lr(self):
        lr = self._lr()
        if self.OPTIMIZER == 'adam':
            lr = adam_lr_t(lr, self.global_step + 1)
        return lr

    def _optimizer(self):
        if self._indirect is not None:       # captured graph: the kernels read the step size from device memory
            return ops.make_opt(OPT_KIND[self.OPTIMIZER], lr=0.0, lr_dev=self._indirect[1])
        return ops.make_opt(OPT_KIND[self.OPTIMIZER], lr=self._step_lr())

    def _to_device(self, a, dtype, slot):
        """Host array -> device tensor of `dtype`: one converting pass into a pinned staging buffer (kept per feed
        and shape), then one copy the host waits for, so the buffer is free again on return."""
        if isinstance(a, torch.Tensor) and a.is_cuda:
            return a.to(dtype).contiguous()
        src = a if isinstance(a, torch.Tensor) else torch.as_tensor(np.ascontiguousarray(np.asarray(a)))
        key = (slot, tuple(src.shape))
        stage = self._stage.get(key)
        if stage is None:
            stage = self._stage[key] = torch.empty(src.shape, dtype=dtype, pin_memory=True)
            while len(self._stage) > 3 * BUFFER_SIZES_PER_MODEL:      # three feeds per batch size
                self._stage.popitem(last=False)
        else:
            self._stage.move_to_end(key)
        stage.copy_(src)
        return stage.to(self.session.device)

    def _upload(self, X_cate, X_cont, y):
        """Feed conversion of the reference placeholders: int64 -> int32, float64 -> float32."""
        idx = self._to_device(X_cate, torch.int32, 'cate')
        xc = self._to_device(X_cont, torch.float32, 'cont')
        yy = self._to_device(y, torch.float32, 'y')
        if idx.dim() != 2 or idx.shape[1] != self.cate_num:
            raise ValueError('X_cate must be [B, %d]' % self.cate_num)
        if xc.dim() != 2 or xc.shape[1] != self.cont_num or xc.shape[0] != idx.shape[0]:
            raise ValueError('X_cont must be [B, %d]' % self.cont_num)
        if yy.dim() != 1 or yy.shape[0] != idx.shape[0]:
            raise ValueError('y must be [B]')
        return idx, xc, yy

    def _buffers(self, B):
        if B in self._bufs:
            self._bufs.move_to_end(B)
        else:
            # a caller with ever-changing batch sizes must not accumulate buffers: drop the least recently used
            # size that no captured graph points at
            free = [b for b in self._bufs if b not in self._graphs and b not in self._train_graphs]
            while len(free) >= BUFFER_SIZES_PER_MODEL:
                del self._bufs[free.pop(0)]
            dev = self.session.device
            bufs = dict(logit=torch.zeros(B, dtype=torch.float32, device=dev),
                        y_out=torch.zeros(B, dtype=torch.float32, device=dev),
                        dlogit=torch.zeros(B, dtype=torch.float32, device=dev),
                        keys=torch.zeros(B * self.cate_num, dtype=torch.int32, device=dev),
                        src=torch.zeros(B * self.cate_num, dtype=torch.int32, device=dev))
            self._alloc(B, bufs, dev)
            self._bufs[B] = bufs
        return self._bufs[B]

    def _alloc(self, B, bufs, dev):
        raise NotImplementedError

    def _drop(self, rate, stream_id, train):
        if not train:
            return None
        if self._indirect is not None:       # captured graph: the step count is added on the device
            return ops.make_dropout(rate, self.session.seed * 1000003, stream_id, seed_add=self._indirect[0])
        return ops.make_dropout(rate, self.session.seed * 1000003 + self.global_step, stream_id)

    def train_step(self, idx, xc, yy):
        """One fwd + bwd + optimizer step on device tensors; returns the device scalar loss
        (computed with the pre-update weights, as the reference's fetch does)."""
        B = idx.shape[0]
        bufs = self._buffers(B)
        self._forward(idx, xc, bufs, True)
        self._loss(bufs, yy, True)
        self._backward(idx, xc, bufs)
        self.session.side_join()                            # weight gradients issued to the side stream
        if self.world > 1:
            self.comm.all_reduce_sum(self.dense.grad)       # loss gradient was pre-scaled by 1/world
        self.dense.apply(self._optimizer())
        self.global_step += 1
        return self._loss_dev

    def train_step_graphed(self, idx, xc, yy):
        """train_step replayed from a captured HIP graph (one per batch size).  What changes from step to step -
        the dropout streams' step count and the (decayed, bias-corrected) step size - lives in two device words
        the kernels read (ctr_dropout_t.seed_add, ctr_opt_t.lr_dev); the host writes them before each replay.
        Same arithmetic and same dropout masks as train_step.  Single-process only."""
        B = idx.shape[0]
        entry = self._train_graphs.get(B)
        if self.world > 1 or (entry is None and len(self._train_graphs) >= GRAPH_SIZES_PER_MODEL):
            return self.train_step(idx, xc, yy)
        if entry is None:
            # first step at this batch size: an ordinary step (buffers and the workspace reach their final size),
            # then the capture - which records launches without running them - for every later step
            loss = self.train_step(idx, xc, yy)
            dev = self.session.device
            step_w = torch.zeros(1, dtype=torch.int64, device=dev)
            lr_w = torch.zeros(1, dtype=torch.float32, device=dev)
            s_idx, s_xc, s_yy = idx.clone(), xc.clone(), yy.clone()
            torch.cuda.synchronize(dev)
            graph = torch.cuda.CUDAGraph()
            step0 = self.global_step
            self._indirect = (step_w, lr_w)
            try:
                with self.session.private_side() as side_objs:
                    _capture(graph, lambda: self.train_step(s_idx, s_xc, s_yy))
            finally:
                self._indirect = None
                self.global_step = step0           # the capture ran no kernel and does not count as a step
            self._train_graphs[B] = (graph, s_idx, s_xc, s_yy, step_w, lr_w, side_objs)
            return loss
        graph, s_idx, s_xc, s_yy, step_w, lr_w, _ = entry
        s_idx.copy_(idx)
        s_xc.copy_(xc)
        s_yy.copy_(yy)
        ops.set_step_words(step_w, self.global_step, lr_w, self._step_lr())
        graph.replay()
        self.global_step += 1
        return self._loss_dev

    def predict_step(self, idx, xc):
        bufs = self._buffers(idx.shape[0])
        self._forward(idx, xc, bufs, False)
        self._loss(bufs, None, False)
        return bufs['y_out']

    def predict_step_graphed(self, idx, xc):
        """predict_step replayed from a captured HIP graph (one per batch size): the forward's launches are
        recorded once on a capture stream and replayed as one graph launch - at serving batch sizes the step is
        launch-bound and this is what removes the per-kernel host cost.  Inputs are copied into the graph's
        static buffers; the result tensor is the graph's static output (read it before the next call).
        Single-process only (a collective cannot be captured here); weights are read in place, so training
        steps between calls are seen by the next replay."""
        B = idx.shape[0]
        entry = self._graphs.get(B)
        if self.world > 1 or (entry is None and len(self._graphs) >= GRAPH_SIZES_PER_MODEL):
            return self.predict_step(idx, xc)
        if entry is None:
            s_idx, s_xc = idx.clone(), xc.clone()
            side = torch.cuda.Stream(self.session.device)
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):             # warm-up off the default stream: buffers and workspace grow here
                self.predict_step(s_idx, s_xc)
                self.predict_step(s_idx, s_xc)
            torch.cuda.current_stream().wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            out = _capture(graph, lambda: self.predict_step(s_idx, s_xc))
            entry = (graph, s_idx, s_xc, out)
            self._graphs[B] = entry
        graph, s_idx, s_xc, out = entry
        s_idx.copy_(idx)
        s_xc.copy_(xc)
        graph.replay()
        return out

    def run(self, session, X_cate, X_cont, y, is_train=False, step=None):
        self._ensure_initialized(session)
        idx, xc, yy = self._upload(X_cate, X_cont, y)
        if idx.shape[0] == 0:
            raise ValueError('empty batch')
        if is_train:
            # the reference's loop feeds small mini-batches (128 by default): launch-bound, so replay a graph
            step = self.train_step_graphed if idx.shape[0] <= GRAPH_MAX_BATCH else self.train_step
            step(idx, xc, yy)
            host = self._scalars.cpu()                  # loss + out-of-range counter in one copy
            if self.strict_ids:
                self._raise_on_bad_ids(int(host[1:].view(torch.int32)[0]))
            return np.float32(host[0].item())
        y_out = self.predict_step_graphed(idx, xc)      # evaluation batches are small: graph replay, not ~15 launches
        # tf.round is round-half-to-even; torch.round too
        pred = torch.round(y_out).cpu().numpy()
        if self.strict_ids:
            self._raise_on_bad_ids(int(self._oor.item()))
        return np.asarray(yy.cpu().numpy(), dtype=np.float32), pred

    def out_of_range_lookups(self, reset=True):
        """How many lookups since the last reset carried an id outside [0, V_f) (standard lookup only; they read a
        zero row and update nothing).  Synchronises the host."""
        n = int(self._oor.item())
        if reset and n:
            self._oor.zero_()
        return n

    def _raise_on_bad_ids(self, n):
        if n:
            self._oor.zero_()
            raise IndexError('%d lookup(s) with an id outside [0, cate_list[f]) in this batch (the rows were read as '
                             'zeros and received no update)' % n)

    def predict_proba(self, session, X_cate, X_cont):
        self._ensure_initialized(session)
        idx, xc, _ = self._upload(X_cate, X_cont, np.zeros(len(X_cate), dtype=np.float32))
        return self.predict_step_graphed(idx, xc).cpu().numpy()

    # hooks
    LOSS_KIND = 0

    def _loss(self, bufs, yy, train):
        if train:
            ops.loss_head(self.LOSS_KIND, bufs['logit'], yy, y_out=bufs['y_out'], loss_out=self._loss_dev,
                          dx=bufs['dlogit'], ws=self.session.ws, grad_scale=1.0 / self.world)
        else:
            ops.loss_head(self.LOSS_KIND, bufs['logit'], None, y_out=bufs['y_out'])

    def _forward(self, idx, xc, bufs, train):
        raise NotImplementedError

    def _backward(self, idx, xc, bufs):
        raise NotImplementedError

    def _dz_route(self, B):
        """Extra keyword arguments for ops.embed_bwd_dz when the front end wants dZ in its exchange layout."""
        route = getattr(self.front, 'dz_route', None)
        return route(B) if route is not None else {}

    def _tables_backward(self, idx, dz, bufs, routed=False):
        kw = dict(dz_in_blocks=True) if routed and hasattr(self.front, 'dz_route') else {}
        self.front.backward_tables(idx, dz, bufs['keys'], bufs['src'], self.session.ws, self._optimizer(),
                                   lazy=(self.adam_mode == 'lazy'), **kw)
