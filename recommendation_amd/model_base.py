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
import json
import os
from collections import OrderedDict

import numpy as np
import torch

from . import ops
from .engine import OPT_KIND, Bags, DenseParams, EmbedFront, EmbeddingTables, adam_lr_t, round_up, truncated_normal
from ._lib import OPT_ADAM


class Saver:
    """Stand-in for tf.train.Saver (DeepFM/DeepFM.py:25, DeepFM/main.py:74-80,129):
    save(sess, path) / restore(sess, path).  Variables are stored by their reference names (plus optimizer
    slots and global_step) in safetensors files.

    One process: one file, `path`.safetensors.  W > 1 ranks (sharded tables): every rank writes its own shard,
    `path`.rank{r}-of-{W}.safetensors (its table rows and their optimizer slots, plus the replicated dense
    variables), and rank 0 writes `path`.manifest.json (world size, layout, which rank holds what).  restore()
    reads whatever layout the checkpoint has and re-shards it into the running model's layout: same layout ->
    each rank reads its own file; another world size or layout -> each rank assembles its rows from the shard
    files that hold them, one table at a time."""

    def __init__(self, model):
        self.model = model

    def save(self, sess, path):
        from safetensors.torch import save_file
        m = self.model
        world, rank = getattr(m, 'world', 1), getattr(m, 'rank', 0)
        d = os.path.dirname(os.path.abspath(path))
        os.makedirs(d, exist_ok=True)
        if world == 1:
            save_file(m._state_dict(), path + '.safetensors')
        else:
            save_file(m._state_dict(), '%s.rank%d-of-%d.safetensors' % (path, rank, world))
            if rank == 0:
                with open(path + '.manifest.json', 'w') as f:
                    json.dump(m._layout(), f)
        if rank == 0:
            with open(os.path.join(d, 'checkpoint'), 'w') as f:
                f.write('model_checkpoint_path: "%s"\n' % os.path.basename(path))
        return path

    def restore(self, sess, path):
        self.model._ensure_initialized(sess)
        if hasattr(self.model, '_load_checkpoint'):
            self.model._load_checkpoint(path)
        else:                                               # single-process model classes with a flat state
            from safetensors.torch import load_file
            self.model._load_state_dict(load_file(path + '.safetensors'))


class _ShardReader:
    """Reads table rows (or their optimizer slots) of one field out of a checkpoint in any layout."""

    def __init__(self, path):
        from safetensors import safe_open
        self._open = lambda p: safe_open(p, 'pt', device='cpu')
        self.path = path
        if os.path.exists(path + '.manifest.json'):
            with open(path + '.manifest.json') as f:
                self.layout = json.load(f)
            W = self.layout['world']
            self.files = ['%s.rank%d-of-%d.safetensors' % (path, r, W) for r in range(W)]
        else:
            self.layout = dict(world=1, shard='table')
            self.files = [path + '.safetensors']
        for p in self.files:
            if not os.path.exists(p):
                raise FileNotFoundError('checkpoint shard %s is missing' % p)
        self._handles = {}

    def handle(self, r):
        if r not in self._handles:
            self._handles[r] = self._open(self.files[r])
        return self._handles[r]

    def _one(self, r, f, slot, cate_list):
        """Rows of field f (weights: slot None; optimizer slot 0/1) held by old rank r, or None if it has no such slot."""
        h = self.handle(r)
        name = 'embedding/embed_cate_w_%d' % f
        if slot is None:
            return h.get_tensor(name)
        per_field = 'tables/slot%d/%d' % (slot, f)
        if per_field in h.keys():
            return h.get_tensor(per_field)
        flat = 'tables/slot%d' % slot                      # one-process files keep the slots of all fields in one array
        if flat in h.keys() and self.layout['world'] == 1:
            off = int(np.sum(cate_list[:f]))
            return h.get_slice(flat)[off:off + int(cate_list[f])]
        return None

    def field(self, f, slot, cate_list):
        """The WHOLE table of field f ([V_f, E]; weights or an optimizer slot), assembled from the shard files."""
        lay, W = self.layout, self.layout['world']
        if W == 1:
            return self._one(0, f, slot, cate_list)
        if lay['shard'] == 'table':
            owner = next(r for r, fs in enumerate(lay['fields_of']) if f in fs)
            return self._one(owner, f, slot, cate_list)
        parts = [self._one(r, f, slot, cate_list) for r in range(W)]
        if parts[0] is None:
            return None
        full = torch.empty(int(cate_list[f]), parts[0].shape[1], dtype=parts[0].dtype)
        for r, part in enumerate(parts):                 # old rank r holds rows (r - f) % W, + W, ...
            full[(r - f) % W::W] = part
        return full

    def same_layout(self, layout):
        return all(self.layout.get(k) == layout.get(k) for k in ('world', 'shard', 'fields_of'))


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
    return full if os.path.exists(full + '.safetensors') or os.path.exists(full + '.manifest.json') else None


# Lifetime of captured graphs.
# What went wrong before: a process that captured many models in turn died with SIGSEGV inside hipGraphLaunch at a
# replay of a freshly captured graph, whenever executables of earlier models had been destroyed - also when they were
# destroyed with the device idle before and after.  The native stack (tools/repro/segv_trace.c) ends four frames below
# at::cuda::CUDAGraph::replay inside libamdhip64, in a loop that walks an array of stream objects by an index with no
# bound and dereferences the element (mov 0x1a8(%rax),%rdi with %rax = array[i]): the launch of an executable with
# parallel branches picks its extra streams from a runtime-wide pool, and destroying another executable with
# branches leaves that pool shorter than the survivor needs.  A graph with a single chain of nodes needs no extra
# stream and never enters that loop.  So captured steps are single-chain graphs (the side stream is switched off
# while capturing), and their executables can be destroyed.
# A graph owns no memory here (_capture asserts that nothing is allocated while
# capturing; every buffer it points at belongs to its model or the session).  A graph executable is never destroyed
# by the garbage collector: a model that is closed or collected hands its graphs - together with the static input
# buffers, the private side stream and the events their nodes refer to - to its Session, which destroys them at a
# quiescent point only (Session.drain_graphs: device idle before and after; called before every new capture and when
# the session closes).  So the number of live graph executables is bounded by the live models (GRAPH_SIZES_PER_MODEL
# batch sizes per model and direction; other sizes run eagerly), and none outlives its session.
GRAPH_SIZES_PER_MODEL = 4
BUFFER_SIZES_PER_MODEL = 8      # per-batch-size activation buffers kept besides the captured sizes (LRU)


def _capture(graph, fn, session=None):
    """Record fn's launches into graph.  The cyclic garbage collector is held off meanwhile: a collection inside
    the capture must not run finalizers that touch the device (torch collects once on entering the capture, before
    it starts).  Graphs of models that are gone are destroyed first, while the device is idle."""
    if session is not None:
        session.drain_graphs()
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
        return out
    finally:
        if was_on:
            gc.enable()


class GraphOwner:
    """Release protocol for a model's captured graphs.  `_graph_tables()` returns the dicts whose values are graph
    entries (sequences that start with the torch.cuda.CUDAGraph and hold everything its nodes point at)."""

    session = None

    def _graph_tables(self):
        return ()

    def _retire_graphs(self):
        """Hands every captured graph to the session (no device call: safe inside the garbage collector)."""
        entries = []
        for table in self._graph_tables():
            entries.extend(table.values())
            table.clear()
        sess = getattr(self, 'session', None)
        if entries and sess is not None:
            sess.retire_graphs(entries)

    def close(self):
        """Waits for the device, then destroys this model's captured graphs.  The model stays usable (the next
        graphed step captures again)."""
        self._retire_graphs()
        sess = getattr(self, 'session', None)
        if sess is not None:
            sess.drain_graphs()

    def __del__(self):
        try:
            self._retire_graphs()
        except Exception:        # interpreter shutdown: the process is about to release everything
            pass


# The pair sort on the side stream beside the forward pass (it reads only the ids).  Off: measured on one box, the
# DeepFM step went 0.838 -> 0.835 ms and the DCN share 1.260 -> 1.271 - the GPU is busy on both streams already, the
# sort's launches take from the gather and the first product what they give back later.
EARLY_SORT = os.environ.get('CTR_EARLY_SORT', '0') == '1'
GRAPH_MAX_BATCH = 16384     # above this a step is GPU-bound and a captured graph gains nothing
# ... and so is a smaller batch through wide towers (batch x dense weights above this): a replayed step is one chain
# of launches, the eager one runs its weight gradients and the pair sort on a second stream (DCN with three 1024-wide
# layers at batch 8192: 1.25 ms eager, 1.29 ms replayed)
GRAPH_MAX_WORK = 6e9


class CTRModel(GraphOwner):
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
        self._bag_pairs = None           # (keys, src) of the multi-hot backward, grown to the largest member count
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
        if getattr(self, 'session', None) is not None:
            self.close()                                # captured graphs point at the storage re-allocated below
        self.session = session
        self._direct_ar = None
        self._bufs.clear()
        self.dense = DenseParams(self.dense_shapes, dev)
        if self.world > 1:
            from .distributed import RowShardedFront, ShardedFront, TorchComm
            if self.comm is None:
                self.comm = TorchComm()
            if self.shard == 'row':
                self.front = RowShardedFront(self.cate_list, self.embed_size, self.cont_num, self.CONT_EMBED, self.comm,
                                             dev, slack=float(getattr(self.args, 'shard_slack', 1.5)))
            else:
                self.front = ShardedFront(self.cate_list, self.embed_size, self.cont_num, self.CONT_EMBED, self.comm, dev)
            self.tables = self.front.t                      # this rank's shard
            # global field number of every local table (row sharding: every rank holds a slice of every field)
            self._own_fields = list(range(self.cate_num)) if self.shard == 'row' else list(self.front.own_fields)
        else:
            self.tables = EmbeddingTables(self.cate_list, self.embed_size, dev)
            self.front = EmbedFront(self.tables, self.mode, self.cont_num, self.CONT_EMBED)
            self._own_fields = list(range(self.cate_num))
        self.tables.init_slots(self.OPTIMIZER, dense_adam=(self.adam_mode == 'tf1'),
                               keep_grad=bool(getattr(self.args, 'table_grads', True)))
        self.dense.init_slots(self.OPTIMIZER)
        # the two words the host reads back after a step share one buffer (one copy): the loss and the
        # out-of-range-id counter of the standard lookup
        self._scalars = torch.zeros(2, dtype=torch.float32, device=dev)
        self._loss_dev = self._scalars[:1]
        self._oor = self._scalars[1:].view(torch.int32)
        if self.mode == 'S':
            # one process: the gather counts; row-sharded: the router counts on the rank whose batch holds the id;
            # table-sharded: the owner's gather counts (out_of_range_lookups sums over the ranks)
            self.front.oor_count = self._oor
        rng = np.random.Generator(np.random.PCG64(seed))
        init = OrderedDict()
        table_names = set('embedding/embed_cate_w_%d' % i for i in range(self.cate_num))
        for k, s in self.dense_shapes.items():
            init[k] = truncated_normal(rng, s, self.init_std)
        shapes = self.variable_shapes()
        if params:
            for k, v in params.items():
                if k not in shapes:
                    raise KeyError('unknown variable %s' % k)
                v = np.asarray(v, dtype=np.float32)
                if tuple(v.shape) != tuple(shapes[k]):
                    raise ValueError('%s: shape %s != %s' % (k, v.shape, shapes[k]))
                init[k] = v
        # tables that were not injected are initialised on the device, shard by shard (a Criteo-1TB-shape shard is
        # tens of GB: it never exists on the host); same distribution, its own seeded stream per (field, rank)
        gen = torch.Generator(device=dev)
        lo = self.init_std * 2.0
        for fl, f in enumerate(self._own_fields):
            if 'embedding/embed_cate_w_%d' % f in init:
                continue
            gen.manual_seed(int(seed) * 1000003 + f * 7919 + self.rank)
            t = self.tables.table(fl)
            if t.numel():
                torch.nn.init.trunc_normal_(t, 0.0, self.init_std, -lo, lo, generator=gen)
        self.set_params(init)
        self.global_step = 0
        self._ready = True
        self._post_init()

    def _post_init(self):
        pass

    def _graph_tables(self):
        return (self._graphs, self._train_graphs)

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
                        t = t[(self.rank - f) % self.world::self.world]
                    self.tables.table(f).copy_(t)
                elif f in self._own_fields:
                    self.tables.table(self._own_fields.index(f)).copy_(t)
            else:
                self.dense.p(k).copy_(t.reshape(self.dense.shapes[k]))

    def get_params(self):
        out = OrderedDict()
        for fl, f in enumerate(self._own_fields):         # sharded: only the fields (or row slices) this rank owns
            out['embedding/embed_cate_w_%d' % f] = self.tables.table(fl).cpu().numpy()
        for k in self.dense.shapes:
            out[k] = self.dense.p(k).cpu().numpy()
        return out

    def get_grads(self):
        """Dense-variable gradients of the last training step (tables: only with adam_mode 'tf1')."""
        out = OrderedDict((k, self.dense.g(k).cpu().numpy()) for k in self.dense.shapes)
        if self.tables.grad is not None:
            for fl, f in enumerate(self._own_fields):
                out['embedding/embed_cate_w_%d' % f] = self.tables.table(fl, self.tables.grad).cpu().numpy()
        return out

    def _layout(self):
        """What a reader needs to find rows in this model's checkpoint shards."""
        lay = dict(world=self.world, shard=self.shard if self.world > 1 else 'table', cate_list=self.cate_list,
                   embed_size=self.embed_size, optimizer=self.OPTIMIZER)
        if self.world > 1 and self.shard == 'table':
            lay['fields_of'] = [list(fs) for fs in self.front.fields_of]
        return lay

    def _state_dict(self):
        sd = OrderedDict((k, torch.as_tensor(v)) for k, v in self.get_params().items())
        sd['optimization/Variable'] = torch.tensor([self.global_step], dtype=torch.int64)
        for name, t in (('dense/slot0', self.dense.slot0), ('dense/slot1', self.dense.slot1)):
            if t is not None:
                sd[name] = t.detach().cpu()
        for s, t in enumerate((self.tables.slot0, self.tables.slot1)):
            if t is None:
                continue
            if self.world == 1:
                sd['tables/slot%d' % s] = t.detach().cpu().contiguous()     # m | v may share one buffer (column blocks)
            else:                                         # per field, so that another layout can pick rows out
                for fl, f in enumerate(self._own_fields):
                    sd['tables/slot%d/%d' % (s, f)] = self.tables.table(fl, t).detach().cpu().contiguous()
        return sd

    def _local_rows(self, f, full):
        """This rank's share of the whole table `full` of field f."""
        if self.world > 1 and self.shard == 'row':
            return full[(self.rank - f) % self.world::self.world]
        return full

    def _load_checkpoint(self, path):
        rd = _ShardReader(path)
        lay = rd.layout
        if 'cate_list' in lay and list(lay['cate_list']) != self.cate_list:
            raise ValueError('checkpoint was written for other table sizes')
        own = rd.handle(self.rank) if rd.same_layout(self._layout()) else None    # same layout: this rank's own file
        h0 = rd.handle(0)
        dense = {k: h0.get_tensor(k).numpy() for k in self.dense.shapes}
        self.set_params(dense)
        self.global_step = int(h0.get_tensor('optimization/Variable')[0])
        for name, t in (('dense/slot0', self.dense.slot0), ('dense/slot1', self.dense.slot1)):
            if t is not None and name in h0.keys():
                t.copy_(h0.get_tensor(name))
        for fl, f in enumerate(self._own_fields):
            for slot, buf in ((None, self.tables.weight), (0, self.tables.slot0), (1, self.tables.slot1)):
                if buf is None:
                    continue
                if own is not None:
                    rows = rd._one(self.rank, f, slot, self.cate_list)
                else:
                    full = rd.field(f, slot, self.cate_list)
                    rows = None if full is None else self._local_rows(f, full)
                if rows is not None:
                    self.tables.table(fl, buf).copy_(rows)
        if self.tables.touched is not None:
            self.tables.touched.fill_(2)      # restored m / v: every row may have optimizer state

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
        xc = self._to_device(X_cont, torch.float32, 'cont')
        yy = self._to_device(y, torch.float32, 'y')
        if isinstance(X_cate, Bags):
            idx = X_cate
        elif isinstance(X_cate, tuple) and len(X_cate) == 2:
            # multi-hot feed (not a reference input): (ids [nnz], offsets [B*F+1]) in CSR form, sum-pooled per field
            idx = Bags.from_host(X_cate[0], X_cate[1], self.cate_num, self.session.device)
        else:
            idx = self._to_device(X_cate, torch.int32, 'cate')
        if isinstance(idx, Bags):
            if self.mode != 'S' or self.world > 1:
                raise ValueError('multi-hot bags need the standard lookup (mode S) in one process')
        if len(idx.shape) != 2 or idx.shape[1] != self.cate_num:
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
        self._presorted = self._sort_early(idx, bufs)
        self._forward(idx, xc, bufs, True)
        self._loss(bufs, yy, True)
        self._backward(idx, xc, bufs)
        self._presorted = False
        self.session.side_join()                            # weight gradients issued to the side stream
        if self.world > 1:                                  # loss gradient was pre-scaled by 1/world
            if getattr(self.args, 'dense_allreduce', 'ring') == 'direct':
                if self._direct_ar is None:
                    from .distributed import DirectAllReduce
                    self._direct_ar = DirectAllReduce(self.comm, self.dense.grad.numel(), self.session.device)
                self._direct_ar(self.dense.grad, self.session.ws)
            else:
                self.comm.all_reduce_sum(self.dense.grad)
        self.dense.apply(self._optimizer())
        self.global_step += 1
        return self._loss_dev

    def graph_pays(self, B):
        """Whether a training step at batch B is launch-bound, i.e. run() replays it from a captured graph."""
        return B <= GRAPH_MAX_BATCH and B * float(self.dense.flat.numel()) <= GRAPH_MAX_WORK

    def _sort_early(self, idx, bufs):
        """The (key, src) pair sort of the table backward reads only the ids: with a side stream it is issued here,
        before the forward pass, and its chain of short launches runs beside the gather and the first tower product
        instead of between the last activation gradient and the row update.  Returns whether it was issued
        (_tables_backward then waits for its mark instead of sorting)."""
        s = self.session
        if not (EARLY_SORT and s.side_enabled and self.world == 1 and isinstance(self.front, EmbedFront)) \
                or isinstance(idx, Bags) or idx.numel() == 0:
            return False
        with s.side_stream():
            self.front.sort_pairs(idx, bufs['keys'], bufs['src'], s.ws_side)
            s.side_mark()
        return True

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
            # The captured step is one chain of launches: no second stream inside a graph.  An executable with
            # parallel branches takes streams from a pool inside the HIP runtime when it is launched, and destroying
            # another such executable shrinks that pool under it (see the note above _capture).
            side_was = self.session.side_enabled
            self.session.side_enabled = False
            try:
                _capture(graph, lambda: self.train_step(s_idx, s_xc, s_yy), self.session)
            finally:
                self.session.side_enabled = side_was
                self._indirect = None
                self.global_step = step0           # the capture ran no kernel and does not count as a step
            self._train_graphs[B] = (graph, s_idx, s_xc, s_yy, step_w, lr_w)
            return loss
        graph, s_idx, s_xc, s_yy, step_w, lr_w = entry
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
            out = _capture(graph, lambda: self.predict_step(s_idx, s_xc), self.session)
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
            # (bags change their member count from batch to batch and run eagerly)
            graphed = self.graph_pays(idx.shape[0]) and not isinstance(idx, Bags)
            step = self.train_step_graphed if graphed else self.train_step
            step(idx, xc, yy)
            host = self._scalars.cpu()                  # loss + out-of-range counter in one copy
            self.check_overflow()
            if self.strict_ids:
                self._raise_on_bad_ids(int(host[1:].view(torch.int32)[0]) if self.world == 1
                                       else self.out_of_range_lookups(reset=False))
            return np.float32(host[0].item())
        # evaluation batches are small: graph replay, not ~15 launches
        y_out = self.predict_step(idx, xc) if isinstance(idx, Bags) else self.predict_step_graphed(idx, xc)
        # tf.round is round-half-to-even; torch.round too
        pred = torch.round(y_out).cpu().numpy()
        self.check_overflow()
        if self.strict_ids:
            self._raise_on_bad_ids(self.out_of_range_lookups(reset=False))
        return np.asarray(yy.cpu().numpy(), dtype=np.float32), pred

    def check_overflow(self):
        """Row-sharded tables send fixed-capacity buckets; raises - on every rank - if one overflowed on any rank
        since the last check.  Host sync; with more than one rank a collective, so every rank must call it at the
        same point (run() and predict_proba() do; callers of train_step()/predict_step() call it themselves)."""
        if hasattr(self.front, 'check_overflow'):
            self.front.check_overflow()

    def out_of_range_lookups(self, reset=True):
        """How many lookups since the last reset carried an id outside [0, V_f) (standard lookup only; they read a
        zero row and update nothing).  Synchronises the host; with more than one rank the count is summed over the
        ranks (a collective: every rank calls it at the same point)."""
        total = self._oor
        if self.world > 1:
            total = self._oor.clone()
            self.comm.all_reduce_sum(total)
        n = int(total.item())
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
        idx, xc, _ = self._upload(X_cate, X_cont, np.zeros(len(X_cont), dtype=np.float32))
        step = self.predict_step if isinstance(idx, Bags) else self.predict_step_graphed
        out = step(idx, xc).cpu().numpy()
        self.check_overflow()
        if self.strict_ids:
            self._raise_on_bad_ids(self.out_of_range_lookups(reset=False))
        return out

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

    def _x_amax_kw(self, x0):
        """Keyword arguments for front.forward(): the tower's input word, raised by the front end where it can
        (engine.EmbedFront); the sharded front ends do not take it, the word is then raised by _x_amax_after()."""
        t = getattr(self, 'tower', None)
        if t is None or t.x_amax is None:
            return {}
        if isinstance(self.front, EmbedFront):
            return dict(amax_out=t.x_amax)
        return {}

    def _x_amax_after(self, x0, width):
        """After front.forward(): raises the tower's input word by one pass over x0[:, :width] when the front end did
        not do it itself (sharded tables)."""
        t = getattr(self, 'tower', None)
        if t is not None and t.x_amax is not None and not isinstance(self.front, EmbedFront):
            ops.absmax(x0[:, :width], t.x_amax)

    def _dz_route(self, B):
        """Extra keyword arguments for ops.embed_bwd_dz when the front end wants dZ in its exchange layout."""
        route = getattr(self.front, 'dz_route', None)
        return route(B) if route is not None else {}

    def _tables_backward(self, idx, dz, bufs, routed=False, fm_g2=None):
        kw = dict(dz_in_blocks=True) if routed and hasattr(self.front, 'dz_route') else {}
        if fm_g2 is not None:
            kw['fm_g2'] = fm_g2
        keys, src = bufs['keys'], bufs['src']
        if isinstance(idx, Bags) and idx.nnz > keys.numel():
            # one pair per bag member: a pair of buffers that only grows, shared by every batch size
            if self._bag_pairs is None or self._bag_pairs[0].numel() < idx.nnz:
                n = max(idx.nnz, 2 * (self._bag_pairs[0].numel() if self._bag_pairs else 0))
                self._bag_pairs = tuple(torch.zeros(n, dtype=torch.int32, device=self.session.device) for _ in range(2))
            keys, src = self._bag_pairs
        if getattr(self, '_presorted', False):
            self.session.wait_mark()
            kw['presorted'] = True
        self.front.backward_tables(idx, dz, keys, src, self.session.ws, self._optimizer(),
                                   lazy=(self.adam_mode == 'lazy'), **kw)
