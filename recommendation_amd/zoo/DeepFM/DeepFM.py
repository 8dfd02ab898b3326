"""DeepFM on the HIP hot path; same class surface as the reference's DeepFM/DeepFM.py:5-131."""
import os

import torch

from recommendation_amd import ops
from recommendation_amd.engine import Tower, ones_padded, round_up, unit_backward
from recommendation_amd.model_base import CTRModel
from recommendation_amd._lib import LOSS_SIGMOID_LOGLOSS


class DeepFM(CTRModel):
    MODEL = 'DeepFM'
    OPTIMIZER = 'adam'                      # DeepFM/DeepFM.py:32
    LOSS_KIND = LOSS_SIGMOID_LOGLOSS        # DeepFM/DeepFM.py:88-91

    def __init__(self, args, cate_num, cont_num, cate_list):
        super().__init__(args, cate_num, cont_num, cate_list)
        self.deep_layers = [int(v) for v in args.deep_layers]
        self.dropout_fm = float(args.dropout_fm)
        self.dropout_deep = float(args.dropout_deep)
        # dZ's FM terms ride the epilogue of the product that forms the tower's dx0 (one pass over [B, N*E] less);
        # CTR_FUSE_DZ=0 keeps the separate pass (the same bits either way)
        self.fuse_dz = os.environ.get('CTR_FUSE_DZ', '1') != '0'
        # CTR_FM_Z_IN_UPDATE=1: of those terms, -g_second * Z is left to the row update - with one id per field a row of
        # Z is the table row, the same for every lookup of it, so the update subtracts (sum of g_second over the row's
        # lookups) * row once and the dx0 product does not read Z (113 MB at the benchmark shape).  Off by default:
        # measured on one box at B = 65536, the product gets 14 us shorter (166 -> 152) and the update 32 us longer
        # (78 -> 110: one more gathered row and a second running sum per lookup), the step 0.856 -> 0.879 ms.
        self.fm_z_in_update = os.environ.get('CTR_FM_Z_IN_UPDATE', '0') == '1'

    def _dense_shapes(self, sh):
        n = len(self.deep_layers)
        self.total_size = self.N * self.embed_size + self.cont_num        # DeepFM/DeepFM.py:18
        d = self.total_size
        for i, L in enumerate(self.deep_layers):
            sh['deep_part/deep_w_%d' % i] = (d, L)
            sh['deep_part/deep_b_%d' % i] = (1, L)
            d = L
        self.concat_width = self.N + self.embed_size + self.deep_layers[-1]
        sh['output_unit/output_w_%d' % (n - 1)] = (self.concat_width, 1)   # DeepFM/DeepFM.py:84 (leaked loop index)
        sh['output_unit/output_b_%d' % (n - 1)] = (1,)

    def _post_init(self):
        self.tower = Tower(self.dense, 'deep_part/deep', len(self.deep_layers))
        if os.environ.get('CTR_GEMM_F16', '1') != '0':
            # the tower's wide products as two-term fp16 splits: every operand's largest magnitude is tracked
            self.tower.enable_amax(self.session.device)

    def _alloc(self, B, bufs, dev):
        N, E, Ls = self.N, self.embed_size, self.deep_layers
        z = lambda *s: torch.zeros(*s, dtype=torch.float32, device=dev)
        D = self.total_size
        bufs['x0'] = z(B, round_up(D + 1, 4))          # one spare column of ones: the bias gradient rides the wgrad
        bufs['x0'][:, D] = 1.0
        bufs['dx0'] = z(B, round_up(D, 4))
        bufs['concat'] = z(B, self.concat_width)
        bufs['dconcat'] = z(B, self.concat_width)
        bufs['s'] = z(B, E)
        padded = [ones_padded(B, L, dev) for L in Ls[:-1]]
        bufs['acts'] = [a for a, _ in padded] + [bufs['concat'][:, N + E:]]
        bufs['ext'] = [bufs['x0'][:, :D + 1]] + [e for _, e in padded]      # layer i's input with its ones column
        bufs['gbufs'] = [z(B, L) for L in Ls[:-1]]

    def _forward(self, idx, xc, bufs, train):
        N, E, n = self.N, self.embed_size, len(self.deep_layers)
        concat = bufs['concat']
        self.tower.begin_step()
        self.front.forward(idx, xc, z_out=bufs['x0'], cont_tail=True, first_out=concat[:, :N], s_out=bufs['s'],
                           second_out=concat[:, N:N + E], drop_first=self._drop(self.dropout_fm, 1, train),
                           drop_second=self._drop(self.dropout_fm, 2, train), **self._x_amax_kw(bufs['x0']))
        self._x_amax_after(bufs['x0'], self.total_size)
        drops = [self._drop(self.dropout_deep, 10 + i, train) for i in range(n)]
        self.tower.forward(bufs['x0'], bufs['acts'], drops, self.session.ws, x_amax=self.tower.x_amax)
        ops.rowdot_fwd(concat, self.dense.p('output_unit/output_w_%d' % (n - 1)).reshape(-1),
                       self.dense.p('output_unit/output_b_%d' % (n - 1)), bufs['logit'])

    def _backward(self, idx, xc, bufs):
        N, E, n = self.N, self.embed_size, len(self.deep_layers)
        B = idx.shape[0]
        ws = self.session.ws
        concat, dconcat, dl = bufs['concat'], bufs['dconcat'], bufs['dlogit']
        wn, bn = 'output_unit/output_w_%d' % (n - 1), 'output_unit/output_b_%d' % (n - 1)
        route = self._dz_route(B)
        d1, d2 = self._drop(self.dropout_fm, 1, True), self._drop(self.dropout_fm, 2, True)
        fuse = self.fuse_dz and not route and E % 4 == 0     # a sharded front end wants dZ in its exchange layout
        # output unit backward + the last tower layer's relu/dropout mask in one pass over concat; when dZ's FM terms
        # ride the dx0 product, the keep masks of the two FM dropouts go onto their gradients here as well
        fused = unit_backward(self.dense, wn, bn, concat, dconcat, dl, ws, mask_from=N + E, rate=self.dropout_deep,
                              keep=(N, d1, E, d2) if fuse and (d1 is not None or d2 is not None) else None)
        fm = dict(g_first=dconcat[:, :N], g_second=dconcat[:, N:N + E], s=bufs['s'], z=bufs['x0'])
        # the term with the looked-up row goes to the row update where that row IS the table row
        from recommendation_amd.engine import Bags, EmbedFront
        z_late = (fuse and self.fm_z_in_update and isinstance(self.front, EmbedFront) and self.mode == 'S'
                  and not isinstance(idx, Bags) and ops.emb_sparse_apply_fm_supported(E, fm['g_second']))
        self.tower.backward(bufs['x0'], bufs['acts'], dconcat[:, N + E:], bufs['gbufs'],
                            [self.dropout_deep] * n, bufs['dx0'], ws, g_last_masked=fused, side=self.session,
                            ext=bufs['ext'], dx_fm=dict(fm, N=N, E=E, z=None if z_late else fm['z']) if fuse else None)
        if not fuse:
            ops.embed_bwd_dz(bufs['dx0'], B=B, N=N, E=E, gz=bufs['dx0'], drop_first=d1, drop_second=d2, **fm, **route)
        self._tables_backward(idx, bufs['dx0'], bufs, routed=True, fm_g2=fm['g_second'] if z_late else None)
