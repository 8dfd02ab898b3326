import sys, numpy as np, torch
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from oracle import ctr_oracle as O
from tests.helpers import no_dropout, rel_err
from tests.test_gpu_models import _build
import recommendation_amd as ra
sess = ra.Session('cuda:0', seed=0)
for variant, B in [('PNN', 65536), ('PNN_outer', 65536), ('AFM', 16384)]:
    name = variant.split('_')[0]
    cfg = no_dropout(O.default_cfg(name, embed_size=16, mode='S'))
    if variant == 'PNN_outer': cfg.inner_product = False
    if variant == 'AFM_pairs': cfg.attention, cfg.canonical_softmax = 'pairs', True
    rng = np.random.default_rng(len(variant))
    cfg.cate_list = [int(v) for v in rng.integers(40, 3000, 26)]
    cfg.cate_num, cfg.cont_num = 26, 13
    cfg.adam_mode = 'tf1'
    P = O.init_params(name, cfg, seed=12, std=0.01 if variant == 'AFM' else 0.05)
    Xc = np.stack([np.minimum(rng.pareto(1.05, B).astype(np.int64), v - 1) for v in cfg.cate_list], 1)
    Xf = rng.standard_normal((B, 13)); y = (Xf[:, 0] + 0.5 * Xf[:, 1] + 0.3 * rng.standard_normal(B) > 0).astype(np.int64)
    if name == 'AFM':
        Xf = np.abs(Xf) * 0.3
        P = {k: (v if 'attention' in k else np.abs(v)) for k, v in P.items()}
    orc = O.OracleModel(name, cfg, P, dtype=torch.float64)
    m = _build(sess, name, cfg, P)
    loss_ref, g_ref, out = orc.loss_and_grads(Xc, Xf, y)
    print(variant, B, 'y_out err', rel_err(m.predict_proba(sess, Xc, Xf), out.y_out.detach().numpy()))
    from recommendation_amd import ops
    prof = ops.KernelTimer(pool=400); ops.set_timer(prof)
    loss_gpu = float(m.train_step(*m._upload(Xc, Xf, y)).item())
    ops.set_timer(None); print('  kernels', sorted(prof.summary()))
    print('  loss', loss_gpu, float(loss_ref))
    errs = {k: rel_err(v, g_ref[k].numpy()) for k, v in m.get_grads().items()}
    for k, e in sorted(errs.items(), key=lambda kv: -kv[1])[:6]:
        print('   %-45s %.3g  |g|max %.3g' % (k, e, float(g_ref[k].abs().max())))
    m.close()
