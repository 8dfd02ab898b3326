"""Data loading and the epoch loop shared by the CSV-family entry points (the counterpart of the
reference's per-model main.py, e.g. DeepFM/main.py:13-43 and :82-129).

Kept from the reference: the 7-tuple of load_dataset, the StandardScaler fit over train+test, the
raw category values used as ids with cate_list = n_unique + 1, file-order mini-batches whose loop
`while end < total` never runs when total <= batch_size, the two stdout line formats, and the
"Accuracy" expression that is 1.0 by construction (len of a boolean array / len).
Changed on purpose: argv is honoured (the reference parses args=[]), and the real log-loss and AUC
are printed on an extra line.
"""
from __future__ import annotations

import os
import time

import numpy as np
import torch

CATE_COLS = ['user_type', 'item_type']
DROP_COLS = ['userid', 'itemid', 'label', 'type']


def load_dataset(file_path):
    """-> X_train_cate, X_train_cont, y_train, X_test_cate, X_test_cont, y_test, cate_list"""
    import pandas as pd
    from sklearn.preprocessing import StandardScaler
    frames = []
    for split in ('train', 'test'):
        df = pd.read_csv(os.path.join(file_path, split + '.csv'))
        df['type'] = split
        frames.append(df)
    both = pd.concat(frames, axis=0, ignore_index=True)
    cont_cols = [c for c in both.columns if c not in DROP_COLS and c not in CATE_COLS]
    cont = StandardScaler().fit_transform(both[cont_cols]) if cont_cols else np.zeros((len(both), 0))
    cate = both[CATE_COLS].values
    is_train = (both['type'] == 'train').values
    label = both['label'].values
    cate_list = [int(np.unique(cate[:, c]).shape[0] + 1) for c in range(cate.shape[1])]
    return (cate[is_train], cont[is_train], label[is_train],
            cate[~is_train], cont[~is_train], label[~is_train], cate_list)


def batches(total, batch_size):
    """(start, end) pairs exactly as the reference's loop produces them."""
    out, end, k = [], batch_size, 0
    while end < total:
        start = k * batch_size
        end = start + batch_size if start + batch_size < total else total
        out.append((start, end))
        k += 1
    return out


def auc_score(y, p):
    """Area under the ROC curve with average ranks for ties."""
    y = np.asarray(y).astype(np.float64)
    p = np.asarray(p).astype(np.float64)
    n_pos, n_neg = y.sum(), (1 - y).sum()
    if n_pos == 0 or n_neg == 0:
        return float('nan')
    order = np.argsort(p, kind='mergesort')
    ranks = np.empty(len(p), dtype=np.float64)
    sp = p[order]
    i = 0
    while i < len(p):
        j = i
        while j + 1 < len(p) and sp[j + 1] == sp[i]:
            j += 1
        ranks[order[i:j + 1]] = 0.5 * (i + j) + 1.0
        i = j + 1
    return float((ranks[y == 1].sum() - n_pos * (n_pos + 1) / 2.0) / (n_pos * n_neg))


def log_loss(y, p, eps=1e-7):
    y = np.asarray(y, dtype=np.float64)
    p = np.asarray(p, dtype=np.float64)
    return float(np.mean(-y * np.log(p + eps) - (1 - y) * np.log(1 - p + eps)))


def fit(model_cls, args, data=None, session=None, log=print):
    """Build the model, restore a checkpoint if there is one, run args.epoch epochs of train + test,
    save after every epoch.  Returns the model."""
    import recommendation_amd as ra
    from recommendation_amd.model_base import get_checkpoint_state
    if data is None:
        data = load_dataset(args.input_dir)
    Xtr_c, Xtr_f, ytr, Xte_c, Xte_f, yte, cate_list = data
    sess = session or ra.Session(getattr(args, 'device', 'cuda:0'), seed=getattr(args, 'seed', 0))
    model = model_cls(args, Xtr_c.shape[1], Xtr_f.shape[1], cate_list)
    model.build()
    ckpt_dir = os.path.join(args.input_dir, args.model_name)
    ckpt = get_checkpoint_state(ckpt_dir)
    if ckpt:
        log('Loading model parameters from %s' % ckpt)
        model.saver.restore(sess, ckpt)
    else:
        log('Creating model with inital parameters')
        model.initialize(sess, seed=getattr(args, 'seed', 0))
    step = 0
    for epoch in range(args.epoch):
        t0 = time.time()
        losses = []
        # pinned, double-buffered feed; the loss of every step stays on the device until the epoch ends
        from recommendation_amd.feeder import Feeder
        model._ensure_initialized(sess)
        dev_losses = []
        for idx, xc, yy in Feeder(sess, Xtr_c, Xtr_f, ytr, batches(Xtr_c.shape[0], args.batch_size)):
            dev_losses.append(model.train_step(idx, xc, yy).clone())
            step += 1
        losses = [float(v) for v in torch.cat(dev_losses).cpu()] if dev_losses else []
        took = time.strftime('%H: %M: %S', time.gmtime(time.time() - t0))
        log('Epoch %d training: loss = %.4f, took: %s' % (epoch + 1, np.mean(losses) if losses else float('nan'), took))
        # evaluation stays on the device: one forward per test batch, then AUC (radix sort + average
        # ranks) and log loss in one call; only three scalars come back to the host
        dev_y, dev_p = [], []
        for idx, xc, yy in Feeder(sess, Xte_c, Xte_f, yte, batches(Xte_c.shape[0], args.batch_size)):
            dev_p.append(model.predict_step(idx, xc).clone())
            dev_y.append(yy.clone())
        took = time.strftime('%H: %M: %S', time.gmtime(time.time() - t0))
        if dev_y:
            actual, prob = torch.cat(dev_y), torch.cat(dev_p)
            predict = torch.round(prob)                                           # tf.round: half to even
            accuracy = float(len(actual == predict)) / float(len(actual))        # == 1.0, as in the reference
            log('Epoch %d testing: Accuracy = %.4f, took: %s' % (epoch + 1, accuracy, took))
            from recommendation_amd import ops
            auc, ll = ops.auc_logloss(prob.clamp(0.0, 1.0).contiguous(), actual.contiguous(), sess.ws)
            log('Epoch %d testing: logloss = %.6f, AUC = %.6f, accuracy(true) = %.4f'
                % (epoch + 1, float(ll), float(auc), float((actual == predict).float().mean())))
        model.saver.save(sess, os.path.join(ckpt_dir, args.model_name))
    return model


def common_flags(parser, model_name):
    """Flag names and defaults of the reference's main.py files (DeepFM/main.py:48-59)."""
    parser.add_argument('--input_dir', help='The directory of input', type=str, default='')
    parser.add_argument('--model_name', help='The directory of model', type=str, default=model_name + '.ckpt')
    parser.add_argument('--embed_size', help='size for embedding user and item', type=int, default=8)
    parser.add_argument('--batch_size', help='size of mini-batch', type=int, default=128)
    parser.add_argument('--epoch', help='number of epochs', type=int, default=10)
    parser.add_argument('--regular_rate', help='regular rate', type=float, default=0.1)
    parser.add_argument('--learning_rate', help='learning rate', type=float, default=0.001)
    parser.add_argument('--decay_steps', help='decay steps', type=int, default=10000)
    parser.add_argument('--decay_rate', help='decay rate', type=float, default=0.99)
    # additions of this build
    parser.add_argument('--mode', help="R = the reference's embedding block, S = standard lookup", default='R',
                        choices=['R', 'S'])
    parser.add_argument('--device', default='cuda:0')
    parser.add_argument('--seed', type=int, default=0)
    return parser


def int_list(text):
    return [int(v) for v in str(text).replace('[', '').replace(']', '').split(',') if v.strip()]
