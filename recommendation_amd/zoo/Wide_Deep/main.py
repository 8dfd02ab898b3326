"""Train/eval entry point of Wide & Deep; loader, flags and loop of the reference's Wide_Deep/main.py
(argv is honoured).

load_dataset (Wide_Deep/main.py:32-85): train.csv + test.csv with columns userid, itemid, label,
user_type, item_type and numeric features; the two type columns are the categorical fields, their
string join '-' is indexed into one cross feature, everything else numeric is standardised over
train + test together; cardinalities are (#distinct values + 1)."""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import recommendation_amd as ra  # noqa: E402
from recommendation_amd import train_loop as tl  # noqa: E402

ra.enable_zoo_imports()
from Wide_Deep import Wide_Deep  # noqa: E402

CATE_COLS = ['user_type', 'item_type']
CROSS_COLS = [['user_type', 'item_type']]


def load_dataset(file_path):
    import pandas as pd
    from sklearn.preprocessing import StandardScaler
    train = pd.read_csv(os.path.join(file_path, 'train.csv'))
    test = pd.read_csv(os.path.join(file_path, 'test.csv'))
    train['type'], test['type'] = 'train', 'test'
    data = pd.concat([train, test], axis=0)
    cont_cols = [c for c in data.columns if c not in ['userid', 'itemid', 'label', 'type'] and c not in CATE_COLS]
    cont = StandardScaler().fit_transform(data[cont_cols])
    cross = []
    for cols in CROSS_COLS:
        joined = data[cols].astype(str).agg('-'.join, axis=1).values
        _, inv = np.unique(joined, return_inverse=True)          # value -> index in sorted order of the values
        cross.append(inv)
    cross = np.stack(cross, axis=1)
    cate = data[CATE_COLS].values
    is_train = (data['type'] == 'train').values
    label = data['label'].values
    cross_list = [int(np.unique(cross[:, i]).shape[0] + 1) for i in range(cross.shape[1])]
    cate_list = [int(np.unique(cate[:, i]).shape[0] + 1) for i in range(cate.shape[1])]
    tr, te = is_train, ~is_train
    return (cross[tr], cate[tr], cont[tr], label[tr], cross[te], cate[te], cont[te], label[te], cross_list, cate_list)


def make_parser():
    p = argparse.ArgumentParser()
    p.add_argument('--input_dir', type=str, default='')
    p.add_argument('--model_name', type=str, default='Wide_Deep.ckpt')
    p.add_argument('--embed_size', type=int, default=8)
    p.add_argument('--hidden_layers', type=tl.int_list, default='[32, 64, 128]')
    p.add_argument('--batch_size', type=int, default=128)
    p.add_argument('--epoch', type=int, default=10)
    p.add_argument('--regular_rate', type=float, default=0.1)
    p.add_argument('--learning_rate', type=float, default=0.001)
    p.add_argument('--dropout', type=float, default=0.2)
    p.add_argument('--decay_steps', type=int, default=10000)
    p.add_argument('--decay_rate', type=float, default=0.99)
    p.add_argument('--device', type=str, default='cuda:0')
    p.add_argument('--seed', type=int, default=0)
    return p


def fit(args, data=None, session=None, log=print):
    from recommendation_amd.model_base import get_checkpoint_state
    if data is None:
        data = load_dataset(args.input_dir)
    Xtr_r, Xtr_c, Xtr_f, ytr, Xte_r, Xte_c, Xte_f, yte, cross_list, cate_list = data
    sess = session or ra.Session(args.device, seed=args.seed)
    model = Wide_Deep.Wide_Deep(args, Xtr_r.shape[1], Xtr_c.shape[1], Xtr_f.shape[1], cross_list, cate_list)
    model.build()
    ckpt_dir = os.path.join(args.input_dir, args.model_name)
    ckpt = get_checkpoint_state(ckpt_dir)
    if ckpt:
        log('Loading model parameters from %s' % ckpt)
        model.saver.restore(sess, ckpt)
    else:
        log('Creating model with inital parameters')
        model.initialize(sess, seed=args.seed)
    step = 0
    for epoch in range(args.epoch):
        t0 = time.time()
        losses = []
        for s, e in tl.batches(Xtr_c.shape[0], args.batch_size):
            losses.append(model.run(sess, Xtr_r[s:e], Xtr_c[s:e], Xtr_f[s:e], ytr[s:e], True, step))
            step += 1
        took = time.strftime('%H: %M: %S', time.gmtime(time.time() - t0))
        log('Epoch %d training: loss = %.4f, took: %s' % (epoch + 1, np.mean(losses) if losses else float('nan'), took))
        actuals, predicts = [], []
        for s, e in tl.batches(Xte_c.shape[0], args.batch_size):
            a, p = model.run(sess, Xte_r[s:e], Xte_c[s:e], Xte_f[s:e], yte[s:e], False)
            actuals.append(a)
            predicts.append(p)
        took = time.strftime('%H: %M: %S', time.gmtime(time.time() - t0))
        if actuals:
            actuals, predicts = np.concatenate(actuals), np.concatenate(predicts)
            accuracy = float(len(actuals == predicts)) / float(len(actuals))     # == 1.0, as in the reference
            log('Epoch %d testing: Accuracy = %.4f, took: %s' % (epoch + 1, accuracy, took))
        model.saver.save(sess, os.path.join(ckpt_dir, args.model_name))
    return model


if __name__ == '__main__':
    a = make_parser().parse_args()
    if isinstance(a.hidden_layers, str):
        a.hidden_layers = tl.int_list(a.hidden_layers)
    fit(a)
