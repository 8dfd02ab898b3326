"""Train/eval entry point of NCF; loader, flags and loop of the reference's NCF/main.py (argv is honoured).

load_dataset (NCF/main.py:19-50): train.csv + test.csv with columns userid, itemid, label; ids are
re-indexed in order of first appearance over train followed by test."""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import recommendation_amd as ra  # noqa: E402
from recommendation_amd import train_loop as tl  # noqa: E402

ra.enable_zoo_imports()
from NCF import NCF  # noqa: E402


def load_dataset(file_path):
    import pandas as pd
    cols = ['userid', 'itemid', 'label']
    train = pd.read_csv(os.path.join(file_path, 'train.csv'))[cols]
    test = pd.read_csv(os.path.join(file_path, 'test.csv'))[cols]
    both = pd.concat([train, test], axis=0)
    user_map = {v: i for i, v in enumerate(both['userid'].unique())}      # first-appearance order
    item_map = {v: i for i, v in enumerate(both['itemid'].unique())}
    enc = lambda df: np.stack([df['userid'].map(user_map).values, df['itemid'].map(item_map).values], axis=1)
    return enc(train), train['label'].values, enc(test), test['label'].values, len(user_map), len(item_map)


def make_parser():
    p = argparse.ArgumentParser()
    p.add_argument('--input_dir', type=str, default='')
    p.add_argument('--model_name', type=str, default='NCF.ckpt')
    p.add_argument('--embed_size', type=int, default=16)
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
    Xtr, ytr, Xte, yte, user_size, item_size = data
    sess = session or ra.Session(args.device, seed=args.seed)
    model = NCF.NCF(args, user_size, item_size)
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
        for s, e in tl.batches(Xtr.shape[0], args.batch_size):
            losses.append(model.run(sess, Xtr[s:e], ytr[s:e], True, step))
            step += 1
        took = time.strftime('%H: %M: %S', time.gmtime(time.time() - t0))
        log('Epoch %d training: loss = %.4f, took: %s' % (epoch + 1, np.mean(losses) if losses else float('nan'), took))
        actuals, predicts = [], []
        for s, e in tl.batches(Xte.shape[0], args.batch_size):
            a, p = model.run(sess, Xte[s:e], yte[s:e], False)
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
    fit(make_parser().parse_args())
