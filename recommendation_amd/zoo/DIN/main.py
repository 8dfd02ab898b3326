"""Train/eval entry point of DIN: flags and defaults of the reference's DIN/main.py:66-72, its GAUC /
AUC bookkeeping (DIN/main.py:12-61) and its epoch loop.  Differences, on purpose: argv is honoured;
the global step is advanced per training call (in the reference it stays 0, so evaluation runs on
every step and then crashes on a missing argument, DIN/main.py:100-104); evaluation happens every
--eval_every steps; the dataset is read with pickle only from a file the user names."""
import argparse
import os
import pickle
import random
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import recommendation_amd as ra  # noqa: E402

ra.enable_zoo_imports()
from DIN import DIN, data  # noqa: E402


def calc_auc(raw_arr):
    """AUC from [is_negative, is_positive, score] records by the trapezoid rule over the score-sorted
    list; -0.5 when (almost) every record has the same class, None when a class is absent."""
    fp = tp = 0.0
    area = 0.0
    for neg, pos, _ in sorted(raw_arr, key=lambda r: r[2]):
        fp_next, tp_next = fp + neg, tp + pos
        area += (fp_next - fp) * (tp_next + tp)
        fp, tp = fp_next, tp_next
    limit = len(raw_arr) - 1e-3
    if tp > limit or fp > limit:
        return -0.5
    if tp * fp > 0.0:
        return 1.0 - area / (2.0 * tp * fp)
    return None


def score_records(p_and_n):
    return [[0, 1, s] for s in p_and_n[:, 0].tolist()] + [[1, 0, s] for s in p_and_n[:, 1].tolist()]


def evaluate(sess, model, test_set, args, state):
    auc_sum, records = 0.0, []
    for _, uij in data.DataInputTest(test_set, args.test_batch_size):
        auc_, score_ = model.eval(sess, uij)
        records += score_records(score_)
        auc_sum += auc_ * len(uij[0])
    gauc = auc_sum / len(test_set)
    auc = calc_auc(records)
    if state['best_auc'] < gauc:
        state['best_auc'] = gauc
        model.save(sess, args.model_name)
    return gauc, auc


def make_parser():
    p = argparse.ArgumentParser()
    p.add_argument('--input_dir', help='The directory of input', type=str, default='')
    p.add_argument('--model_name', help='The directory of model', type=str, default='DIN.ckpt')
    p.add_argument('--train_batch_size', help='size of train batch', type=int, default=32)
    p.add_argument('--test_batch_size', help='size for test batch', type=int, default=512)
    p.add_argument('--hidden_units', help='hidden units', type=int, default=32)
    p.add_argument('--epoch', help='number of epochs', type=int, default=10)
    p.add_argument('--learning_rate', help='learning rate', type=float, default=0.1)
    p.add_argument('--eval_every', type=int, default=1000)
    p.add_argument('--device', default='cuda:0')
    return p


def run(args, train_set, test_set, cate_list, counts, log=print):
    user_count, item_count, cate_count = counts
    state = {'best_auc': 0.0}
    with ra.Session(args.device) as sess:
        model = DIN.DIN(args, user_count, item_count, cate_count, cate_list)
        model.initialize(sess)
        t0 = time.time()
        for _ in range(args.epoch):
            random.shuffle(train_set)
            loss_sum = 0.0
            for _, uij in data.DataInput(train_set, args.train_batch_size):
                loss_sum += model.train(sess, uij, args.learning_rate)
                model.global_step += 1
                if model.global_step % args.eval_every == 0:
                    gauc, auc = evaluate(sess, model, test_set, args, state)
                    log('Epoch %d Global_step %d\tTrain_loss: %.4f\tEval_GAUC: %.4f\tEval_AUC: %.4f'
                        % (model.global_epoch_step, model.global_step, loss_sum / args.eval_every, gauc,
                           auc if auc is not None else float('nan')))
                    loss_sum = 0.0
            log('Epoch %d DONE\tCost time: %.2f' % (model.global_epoch_step, time.time() - t0))
            model.global_epoch_step += 1
        log('best test_gauc: %s' % state['best_auc'])
    return model


if __name__ == '__main__':
    args = make_parser().parse_args()
    with open(args.input_dir, 'rb') as f:       # the user's own dataset file, as in the reference
        train_set = pickle.load(f)
        test_set = pickle.load(f)
        cate_list = pickle.load(f)
        counts = pickle.load(f)
    run(args, train_set, test_set, cate_list, counts)
