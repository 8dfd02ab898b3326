"""Criteo text -> ids / dense / labels (host code of the library; runs without a GPU): against a line-by-line Python
restatement of the format, with empty fields, a missing trailing newline, CRLF, short lines, negative counts,
upper-case tokens, every thread count, a row limit and block-wise reading of a file."""
import numpy as np
import pytest

from recommendation_amd import criteo

CARDS = [1460, 583, 10131227, 2202608, 305, 24, 12517, 633, 3, 93145, 5683, 8351593, 3194, 27, 14992, 5461306, 10, 5652,
         2173, 4, 7046547, 18, 15, 286181, 105, 142572]


def _make(n, seed, crlf=False, trailing_newline=True):
    rng = np.random.default_rng(seed)
    lines = []
    for r in range(n):
        f = [str(int(rng.integers(0, 2)))]
        for j in range(13):
            u = rng.random()
            f.append('' if u < 0.2 else str(int(rng.integers(-3, 100000)) if u < 0.9 else int(rng.integers(0, 2 ** 31))))
        for j in range(26):
            u = rng.random()
            tok = '' if u < 0.15 else '%08x' % int(rng.integers(0, 2 ** 32))
            f.append(tok.upper() if u > 0.9 else (tok.lstrip('0') if u > 0.8 else tok))
        if r % 97 == 5:
            f = f[:int(rng.integers(1, 40))]                       # a short line: the rest counts as empty
        lines.append('\t'.join(f))
        if r % 131 == 7:
            lines.append('')                                       # an empty line carries no sample
    sep = '\r\n' if crlf else '\n'
    return (sep.join(lines) + (sep if trailing_newline else '')).encode(), [ln for ln in lines if ln != '']


def _reference(lines, log_dense):
    ids = np.zeros((len(lines), 26), np.int32)
    dense = np.zeros((len(lines), 13), np.float32)
    label = np.zeros(len(lines), np.float32)
    for r, ln in enumerate(lines):
        f = ln.split('\t') + [''] * 40
        label[r] = 1.0 if f[0] not in ('', '0') else 0.0
        for j in range(13):
            v = float(int(f[1 + j])) if f[1 + j] not in ('', '-') else 0.0
            dense[r, j] = np.log1p(np.float32(max(v, 0.0))) if log_dense else np.float32(v)
        for k in range(26):
            ids[r, k] = (int(f[14 + k], 16) if f[14 + k] else 0) % CARDS[k]
    return ids, dense, label


@pytest.mark.parametrize('threads', [1, 2, 3, 8])
@pytest.mark.parametrize('log_dense', [False, True])
def test_parse_matches_the_line_by_line_restatement(threads, log_dense):
    text, lines = _make(3000, 1)
    ids, dense, label = criteo.parse_text(text, CARDS, log_dense=log_dense, threads=threads)
    r_ids, r_dense, r_label = _reference(lines, log_dense)
    assert np.array_equal(ids, r_ids) and np.array_equal(label, r_label)
    assert np.allclose(dense, r_dense, rtol=1e-6, atol=0) if log_dense else np.array_equal(dense, r_dense)
    assert all(int(ids[:, k].max()) < CARDS[k] and int(ids[:, k].min()) >= 0 for k in range(26))


def test_line_endings_row_limit_and_empty_input():
    text, lines = _make(500, 2, crlf=True, trailing_newline=False)
    ids, dense, label = criteo.parse_text(text, CARDS, log_dense=False, threads=4)
    r_ids, r_dense, r_label = _reference(lines, False)
    assert np.array_equal(ids, r_ids) and np.array_equal(dense, r_dense) and np.array_equal(label, r_label)
    ids2, dense2, label2 = criteo.parse_text(text, CARDS, log_dense=False, threads=4, max_rows=123)
    assert len(label2) == 123 and np.array_equal(ids2, r_ids[:123]) and np.array_equal(dense2, r_dense[:123])
    e = criteo.parse_text(b'', CARDS)
    assert e[0].shape == (0, 26) and e[1].shape == (0, 13) and e[2].shape == (0,)
    with pytest.raises(ValueError):
        criteo.parse_text(text, CARDS[:-1])
    with pytest.raises(ValueError, match='cards'):
        criteo.parse_text(text, CARDS[:-1] + [0])


def test_block_wise_file_reading_and_the_loader_tuple(tmp_path):
    text, lines = _make(4000, 3)
    (tmp_path / 'train.txt').write_bytes(text)
    t2, l2 = _make(700, 4)
    (tmp_path / 'test.txt').write_bytes(t2)
    whole = criteo.read_file(str(tmp_path / 'train.txt'), CARDS, block_bytes=50000, log_dense=True, threads=3)
    r_ids, r_dense, r_label = _reference(lines, True)
    assert np.array_equal(whole[0], r_ids) and np.array_equal(whole[2], r_label) and np.allclose(whole[1], r_dense, rtol=1e-6)
    tup = criteo.load_dataset(str(tmp_path), CARDS, threads=2)
    assert len(tup) == 7 and tup[6] == CARDS and tup[0].shape == (4000, 26) and tup[3].shape == (700, 26)
    both = np.concatenate([tup[1], tup[4]]).astype(np.float64)
    assert np.allclose(both.mean(0), 0, atol=1e-4) and np.allclose(both.std(0), 1, atol=1e-3)
