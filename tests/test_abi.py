"""The C-ABI library loads on a CPU-only machine and exports every symbol include/ctrhip.h
declares, with the bindings in recommendation_amd/_lib.py covering exactly that set.
No compute call is made here (there is no GPU)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, 'include', 'ctrhip.h')


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(ctr_[a-z0-9_]+)\s*\(', text)))


def test_header_declares_the_hot_path():
    syms = declared_symbols()
    for must in ('ctr_embed_fwd', 'ctr_emb_sort', 'ctr_emb_sparse_apply', 'ctr_gemm_f32', 'ctr_cross_fwd',
                 'ctr_cross_bwd', 'ctr_loss_head', 'ctr_dense_update', 'ctr_pnn_inner_fwd', 'ctr_afm_pair_fwd'):
        assert must in syms


def test_library_exports_every_declared_symbol():
    from recommendation_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), 'run __graft_entry__.build() first'
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for s in declared_symbols():
        assert hasattr(lib, s), 'libctrhip.so does not export %s' % s


def test_bindings_cover_exactly_the_header():
    from recommendation_amd import _lib
    assert sorted(_lib.SIGNATURES) == declared_symbols()
    lib = _lib.load()
    assert lib.ctr_version() >= 100
    assert lib.ctr_last_error() is not None


def test_host_only_queries_need_no_gpu():
    from recommendation_amd import _lib
    lib = _lib.load()
    assert lib.ctr_emb_sort_workspace_bytes(1000) > 8000
    assert lib.ctr_gemm_f32_workspace_bytes(64, 32, 65536) > 0          # split-K slab
    assert lib.ctr_gemm_f32_workspace_bytes(65536, 256, 429) == 0
    assert lib.ctr_keep_hash24(1, 2, 3) < (1 << 24)


def test_no_cpu_fallback():
    """Sessions refuse to start without a GPU and operators refuse CPU tensors."""
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    from recommendation_amd import Session
    with pytest.raises(RuntimeError):
        Session('cuda:0')
