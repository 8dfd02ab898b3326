"""ctr_emb_sort_static (scratch built into the library, no allocation) against ctr_emb_sort (caller workspace),
through the C ABI: bit-identical sorted pairs on two calls in a row, the second with a smaller input; the size
limit is enforced; a call after the first leaves the device's free memory untouched and can be captured into a HIP
graph (stream capture refuses an allocation or a synchronisation), whose replay sorts whatever the input holds."""
import ctypes as C

import numpy as np
import pytest
import torch

from tests.helpers import assert_sorted_pairs

pytestmark = pytest.mark.gpu


def _sort_both(lib, idx, row_offset, N, total_rows, rmode):
    B, F = idx.shape
    n = B * F
    dev = idx.device
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    outs = []
    for static in (False, True):
        keys = torch.full((n,), -1, dtype=torch.int32, device=dev)
        src = torch.full((n,), -1, dtype=torch.int32, device=dev)
        args = [C.c_void_p(idx.data_ptr()), C.c_void_p(row_offset.data_ptr()), B, F, N, total_rows, rmode,
                C.c_void_p(keys.data_ptr()), C.c_void_p(src.data_ptr())]
        if static:
            rc = lib.ctr_emb_sort_static(*args, stream)
        else:
            # the workspace with which ctr_emb_sort takes the same path as the built-in scratch allows
            nbytes = lib.ctr_emb_sort_workspace_bytes_fields(B, F, total_rows, rmode)
            ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            rc = lib.ctr_emb_sort(*args, C.c_void_p(ws.data_ptr()), nbytes, stream)
        assert rc == 0, lib.ctr_last_error()
        torch.cuda.synchronize()
        outs.append((keys.cpu().numpy(), src.cpu().numpy()))
    return outs


def test_static_scratch_sort_matches_workspace_sort_on_two_calls_in_a_row():
    from recommendation_amd import _lib
    lib = _lib.load()
    dev = torch.device('cuda:0')
    rng = np.random.default_rng(5)
    cards = [1460, 583, 1013122, 220260, 305, 24, 12517, 633, 3, 93145, 5683, 835159, 3194]
    offs = np.concatenate([[0], np.cumsum(cards)]).astype(np.int64)
    row_offset = torch.as_tensor(offs, device=dev)
    total = int(offs[-1])
    for B in (40000, 700):                      # the second call is smaller: stale scratch content must not matter
        ids = np.stack([np.minimum(rng.pareto(1.05, B).astype(np.int64), v - 1) for v in cards], 1).astype(np.int32)
        ids[rng.integers(0, B, 5), rng.integers(0, len(cards), 5)] = -1            # out-of-range ids
        idx = torch.as_tensor(ids, device=dev)
        (k0, s0), (k1, s1) = _sort_both(lib, idx, row_offset, len(cards), total, 0)
        assert np.array_equal(k0, k1) and np.array_equal(s0, s1)
        ku = k0.view(np.uint32)
        assert np.all(ku[:-1] <= ku[1:])
        # against the definition: stable sort of (key, src) by key
        key = np.where((ids >= 0) & (ids < np.asarray(cards)), offs[:-1] + ids, total).reshape(-1).astype(np.uint32)
        assert_sorted_pairs(ku, s0.view(np.uint32), key, np.arange(key.size), total)
    # reference-exact (one-hot) pair form too
    idx = torch.as_tensor(rng.integers(0, 3, (513, 4)).astype(np.int32), device=dev)
    ro = torch.as_tensor(np.array([0, 3, 6, 9, 12], dtype=np.int64), device=dev)
    (k0, s0), (k1, s1) = _sort_both(lib, idx, ro, 12, 12, 1)
    assert np.array_equal(k0, k1) and np.array_equal(s0, s1)


def test_static_scratch_sort_allocates_nothing_and_replays_from_a_captured_graph():
    from recommendation_amd import _lib
    lib = _lib.load()
    dev = torch.device('cuda:0')
    rng = np.random.default_rng(9)
    cards = [50000, 7, 1200, 300000]
    offs = np.concatenate([[0], np.cumsum(cards)]).astype(np.int64)
    row_offset = torch.as_tensor(offs, device=dev)
    total, B, F = int(offs[-1]), 6000, len(cards)
    n = B * F

    def draw():
        return np.stack([rng.integers(0, v, B) for v in cards], 1).astype(np.int32)

    idx = torch.as_tensor(draw(), device=dev)
    keys = torch.zeros(n, dtype=torch.int32, device=dev)
    src = torch.zeros(n, dtype=torch.int32, device=dev)

    def call(stream):
        rc = lib.ctr_emb_sort_static(C.c_void_p(idx.data_ptr()), C.c_void_p(row_offset.data_ptr()), B, F, F, total, 0,
                                     C.c_void_p(keys.data_ptr()), C.c_void_p(src.data_ptr()), C.c_void_p(stream))
        assert rc == 0, lib.ctr_last_error()

    call(torch.cuda.current_stream().cuda_stream)          # the first call on a device looks the scratch's address up
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info(dev)[0]
    held0 = torch.cuda.memory_reserved(dev)
    call(torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info(dev)[0] == free0 and torch.cuda.memory_reserved(dev) == held0

    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        call(torch.cuda.current_stream().cuda_stream)
    for _ in range(2):                                      # new ids into the captured input, replay, compare
        ids = draw()
        idx.copy_(torch.as_tensor(ids, device=dev))
        keys.fill_(-1)
        src.fill_(-1)
        graph.replay()
        torch.cuda.synchronize()
        key = (offs[:-1] + ids).reshape(-1).astype(np.uint32)
        order = np.argsort(key, kind='stable')
        assert np.array_equal(keys.cpu().numpy().view(np.uint32), key[order])
        assert np.array_equal(src.cpu().numpy().view(np.uint32), order.astype(np.uint32))
    torch.cuda.synchronize()
    del graph


def test_static_scratch_sort_rejects_inputs_beyond_its_limit():
    from recommendation_amd import _lib
    lib = _lib.load()
    # nothing is dereferenced before the size check: 2^21 + 64 lookups
    rc = lib.ctr_emb_sort_static(None, None, (1 << 15) + 1, 64, 64, 1000, 0, None, None, None)
    assert rc == -1
    assert b'exceeds the built-in scratch' in lib.ctr_last_error()
