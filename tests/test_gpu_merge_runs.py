"""lsf_merge_sorted_runs (round 4): the ascending merge of two ascending runs of distinct voxel indices, several pairs per
launch -- what the compact-face plan of a slab run uses instead of torch.sort(torch.cat(...)) (a face's band voxels are the
face slices' entries of the INTERIOR and of the BOUNDARY list).  Against torch.sort, with empty runs, one-sided pairs and
four pairs at once."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu


def test_merge_equals_sort():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from levelsetfusion_python_amd import _lib, device as dev
    g = torch.Generator(device="cpu").manual_seed(7)
    cases = []
    for na, nb in ((1000, 37), (0, 513), (700, 0), (50_000, 91_234)):
        pool = torch.randperm(4 * (na + nb) + 8, generator=g)[:na + nb].to(torch.int32)
        a = torch.sort(pool[:na]).values.cuda()
        b = torch.sort(pool[na:]).values.cuda()
        out = torch.full((na + nb,), -1, dtype=torch.int32, device="cuda")
        cases.append((a, b, out))
    n = len(cases)
    arr = lambda k: (ctypes.c_void_p * n)(*[c[k].data_ptr() if c[k].numel() else None for c in cases])
    cnt = lambda k: (ctypes.c_int64 * n)(*[c[k].numel() for c in cases])
    _lib.check(_lib.lib.lsf_merge_sorted_runs(arr(0), cnt(0), arr(1), cnt(1), arr(2), n, dev.stream_ptr()),
               "lsf_merge_sorted_runs")
    for a, b, out in cases:
        assert torch.equal(out, torch.sort(torch.cat([a, b])).values)
    # argument errors are reported, nothing is launched
    assert _lib.lib.lsf_merge_sorted_runs(arr(0), cnt(0), arr(1), cnt(1), arr(2), 5, None) == -1
    assert _lib.lib.lsf_merge_sorted_runs(None, cnt(0), arr(1), cnt(1), arr(2), 1, None) == -1
    assert _lib.lib.lsf_merge_sorted_runs(None, None, None, None, None, 0, None) == 0
