"""amber_mpn_csr_build_static: the CSR build with its scratch in a file-scope device array (no workspace argument,
no allocation) against amber_mpn_csr_build, bit for bit, through the C ABI.  Collected last on purpose (file name):
the round's end runs `pytest -x`."""

import pytest
import torch

from amber_amd import _lib
from amber_amd import csr as csr_mod
from amber_amd import synthetic as syn

pytestmark = pytest.mark.gpu


def _static_build(ei: torch.Tensor, n: int):
    """(rc, rowptr, src_sorted, dst_sorted, perm, status) of one call; outputs allocated BEFORE the call."""
    lib = _lib.load()
    dev = ei.device
    e = int(ei.shape[1])
    out = [torch.full((n + 1,), -7, dtype=torch.int32, device=dev)] + \
          [torch.full((e,), -7, dtype=torch.int32, device=dev) for _ in range(3)] + \
          [torch.full((1,), -7, dtype=torch.int32, device=dev)]
    with torch.cuda.device(dev):
        rc = lib.amber_mpn_csr_build_static(ei.data_ptr(), e, n, *[t.data_ptr() for t in out], _lib.stream_ptr(dev))
    return (rc, *out)


def _check_against_workspace_build(ei, n):
    want = csr_mod.build_csr(ei, n, validate=False)
    rc, rowptr, src, dst, perm, status = _static_build(ei, n)
    assert rc == 0, _lib.load().amber_mpn_last_error()
    assert int(status.item()) == 0
    for got, ref, name in ((rowptr, want.rowptr, "rowptr"), (src, want.src_sorted, "src_sorted"),
                           (dst, want.dst_sorted, "dst_sorted"), (perm, want.perm, "perm")):
        assert torch.equal(got, ref), f"{name} differs from amber_mpn_csr_build"
    # and against the definition: a stable sort by destination
    order = torch.sort(ei[1], stable=True).indices.to(torch.int32)
    assert torch.equal(perm, order)


def test_static_csr_two_calls_second_smaller(gpu_device):
    """Two calls in a row on one GPU share the array, which keeps the first call's contents: the second, smaller graph
    must not see them (counts and cursors are cleared by the call itself)."""
    big = syn.grid_batch(num_graphs=3, nx=60, ny=50, seed=3).to(gpu_device)
    small = syn.grid_batch(num_graphs=1, nx=9, ny=7, seed=4).to(gpu_device)
    _check_against_workspace_build(big.edge_index, int(big.x.shape[0]))
    _check_against_workspace_build(small.edge_index, int(small.x.shape[0]))
    # a hub row longer than a tile (the wave-per-row sort) and isolated nodes, then the small graph once more
    g = torch.Generator().manual_seed(11)
    n = 5000
    ei = torch.stack([torch.randint(0, n, (60000,), generator=g), torch.randint(0, 40, (60000,), generator=g)]).to(gpu_device)
    _check_against_workspace_build(ei, n)
    _check_against_workspace_build(small.edge_index, int(small.x.shape[0]))


def test_static_csr_edge_cases(gpu_device):
    lib = _lib.load()
    empty = torch.zeros(2, 0, dtype=torch.int64, device=gpu_device)
    rc, rowptr, *_ = _static_build(empty, 5)
    assert rc == 0 and rowptr.tolist() == [0] * 6
    # an index outside [0, N) raises the status word, as in amber_mpn_csr_build
    bad = torch.tensor([[0, 1, 2], [1, 9, 0]], dtype=torch.int64, device=gpu_device)
    rc, *_, status = _static_build(bad, 3)
    assert rc == 0 and int(status.item()) == 1
    # beyond the array: rejected before anything is launched, outputs untouched
    n, e = 3_000_000, 10
    assert lib.amber_mpn_csr_workspace_bytes(n, e) - 256 > _lib.CSR_STATIC_SCRATCH_BYTES
    ei = torch.zeros(2, e, dtype=torch.int64, device=gpu_device)
    rc, rowptr, *_ = _static_build(ei, n)
    assert rc == -3 and b"static array holds" in lib.amber_mpn_last_error()
    assert int(rowptr[0].item()) == -7
    # the largest graph that fits still works (N + E bound: 2 N + E + scan words <= 4 Mi int32)
    n, e = 1_000_000, 2_150_000
    assert lib.amber_mpn_csr_workspace_bytes(n, e) - 256 <= _lib.CSR_STATIC_SCRATCH_BYTES
    g = torch.Generator().manual_seed(5)
    ei = torch.stack([torch.randint(0, n, (e,), generator=g), torch.randint(0, n, (e,), generator=g)]).to(gpu_device)
    _check_against_workspace_build(ei, n)
