"""Allocation-free entry point mivae_colsum_static (the file-scope device scratch array) vs mivae_colsum with a caller
workspace: bit-identical, two calls in a row (the second on a smaller input), and the size limit is enforced."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_colsum_static_matches_colsum_two_calls():
    import nonauto_lm_amd as M
    L = M._lib
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(5)
    for rows, D, dtype in ((16384, 64, torch.bfloat16), (3000, 37, torch.float32)):      # second call: smaller input
        x = (torch.randn(rows, D, generator=g) * 3).to(dtype).to(dev)
        code = L.dtype_code(dtype)
        need = L.query("mivae_col_workspace_floats", rows, D, code)
        assert need <= L.query("mivae_colsum_static_max_floats")
        ws = torch.empty(need, dtype=torch.float32, device=dev)
        ref = torch.full((D,), float("nan"), device=dev)
        got = torch.full((D,), float("nan"), device=dev)
        L.call("mivae_colsum", L.ptr(x), L.ptr(ref), L.ptr(ws), rows, D, code, L.stream())
        torch.cuda.synchronize()
        before = torch.cuda.memory_allocated()
        L.call("mivae_colsum_static", L.ptr(x), L.ptr(got), rows, D, code, L.stream())
        torch.cuda.synchronize()
        assert torch.cuda.memory_allocated() == before
        assert torch.equal(got, ref), (rows, D, dtype)
        assert (got.double().cpu() - x.double().cpu().sum(0)).abs().max() <= 1e-3 * x.double().abs().sum(0).max().cpu()


def test_colsum_static_under_graph_capture():
    """the entry point is meant for HIP-graph capture -- after mivae_colsum_static_init() (the array's address
    resolved eagerly) a captured call replays correctly, allocates nothing, and the stream's device decides the array."""
    import nonauto_lm_amd as M
    L = M._lib
    dev = torch.device("cuda:0")
    L.call("mivae_colsum_static_init")
    rows, D = 9000, 48
    x = torch.randn(rows, D, generator=torch.Generator().manual_seed(9)).to(dev)
    ws = torch.empty(L.query("mivae_col_workspace_floats", rows, D, 0), dtype=torch.float32, device=dev)
    ref = torch.empty(D, device=dev)
    got = torch.full((D,), float("nan"), device=dev)
    L.call("mivae_colsum", L.ptr(x), L.ptr(ref), L.ptr(ws), rows, D, 0, L.stream())
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        L.call("mivae_colsum_static", L.ptr(x), L.ptr(got), rows, D, 0, L.stream())
    x.mul_(2.0)                      # replay on new values of the static input
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(got, 2.0 * ref)


def test_colsum_static_rejects_oversize():
    import nonauto_lm_amd as M
    L = M._lib
    rows, D = 1 << 16, 4096                                     # 1024 blocks x 2 x 4096 floats > 2^20
    assert L.query("mivae_col_workspace_floats", rows, D, 0) > L.query("mivae_colsum_static_max_floats")
    rc = L.lib().mivae_colsum_static(None, None, rows, D, 0, None)
    assert rc == -3 and b"static buffer" in L.lib().mivae_last_error()
