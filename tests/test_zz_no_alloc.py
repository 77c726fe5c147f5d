"""vitmi_layernorm_bwd_static (the entry point that allocates nothing: its per-block partial sums live in a device
array of the library) against vitmi_layernorm_bwd with a caller workspace, through the C ABI: bit-identical dx,
dgamma, dbeta and residual column sums on two calls in a row, the second with a smaller input (fewer blocks and a
shorter row, so stale words of the first call sit beside the second call's in the shared array), and a clear error
beyond the array's size.  Collected last on purpose (file name)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _case(rows, D, dtype, seed, with_res):
    from visiontransformer_amd import ops
    from visiontransformer_amd._lib import check, lib
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(rows, D, generator=g).to(DEV, dtype)
    dy = torch.randn(rows, D, generator=g).to(DEV, dtype)
    res = torch.randn(rows, D, generator=g).to(DEV, dtype) if with_res else None
    gamma = (1 + 0.1 * torch.randn(D, generator=g)).to(DEV)
    beta = torch.zeros(D, device=DEV)
    _, mean, rstd = ops.layernorm_fwd(x, gamma, beta, 1e-5)
    ref = ops.layernorm_bwd(dy, x, gamma, mean, rstd, dres=res, want_dres_colsum=with_res)
    dx = torch.full_like(x, float("nan"))
    dg = torch.full((D,), float("nan"), device=DEV)
    db = torch.full((D,), float("nan"), device=DEV)
    rs = torch.full((D,), float("nan"), device=DEV) if with_res else None
    p = ops._p
    check(lib.vitmi_layernorm_bwd_static(p(dy), p(x), x.stride(0), p(gamma), p(mean), p(rstd), p(res), p(dx), p(dg),
                                         p(db), p(rs), ops._dt(x), rows, D, ops._stream()),
          "vitmi_layernorm_bwd_static")
    torch.cuda.synchronize()
    got = (dx, dg, db) + ((rs,) if with_res else ())
    for a, b in zip(got, ref):
        assert torch.equal(a, b)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_static_scratch_layernorm_backward_matches_workspace_version(dtype):
    _case(50432, 768, dtype, 1, True)        # 768 blocks: the whole array at D = 768
    _case(1000, 192, dtype, 2, True)         # smaller grid and row, right after
    _case(777, 1024, dtype, 3, False)        # the widest row the array holds, no residual sums
    _case(3, 50, dtype, 4, True)             # unaligned row length: scalar kernel


def test_static_scratch_rejects_rows_beyond_its_size():
    from visiontransformer_amd import ops
    from visiontransformer_amd._lib import VitmiError, check, lib
    D = 1028
    x = torch.zeros(4, D, device=DEV)
    v = torch.zeros(4, device=DEV)
    g = torch.ones(D, device=DEV)
    p = ops._p
    with pytest.raises(VitmiError, match="exceeds the static scratch"):
        check(lib.vitmi_layernorm_bwd_static(p(x), p(x), D, p(g), p(v), p(v), None, p(x), p(g), p(g), None, 0, 4, D,
                                             ops._stream()), "vitmi_layernorm_bwd_static")


def test_static_scratch_call_allocates_nothing_and_runs_under_stream_capture():
    """What the entry point exists for: the call takes no device memory (the allocator's and the driver's counts do not
    move across it) and is legal while a stream is being captured, where hipMalloc is not; the replayed graph gives the
    eager result bit for bit.  One eager call comes first: the first call on a device looks the array's address up
    (hipGetSymbolAddress), one-time set-up that callers do before they start capturing, like a warm-up of any other
    kernel."""
    from visiontransformer_amd import ops
    from visiontransformer_amd._lib import check, lib
    rows, D = 4000, 768
    g = torch.Generator().manual_seed(9)
    x = torch.randn(rows, D, generator=g).to(DEV, torch.bfloat16)
    dy = torch.randn(rows, D, generator=g).to(DEV, torch.bfloat16)
    res = torch.randn(rows, D, generator=g).to(DEV, torch.bfloat16)
    gamma = (1 + 0.1 * torch.randn(D, generator=g)).to(DEV)
    _, mean, rstd = ops.layernorm_fwd(x, gamma, torch.zeros(D, device=DEV), 1e-5)
    outs = [torch.zeros_like(x), torch.zeros(D, device=DEV), torch.zeros(D, device=DEV), torch.zeros(D, device=DEV)]
    p = ops._p

    def call(stream):
        check(lib.vitmi_layernorm_bwd_static(p(dy), p(x), x.stride(0), p(gamma), p(mean), p(rstd), p(res), p(outs[0]),
                                             p(outs[1]), p(outs[2]), p(outs[3]), ops._dt(x), rows, D, stream),
              "vitmi_layernorm_bwd_static")

    call(ops._stream())
    torch.cuda.synchronize()
    eager = [o.clone() for o in outs]
    free0, alloc0, reserved0 = torch.cuda.mem_get_info()[0], torch.cuda.memory_allocated(), torch.cuda.memory_reserved()
    call(ops._stream())
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info()[0] == free0
    assert torch.cuda.memory_allocated() == alloc0 and torch.cuda.memory_reserved() == reserved0
    for o in outs:
        o.zero_()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        call(ops._stream())                      # torch's current stream is the capturing one here
    torch.cuda.synchronize()
    assert all(float(o.float().abs().sum()) == 0.0 for o in outs)      # captured, not run
    graph.replay()
    torch.cuda.synchronize()
    for a, b in zip(outs, eager):
        assert torch.equal(a, b)
