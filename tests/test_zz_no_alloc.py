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
