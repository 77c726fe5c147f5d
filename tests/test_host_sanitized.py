"""The host side of the C-ABI library under UndefinedBehaviorSanitizer (trap mode: a finding kills the process with
SIGILL).  Runs in a child process so that a trap is a failed test, not a dead test runner.  Needs
`python visiontransformer_amd/build.py --host-sanitize`; skipped when that library has not been built.  No GPU: only
entry points and arguments that return before the first device call (size queries, argument checks)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "visiontransformer_amd", "libvitmi_hip_hostsan.so")

CHILD = r"""
import ctypes, os, sys
sys.path.insert(0, sys.argv[2])
os.environ["VITMI_LIB"] = sys.argv[1]
from visiontransformer_amd import _lib
lib = _lib.lib
big = 2 ** 31 - 1
# workspace arithmetic at the benchmark's sizes and at the largest ints the signatures take: no signed overflow
for dt in (_lib.F32, _lib.BF16):
    assert lib.vitmi_attn_bwd_workspace_bytes(dt, 256, 197, 12, 64) > 0
    assert lib.vitmi_attn_bwd_dprobs_workspace_bytes(dt, 256, 197, 12, 64) > 0
    assert lib.vitmi_gemm_wgrad_workspace_bytes(dt, 3072, 768, 50432) >= 0
    assert lib.vitmi_ff_block_bwd_workspace_bytes(dt, 50432, 768, 3072) > 0
    assert lib.vitmi_attn_block_bwd_workspace_bytes(dt, 256, 197, 768, 12, 64) > 0
    assert lib.vitmi_wgrad_sqnorm_workspace_bytes(dt, 3072, 768, 50432, 197) >= 0
    lib.vitmi_attn_bwd_workspace_bytes(dt, big, 608, 64, 64)
    lib.vitmi_attn_bwd_dprobs_workspace_bytes(dt, 4096, 256, 64, 64)
    lib.vitmi_patch_embed_bwd_w_workspace_bytes(dt, 4096, 3, 224, 224, 16, 768)
# the paired weight gradients: plan arithmetic for the benchmark's pairs, with compute units reserved, for shapes that
# fall back to single products, and at the largest token counts
for res in (0, 8, 64, 128):
    assert lib.vitmi_set_reserved_cus(res) == 0
    assert lib.vitmi_gemm_wgrad_pair_workspace_bytes(_lib.BF16, 50432, 768, 3072, 3072, 768) > 0
    assert lib.vitmi_gemm_wgrad_pair_workspace_bytes(_lib.BF16, 50432, 768, 768, 2304, 768) > 0
    lib.vitmi_gemm_wgrad_pair_workspace_bytes(_lib.BF16, big - 63, 768, 768, 2304, 768)
    lib.vitmi_gemm_wgrad_pair_workspace_bytes(_lib.BF16, 64 * 40, 256, 256, 256, 256)
    lib.vitmi_gemm_wgrad_pair_workspace_bytes(_lib.BF16, 50432, 32768, 32768, 32768, 32768)
assert lib.vitmi_set_reserved_cus(0) == 0
assert lib.vitmi_gemm_wgrad_pair_workspace_bytes(_lib.BF16, 50400, 768, 768, 2304, 768) >= 0
# fp32 weight gradients split the token axis into slabs (few output tiles, K = tokens): a workspace of their own
assert lib.vitmi_gemm_wgrad_pair_workspace_bytes(_lib.F32, 50432, 768, 768, 2304, 768) > 0
assert lib.vitmi_gemm_wgrad_workspace_bytes(_lib.F32, 192, 512, 4160) > 0
assert lib.vitmi_gemm_wgrad_workspace_bytes(_lib.F32, 3072, 768, 100) == 0
lib.vitmi_gemm_wgrad_workspace_bytes(_lib.F32, 4096, 4096, big)
# the 32-pixel geometry: block workspaces with the one-launch kernels' partial sums, patches narrower than 8 pixels
for dt in (_lib.F32, _lib.BF16):
    assert lib.vitmi_ff_block_bwd_workspace_bytes(dt, 16640, 192, 512) > 0
    assert lib.vitmi_attn_block_bwd_workspace_bytes(dt, 256, 65, 192, 12, 8) > 0
assert lib.vitmi_patch_wide_group(_lib.BF16, _lib.BF16, 256, 3, 32, 32, 4, 192) == 2
assert lib.vitmi_patch_wide_group(_lib.BF16, _lib.BF16, 256, 3, 20, 20, 4, 192) == 0
assert lib.vitmi_patch_wide_group(_lib.F32, _lib.F32, 256, 3, 32, 32, 4, 192) == 0
assert lib.vitmi_patch_embed_bwd_w_wide_workspace_bytes(256, 3, 32, 32, 4, 192) > 0
lib.vitmi_patch_embed_bwd_w_wide_workspace_bytes(4096, 3, 224, 224, 4, 768)
assert lib.vitmi_patch_embed_fwd_wide(None, None, None, None, 1, 3, 32, 32, 4, 192, 0, None) != 0
assert lib.vitmi_patch_wide_weight(None, None, None, None, 192, 3, 4, None) != 0
assert lib.vitmi_gemm_wgrad_pair(_lib.BF16, 64, 256, 256, None, 256, None, 256, None, 256, 256, None, 256, None, 256, None,
                                 None, None) != 0
assert lib.vitmi_attn_block_infer_workspace_bytes(256, 197, 768, 12, 64) > 0
assert lib.vitmi_attn_block_infer_workspace_bytes(big // 608, 608, 768, 12, 64) > 0
assert lib.vitmi_ff_block_infer_workspace_bytes(50432, 768, 3072) > 0
assert lib.vitmi_ff_block_infer_workspace_bytes(big, 4096, 16384) > 0
assert lib.vitmi_gemm_ln(0, 1, 1, None, 0, None, None, None, None, None, None, 0, 5, None) != 0
assert lib.vitmi_row_stats(None, 0, None, None, _lib.BF16, 1, 1, 1e-5, None) != 0
assert lib.vitmi_gemm_colsum_workspace_bytes(50432, 3072) > 0
assert lib.vitmi_colsum_workspace_bytes(2 ** 40, 4096) > 0
assert lib.vitmi_layernorm_bwd_workspace_bytes(2 ** 40, 1024) > 0
assert lib.vitmi_rollout_workspace_bytes(12, 256, 197) > 0
assert lib.vitmi_tokens_assemble_bwd_workspace_bytes(197, 768) > 0
# the multi-job weight gradient: size query over a host job table (chunks of 16), misaligned job rejected
import ctypes
from visiontransformer_amd.ops import _WgradJob
jobs = (_WgradJob * 20)()
for j in jobs:
    j.U, j.ldu, j.V, j.ldv, j.C, j.ldc, j.P, j.Q = 4096, 512, 8192, 192, 16384, 192, 512, 192
assert lib.vitmi_gemm_wgrad_multi_workspace_bytes(ctypes.addressof(jobs), 20, 16640) > 0
assert lib.vitmi_gemm_wgrad_multi_workspace_bytes(ctypes.addressof(jobs), 20, big) > 0
assert lib.vitmi_gemm_wgrad_multi_workspace_bytes(None, 20, 16640) == 0
jobs[3].P = 100
assert lib.vitmi_gemm_wgrad_multi(ctypes.addressof(jobs), 20, 16640, None, None) != 0 and b"job 3" in lib.vitmi_last_error()
assert lib.vitmi_gemm_wgrad_multi(None, 0, 16640, None, None) == 0
assert lib.vitmi_ff_block_bwd_du_offset(_lib.BF16, 16640, 192, 512) == 0
assert lib.vitmi_attn_block_bwd_dqkv_offset(_lib.BF16, 256, 65, 192, 12, 8) == 256 * 65 * 96 * 2
# the gathered vector sums: where the partial sums sit, how many strips, and the argument checks of the late launch
assert lib.vitmi_ff_block_bwd_sums_offset(_lib.BF16, 16640, 192, 512) == 16640 * 512 * 2 + 16640 * 192 * 2
assert lib.vitmi_ff_block_bwd_sums_strips(_lib.BF16, 16640, 192, 512) > 0
assert lib.vitmi_ff_block_bwd_sums_strips(_lib.BF16, 50432, 768, 3072) == 0
assert lib.vitmi_ff_block_bwd_sums_strips(_lib.F32, 16640, 192, 512) == 0
assert lib.vitmi_attn_block_bwd_sums_offset(_lib.BF16, 256, 65, 192, 12, 8) == 256 * 65 * (96 + 288 + 192) * 2
assert lib.vitmi_attn_block_bwd_sums_strips(_lib.BF16, 256, 65, 192, 12, 8) == 256
assert lib.vitmi_attn_block_bwd_sums_strips(_lib.BF16, 256, 197, 768, 12, 64) == 0
assert lib.vitmi_ff_block_bwd_sums_offset(_lib.BF16, big // 4096, 256, 1024) > 0
from visiontransformer_amd.ops import _SumsJob
sj = (_SumsJob * 40)()
assert lib.vitmi_small_sums_multi(None, 0, None) != 0
assert lib.vitmi_small_sums_multi(ctypes.addressof(sj), 40, None) != 0 and b"job 0" in lib.vitmi_last_error()
assert lib.vitmi_adam_step_matrices(None, None, 1, 1, 1e-3, 0.9, 0.999, 1e-8, 0.0, 1, 1.0, None) != 0
# argument checks: a clean error code and a message, never a crash
assert lib.vitmi_attn_fwd(None, None, None, None, _lib.BF16, 1, 1, 1, 64, 0, None) != 0 and lib.vitmi_last_error()
assert lib.vitmi_attn_bwd(None, None, None, None, None, None, None, _lib.BF16, 1, 1, 1, 64, 0, None) != 0
assert lib.vitmi_set_reserved_cus(7) != 0 and b"multiple of 8" in lib.vitmi_last_error()
assert lib.vitmi_set_reserved_cus(0) == 0
for knob in ("vitmi_debug_set_nt_group", "vitmi_debug_set_nt_tail_split", "vitmi_debug_set_gelu_pair",
             "vitmi_debug_set_attn_head_groups", "vitmi_debug_set_tn_batch", "vitmi_debug_set_small_fused",
             "vitmi_debug_set_wgrad_overlap", "vitmi_debug_set_qkv_planes", "vitmi_debug_set_ghost_norms"):
    getattr(lib, knob)(1)
print("host-sanitized checks ok")
"""


@pytest.mark.skipif(not os.path.exists(LIB), reason="libvitmi_hip_hostsan.so not built (build.py --host-sanitize)")
def test_size_queries_and_argument_checks_under_ubsan():
    r = subprocess.run([sys.executable, "-c", CHILD, LIB, ROOT], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "host-sanitized checks ok" in r.stdout, (r.returncode, r.stdout[-2000:], r.stderr[-2000:])
