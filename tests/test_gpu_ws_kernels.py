"""The weight-stationary kernels (csrc/gemm_ws.cuh: forward, data gradient with BatchNorm backward in the epilogue, data gradient
behind a dropout with the mask and the BatchNorm-backward sums, the projection's twice-computed rank-16 gradient) and the
weight-gradient kernel, each against a plain torch fp32 recomputation of its output from its own stored inputs, at a size
whose last 32-, 48- and 64-row tiles are ragged (975 groups = 39,975 rows), with and without dropout (without: every data
gradient is the BatchNorm-fused kind).  Until round 3 this file compared these kernels element-wise with the tile-staged
kernels they had replaced; those left the product library (tools-only build: make -C csrc variants), so the comparison is with the arithmetic
itself -- tests/test_gpu_fullsize.py's recompute_check, which is what found the "lanes 12-15" faults of round 2 at bench size."""
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dp", [0.0, 0.0635])
def test_ws_kernels_against_fp32_recompute_ragged(dp):
    from test_gpu_fullsize import recompute_check
    recompute_check(975, dp, data_seed=3)
