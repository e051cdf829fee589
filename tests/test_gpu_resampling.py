"""LINEAR resampling strategy: math_utils/resampling.py on the GPU against the reference's known
answers (tests/test_math.py:54-221), the reference's outputs, the oracle, and inside HierarchicalOptimizer3d."""
import numpy as np
import pytest
import torch

from oracle import lsf_oracle as O

pytestmark = pytest.mark.gpu


def test_linear_resampling(ref_literals, ref_leaf):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import levelsetfusion_python_amd as lsf
    from levelsetfusion_python_amd.math_utils import resampling
    T = ref_literals
    for t, (a, u, d) in {"01": ("x", "up_x", "down_x"), "02": ("y", "up_y", "down_y"),
                         "03": ("t", "up_t", "down_t")}.items():
        x, up, down = (T["math.test_resampling%s.%s" % (t, k)] for k in (a, u, d))
        assert np.allclose(resampling.upsample2x_linear(x), up)           # the reference's assertions
        assert np.allclose(resampling.downsample2x_linear(up), down)
    vol = ref_leaf["resampling.vol"]
    assert np.abs(resampling.upsample2x_linear(vol) - ref_leaf["resampling.up"]).max() <= 1e-6
    assert np.abs(resampling.downsample2x_linear(vol) - ref_leaf["resampling.down"]).max() <= 1e-6
    rng = np.random.default_rng(4)
    f = rng.standard_normal((6, 10, 12)).astype(np.float32)
    assert np.array_equal(resampling.upsample2x_linear(f), O.upsample2x_linear(f))
    assert np.array_equal(resampling.downsample2x_linear(f), O.downsample2x_linear(f).astype(np.float32))
    with pytest.raises(ValueError):
        resampling.downsample2x_linear(np.zeros((3, 4, 4), np.float32))
    with pytest.raises(NotImplementedError):
        resampling.upsample2x_linear(np.zeros((4, 4), np.float32))
    # the strategy inside the 3-D optimizer == oracle, bit for bit
    c, l = O.sphere_pair(32, d=3)
    kw = dict(tikhonov_term_enabled=True, gradient_kernel_enabled=False, maximum_chunk_size=4, rate=0.1,
              maximum_iteration_count=3, maximum_warp_update_threshold=0.0, tikhonov_strength=0.05)
    warp = lsf.HierarchicalOptimizer3d(resampling_strategy=lsf.HierarchicalOptimizer3d.ResamplingStrategy.LINEAR,
                                       **kw).optimize(c, l)
    ref = O.HierarchicalOracle(linear_resampling=True, **kw).optimize(c, l)
    assert np.abs(warp - ref).max() == 0.0
    nearest = lsf.HierarchicalOptimizer3d(**kw).optimize(c, l)
    assert np.abs(warp - nearest).max() > 1e-4  # the two strategies really differ
