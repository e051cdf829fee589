"""Per-iteration telemetry of the hierarchical optimizer: the reference's own known answer for the
level-3 / iteration-50 warp field (tests/test_hierarchical_optimizer2d.py:71-102), per-iteration warps captured from the
reference, and the telemetry_log.npz round trip."""
import numpy as np
import pytest
import torch

from oracle import lsf_oracle as O

pytestmark = pytest.mark.gpu


def test_iteration_data_known_answer_and_roundtrip(ref_literals, ref_hierarchical, tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import levelsetfusion_python_amd as lsf
    from levelsetfusion_python_amd.nonrigid_opt.hierarchical import telemetry
    T, H = ref_literals, ref_hierarchical
    canon, live = T["hierarchical_data.canonical_field"], T["hierarchical_data.live_field"]
    logging = lsf.HierarchicalOptimizer2d.LoggingParameters(collect_per_level_convergence_reports=True,
                                                            collect_per_level_iteration_data=True)
    opt = lsf.HierarchicalOptimizer2d(tikhonov_term_enabled=False, gradient_kernel_enabled=False, maximum_chunk_size=8,
                                      rate=0.2, maximum_iteration_count=100, maximum_warp_update_threshold=0.001,
                                      data_term_amplifier=1.0, tikhonov_strength=0.0,
                                      kernel=lsf.generate_1d_sobolev_kernel(size=7, strength=0.1),
                                      logging_parameters=logging)
    warp = opt.optimize(canon, live)
    data = opt.get_per_level_iteration_data()
    assert [d.get_frame_count() for d in data] == [1, 100, 100, 100]
    vec = data[3].get_warp_fields()
    assert np.allclose(vec[50], T["hierarchical_data.iteration50_warp_field"], atol=1e-6)  # the reference's check
    assert np.array_equal(vec[-1], warp)
    assert np.allclose(warp, T["hierarchical_data.warp_field"], atol=1e-5)
    assert len(opt.get_per_level_convergence_reports()) == 4
    # Tikhonov run: every per-iteration warp equals the reference's capture; gradient snapshots equal the oracle's terms
    kw = dict(tikhonov_term_enabled=True, gradient_kernel_enabled=False, maximum_chunk_size=8, rate=0.2,
              maximum_iteration_count=4, maximum_warp_update_threshold=0.0, tikhonov_strength=0.2)
    opt = lsf.HierarchicalOptimizer2d(logging_parameters=logging, check_interval=3, **kw)
    opt.optimize(canon, live)
    data = opt.get_per_level_iteration_data()
    for level, d in enumerate(data):
        assert d.get_frame_count() == 4 and len(d.get_tikhonov_term_gradients()) == 4
        for it in range(4):
            key = "g16.tik1.ker0.chunk8.L%d.it%d.warp" % (level, it)
            assert np.abs(d.get_warp_fields()[it] - H[key]).max() == 0.0
    o = O.HierarchicalOracle(**kw)
    seen = {}
    o.iteration_hook = lambda level, it, w, g, m: seen.__setitem__((level, it), g.copy())
    o.optimize(canon, live)
    lvl = 3
    for it in range(1, 4):  # gradient_it = data_it - 0.2 * laplace(gradient_{it-1});  snapshots hold both terms
        dgrad, tgrad = data[lvl].get_data_term_gradients()[it], data[lvl].get_tikhonov_term_gradients()[it]
        assert np.abs(tgrad[..., 0] - O.laplace_replicate(seen[(lvl, it - 1)][..., 0])).max() == 0.0
        assert np.abs((np.float32(1.0) * dgrad - np.float32(0.2) * tgrad) - seen[(lvl, it)]).max() == 0.0
    # npz round trip
    telemetry.save_telemetry_log(data, str(tmp_path))
    back = telemetry.load_telemetry_log(str(tmp_path))
    assert len(back) == 4 and back[2].get_frame_count() == 4
    assert np.array_equal(back[3].get_warp_fields()[2], data[3].get_warp_fields()[2])
    assert np.array_equal(back[1].get_tikhonov_term_gradients()[3], data[1].get_tikhonov_term_gradients()[3])
