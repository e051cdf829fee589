"""Multi-pair driver on the GPU: independent TSDF pairs -> HierarchicalOptimizer -> per-level convergence reports ->
table / analysis files (reference run_hierarchical_optimizer3d_multipair.py:403-441)."""
import numpy as np
import pytest
import torch

from oracle import lsf_oracle as O

pytestmark = pytest.mark.gpu


def test_multipair_run(tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import levelsetfusion_python_amd as lsf
    from levelsetfusion_python_amd.experiment import multipair as mp
    from levelsetfusion_python_amd.nonrigid_opt import field_warping as fw
    pairs_dir = str(tmp_path / "pairs")
    base_c, base_l = O.sphere_pair(32, d=2)
    for k in range(3):
        mp.save_pair(pairs_dir, 10 + k, 240, base_c, np.roll(base_l, k, axis=1))
    pairs = mp.load_pairs(pairs_dir)
    assert len(pairs) == 3
    kw = dict(tikhonov_term_enabled=False, gradient_kernel_enabled=False, maximum_chunk_size=4, rate=0.2,
              maximum_iteration_count=30, maximum_warp_update_threshold=0.05)
    opt = lsf.HierarchicalOptimizer2d(
        logging_parameters=lsf.HierarchicalOptimizer2d.LoggingParameters(collect_per_level_convergence_reports=True),
        **kw)
    df = mp.run_experiment(opt, pairs, str(tmp_path / "out"))
    assert len(df) == 3 and len(df.columns) == 2 + 17 * 3
    assert list(df["canonical_frame"]) == [10, 11, 12]
    for name in ("convergence_reports.csv", "convergence_reports.pkl", "analysis.txt", "bad_cases.csv",
                 "all_cases.csv"):
        assert (tmp_path / "out" / name).exists()
    # the table's numbers against the oracle for the last pair
    o = O.HierarchicalOracle(**kw)
    warp = o.optimize(pairs[2][2], pairs[2][3])
    assert [int(df["l%d_iter_count" % i][2]) for i in range(3)] == o.per_level_iteration_counts
    assert [bool(df["l%d_iter_lim_reached" % i][2]) for i in range(3)] == \
        [c >= 30 for c in o.per_level_iteration_counts]
    assert np.isclose(df["l2_warp_delta_max"][2], o.per_level_max_updates[2][-1], atol=1e-7)
    resampled = O.warp_field(pairs[2][3], warp)
    diff = np.abs(pairs[2][2].astype(np.float64) - resampled)
    assert np.isclose(df["l2_diff_delta_max"][2], diff.max(), atol=1e-6)
    assert np.isclose(df["l2_diff_delta_mean"][2], diff.mean(), atol=1e-7)
    at = np.unravel_index(int(np.argmax(diff)), diff.shape)
    assert (int(df["l2_diff_max_x"][2]), int(df["l2_diff_max_y"][2])) == (at[1], at[0])
    # an optimizer without report collection is rejected loudly
    with pytest.raises(ValueError):
        mp.run_pairs(lsf.HierarchicalOptimizer2d(**kw), pairs[:1])


def test_pairs_in_flight_give_the_same_table(tmp_path):
    """run_pairs with a sequence of optimizers: two / three pairs in flight at once (a thread and a stream per optimizer)
    -- the same report table as one optimizer pair after pair, 2-D and 3-D (the reference's loop is embarrassingly
    parallel: run_hierarchical_optimizer3d_multipair.py:403-432)"""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import levelsetfusion_python_amd as lsf
    from levelsetfusion_python_amd.experiment import multipair as mp
    for d, n, cls, extra in ((2, 64, lsf.HierarchicalOptimizer2d, {}),
                             (3, 32, lsf.HierarchicalOptimizer3d, {})):
        base_c, base_l = O.sphere_pair(n, d=d)
        pairs = [(100 + k, 7, base_c, np.roll(base_l, k % 4, axis=d - 1)) for k in range(7)]
        kw = dict(tikhonov_term_enabled=True, tikhonov_strength=0.2, gradient_kernel_enabled=False, maximum_chunk_size=4,
                  rate=0.2, maximum_iteration_count=25, maximum_warp_update_threshold=0.02, **extra)

        def make():
            return cls(logging_parameters=cls.LoggingParameters(collect_per_level_convergence_reports=True), **kw)
        one = mp.post_process_convergence_report_sets(*mp.run_pairs(make(), pairs))
        seen = []
        for lanes in (2, 3):
            many = mp.post_process_convergence_report_sets(
                *mp.run_pairs([make() for _ in range(lanes)], pairs, progress=lambda k, total: seen.append(k)))
            # the two standard deviations come out of float64 sums whose order varies from launch to launch (1e-14,
            # also between two runs of ONE optimizer); every other column is exact
            loose = [c for c in one.columns if c.endswith("_std")]
            exact = [c for c in one.columns if c not in loose]
            assert many[exact].equals(one[exact]), "pairs in flight changed the table (%d-D, %d lanes)" % (d, lanes)
            assert np.allclose(many[loose].to_numpy(dtype=np.float64), one[loose].to_numpy(dtype=np.float64),
                               rtol=1e-11, atol=1e-13)
        assert sorted(seen) == sorted(list(range(7)) * 2)
    # a failure inside a lane reaches the caller
    with pytest.raises(ValueError):
        mp.run_pairs([lsf.HierarchicalOptimizer2d(**{k: v for k, v in kw.items()}) for _ in range(2)],
                     [(1, 1) + O.sphere_pair(64, d=2)] * 3)
