"""Import harness for the *reference* (Algomorph/LevelSetFusion-Python) -- used ONLY in the build
container, ONLY by tests/golden/make_golden.py, to generate golden vectors.

The reference's Python hot path imports five modules that are absent in this image (cv2, sktensor,
progressbar, lxml, and its own un-vendored C++ extension `level_set_fusion_optimization`).  This module
registers inert stand-ins for them in sys.modules *before* the reference is imported, so that the
reference's own numpy code runs unmodified.  None of the stand-ins compute anything that ends up in a
golden vector, with one exception that is a pure delegation to the reference's own Python twin
(`warp_field_advanced`, see below).

Nothing in here travels to the GPU box as a dependency of a test: the `-m gpu` tests, smoke() and
bench.py never import this file (it needs /root/reference, which does not exist there).
"""
import enum
import os
import sys
import types

import numpy as np

REFERENCE_ROOT = os.environ.get("LSF_REFERENCE_ROOT", "/root/reference")


def _module(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


class _Bag:
    """attribute bag accepting any ctor arguments (used for report/parameter structs)"""

    def __init__(self, *args, **kwargs):
        self.args = args
        self.__dict__.update(kwargs)

    def __eq__(self, other):
        return True


def install():
    if "level_set_fusion_optimization" in sys.modules:
        return
    os.environ.setdefault("MPLBACKEND", "Agg")
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)

    # ---- cv2: only import-time names + a PIL-backed imread ------------------------------------------
    class _VideoWriter:
        def __init__(self, *a, **k):
            pass

        def write(self, *a, **k):
            pass

        def release(self):
            pass

    def _imread(path, flags=None):
        from PIL import Image
        return np.array(Image.open(path))

    _module("cv2", VideoWriter=_VideoWriter, VideoWriter_fourcc=lambda *a: 0,
            cvtColor=lambda img, code: img, putText=lambda *a, **k: None, imread=_imread,
            imwrite=lambda *a, **k: True, resize=lambda img, *a, **k: img,
            IMREAD_UNCHANGED=-1, COLOR_GRAY2BGR=0, COLOR_BGR2GRAY=1, COLOR_RGB2BGR=2, COLOR_BGR2RGB=3,
            FONT_HERSHEY_PLAIN=0, FONT_HERSHEY_SIMPLEX=0, INTER_LINEAR=1, INTER_NEAREST=0, remap=None,
            Rodrigues=None, BORDER_REPLICATE=1, GaussianBlur=None, LINE_AA=16)

    # ---- sktensor: only the rank-0 (plain SVD) branch of tucker.hooi is exercised --------------------
    def _ttm(t, mat, mode, transp=False, without=False):
        m = mat.T if transp else mat
        return np.moveaxis(np.tensordot(m, np.asarray(t), axes=(1, mode)), 0, mode)

    sk = _module("sktensor", dtensor=np.asarray)
    core = _module("sktensor.core", tensor_mixin=object, ttm=_ttm, nvecs=None,
                   norm=lambda x: float(np.linalg.norm(np.asarray(x))))
    sk.core = core
    pyu = _module("sktensor.pyutils", is_number=lambda x: isinstance(x, (int, float, np.integer, np.floating)))
    sk.pyutils = pyu

    _module("progressbar")
    lx = _module("lxml")
    lx.etree = _module("lxml.etree")

    # ---- the un-vendored C++ extension ------------------------------------------------------------------
    class FilteringMethod(enum.Enum):
        NONE = 0
        BILINEAR_IMAGE_SPACE = 1
        BILINEAR_VOXEL_SPACE = 2
        EWA_IMAGE_SPACE = 3
        EWA_VOXEL_SPACE = 4
        EWA_VOXEL_SPACE_INCLUSIVE = 5

    tsdf_ns = types.SimpleNamespace(FilteringMethod=FilteringMethod, Parameters2d=_Bag, Parameters3d=_Bag,
                                    Generator2d=_Bag, Generator3d=_Bag)

    class _CppHO2d(_Bag):
        VerbosityParameters = _Bag
        LoggingParameters = _Bag

        class ResamplingStrategy(enum.Enum):
            NEAREST_AND_AVERAGE = 0
            LINEAR = 1

    captured = {}

    def _warp_field_advanced(warped_live, canonical, u, v, band_union_only=False, known_values_only=False,
                             substitute_original=False):
        # pure delegation to the reference's own Python twin (nonrigid_opt/field_warping.py:112), with the
        # argument reorder of slavcheva_optimizer2d.py:224-234 vs :324-327
        from nonrigid_opt import field_warping
        warp = np.stack((u, v), axis=2)
        grad = np.zeros_like(warp)
        new_live = field_warping.warp_field_advanced(canonical, warped_live, warp, grad, band_union_only,
                                                     known_values_only, substitute_original)
        return new_live, (warp[:, :, 0].copy(), warp[:, :, 1].copy())

    def _build_warp_stats(*args):
        captured["warp_stats_args"] = [np.array(a, copy=True) if isinstance(a, np.ndarray) else a for a in args]
        return _Bag(*args)

    def _build_tsdf_stats(*args):
        captured["tsdf_stats_args"] = [np.array(a, copy=True) if isinstance(a, np.ndarray) else a for a in args]
        return _Bag(*args)

    _module("level_set_fusion_optimization", tsdf=tsdf_ns, Vector2i=_Bag, Vector3i=_Bag,
            ConvergenceReport2d=_Bag, WarpDeltaStatistics2d=_Bag, TsdfDifferenceStatistics2d=_Bag,
            data_term_at_location=None, HierarchicalOptimizer2d=_CppHO2d, HierarchicalOptimizer3d=_CppHO2d,
            warp_field_advanced=_warp_field_advanced, build_warp_delta_statistics_2d=_build_warp_stats,
            build_tsdf_difference_statistics_2d=_build_tsdf_stats, captured=captured)

    # ---- silence the reference's always-on visualisers (I/O side effects only) ------------------------
    import nonrigid_opt.slavcheva.slavcheva_visualizer as sviz

    for name in ("make_vector_field_plot", "warp_field_to_heatmap", "make_3d_plots"):
        if hasattr(sviz, name):
            setattr(sviz, name, lambda *a, **k: np.zeros((4, 4, 3), dtype=np.uint8))
    for name in [n for n in dir(sviz.SlavchevaVisualizer) if n.startswith("write_")]:
        setattr(sviz.SlavchevaVisualizer, name, lambda self, *a, **k: None)
