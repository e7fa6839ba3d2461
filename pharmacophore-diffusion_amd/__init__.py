"""pharmacophore-diffusion_amd: MI355X-native denoising hot path of PharmacoForge.

The directory name is not a valid Python identifier; import it as ``pharmacoforge_amd`` (a tiny
alias module at the repository root) or via ``importlib.import_module('pharmacophore-diffusion_amd')``.

Contents: csrc/ (HIP kernels + the C ABI of include/pfdyn.h), a ctypes binding, and the host-side
mirror of the reference's Python interface for this path (PharmRecDynamicsGVP, PharmacophoreDiff,
sample_given_receptor / sample, SampledPharmacophore)."""
from . import _lib, schedule, synthetic
from .engine import PfEngine
from ._lib import PfError

__all__ = ["PfEngine", "PfError", "_lib"]

try:  # host-side mirror of the reference API (needs only torch)
    from .graph import PocketGraph, batch, unbatch, build_initial_complex_graph, copy_graph  # noqa: F401
    from .models import (GVP, GVPLayerNorm, GVPMultiEdgeConv, NoisePredictionBlock, PharmRecGVP,  # noqa: F401
                         PharmRecDynamicsGVP, PharmacophoreDiff, PredefinedNoiseSchedule, PharmSizeDistribution,
                         FlatAdam, model_from_config)
    from .analysis import SampledPharmacophore, SampleAnalyzer, write_pharmacophore_file  # noqa: F401
    __all__ += ["PocketGraph", "batch", "unbatch", "build_initial_complex_graph", "copy_graph", "PharmRecDynamicsGVP",
                "PharmacophoreDiff", "FlatAdam", "PredefinedNoiseSchedule", "SampledPharmacophore", "SampleAnalyzer",
                "model_from_config"]
except ImportError as _e:   # pragma: no cover  (only while the package is being bootstrapped)
    _host_api_error = _e
