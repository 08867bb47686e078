"""vf_nerf_amd — MI355X-native implementation of the VF-NeRF volume-rendering hot path
(``VectorFieldNerf.render``) behind the reference's own Python interface.

Import order matters: ``torch`` first (its HIP runtime is the one ``libvfn.so`` binds to)."""
import torch  # noqa: F401  (must precede the ctypes load of libvfn.so)

from .settings import (CudaConfig, DensityConfig, RaySamplerConfig, RenderingNetConfig, SchedulerConfig,  # noqa: F401
                     VFNerfConfig, VFNetConfig, shipped_config)
from .nerf import VectorFieldNerf  # noqa: F401
from .render_output import NerfOutput  # noqa: F401

__all__ = ["VectorFieldNerf", "NerfOutput", "VFNerfConfig", "VFNetConfig", "RenderingNetConfig", "RaySamplerConfig",
           "CudaConfig", "SchedulerConfig", "DensityConfig", "shipped_config"]
