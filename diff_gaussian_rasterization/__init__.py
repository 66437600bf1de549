"""Drop-in module name for SplatFields: reference gaussian_renderer/__init__.py:14 does
``from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer``.
Putting this repository on ``sys.path`` (instead of pip-installing the CUDA extension of reference
README.md:28) routes that import to the MI355X implementation in ``splatfields_amd``."""
from splatfields_amd.rasterizer import (GaussianRasterizationSettings, GaussianRasterizer,  # noqa: F401
                                        rasterize_gaussians)

__all__ = ["GaussianRasterizationSettings", "GaussianRasterizer", "rasterize_gaussians"]
