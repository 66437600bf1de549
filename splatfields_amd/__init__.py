"""MI355X-native differentiable Gaussian-splat rasterizer behind SplatFields' render() boundary."""
from .rasterizer import GaussianRasterizationSettings, GaussianRasterizer, rasterize_gaussians  # noqa: F401
from .render import render  # noqa: F401
