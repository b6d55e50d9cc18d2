"""Import-time placeholder for `nerfacc` (absent from this image).

The reference's ray_samplers.py / renderers.py import these names at module top but never execute
them on the nerfacto / SAM path (they serve the occupancy-grid VolumetricSampler only).
Used ONLY by tests/golden/make_golden.py in the build container.
"""


class OccupancyGrid:  # noqa: D101
    pass


class ContractionType:  # noqa: D101
    pass


def _not_on_this_path(*args, **kwargs):
    raise NotImplementedError("nerfacc is not executed on the nerfacto/SAM hot path")


ray_marching = accumulate_along_rays = contract = _not_on_this_path
