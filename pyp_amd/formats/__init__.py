"""File formats on the refine3d / reconstruct3d / merge3d call surface."""
from . import cistem, mrc, parfile  # noqa: F401
