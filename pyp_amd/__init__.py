"""pyp_amd — MI355X-native projection matching + Fourier reconstruction behind PYP's
refine3d / reconstruct3d / merge3d call surface."""
__version__ = "0.1.0"
