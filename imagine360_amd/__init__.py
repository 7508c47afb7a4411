"""imagine360_amd -- MI355X-native (gfx950) implementation of Imagine360's dual-branch
denoising hot path behind the reference's Python API.  See DESIGN.md."""
__version__ = "0.1.0"
