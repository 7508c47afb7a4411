"""im360_oracle -- CPU restatement of the Imagine360 dual-branch denoising hot path.

TEST INFRASTRUCTURE, NOT PRODUCT.  Plain torch-CPU fp32, functional style, operating on a
flat ``state_dict`` (name -> tensor) that uses the reference's parameter names.  Only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this package; ``imagine360_amd`` never does.

Parity pin: every function here was checked in the authoring container against the real
reference imported from /root/reference (oracle/tools/ref_shims.py +
oracle/tools/gen_golden.py) and against the fixtures that script committed under
tests/golden/.  Arithmetic that the reference delegates to un-vendored third parties
(xformers 0.0.28.post1 attention, kornia remap / gaussian_blur2d / create_meshgrid,
cv2.Rodrigues) is restated from the documented semantics -- that part is "parity
unpinned" by the reference itself.

Each function cites the reference file:line it follows (paths relative to the reference
repository root).
"""
from .cfg import UNetCfg, VAECfg, sd21_unet_cfg, sd21_vae_cfg  # noqa: F401
