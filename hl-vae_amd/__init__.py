"""hlvae_amd -- MI355X-native HL-VAE ELBO training hot path (hand-written HIP behind a C ABI).

Only what the hot path needs lives here: the drop-in ``HLVAE`` class, the
column plan, the GP-prior KL, the training step and the data-parallel glue.  The compute is in
``csrc/`` (gfx950 HIP kernels, exported through include/hlvae_hip.h); this package never falls
back to a CPU implementation.
"""
from . import layout, synthetic  # noqa: F401

__all__ = ["layout", "synthetic"]
