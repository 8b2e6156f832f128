"""rald_amd - MI355X-native (gfx950) implementation of RaLD's hot path: the radar-conditioned
latent denoiser + EDM/Heun sampler and the set-latent autoencoder encode/decode, as
hand-written HIP kernels behind a C-ABI library (include/rald_hip.h), wrapped by modules that
keep the reference's names, signatures and checkpoint keys.

Importing the package is cheap and GPU-free (specs, seeded weights, synthetic inputs);
the HIP library is loaded on first use by ``rald_amd._lib`` and fails loudly when missing.
"""
__version__ = "0.1.0"
