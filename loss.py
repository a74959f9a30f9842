"""Losses of the sin-inn training step (drop-in for the reference's loss.py): HIP kernels with autograd.
reconstruction = mean squared error (reference loss.py:3-5), mmd = inverse-multiquadric kernel MMD over the batch
(loss.py:9-36; device-agnostic here -- the reference hard-codes 'cuda'), latent_nll = mean z^2 (loss.py:38-39)."""
from sin_inn_amd.functional import latent_nll, mmd, reconstruction   # noqa: F401
