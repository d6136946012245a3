"""MI355X-native engine for the AR-DAE-VAE inner training loop (reference: ivae_ardae.py:546-846).

Drop-in surface (same names as the reference's `models` / `utils` re-exports used by ivae_ardae.py):
    MNISTIPVAE, ToyIPVAE, ToyAuxIPVAE, ConvIPVAE, MNISTAuxIPVAE, MNISTConvAuxIPVAE, ResConvIPVAE, MNISTResConvAuxIPVAE, MNISTResConvAuxIPVAEClipped, MLPGradCARDAE, MLPResCARDAE, Adam, RMSprop, normal_energy_func, annealing_func
Fused path:
    ArdaeEngine, TrainConfig  -- one train step as a straight line of C-ABI calls (what bench.py times)
    ScalarLog                 -- the reference's per-step scalars through a device ring buffer (no host sync in the step)
The compute is libardae_hip.so (hand-written HIP for gfx950, C ABI in include/ardae_hip.h); there is no fallback.
"""
from . import _lib  # noqa: F401
from . import rng  # noqa: F401
from . import data  # noqa: F401
from .rng import manual_seed  # noqa: F401
from .modules import (MNISTIPVAE, ToyIPVAE, ToyAuxIPVAE, ConvIPVAE, MNISTAuxIPVAE, MNISTConvAuxIPVAE, ResConvIPVAE, MNISTResConvAuxIPVAE, MNISTResConvAuxIPVAEClipped, MLPGradCARDAE, MLPResCARDAE, ImplicitPosteriorVAE, ConditionalARDAE,  # noqa: F401
                      normal_energy_func)
from .optim import Adam, RMSprop  # noqa: F401
from .engine import ArdaeEngine, TrainConfig, annealing_func  # noqa: F401
from .scalar_log import ScalarLog  # noqa: F401
