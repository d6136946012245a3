"""MI355X-native engine for the AR-DAE-VAE inner training loop (reference: ivae_ardae.py:546-846)."""
from . import _lib  # noqa: F401
