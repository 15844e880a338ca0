"""sushi_amd -- MI355X-native audio template matching for tp7/Sushi.

Drop-in for the one compute-heavy path of Sushi: ``wav.WavStream.find_substream``
(reference wav.py:177-188, called from sushi.py:432,450-452,460-462), plus a batched form
of the same operation.  The matching runs as hand-written HIP kernels (gfx950) behind the
C ABI in ``include/sushi_hip.h``; there is no CPU fallback.
"""
from .common import SushiError, clip  # noqa: F401

__version__ = "0.1.0"
