"""pips_amd -- MI355X-native (gfx950) inference path of the PIPs point tracker.

``from pips_amd import Pips`` is the drop-in for ``from nets.pips import Pips``
(reference nets/pips.py:400-611); all arithmetic runs in libpips_hip.so behind the C ABI
declared in include/pips_hip.h.
"""
from .pips import Pips, FeatureCache  # noqa: F401
from ._lib import PipsHipError  # noqa: F401

__all__ = ["Pips", "FeatureCache", "PipsHipError"]
