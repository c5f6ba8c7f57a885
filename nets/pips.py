"""``from nets.pips import Pips`` -- the reference's import path (nets/pips.py:400), served by the HIP path.

Put this repository ahead of the reference checkout on ``sys.path`` and the unmodified callers pick up
``pips_amd.Pips``: same constructor, state dict, forward signature and return tuple."""
from pips_amd.pips import Pips, FeatureCache          # noqa: F401
from pips_amd.pips import sequence_loss, balanced_ce_loss   # noqa: F401  (nets/pips.py:14-56, evaluation only)
