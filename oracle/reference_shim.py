"""Import the UNMODIFIED reference model from /root/reference (build container only).

TEST INFRASTRUCTURE.  /root/reference does not exist on the GPU box, so nothing that
runs there (``-m gpu`` tests, smoke(), bench.py) may import this module; it is used by
``oracle/check_against_reference.py`` and ``tests/golden/make_golden.py`` to pin the
oracle and to produce the committed golden vectors.
"""
from __future__ import annotations

import os
import sys

REFERENCE_ROOT = "/root/reference"


def available() -> bool:
    return os.path.isfile(os.path.join(REFERENCE_ROOT, "nets", "pips.py"))


def reference_module():
    """The unmodified ``nets/pips.py`` as a module (its functions: score_map_loss, balanced_ce_loss, ...)."""
    load = load_reference_pips.__globals__["_load_module"]
    return load()


def _load_module():
    import torch

    sys.dont_write_bytecode = True            # the mount is read-only by contract
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    if not torch.cuda.is_available():
        # nets/pips.py:429 does torch.tensor(0.0).cuda() (dead value); without a GPU make
        # .cuda() the identity for the duration of this process.
        torch.Tensor.cuda = lambda self, *a, **k: self
    # by file path: this repository ships its own ``nets/pips.py`` (the drop-in import path; ``nets`` is a namespace
    # package on both sides), which wins the name ``nets.pips`` whenever the repository is ahead on sys.path
    import importlib.util
    name = "_reference_nets_pips"
    if name not in sys.modules:
        spec = importlib.util.spec_from_file_location(name, os.path.join(REFERENCE_ROOT, "nets", "pips.py"))
        mod = importlib.util.module_from_spec(spec)
        sys.modules[name] = mod
        spec.loader.exec_module(mod)
    return sys.modules[name]


def load_reference_pips(sd, stride=8, S=8):
    """Instantiate ``nets.pips.Pips`` and load ``sd`` into it (strict)."""
    Pips = _load_module().Pips
    m = Pips(S=S, stride=stride).eval()
    missing = m.load_state_dict(sd, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    return m
