"""Pin the oracle restatement to the real reference (run in the build container).

    python oracle/check_against_reference.py

Runs /root/reference/nets/pips.py (unmodified) and oracle/pips_oracle.py on the same
seeded weights/inputs for several shapes and prints max |difference| per output.
TEST INFRASTRUCTURE.
"""
from __future__ import annotations

import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from oracle import pips_oracle as O            # noqa: E402
from oracle import reference_shim as R         # noqa: E402
from pips_amd.weights import init_state_dict   # noqa: E402


def make_inputs(B, N, H, W, seed=1, S=8):
    g = torch.Generator().manual_seed(seed)
    rgbs = torch.randint(0, 256, (B, S, 3, H, W), generator=g).float()
    xys = torch.rand(B, N, 2, generator=g) * torch.tensor([W - 1.0, H - 1.0])
    return xys, rgbs


def main():
    assert R.available(), "reference not mounted"
    worst = 0.0
    for (B, N, H, W, stride, iters, tamed) in [
        (1, 16, 128, 160, 8, 3, False),
        (2, 5, 136, 200, 8, 2, True),
        (1, 16, 96, 128, 4, 2, False),
    ]:
        sd = init_state_dict(0, tamed=tamed)
        xys, rgbs = make_inputs(B, N, H, W)
        ref = R.load_reference_pips(sd, stride=stride)
        with torch.no_grad():
            p, p2, vis, ff, _ = ref(xys, rgbs, iters=iters, return_feat=True)
        q, q2, qvis, qff = O.forward(sd, xys, rgbs, iters=iters, stride=stride)
        d = max((a - b).abs().max().item() for a, b in zip(p + p2 + [vis, ff], q + q2 + [qvis, qff]))
        print(f"B={B} N={N} {H}x{W} stride={stride} iters={iters} tamed={tamed}: max|ref-oracle| = {d:.3e}")
        worst = max(worst, d)
        # feat_init / coords_init path (chain_demo.py:54)
        ci = p[-1] + 0.25
        with torch.no_grad():
            p, p2, vis, ff, _ = ref(xys, rgbs, iters=1, coords_init=ci, feat_init=ff, return_feat=True)
        q, q2, qvis, qff = O.forward(sd, xys, rgbs, iters=1, stride=stride, coords_init=ci, feat_init=ff)
        d = max((a - b).abs().max().item() for a, b in zip(p + p2 + [vis, ff], q + q2 + [qvis, qff]))
        print(f"   with coords_init/feat_init: {d:.3e}")
        worst = max(worst, d)
    print("worst", worst)
    return 0 if worst == 0.0 or worst < 1e-5 else 1


if __name__ == "__main__":
    sys.exit(main())
