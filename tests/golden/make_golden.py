"""Generate the golden vectors by running the UNMODIFIED reference (build container only).

    python tests/golden/make_golden.py              (--windows: only the Pips(S != 8) cases)

For every case of cases.py: load seeded weights into /root/reference/nets/pips.py's Pips,
run forward on the seeded inputs, store coord_predictions / vis_e / ffeat as float32 in
tests/golden/<case>.npz.  Also writes state_dict_keys.json (names + shapes of the
reference state dict) and demo_half_frames.npz (first 8 demo frames, PIL-resized to
320x180, the only stored input).
"""
from __future__ import annotations

import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

from oracle import reference_shim as R            # noqa: E402
from pips_amd.weights import init_state_dict      # noqa: E402
import cases as G                                 # noqa: E402


def make_demo_frames():
    from PIL import Image
    d = os.path.join(R.REFERENCE_ROOT, "demo_images")
    names = sorted(f for f in os.listdir(d) if f.endswith(".jpg"))[:8]
    frames = [np.asarray(Image.open(os.path.join(d, n)).convert("RGB").resize((320, 180), Image.BILINEAR)) for n in names]
    np.savez_compressed(os.path.join(HERE, "demo_half_frames.npz"), frames=np.stack(frames).astype(np.uint8))
    # BASELINE configs[0]: the same eight frames at the size demo.py:24-27 runs them (640x360 on disk = H_, W_: its resize is the identity)
    full = [np.asarray(Image.open(os.path.join(d, n)).convert("RGB")) for n in names]
    assert full[0].shape == (360, 640, 3)
    np.savez_compressed(os.path.join(HERE, "demo_full_frames.npz"), frames=np.stack(full).astype(np.uint8))


def run_case(name, case):
    """one forward of the unmodified reference -> tests/golden/<name>.npz; returns the reference module"""
    S = case.get("S", 8)
    sd = init_state_dict(0, S=S, tamed=case["tamed"])
    xys, rgbs, ci, fi = G.make_inputs(case)
    ref = R.load_reference_pips(sd, stride=case["stride"], S=S)
    with torch.no_grad():
        preds, preds2, vis, ffeat, losses = ref(xys, rgbs, coords_init=ci, feat_init=fi, iters=case["iters"],
                                                return_feat=True)
    assert losses is None and len(preds2) == case["iters"] + 4
    np.savez_compressed(os.path.join(HERE, name + ".npz"),
                        trajs=torch.stack(preds).numpy().astype(np.float32),
                        traj0=preds2[0].numpy().astype(np.float32),
                        vis=vis.numpy().astype(np.float32),
                        ffeat=ffeat.numpy().astype(np.float32))
    print(name, "trajs", tuple(torch.stack(preds).shape), "max|disp| px",
          float((preds[-1] - preds2[0]).abs().max()))
    return ref


def main():
    assert R.available(), "reference not mounted at /root/reference"
    make_demo_frames()
    keys = None
    for name, case in G.CASES.items():
        ref = run_case(name, case)
        if keys is None:
            keys = {k: list(v.shape) for k, v in ref.state_dict().items()}
    with open(os.path.join(HERE, "state_dict_keys.json"), "w") as f:
        json.dump(keys, f, indent=0)


def main_windows():
    """Pips(S != 8) cases (cases.WINDOW_CASES)"""
    assert R.available(), "reference not mounted at /root/reference"
    for name, case in G.WINDOW_CASES.items():
        run_case(name, case)


def main_losses(name="s8_raw_i3"):
    """(seq_loss, vis_loss, ce_loss) of the reference forward called with trajs_g / vis_g / valids
    (nets/pips.py:600-606, the way test_on_flt.py:87 calls it) -> <case>_losses.npz."""
    assert R.available(), "reference not mounted at /root/reference"
    case = G.CASES[name] if name in G.CASES else G.WINDOW_CASES[name]
    S = case.get("S", 8)
    sd = init_state_dict(0, S=S, tamed=case["tamed"])
    xys, rgbs, ci, fi = G.make_inputs(case)
    trajs_g, vis_g, valids = G.make_targets(case)
    ref = R.load_reference_pips(sd, stride=case["stride"], S=S)
    with torch.no_grad():
        out = ref(xys, rgbs, coords_init=ci, feat_init=fi, iters=case["iters"], trajs_g=trajs_g, vis_g=vis_g, valids=valids)
    seq, vis, ce = out[3]
    np.savez_compressed(os.path.join(HERE, name + "_losses.npz"), seq_loss=np.float32(seq), vis_loss=np.float32(vis),
                        ce_loss=np.float32(ce))
    print(name, "losses: seq", float(seq), "vis", float(vis), "ce", float(ce))


if __name__ == "__main__":
    if "--demo-full" in sys.argv:            # only BASELINE configs[0] at its own size
        assert R.available(), "reference not mounted at /root/reference"
        make_demo_frames()
        run_case("demo_full_s4_i2", G.CASES["demo_full_s4_i2"])
        sys.exit(0)
    if "--window" in sys.argv:               # ONE Pips(S != 8) fixture by name (the others stay byte for byte what they are)
        assert R.available(), "reference not mounted at /root/reference"
        nm = sys.argv[sys.argv.index("--window") + 1]
        run_case(nm, G.WINDOW_CASES[nm])
        sys.exit(0)
    if "--windows" in sys.argv:              # only the Pips(S != 8) fixtures
        main_windows()
        main_losses("w5_tamed_i3")
        sys.exit(0)
    if "--losses" not in sys.argv:
        main()
        main_windows()
    main_losses("s8_raw_i3")
    main_losses("s8_tamed_i6")
    main_losses("w5_tamed_i3")
