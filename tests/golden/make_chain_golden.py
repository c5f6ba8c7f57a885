"""Pin the chaining loop to the reference's OWN TEXT (build container only).

    python tests/golden/make_chain_golden.py

chain_demo.py cannot be imported here (cv2 / imageio / tensorboardX are absent), but its loop can be executed: this
script reads /root/reference/chain_demo.py, takes lines 39-83 -- from ``trajs_e = torch.zeros(...)`` to
``trajs_e[:,:,n] = traj_e``, the whole per-keypoint window/hop loop -- VERBATIM, rewrites only ``device='cuda'`` to
``device='cpu'``, and ``exec``s that text with the unmodified reference ``nets.pips.Pips`` (oracle/reference_shim.py)
as ``model`` on the seeded video of cases.CHAIN_CASE.  The model is wrapped only to record which window each call
received (the hop sequence).  Output: tests/golden/chain_t13.npz = the reference loop's ``trajs_e`` and hops, which
tests/test_oracle_golden.py holds oracle/chain_oracle.py to.
"""
from __future__ import annotations

import os
import sys
import textwrap

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

from oracle import reference_shim as R            # noqa: E402
from pips_amd.weights import init_state_dict      # noqa: E402
import cases as G                                 # noqa: E402

FIRST, LAST = 39, 83                              # chain_demo.py line numbers (1-based, inclusive)


def loop_text():
    lines = open(os.path.join(R.REFERENCE_ROOT, "chain_demo.py")).read().split("\n")
    body = lines[FIRST - 1:LAST]
    assert body[0].strip().startswith("trajs_e = torch.zeros((B, S, N, 2)") and body[-1].strip() == "trajs_e[:,:,n] = traj_e", \
        "chain_demo.py moved: re-check the slice"
    text = textwrap.dedent("\n".join(body))
    assert text.count("device='cuda'") == 2
    return text.replace("device='cuda'", "device='cpu'")


def main():
    assert R.available(), "reference not mounted at /root/reference"
    case = G.CHAIN_CASE
    assert case["iters"] == 6                      # the reference text calls model(..., iters=6, ...)
    video, xy0 = G.make_chain_inputs(case)
    ref = R.load_reference_pips(init_state_dict(0, tamed=case["tamed"]), stride=case["stride"])
    starts = []

    def model(xy, rgb_seq, **kw):                  # records the window's first frame, then the real forward
        t = [i for i in range(video.shape[1]) if torch.equal(rgb_seq[0, 0], video[0, i])]
        starts.append((len(t) and t[0], float(xy[0, 0, 0]), float(xy[0, 0, 1])))
        with torch.no_grad():
            return ref(xy, rgb_seq, **kw)

    B, S, N = 1, case["T"], case["N"]
    ns = {"torch": torch, "model": model, "rgbs": video, "xy0": xy0, "B": B, "S": S, "N": N}
    exec(compile(loop_text(), "chain_demo.py[39:83]", "exec"), ns)
    trajs_e = ns["trajs_e"]
    # hop sequence per particle: a call whose window starts at frame 0 opens a new particle
    hops, cur = [], None
    for (t0, _, _) in starts:
        if t0 == 0:
            cur = [0]
            hops.append(cur)
        else:
            cur.append(t0)
    hop_steps = []
    for seq, n in zip(hops, range(N)):
        # window starts -> the si of every hop but the last (which left the video: not recoverable from starts alone)
        hop_steps.append([b - a for a, b in zip(seq[:-1], seq[1:])])
    flat = np.array([s for h in hop_steps for s in h] or [0], dtype=np.int32)
    cnt = np.array([len(h) for h in hop_steps], dtype=np.int32)
    np.savez_compressed(os.path.join(HERE, "chain_t13.npz"), trajs_e=trajs_e.numpy().astype(np.float32),
                        hop_steps=flat, hops_per_particle=cnt, window_starts=np.array([s[0] for s in starts], dtype=np.int32))
    print("chain golden: trajs_e", tuple(trajs_e.shape), "window starts", [s[0] for s in starts], "hops", hop_steps)


if __name__ == "__main__":
    main()
