#!/usr/bin/env python3
"""What the rounding of a 16-bit tower does to a search, measured where it matters: the same positions searched with the
fp32 tower and with the bf16 tower of this library (same weights, same replayed random draws), compared at the root.

The bf16 tower's value output is 1e-2 ... 4e-2 away from the fp32 value on a 20-block network (its priors within 3e-3);
this tool says what that does to the visit distribution the self-play data is made of, to the move that gets played and
to the root's value.

    python tools/bf16_search_effect.py [positions=48] [sims=800] [blocks=20] [sharpen=1] [dtypes=bf16]

Positions: Go 19x19, slot g after 2 * (g % 32) uniformly random legal stone moves (0 ... 62 plies), identical in every
engine.  Network: the bench's architecture (stem + `blocks` residual blocks x 256 channels), seeded random weights, batch-norm
statistics calibrated on random boards as a trained network's are; `sharpen` multiplies the policy head's last layer (1 =
flat priors of a random network, 30 = most of the mass on a few points, as after training).  Draws (transform ids, root
noise, the move-choice uniform) are replayed, the same in every engine.  Yardstick: the fp32 tower searched once more with
OTHER transform ids (the reference draws a random dihedral transform for every playout, search/mcts.py:208, so this is
how much two runs of the reference itself differ when nothing but those draws changes).  Writes a text summary to stdout
and, with AZ_EFFECT_JSON=path, the numbers as JSON."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from alphazero_amd.engine import Engine  # noqa: E402
from alphazero_amd.network.model import calibrate_bn, fold_to_blob, random_net  # noqa: E402

G = int(sys.argv[1]) if len(sys.argv) > 1 else 48
SIMS = int(sys.argv[2]) if len(sys.argv) > 2 else 800
NB = int(sys.argv[3]) if len(sys.argv) > 3 else 20
SHARPEN = float(sys.argv[4]) if len(sys.argv) > 4 else 1.0
DTYPES = (sys.argv[5] if len(sys.argv) > 5 else "bf16").split(",")
H, C = 19, 256
O = H * H + 1

rng = np.random.RandomState(7)
net = random_net(H, H, O, NB, C, seed=0)
calibrate_bn(net, torch.from_numpy((rng.rand(6, 17, H, H) < 0.3).astype(np.float32)))
if SHARPEN != 1.0:
    with torch.no_grad():
        net.p_fc.weight.mul_(SHARPEN)
        net.p_fc.bias.mul_(SHARPEN)
blob = fold_to_blob(net)

YARD = "fp32, other transform ids"
names = ["fp32"] + DTYPES + [YARD]
eng = {}
for dt in names:
    eng[dt] = Engine("go", H, H, n_slots=G, max_playout=SIMS, transform_types=8, rng_replay=True, num_blocks=NB, channels=C,
                     tower_dtype="fp32" if dt == YARD else dt, seed=1)
    eng[dt].tower_load(blob)

# positions: the same random legal stone moves in every engine
n_plies = np.array([2 * (g % 32) for g in range(G)])
for t in range(int(n_plies.max())):
    slots = np.flatnonzero(n_plies > t)
    masks = eng["fp32"].legal_mask(slots)[:, :H * H]
    moves = np.array([rng.choice(np.flatnonzero(m)) for m in masks], np.int32)
    for dt in names:
        st, _ = eng[dt].do_move(slots, moves)
        assert (st == 0).all()
# draws
streams = [(rng.randint(0, 8, size=SIMS + 2).astype(np.int32), rng.gamma(0.03, 1.0, size=O), rng.rand(2)) for _ in range(G)]
uni = rng.rand(G)
other_tids = [rng.randint(0, 8, size=SIMS + 2).astype(np.int32) for _ in range(G)]
res = {}
for dt in names:
    e = eng[dt]
    for g in range(G):
        if dt == YARD:
            e.rng_inject(g, other_tids[g], streams[g][1], streams[g][2])
        else:
            e.rng_inject(g, *streams[g])
    e.search_begin(np.arange(G), np.zeros(G, np.int32), np.ones(G, np.int32))
    t0 = time.perf_counter()
    e.search_run(SIMS + 1)
    while e.step_select() > 0:
        e.step_finish()
    dt_s = time.perf_counter() - t0
    rows = []
    for g in range(G):
        a, n, w, p, rootN = e.root_stats(g, 0)
        rows.append(dict(a=a.copy(), n=n.astype(np.float64), w=w.astype(np.float64), p=p.copy(), move=e.choose(g, 0, float(uni[g]))))
    # the network's own outputs at the root positions (no transform)
    pr, vr = e.tower_forward(e.features(np.arange(G)))
    res[dt] = dict(rows=rows, seconds=dt_s, root_priors=pr, root_value=vr)
    print("%s tower: %d positions x %d simulations in %.1f s" % (dt, G, SIMS, dt_s), file=sys.stderr)

out = {"positions": G, "sims": SIMS, "blocks": NB, "channels": C, "board": H, "policy_head_scale": SHARPEN,
       "plies_before_search": "0 ... %d" % int(n_plies.max()), "compared_with": "fp32 tower of the same library, same weights and draws"}
print("Go %dx%d, %d positions (0 ... %d plies in), %d simulations each, stem + %d blocks x %d ch, policy head x%g" %
      (H, H, G, int(n_plies.max()), SIMS, NB, C, SHARPEN))
ref = res["fp32"]
pmax = np.array([(r["n"] / r["n"].sum()).max() for r in ref["rows"]])
print("  fp32 search: visit share of the most visited move: mean %.3f, max %.3f" % (pmax.mean(), pmax.max()))
out["fp32_top_visit_share_mean"] = float(pmax.mean())
for dt in DTYPES + [YARD]:
    cur = res[dt]
    d_pi, tv, arg, mv, dq, same_children = [], [], [], [], [], []
    for r0, r1 in zip(ref["rows"], cur["rows"]):
        same = np.array_equal(np.sort(r0["a"]), np.sort(r1["a"]))   # (the order follows the root's transform id)
        same_children.append(same)
        pi0 = np.zeros(O); pi1 = np.zeros(O)
        pi0[r0["a"]] = r0["n"] / r0["n"].sum(); pi1[r1["a"]] = r1["n"] / r1["n"].sum()
        d_pi.append(np.abs(pi0 - pi1).max()); tv.append(0.5 * np.abs(pi0 - pi1).sum())
        arg.append(int(np.argmax(pi0) != np.argmax(pi1))); mv.append(int(r0["move"] != r1["move"]))
        dq.append(abs(r0["w"].sum() / r0["n"].sum() - r1["w"].sum() / r1["n"].sum()))
    dv = np.abs(ref["root_value"] - cur["root_value"]); dp = np.abs(ref["root_priors"] - cur["root_priors"]).max(axis=1)
    rec = {"max_abs_dpi": float(np.max(d_pi)), "mean_abs_dpi_max": float(np.mean(d_pi)), "total_variation_mean": float(np.mean(tv)),
           "total_variation_max": float(np.max(tv)), "argmax_move_differs_frac": float(np.mean(arg)),
           "sampled_move_differs_frac": float(np.mean(mv)), "root_q_abs_diff_mean": float(np.mean(dq)), "root_q_abs_diff_max": float(np.max(dq)),
           "network_value_abs_diff_max": float(dv.max()), "network_value_abs_diff_mean": float(dv.mean()),
           "network_priors_abs_diff_max": float(dp.max()), "same_root_children": bool(all(same_children)), "seconds": cur["seconds"],
           "fp32_seconds": ref["seconds"]}
    out[dt] = rec
    print("  %s against fp32:" % dt)
    print("    network at the root positions: |dvalue| max %.2e mean %.2e, |dprior| max %.2e" % (dv.max(), dv.mean(), dp.max()))
    print("    visit distribution pi = N / sum N: max |dpi| %.4f (mean over positions of the max %.4f), total variation mean %.4f max %.4f"
          % (rec["max_abs_dpi"], rec["mean_abs_dpi_max"], rec["total_variation_mean"], rec["total_variation_max"]))
    print("    most visited move differs in %.1f %% of the positions; the move sampled with the same uniform differs in %.1f %%"
          % (100 * rec["argmax_move_differs_frac"], 100 * rec["sampled_move_differs_frac"]))
    print("    root value of the search (sum W / sum N over the root's children): |dQ| mean %.2e max %.2e" % (rec["root_q_abs_diff_mean"], rec["root_q_abs_diff_max"]))
if os.environ.get("AZ_EFFECT_JSON"):
    with open(os.environ["AZ_EFFECT_JSON"], "w") as f:
        json.dump(out, f, indent=1)
for e in eng.values():
    e.close()
