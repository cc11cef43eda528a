"""Error budget of the towers at depth (GPU; a study script, not collected by pytest): HIP bf16 / f16 / fp32 outputs against the
fp32 and float64 evaluations of the network restatement and against a CPU model of the bf16 arithmetic.  One line per case.
    python tests/study_tower_depth_errors.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from alphazero_amd.engine import Engine
from alphazero_amd.network.model import fold_to_blob, random_net, calibrate_bn
from oracle import tower_ref

def boards(B, H, seed):
    return (np.random.RandomState(seed).rand(B, 17, H, H) < 0.3).astype(np.int8)

for game, H, NB, B in [("go", 19, 20, 2), ("go", 19, 20, 96), ("go", 9, 10, 410), ("reversi", 8, 6, 520), ("go", 19, 10, 96), ("go", 19, 5, 96)]:
    O = H * H + 1
    for seed in (1, 2):
        net = random_net(H, H, O, NB, 256, seed=seed)
        calibrate_bn(net, torch.from_numpy(boards(6, H, seed + 100)).float())
        blob = fold_to_blob(net)
        x = boards(B, H, seed)
        pick = np.array(sorted(set([0, B // 2, B - 1])))
        xt = torch.from_numpy(x[pick]).float()
        r32 = tower_ref.forward(net.state_dict(), xt, NB)
        r64 = tower_ref.forward(net.state_dict(), xt, NB, dtype=torch.float64)
        rbf = tower_ref.forward_bf16_model(blob, xt, H, H, O, NB, 256)
        for dt in ("bf16", "f16", "fp32"):
            eng = Engine(game, H, H, n_slots=max(B, 4), num_blocks=NB, channels=256, tower_dtype=dt)
            eng.tower_load(blob)
            p, v = eng.tower_forward(x)
            eng.close()
            p, v = p[pick], v[pick]
            def e(r):
                return "%.2e/%.2e" % (np.abs(p - r[0].numpy()).max(), np.abs(v - r[1].numpy()).max())
            print("%s %dx%d NB=%d B=%d seed=%d %s: vs fp32 %s  vs f64 %s  vs bf16-model %s   (p/v; pmax %.3f)" %
                  (game, H, H, NB, B, seed, dt, e(r32), e(r64), e(rbf), float(r64[0].max())), flush=True)
