"""TEST INFRASTRUCTURE - NOT PRODUCT CODE.

fp32 CPU restatement of the reference's network forward pass
(AlphaZero/network/model.py:39-88, network/util.py:18-33), written against a plain
state_dict with torch.nn.functional ops and un-folded batch norm.

Parity unpinned: the reference's arithmetic is TensorFlow 1.x (requirements.txt:6-7), which is
not installed here, ships no checkpoint and is covered by none of the reference's tests.  This
file is the anchor the HIP tower is compared with (1e-5 abs for fp32, 1e-2 abs for bf16).
"""
import torch
import torch.nn.functional as F

BN_EPS = 1e-3


def _bn(x, sd, prefix):
    mean, var, beta = sd[prefix + ".running_mean"], sd[prefix + ".running_var"], sd[prefix + ".beta"]
    return (x - mean[None, :, None, None]) / torch.sqrt(var[None, :, None, None] + BN_EPS) + beta[None, :, None, None]


def forward(sd, x, num_blocks):
    """sd: AZNet.state_dict() (float32 CPU); x: float32 (B,17,H,W) -> (priors (B,O), values (B,))."""
    with torch.no_grad():
        r = F.relu(_bn(F.conv2d(x, sd["stem.weight"], padding=1), sd, "stem_bn"))
        for i in range(num_blocks):
            r1 = F.relu(_bn(F.conv2d(r, sd["conv1.%d.weight" % i], padding=1), sd, "bn1.%d" % i))
            r2 = _bn(F.conv2d(r1, sd["conv2.%d.weight" % i], padding=1), sd, "bn2.%d" % i)
            r = F.relu(r + r2)
        p = F.relu(_bn(F.conv2d(r, sd["p_conv.weight"]), sd, "p_bn")).reshape(x.shape[0], -1)
        logits = p @ sd["p_fc.weight"].t() + sd["p_fc.bias"]
        v = F.relu(_bn(F.conv2d(r, sd["v_conv.weight"]), sd, "v_bn")).reshape(x.shape[0], -1)
        v = F.relu(v @ sd["v_fc1.weight"].t() + sd["v_fc1.bias"])
        v = torch.tanh((v @ sd["v_fc2.weight"].t() + sd["v_fc2.bias"]).squeeze(-1))
        return F.softmax(logits, dim=1), v
