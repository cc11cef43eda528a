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


def forward(sd, x, num_blocks, dtype=None):
    """sd: AZNet.state_dict() (float32 CPU); x: float32 (B,17,H,W) -> (priors (B,O), values (B,)).
    dtype=torch.float64 evaluates the same formula in double precision (weights and input are converted exactly): the
    value every float32 implementation of it - TensorFlow's, PyTorch's, the HIP tower's - approximates to rounding."""
    if dtype is not None:
        sd = {k: v.to(dtype) for k, v in sd.items()}
        x = x.to(dtype)
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


def _to_bf16(t):
    return t.to(torch.bfloat16).to(torch.float32)


def forward_bf16_model(blob, x, H, W, O, num_blocks, C):
    """What the bf16 tower computes, restated on the CPU from the folded weight blob (layout: alphazero_amd/network/
    model.py): 3x3 convolutions with bf16 weights and bf16 activations, float32 accumulation, float32 bias / residual /
    ReLU, activations rounded to bf16 (round to nearest even) after every layer; heads in float32 on the last bf16
    activations.  Separates kernel errors from the rounding a bf16 tower has by design."""
    blob = torch.as_tensor(blob, dtype=torch.float32)
    pos = [0]

    def take(*shape):
        n = 1
        for d in shape:
            n *= d
        t = blob[pos[0]:pos[0] + n].reshape(*shape)
        pos[0] += n
        return t

    def conv(a, cin):
        w = _to_bf16(take(C, 9, cin)).reshape(C, 3, 3, cin).permute(0, 3, 1, 2).contiguous()
        b = take(C)
        return F.conv2d(a, w, padding=1) + b[None, :, None, None]

    HW = H * W
    with torch.no_grad():
        r = _to_bf16(F.relu(conv(x.float(), 17)))
        for _ in range(num_blocks):
            t = _to_bf16(F.relu(conv(r, C)))
            r = _to_bf16(F.relu(conv(t, C) + r))
        wh, bh = take(3, C), take(3)
        hd = F.relu(torch.einsum("bchw,kc->bkhw", r, wh) + bh[None, :, None, None])
        wp, bp = take(2 * HW, O), take(O)
        logits = hd[:, :2].reshape(x.shape[0], -1) @ wp + bp
        w1, b1, w2, b2 = take(HW, 256), take(256), take(256), take(1)
        v = torch.tanh(F.relu(hd[:, 2].reshape(x.shape[0], -1) @ w1 + b1) @ w2 + b2)
        assert pos[0] == blob.numel()
        return F.softmax(logits, dim=1), v
