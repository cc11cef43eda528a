"""Dual-head ResNet weights (PyTorch holds them) and the folded blob the HIP tower loads.

Architecture follows AlphaZero/network/model.py:39-88 and network/util.py:18-33:
3x3 conv (no bias) -> BN(center only, eps 1e-3) -> ReLU; `num_blocks` x [conv-BN-ReLU-conv-BN-add-ReLU];
policy head 1x1 conv->2, BN, ReLU, NCHW flatten, FC->O, softmax; value head 1x1 conv->1, BN, ReLU,
FC->256, ReLU, FC->1, tanh.  TF's default initialiser (glorot uniform) and fresh BN statistics
(mean 0, var 1, beta 0) are used for random initialisation.

Blob layout (float32, BN already folded; tap = (dh+1)*3 + (dw+1) over the (H, W) axes):
    stem_w[C][9][17] stem_b[C]
    per block: w1[C][9][C] b1[C] w2[C][9][C] b2[C]
    head_w[3][C] (policy ch0, policy ch1, value) head_b[3]
    pfc_w[2*H*W][O] pfc_b[O]
    v1_w[H*W][256] v1_b[256] v2_w[256] v2_b[1]
"""
import math

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

BN_EPS = 1e-3  # tf.contrib.layers.batch_norm default epsilon


class CenterBN(nn.Module):
    """Inference/training batch norm with an offset but no scale (center=True, scale=False)."""

    def __init__(self, c, decay=0.9):
        super().__init__()
        self.beta = nn.Parameter(torch.zeros(c))
        self.register_buffer("running_mean", torch.zeros(c))
        self.register_buffer("running_var", torch.ones(c))
        self.decay = decay

    def forward(self, x):
        if self.training:
            mean = x.mean(dim=(0, 2, 3))
            var = x.var(dim=(0, 2, 3), unbiased=False)
            with torch.no_grad():
                self.running_mean.mul_(self.decay).add_(mean * (1 - self.decay))
                self.running_var.mul_(self.decay).add_(var * (1 - self.decay))
        else:
            mean, var = self.running_mean, self.running_var
        return (x - mean[None, :, None, None]) / torch.sqrt(var[None, :, None, None] + BN_EPS) + self.beta[None, :, None, None]


def _glorot_conv(conv):
    co, ci, kh, kw = conv.weight.shape
    limit = math.sqrt(6.0 / (ci * kh * kw + co * kh * kw))
    nn.init.uniform_(conv.weight, -limit, limit)


def _glorot_linear(lin):
    limit = math.sqrt(6.0 / (lin.in_features + lin.out_features))
    nn.init.uniform_(lin.weight, -limit, limit)
    nn.init.zeros_(lin.bias)


class AZNet(nn.Module):
    def __init__(self, board_height, board_width, flat_move_output, num_blocks, channels=256, planes=17):
        super().__init__()
        self.H, self.W, self.O, self.NB, self.C, self.F = board_height, board_width, flat_move_output, num_blocks, channels, planes
        C = channels
        self.stem = nn.Conv2d(planes, C, 3, padding=1, bias=False)
        self.stem_bn = CenterBN(C)
        self.conv1 = nn.ModuleList([nn.Conv2d(C, C, 3, padding=1, bias=False) for _ in range(num_blocks)])
        self.bn1 = nn.ModuleList([CenterBN(C) for _ in range(num_blocks)])
        self.conv2 = nn.ModuleList([nn.Conv2d(C, C, 3, padding=1, bias=False) for _ in range(num_blocks)])
        self.bn2 = nn.ModuleList([CenterBN(C) for _ in range(num_blocks)])
        self.p_conv = nn.Conv2d(C, 2, 1, bias=False)
        self.p_bn = CenterBN(2)
        self.p_fc = nn.Linear(2 * board_height * board_width, flat_move_output)
        self.v_conv = nn.Conv2d(C, 1, 1, bias=False)
        self.v_bn = CenterBN(1)
        self.v_fc1 = nn.Linear(board_height * board_width, 256)
        self.v_fc2 = nn.Linear(256, 1)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                _glorot_conv(m)
            elif isinstance(m, nn.Linear):
                _glorot_linear(m)

    def forward(self, x):
        r = F.relu(self.stem_bn(self.stem(x)))
        for i in range(self.NB):
            r1 = F.relu(self.bn1[i](self.conv1[i](r)))
            r2 = self.bn2[i](self.conv2[i](r1))
            r = F.relu(r + r2)
        p = F.relu(self.p_bn(self.p_conv(r))).reshape(x.shape[0], -1)
        logits = self.p_fc(p)
        v = F.relu(self.v_bn(self.v_conv(r))).reshape(x.shape[0], -1)
        v = torch.tanh(self.v_fc2(F.relu(self.v_fc1(v))).squeeze(-1))
        return F.softmax(logits, dim=1), v, logits


def _fold(conv, bn):
    w = conv.weight.detach().double()
    s = 1.0 / torch.sqrt(bn.running_var.detach().double() + BN_EPS)
    wf = w * s[:, None, None, None]
    b = bn.beta.detach().double() - bn.running_mean.detach().double() * s
    co, ci = wf.shape[0], wf.shape[1]
    # [co][ci][kh][kw] -> [co][kh*kw][ci]
    return wf.permute(0, 2, 3, 1).reshape(co, -1, ci), b


def fold_to_blob(net):
    """BN-folded float32 weight blob in the layout az_tower_load_weights expects."""
    parts = []

    def put(t):
        parts.append(np.asarray(t.detach().cpu().double().numpy() if torch.is_tensor(t) else t, dtype=np.float64).ravel())

    w, b = _fold(net.stem, net.stem_bn)
    put(w); put(b)
    for i in range(net.NB):
        w, b = _fold(net.conv1[i], net.bn1[i]); put(w); put(b)
        w, b = _fold(net.conv2[i], net.bn2[i]); put(w); put(b)
    wp, bp = _fold(net.p_conv, net.p_bn)
    wv, bv = _fold(net.v_conv, net.v_bn)
    put(torch.cat([wp.reshape(2, -1), wv.reshape(1, -1)], 0)); put(torch.cat([bp, bv]))
    put(net.p_fc.weight.t().contiguous()); put(net.p_fc.bias)
    put(net.v_fc1.weight.t().contiguous()); put(net.v_fc1.bias)
    put(net.v_fc2.weight.reshape(-1)); put(net.v_fc2.bias)
    return np.concatenate(parts).astype(np.float32)


def random_net(board_height, board_width, flat_move_output, num_blocks, channels=256, seed=0, perturb_bn=False):
    """Seeded glorot-uniform network; perturb_bn randomises BN statistics so folding is exercised."""
    g = torch.Generator().manual_seed(seed)
    state = torch.random.get_rng_state()
    torch.manual_seed(seed)
    net = AZNet(board_height, board_width, flat_move_output, num_blocks, channels)
    if perturb_bn:
        for m in net.modules():
            if isinstance(m, CenterBN):
                m.running_mean.copy_(torch.randn(m.running_mean.shape, generator=g) * 0.1)
                m.running_var.copy_(torch.rand(m.running_var.shape, generator=g) * 0.5 + 0.75)
                m.beta.data.copy_(torch.randn(m.beta.shape, generator=g) * 0.1)
    torch.random.set_rng_state(state)
    return net.eval()


def calibrate_bn(net, x):
    """Sets every batch norm's moving mean / variance to the statistics of its input on the batch x (float (B,17,H,W)),
    layer by layer, as the moving averages of a network that has trained on such data would be.  A randomly initialised
    tower with untouched statistics (mean 0, variance 1) roughly doubles its activation variance per residual block
    (there is no BN scale), which after 20 blocks leaves logits of magnitude 50 and more; calibrated, every layer's
    output is unit-scale."""
    bns = [m for m in net.modules() if isinstance(m, CenterBN)]
    old = [m.decay for m in bns]
    for m in bns:
        m.decay = 0.0          # running statistics <- this batch's statistics
    net.train()
    with torch.no_grad():
        net(x)
    for m, d in zip(bns, old):
        m.decay = d
    return net.eval()
