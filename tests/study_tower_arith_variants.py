#!/usr/bin/env python3
"""CPU model of the 16-bit tower's arithmetic in several variants, each against the float64 value of the network formula
(oracle/tower_ref.forward): which rounding costs what at the depths the configurations name.  Runs without a GPU.

    python tests/study_tower_arith_variants.py <board> <blocks> <seeds>

Variants: the device's bf16 arithmetic (conv output rounded to bf16 before the residual is added, then again after), one
rounding per layer, a float32 residual stream, float32 heads, float32 weights with bf16 activations, and the same with
IEEE half precision (f16: 11 significant bits against bf16's 8) in place of bf16.  Networks: 256 channels, seeded random
weights, batch-norm statistics calibrated on random boards."""
import sys, numpy as np, torch, torch.nn.functional as F
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from oracle import tower_ref
from alphazero_amd.network.model import CenterBN, calibrate_bn, random_net, fold_to_blob
torch.set_num_threads(8)
def boards(B,H,seed):
    return (np.random.RandomState(seed).rand(B,17,H,H) < 0.3).astype(np.int8)
def deep_net(H,O,NB,seed):
    net = random_net(H,H,O,NB,256,seed=seed)
    calibrate_bn(net, torch.from_numpy(boards(6,H,seed+100)).float())
    g = torch.Generator().manual_seed(seed)
    for m in net.modules():
        if isinstance(m, CenterBN): m.beta.data.copy_(torch.randn(m.beta.shape, generator=g)*0.1)
    return net
def rnd(t, fmt):
    if fmt=='bf16': return t.to(torch.bfloat16).to(torch.float32)
    if fmt=='f16': return t.to(torch.float16).to(torch.float32)
    return t
def model(blob,x,H,O,NB,C,fmt='bf16',double_round=False,res32=False,head32=False, wfmt=None):
    wfmt = wfmt or fmt
    blob = torch.as_tensor(blob,dtype=torch.float32); pos=[0]
    def take(*shape):
        n=int(np.prod(shape)); t=blob[pos[0]:pos[0]+n].reshape(*shape); pos[0]+=n; return t
    def conv(a,cin):
        w = rnd(take(C,9,cin),wfmt).reshape(C,3,3,cin).permute(0,3,1,2).contiguous(); b=take(C)
        return F.conv2d(a,w,padding=1)+b[None,:,None,None]
    HW=H*H
    with torch.no_grad():
        r = F.relu(conv(x.float(),17))
        if not res32: r = rnd(r,fmt)
        for _ in range(NB):
            t = rnd(F.relu(conv(rnd(r,fmt),C)),fmt)
            c2 = conv(t,C)
            if double_round: c2 = rnd(c2,fmt)
            r = F.relu(c2 + r)
            if not res32: r = rnd(r,fmt)
        rh = r if head32 else rnd(r,fmt)
        wh,bh = take(3,C),take(3)
        hd = F.relu(torch.einsum("bchw,kc->bkhw",rh,wh)+bh[None,:,None,None])
        wp,bp = take(2*HW,O),take(O)
        logits = hd[:,:2].reshape(x.shape[0],-1)@wp+bp
        w1,b1,w2,b2 = take(HW,256),take(256),take(256),take(1)
        v = torch.tanh(F.relu(hd[:,2].reshape(x.shape[0],-1)@w1+b1)@w2+b2)
        return F.softmax(logits,dim=1), v
if __name__=="__main__":
    H,NB = int(sys.argv[1]), int(sys.argv[2]); O=H*H+1
    variants = [("bf16 double-round (device today)",dict(fmt='bf16',double_round=True)),
                ("bf16 single-round",dict(fmt='bf16')),
                ("bf16 + fp32 residual",dict(fmt='bf16',res32=True)),
                ("bf16 + fp32 residual + fp32 heads",dict(fmt='bf16',res32=True,head32=True)),
                ("bf16 acts, fp32 weights",dict(fmt='bf16',wfmt='f32')),
                ("f16 double-round",dict(fmt='f16',double_round=True)),
                ("f16 single-round",dict(fmt='f16')),
                ("f16 + fp32 residual",dict(fmt='f16',res32=True,head32=True))]
    errs = {n:[[],[]] for n,_ in variants}
    for seed in range(int(sys.argv[3])):
        net = deep_net(H,O,NB,seed=NB+seed); blob = fold_to_blob(net)
        x = torch.from_numpy(boards(3,H,seed=NB+seed)).float()
        rp,rv = tower_ref.forward(net.state_dict(), x, NB, dtype=torch.float64)
        for n,kw in variants:
            p,v = model(blob,x,H,O,NB,256,**kw)
            errs[n][0].append(float((p.double()-rp).abs().max())); errs[n][1].append(float((v.double()-rv).abs().max()))
    print("H=%d NB=%d seeds=%s (3 boards each): max |dp|, max |dv|, mean |dv|max"%(H,NB,sys.argv[3]))
    for n,_ in variants:
        print("  %-40s p %.2e  v max %.2e  v mean-of-max %.2e"%(n,max(errs[n][0]),max(errs[n][1]),np.mean(errs[n][1])))
