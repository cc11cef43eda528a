"""Weight interchange with the reference's TensorFlow graph (network/model.py:47-84, network/util.py:18-33).

The reference stores its weights as TF1 variables under the scope "model" (network/main.py:58); a
checkpoint is a name -> array map.  TensorFlow itself is not a dependency here: `to_tf_named` /
`from_tf_named` convert between an `AZNet` and a plain dict of NumPy arrays keyed by the reference's
variable names and laid out as TF lays them out, which `np.savez` can store and a three-line TF script
(`tf.train.load_variable` per name / `tf.assign` per name) can read or produce.

    conv kernels      HWIO  [kh, kw, in, out]          (PyTorch: OIHW)
    linear            [in, out]                        (PyTorch: [out, in])
    batch norm        <scope>/BatchNorm/{beta, moving_mean, moving_variance}  (center only, no gamma)
    policy flatten    NCHW order (c, h, w): every call site builds the graph with data_format='NCHW'

Parity note: no TF checkpoint ships with the reference and TF is absent, so the name list follows the
scopes in the source (tf.contrib.layers.batch_norm's default inner scope is "BatchNorm") and is
unpinned against a real checkpoint.
"""
import numpy as np
import torch


def _bn_items(prefix, bn):
    return [(prefix + "/BatchNorm/beta", bn.beta), (prefix + "/BatchNorm/moving_mean", bn.running_mean),
            (prefix + "/BatchNorm/moving_variance", bn.running_var)]


def _entries(net):
    """[(tf_name, tensor, kind)] in graph construction order; kind: conv | linear_w | vec."""
    out = [("model/W0", net.stem.weight, "conv")]
    out += [(n, t, "vec") for n, t in _bn_items("model/bn", net.stem_bn)]
    for i in range(net.NB):
        s = "model/resblock_%d" % i
        out.append((s + "/W1", net.conv1[i].weight, "conv"))
        out.append((s + "/W2", net.conv2[i].weight, "conv"))
        out += [(n, t, "vec") for n, t in _bn_items(s + "/B1", net.bn1[i])]
        out += [(n, t, "vec") for n, t in _bn_items(s + "/B2", net.bn2[i])]
    out.append(("model/policy_head/W0", net.p_conv.weight, "conv"))
    out += [(n, t, "vec") for n, t in _bn_items("model/policy_head/bn", net.p_bn)]
    out.append(("model/policy_head/linear/W", net.p_fc.weight, "linear_w"))
    out.append(("model/policy_head/linear/b", net.p_fc.bias, "vec"))
    out.append(("model/value_head/W0", net.v_conv.weight, "conv"))
    out += [(n, t, "vec") for n, t in _bn_items("model/value_head/bn", net.v_bn)]
    out.append(("model/value_head/F1/W", net.v_fc1.weight, "linear_w"))
    out.append(("model/value_head/F1/b", net.v_fc1.bias, "vec"))
    out.append(("model/value_head/F2/W", net.v_fc2.weight, "linear_w"))
    out.append(("model/value_head/F2/b", net.v_fc2.bias, "vec"))
    return out


def to_tf_named(net):
    """AZNet -> {tf variable name: float32 array in TF layout}."""
    d = {}
    for name, t, kind in _entries(net):
        a = t.detach().cpu().float()
        if kind == "conv":
            a = a.permute(2, 3, 1, 0)      # OIHW -> HWIO
        elif kind == "linear_w":
            a = a.t()                      # [out, in] -> [in, out]
        d[name] = np.ascontiguousarray(a.numpy())
    return d


def from_tf_named(net, named):
    """Loads a {tf variable name: array} map into `net` in place; every expected name must be present."""
    with torch.no_grad():
        for name, t, kind in _entries(net):
            if name not in named:
                raise KeyError("missing TF variable %r" % name)
            a = torch.from_numpy(np.asarray(named[name], dtype=np.float32))
            if kind == "conv":
                a = a.permute(3, 2, 0, 1)  # HWIO -> OIHW
            elif kind == "linear_w":
                a = a.t()
            if tuple(a.shape) != tuple(t.shape):
                raise ValueError("%s: shape %s does not fit %s" % (name, tuple(a.shape), tuple(t.shape)))
            t.copy_(a)
    return net
