"""``ME.utils`` subset: batched_coordinates (train_joint.py:82), sparse_quantize
(utils/dataloader.py:197), kaiming_normal_ (utils/resnet.py:112)."""
import math

import numpy as np
import torch


def batched_coordinates(coords, dtype=torch.int32, device=None):
    """list of [Ni, 3] coordinates -> [sum Ni, 4] with the batch index in column 0; floats are floored."""
    out = []
    for b, c in enumerate(coords):
        c = torch.as_tensor(np.asarray(c) if not torch.is_tensor(c) else c)
        if c.is_floating_point():
            c = torch.floor(c)
        c = c.to(dtype)
        out.append(torch.cat([torch.full((c.shape[0], 1), b, dtype=dtype), c], 1))
    r = torch.cat(out, 0)
    return r.to(device) if device is not None else r


def sparse_quantize(coordinates, features=None, labels=None, quantization_size=None, return_index=False,
                    **_):
    """floor(coords / quantization_size), one (the first) point per voxel."""
    c = np.asarray(coordinates)
    if quantization_size is not None:
        c = np.floor(c / quantization_size)
    c = c.astype(np.int32)
    _, idx = np.unique(c, axis=0, return_index=True)
    idx = np.sort(idx)
    if return_index:
        return c[idx], idx
    if features is None:
        return c[idx]
    if labels is None:
        return c[idx], np.asarray(features)[idx]
    return c[idx], np.asarray(features)[idx], np.asarray(labels)[idx]


def kaiming_normal_(tensor, a=0, mode="fan_in", nonlinearity="leaky_relu"):
    """ME.utils.kaiming_normal_ on a conv ``kernel`` [K, Cin, Cout] / [Cin, Cout]:
    fan_in = Cin * K, fan_out = Cout * K."""
    if tensor.dim() == 3:
        K, cin, cout = tensor.shape
    else:
        (cin, cout), K = tensor.shape, 1
    fan = cin * K if mode == "fan_in" else cout * K
    gain = torch.nn.init.calculate_gain(nonlinearity, a)
    std = gain / math.sqrt(fan)
    with torch.no_grad():
        return tensor.normal_(0, std)
