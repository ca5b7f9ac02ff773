"""Inference-time Conv+BatchNorm folding shared by the BEV backbone, the SA MLPs and the RoI head.

w' = w * gamma / sqrt(running_var + eps) (per output channel), b' = beta - running_mean * that (+ conv bias * that).
The folded tensors are cached on the conv module and keyed on (data_ptr, _version) of every tensor they are made from:
an optimizer step, load_state_dict or .to(device) changes the key, so a stale fold is never used. No reference
counterpart (the reference runs the BN layers as modules); values agree up to f32 rounding, tested at rtol 1e-4."""
import torch
import torch.nn as nn


def _key(tensors):
    return tuple((t.data_ptr(), t._version) for t in tensors if t is not None)


def fold_conv_bn(conv, bn, transform=None):
    """-> (weight', bias') of conv followed by eval-mode bn; `transform(w, b)` post-processes once, inside the cache"""
    src = (conv.weight, conv.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var)
    key = _key(src)
    slot = '_crb_fold' if transform is None else '_crb_fold_' + transform.__name__
    hit = conv.__dict__.get(slot)
    if hit is not None and hit[0] == key:
        return hit[1]
    with torch.no_grad():
        scale = bn.weight * torch.rsqrt(bn.running_var + bn.eps)
        shift = bn.bias - bn.running_mean * scale
        if conv.bias is not None:
            shift = shift + conv.bias * scale
        if isinstance(conv, (nn.ConvTranspose1d, nn.ConvTranspose2d, nn.ConvTranspose3d)):
            shape = (1, -1) + (1,) * (conv.weight.dim() - 2)            # (Cin, Cout/groups, ...)
        else:
            shape = (-1, 1) + (1,) * (conv.weight.dim() - 2)
        w = conv.weight * scale.view(shape)                 # keeps the weight's memory format (channels_last stays)
        val = (w, shift)
        if transform is not None:
            val = transform(w, shift)
    conv.__dict__[slot] = (key, val)
    return val
