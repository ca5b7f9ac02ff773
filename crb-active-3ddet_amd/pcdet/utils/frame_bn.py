"""Per-frame BatchNorm statistics for a batch of G frames in train mode — what G separate bs=1 passes compute
(CRB stage 2, pcdet/query_strategies/crb_sampling.py:174-212: every frame is its own batch, SURVEY finding 11).

Two mechanisms, used together by `per_frame_batchnorm(model, G)`:
  * the fused HIP BatchNorm entry points (crbhip.bnrelu: sparse backbone, BEV backbone rows, set-abstraction rows) run
    their kernels once per frame row range (crbhip.bnrelu.frame_groups);
  * every plain nn.BatchNorm1d module (keypoint feature fusion, point head, RoI-head FC stack) gets a forward that
    normalises each frame's rows with that frame's statistics: the G frames become G x C "channels" of one
    F.batch_norm call (differentiable; the RoI-head gradient embeddings flow through it).
Running statistics: the fused kernels update them once per frame in frame order like G passes would; the patched
nn.BatchNorm1d modules leave them untouched (train-mode outputs never read them; the active loop reloads the initial
weights after every selection round, train_active_utils.py:321-322)."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from crbhip import bnrelu


def frame_batch_norm_1d(x, bn, G, offsets=None):
    """x (N,C) or (N,C,L) with N = rows of G frames -> BatchNorm with per-frame batch statistics"""
    if offsets is not None:                                  # ragged rows per frame: one call per frame
        outs = [F.batch_norm(x[a:b], None, None, bn.weight, bn.bias, True, 0.0, bn.eps)
                for a, b in zip(offsets[:-1], offsets[1:]) if b > a]
        return torch.cat(outs, 0)
    N, C = x.shape[0], x.shape[1]
    if N % G:
        raise ValueError('%d rows do not split into %d frames' % (N, G))
    n = N // G
    three = x.dim() == 3
    L = x.shape[2] if three else 1
    xg = x.reshape(G, n, C, L).permute(0, 2, 1, 3).reshape(1, G * C, n * L)
    w = bn.weight.repeat(G) if bn.affine else None
    b = bn.bias.repeat(G) if bn.affine else None
    y = F.batch_norm(xg, None, None, w, b, True, 0.0, bn.eps)
    y = y.reshape(G, C, n, L).permute(0, 2, 1, 3).reshape(N, C, L)
    return y if three else y.reshape(N, C)


class per_frame_batchnorm(object):
    def __init__(self, model, G):
        self.model, self.G = model, int(G)
        self.groups = bnrelu.frame_groups(self.G)
        self.patched = []

    def __enter__(self):
        self.groups.__enter__()
        G, groups = self.G, self.groups
        for m in self.model.modules():
            if isinstance(m, nn.BatchNorm1d) and m.training:
                def fwd(x, _m=m):
                    return frame_batch_norm_1d(x, _m, G, groups.noted(x))
                self.patched.append(m)
                m.forward = fwd
            elif isinstance(m, nn.BatchNorm2d) and m.training:
                def fwd2(x, _m=m):
                    raise RuntimeError('BatchNorm2d module called directly under per_frame_batchnorm: only the fused row '
                                       'paths (crbhip.bnrelu) keep per-frame statistics for image-shaped tensors')
                self.patched.append(m)
                m.forward = fwd2
        return self

    def __exit__(self, *exc):
        for m in self.patched:
            del m.forward                                     # back to the class's forward
        self.patched = []
        return self.groups.__exit__(*exc)
