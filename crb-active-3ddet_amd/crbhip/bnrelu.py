"""Host side of the fused sparse BatchNorm1d(+ReLU) kernels (C-ABI: crb_bn_relu_*)."""
import torch

from ._lib import lib, check, ptr, cur_stream, require_cuda


FUSE_RUNNING = __import__('os').environ.get('CRB_BN_FUSE_RUNNING', '1') == '1'


def supported(x, bn):
    C = x.shape[1]
    return (x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.shape[0] >= 2 and C % 4 == 0 and
            256 % (C // 4) == 0 and bn.affine and bn.track_running_stats)


class _BNReLUTrain(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, eps, relu, running_mean=None, running_var=None, momentum=0.0):
        require_cuda(x, gamma, beta)
        x = x.contiguous()
        n, C = x.shape
        dev = x.device
        z = torch.empty_like(x)
        mean = torch.empty((C,), dtype=torch.float32, device=dev)
        var = torch.empty_like(mean)
        invstd = torch.empty_like(mean)
        wsb = lib.crb_bn_workspace_bytes(n, C)
        ws = torch.empty((wsb,), dtype=torch.uint8, device=dev)
        g, b = gamma.contiguous().float(), beta.contiguous().float()
        check(lib.crb_bn_relu_forward(ptr(x), n, C, ptr(g), ptr(b), float(eps), int(relu), ptr(z), ptr(mean), ptr(var),
                                      ptr(invstd), ptr(running_mean), ptr(running_var), float(momentum), ptr(ws), wsb,
                                      cur_stream(dev)), 'crb_bn_relu_forward')
        ctx.save_for_backward(x, mean, invstd, g, b)
        ctx.relu = int(relu)
        ctx.mark_non_differentiable(mean, var)
        return z, mean, var

    @staticmethod
    def backward(ctx, dz, _dm, _dv):
        x, mean, invstd, g, b = ctx.saved_tensors
        n, C = x.shape
        dev = x.device
        dz = dz.contiguous().float()
        dx = torch.empty_like(x)
        dgamma = torch.empty((C,), dtype=torch.float32, device=dev)
        dbeta = torch.empty_like(dgamma)
        wsb = lib.crb_bn_workspace_bytes(n, C)
        ws = torch.empty((wsb,), dtype=torch.uint8, device=dev)
        check(lib.crb_bn_relu_backward(ptr(x), ptr(dz), n, C, ptr(mean), ptr(invstd), ptr(g), ptr(b), ctx.relu, ptr(dx),
                                       ptr(dgamma), ptr(dbeta), ptr(ws), wsb, cur_stream(dev)), 'crb_bn_relu_backward')
        return dx, dgamma, dbeta, None, None, None, None, None


def bn_relu(x, bn, relu=True):
    """x (N,C) cuda f32; bn: nn.BatchNorm1d. Same semantics as relu(bn(x)) incl. the running-statistics update
    (momentum, unbiased running variance, num_batches_tracked)."""
    n, C = x.shape
    if bn.training:
        with torch.no_grad():
            bn.num_batches_tracked += 1
        if FUSE_RUNNING and bn.momentum is not None and bn.running_mean.is_contiguous() and \
                bn.running_var.is_contiguous():
            # the running statistics are updated inside the finalize launch of the forward
            z, mean, var = _BNReLUTrain.apply(x, bn.weight, bn.bias, bn.eps, relu, bn.running_mean, bn.running_var,
                                              float(bn.momentum))
            return z
        z, mean, var = _BNReLUTrain.apply(x, bn.weight, bn.bias, bn.eps, relu)
        with torch.no_grad():                      # cumulative moving average (momentum=None): factor known on the host
            m = bn.momentum if bn.momentum is not None else 1.0 / float(bn.num_batches_tracked)
            bn.running_mean.mul_(1 - m).add_(mean, alpha=m)
            bn.running_var.mul_(1 - m).add_(var, alpha=m * n / (n - 1))
        return z
    if torch.is_grad_enabled() and (x.requires_grad or bn.weight.requires_grad):
        z = torch.nn.functional.batch_norm(x, bn.running_mean, bn.running_var, bn.weight, bn.bias, False, 0.0, bn.eps)
        return torch.relu(z) if relu else z
    x = x.contiguous()
    invstd = torch.rsqrt(bn.running_var + bn.eps)
    z = torch.empty_like(x)
    check(lib.crb_bn_relu_apply(ptr(x), n, C, ptr(bn.running_mean.contiguous()), ptr(invstd), ptr(bn.weight.contiguous()),
                                ptr(bn.bias.contiguous()), int(relu), ptr(z), cur_stream(x.device)), 'crb_bn_relu_apply')
    return z


@torch.no_grad()
def bn_apply_(x, bn, relu=True):
    """in-place inference BatchNorm(+ReLU) on (N, C) rows from the running statistics: one read + one write of x"""
    require_cuda(x)
    n, C = x.shape
    assert x.is_contiguous()
    invstd = torch.rsqrt(bn.running_var + bn.eps)
    check(lib.crb_bn_relu_apply(ptr(x), n, C, ptr(bn.running_mean.contiguous()), ptr(invstd), ptr(bn.weight.contiguous()),
                                ptr(bn.bias.contiguous()), int(relu), ptr(x), cur_stream(x.device)), 'crb_bn_relu_apply')
    return x
