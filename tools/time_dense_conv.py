#!/usr/bin/env python
"""The dense 3x3 convolution of the BEV backbone (16 x 128 x 200 x 176, 128 -> 128, f32) through MIOpen in both memory
layouts, forward / backward, with and without torch.backends.cudnn.benchmark (MIOpen's find): which solver families can it
reach and what do they cost?  usage: python tools/time_dense_conv.py"""
import os
import sys
import torch

if __name__ == '__main__':
    dev = torch.device('cuda', 0)
    for bench in (False, True):
        torch.backends.cudnn.benchmark = bench
        for cl in (False, True):
            for (C, H, W) in ((128, 200, 176), (256, 100, 88)):
                x = torch.randn(16, C, H, W, device=dev, requires_grad=True)
                conv = torch.nn.Conv2d(C, C, 3, padding=1, bias=False).to(dev)
                if cl:
                    x = x.detach().to(memory_format=torch.channels_last).requires_grad_(True)
                    conv = conv.to(memory_format=torch.channels_last)
                for _ in range(5):
                    y = conv(x)
                    y.backward(torch.ones_like(y))
                torch.cuda.synchronize()
                e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
                n = 10
                e[0].record()
                for _ in range(n):
                    y = conv(x)
                e[1].record()
                g = torch.ones_like(y)
                for _ in range(n):
                    y = conv(x)
                    y.backward(g)
                e[2].record()
                torch.cuda.synchronize()
                fwd = e[0].elapsed_time(e[1]) / n
                fb = e[1].elapsed_time(e[2]) / n
                gf = 2.0 * 16 * C * C * 9 * H * W / 1e9
                print('benchmark=%s %s C=%d %dx%d: fwd %.3f ms (%.0f TF), fwd+bwd %.3f ms (%.0f TF over 3 convs)' % (
                    bench, 'NHWC' if cl else 'NCHW', C, H, W, fwd, gf / fwd, fb, 3 * gf / fb), flush=True)
