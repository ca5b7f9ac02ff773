"""Conv+BN folding helper (host logic, CPU): values and cache invalidation."""
import torch
import torch.nn as nn

from pcdet.utils.fold_utils import fold_conv_bn


def _rand_bn(bn):
    bn.running_mean.normal_(0, 0.3)
    bn.running_var.uniform_(0.5, 2.0)
    bn.weight.data.uniform_(0.5, 1.5)
    bn.bias.data.normal_(0, 0.2)


def test_fold_matches_module_and_cache_tracks_versions():
    torch.manual_seed(0)
    for conv, bn, x in ((nn.Conv2d(5, 7, 3, padding=1, bias=False), nn.BatchNorm2d(7), torch.randn(2, 5, 9, 8)),
                        (nn.ConvTranspose2d(5, 6, 2, stride=2, bias=True), nn.BatchNorm2d(6), torch.randn(2, 5, 4, 4)),
                        (nn.Conv1d(4, 3, 1, bias=True), nn.BatchNorm1d(3), torch.randn(6, 4, 1))):
        _rand_bn(bn)
        bn.eval()
        f = {nn.Conv2d: torch.nn.functional.conv2d, nn.Conv1d: torch.nn.functional.conv1d}.get(type(conv))
        with torch.no_grad():
            w, b = fold_conv_bn(conv, bn)
            ref = bn(conv(x))
            got = f(x, w, b, conv.stride, conv.padding) if f else \
                torch.nn.functional.conv_transpose2d(x, w, b, conv.stride, conv.padding)
            torch.testing.assert_close(got, ref, rtol=1e-4, atol=1e-5)
            assert fold_conv_bn(conv, bn)[0] is w                    # cache hit
            conv.weight.mul_(1.5)                                    # in-place update (optimizer / load_state_dict)
            w2, _ = fold_conv_bn(conv, bn)
            assert w2 is not w
            torch.testing.assert_close(w2, w * 1.5, rtol=1e-6, atol=1e-7)
            bn.running_var.add_(0.1)
            assert fold_conv_bn(conv, bn)[0] is not w2
