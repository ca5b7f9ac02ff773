"""In-situ check of the Winograd convolutions inside a real training step.

A model-level gradient comparison goes through the BEV backbone's eleven train-mode BatchNorm layers, which amplify any f32
rounding difference to several 1e-3 at small batch sizes: such a pin cannot see a convolution bug of relative size 1e-3. This
module checks every Winograd launch of a step WHERE IT HAPPENS: inside `WinogradInSitu()` each forward, input-gradient and
weight-gradient call of crbhip.winograd is compared against an f64 convolution (torch, on the device) of the very tensors it was
handed, before anything downstream can amplify the difference. Errors are max |got - f64| / max |f64| per launch.

    with WinogradInSitu() as chk:
        loss = model(batch)[0]['loss']; loss.backward()
    chk.assert_all(2e-5)

Nothing here is a fallback: the kernels run exactly as in the product path, the f64 convolutions are only the yardstick."""
import torch
import torch.nn.functional as F

from . import winograd


class WinogradInSitu:
    def __init__(self, every=1, kinds=('fwd', 'dgrad', 'wgrad')):
        self.records = []          # (kind, (N, Cin, Cout, H, W), error)
        self.every = max(1, int(every))
        self.kinds = set(kinds)
        self._count = {}
        self._saved = {}

    # ---- patching -------------------------------------------------------------------------------------------------------------
    def __enter__(self):
        names = ('conv3x3', 'conv3x3_stats', 'conv3x3_U2', 'conv3x3_wgrad', 'weights_input_grad2')
        self._saved = {n: getattr(winograd, n) for n in names}
        orig = self._saved
        chk = self

        def conv3x3(x, weight, bias=None):
            y = orig['conv3x3'](x, weight, bias)
            chk._fwd(x, weight, bias, y)
            return y

        def conv3x3_stats(x, weight):
            y, st = orig['conv3x3_stats'](x, weight)
            chk._fwd(x, weight, None, y)
            return y, st

        def weights_input_grad2(weight):
            U = orig['weights_input_grad2'](weight)
            U._insitu_weight = weight.detach()
            return U

        def conv3x3_U2(x, U, bias=None, relu=False):
            y = orig['conv3x3_U2'](x, U, bias, relu)
            w = getattr(U, '_insitu_weight', None)
            if w is not None and bias is None and not relu:
                chk._dgrad(x, w, y)
            return y

        def conv3x3_wgrad(x, dy, like):
            dw = orig['conv3x3_wgrad'](x, dy, like)
            chk._wgrad(x, dy, dw)
            return dw
        for n, f in (('conv3x3', conv3x3), ('conv3x3_stats', conv3x3_stats), ('weights_input_grad2', weights_input_grad2),
                     ('conv3x3_U2', conv3x3_U2), ('conv3x3_wgrad', conv3x3_wgrad)):
            setattr(winograd, n, f)
        return self

    def __exit__(self, *exc):
        for n, f in self._saved.items():
            setattr(winograd, n, f)
        return False

    # ---- the yardsticks -------------------------------------------------------------------------------------------------------
    def _take(self, kind):
        if kind not in self.kinds:
            return False
        k = self._count.get(kind, 0)
        self._count[kind] = k + 1
        return k % self.every == 0

    @staticmethod
    def _err(got, want):
        return float((got.double() - want).abs().max() / want.abs().max().clamp_min(1e-300))

    def _fwd(self, x, weight, bias, y):
        if not self._take('fwd'):
            return
        with torch.no_grad():
            want = F.conv2d(x.detach().double(), weight.detach().double(), None if bias is None else bias.detach().double(), padding=1)
            self.records.append(('fwd', (x.shape[0], x.shape[1], weight.shape[0], x.shape[2], x.shape[3]), self._err(y.detach(), want)))

    def _dgrad(self, dy, weight, dx):
        if not self._take('dgrad'):
            return
        with torch.no_grad():
            want = F.conv_transpose2d(dy.detach().double(), weight.double(), padding=1)
            self.records.append(('dgrad', (dy.shape[0], weight.shape[1], weight.shape[0], dy.shape[2], dy.shape[3]),
                                 self._err(dx.detach(), want)))

    def _wgrad(self, x, dy, dw):
        if not self._take('wgrad'):
            return
        with torch.no_grad():
            want = torch.nn.grad.conv2d_weight(x.detach().double(), dw.shape, dy.detach().double(), padding=1)
            self.records.append(('wgrad', (x.shape[0], x.shape[1], dy.shape[1], x.shape[2], x.shape[3]), self._err(dw.detach(), want)))

    # ---- results --------------------------------------------------------------------------------------------------------------
    def worst(self):
        out = {}
        for kind, shape, e in self.records:
            if kind not in out or e > out[kind][1]:
                out[kind] = (shape, e)
        return out

    def summary(self):
        w = self.worst()
        n = {k: sum(1 for r in self.records if r[0] == k) for k in w}
        return ', '.join('%s %d launches worst %.2e @%s' % (k, n[k], w[k][1], 'x'.join(str(v) for v in w[k][0])) for k in sorted(w))

    def assert_all(self, tol, need=('fwd', 'dgrad', 'wgrad')):
        missing = [k for k in need if not any(r[0] == k for r in self.records)]
        assert not missing, 'no Winograd %s launch was seen inside the step (is the Winograd path on?)' % missing
        bad = [r for r in self.records if not r[2] <= tol]
        assert not bad, 'Winograd launches off their f64 convolution by more than %.0e: %s' % (tol, bad[:6])
