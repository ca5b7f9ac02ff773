"""Loader + ctypes signatures for libcrbhip.so.

Signatures are derived from include/crb_hip.h itself (the single source of truth for the C-ABI), so the
binding cannot drift from the header: every prototype there must resolve in the shared object.
"""
import ctypes
import os
import re

import torch  # noqa: F401  (loads libamdhip64 before our library)

_HERE = os.path.dirname(os.path.abspath(__file__))
_PKG = os.path.dirname(_HERE)
# CRB_MEASURE_LIB=1 (tools/ only): the -DCRB_MEASURE build, which adds the measurement-only entry points of
# include/crb_hip_measure.h. The product library exports none of them.
MEASURE = os.environ.get('CRB_MEASURE_LIB', '0') == '1'
lib_path = os.path.join(_PKG, 'lib', 'libcrbhip_measure.so' if MEASURE else 'libcrbhip.so')
header_path = os.path.join(os.path.dirname(_PKG), 'include', 'crb_hip.h')
measure_header_path = os.path.join(os.path.dirname(_PKG), 'include', 'crb_hip_measure.h')


class CrbHipError(RuntimeError):
    pass


_ERR = {-1: 'CRB_ERR_ARG', -2: 'CRB_ERR_WORKSPACE', -3: 'CRB_ERR_LAUNCH', -4: 'CRB_ERR_UNSUPPORTED'}

_CT = {'uint8_t': ctypes.c_uint8, 'int': ctypes.c_int, 'int32_t': ctypes.c_int32, 'int64_t': ctypes.c_int64, 'float': ctypes.c_float,
       'double': ctypes.c_double, 'uint32_t': ctypes.c_uint32, 'void': None}


def parse_header(path=header_path):
    """-> {name: (restype, [argtypes])} for every prototype in the header"""
    src = open(path).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    src = re.sub(r'//[^\n]*', '', src)
    protos = {}
    for m in re.finditer(r'\b(int|int64_t|void|float)\s+(crb_\w+)\s*\(([^)]*)\)\s*;', src):
        ret, name, args = m.group(1), m.group(2), m.group(3).strip()
        argtypes = []
        if args and args != 'void':
            for a in args.split(','):
                a = a.strip()
                if '*' in a:
                    argtypes.append(ctypes.c_void_p)
                else:
                    base = a.replace('const', '').split()[0]
                    argtypes.append(_CT[base])
        protos[name] = (_CT[ret], argtypes)
    return protos


def _load():
    if not os.path.exists(lib_path):
        raise CrbHipError(
            f'{lib_path} not found: build it with `make -C crb-active-3ddet_amd/csrc{" measure" if MEASURE else ""}` '
            f'(or __graft_entry__.build()). The HIP extension is mandatory; there is no CPU fallback.')
    L = ctypes.CDLL(lib_path, mode=ctypes.RTLD_GLOBAL)
    protos = parse_header()
    if MEASURE:
        protos.update(parse_header(measure_header_path))
    for name, (ret, argtypes) in protos.items():
        try:
            fn = getattr(L, name)
        except AttributeError as e:
            raise CrbHipError(f'{lib_path} does not export {name} declared in include/crb_hip*.h') from e
        fn.restype = ret
        fn.argtypes = argtypes
    return L


lib = _load()


def require_measure(name):
    """entry points that live in the measurement library only (include/crb_hip_measure.h): first Winograd design, split-bf16
    gather-GEMM, kernel-variant knobs. The product library does not carry them."""
    if not hasattr(lib, name) or not MEASURE:
        raise CrbHipError('%s is part of the measurement library only (include/crb_hip_measure.h): set CRB_MEASURE_LIB=1 before '
                          'importing crbhip (tools/ do); the product library libcrbhip.so does not carry it' % name)


# Tensors whose address went into the argument list of the call being assembled. `ptr(x.contiguous())` or `ptr(_i32(cnt))`
# may be handed a temporary: without a reference it is freed the moment ptr() returns and the caching allocator can give
# the same block to the next temporary of the SAME argument list. They are released once the launch call has returned
# (check()); from then on stream order protects the memory like any other torch tensor.
_keepalive = []


def check(rc, what):
    _keepalive.clear()
    if rc != 0:
        raise CrbHipError(f'{what} failed: {_ERR.get(rc, rc)}')


def ptr(t):
    """device (or host) pointer of a contiguous tensor, None -> NULL"""
    if t is None:
        return None
    if not t.is_contiguous():
        raise CrbHipError('C-ABI needs contiguous tensors')
    if len(_keepalive) > 256:                      # ptr() used outside a check(...) call: stay bounded
        del _keepalive[:128]
    _keepalive.append(t)
    return ctypes.c_void_p(t.data_ptr())


def cur_stream(device=None):
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise CrbHipError('libcrbhip ops need device tensors: the HIP path has no CPU fallback')


def host_i32x3(v):
    a = (ctypes.c_int32 * 3)(*[int(x) for x in v])
    return a


def host_f32x3(v):
    a = (ctypes.c_float * 3)(*[float(x) for x in v])
    return a
