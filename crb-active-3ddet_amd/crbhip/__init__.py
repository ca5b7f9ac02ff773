"""ctypes binding of libcrbhip.so (the gfx950 C-ABI declared in include/crb_hip.h).

PyTorch is used only for device memory, streams and autograd plumbing; every hot op is a hand-written HIP
kernel reached through the C-ABI. There is NO CPU fallback: a missing library or a failed call raises
CrbHipError.
"""
from ._lib import (lib, CrbHipError, check, ptr, cur_stream, lib_path, header_path, parse_header,  # noqa: F401
                   require_cuda, host_i32x3, host_f32x3)
