#!/usr/bin/env python
"""Driver for the FETCH_SIZE calibration passes (tools/pmc_kernel.sh ... probe_lds_dma): launches crb_probe_lds_dma on a 1.5 GB buffer
(past the 256 MB Infinity Cache) in the patterns given on the command line, each 3 times:  python tools/pmc_lds_dma_calib.py
prints one line per pattern with the bytes it REQUESTS per launch; the counter summary of the same run lists the launches in order."""
import os
os.environ.setdefault('CRB_MEASURE_LIB', '1')
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'crb-active-3ddet_amd'))
import torch  # noqa: E402

if __name__ == '__main__':
    from crbhip import lib, check, ptr, cur_stream
    dev = torch.device('cuda', 0)
    nbytes = 1536 * 2**20
    src = torch.ones(nbytes // 4, device=dev)
    sink = torch.zeros(256, device=dev)
    st = cur_stream(dev)
    pats = [('dense 32-byte pieces (stride 32), one sweep', 32, 1), ('stride 512, one 8-channel chunk of every line', 512, 1),
            ('stride 512, all four chunks of every line in four sweeps', 512, 15), ('stride 128 (whole lines), one piece per line', 128, 1)]
    for name, stride, mask in pats:
        pieces = nbytes // stride
        for _ in range(3):
            check(lib.crb_probe_lds_dma(ptr(src), pieces, stride, mask, ptr(sink), st), 'probe')
        torch.cuda.synchronize()
        lines128 = nbytes if stride <= 128 else pieces * 128           # bytes of the distinct 128-byte lines the launch touches
        sect64 = nbytes if stride == 32 else pieces * 64 * (2 if mask == 15 else 1)
        print('PATTERN %-62s requested %8.1f MB per launch (distinct 64-byte sectors %8.1f MB, distinct 128-byte lines %8.1f MB)' % (
            name, pieces * 32 * bin(mask).count('1') / 1e6, sect64 / 1e6, lines128 / 1e6))
