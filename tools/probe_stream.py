"""GPU, measurement library: how fast one workgroup per CU streams an L2-resident image into LDS (crb_probe_stream)."""
import os
import sys
os.environ['CRB_MEASURE_LIB'] = '1'
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'crb-active-3ddet_amd'))
import torch  # noqa: E402
from crbhip import lib, check, cur_stream  # noqa: E402

dev = torch.device('cuda:0')
sink = torch.zeros(4, device=dev)
for mb in (1.5, 6.0, 96.0):
    nbytes = int(mb * 2 ** 20)
    src = torch.randn(nbytes // 4, device=dev)
    for variant in (0, 1):
        for threads in (256, 512):
            for depth in (4, 8):
                for cus in (256, 64):
                    iters = max(1, int(32 * 2 ** 20 // nbytes))
                    def run():
                        check(lib.crb_probe_stream(variant, threads, depth, cus, src.data_ptr(), nbytes, iters, sink.data_ptr(), cur_stream(dev)), 'probe')
                    run(); torch.cuda.synchronize()
                    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(3):
                        run()
                    e1.record(); torch.cuda.synchronize()
                    ms = e0.elapsed_time(e1) / 3
                    per_cu = nbytes * iters / (ms * 1e-3) / 1e9
                    print('image %.1f MB, %s, %d threads, depth %d, %d workgroups: %.1f GB/s per CU (%.1f B/clk at 2.4 GHz), %.2f TB/s total'
                          % (mb, 'LDS-DMA' if variant == 0 else 'register staging', threads, depth, cus, per_cu, per_cu / 2.4, per_cu * cus / 1e3), flush=True)
