#!/usr/bin/env python
"""Run only the CRB stage-1 scoring pass of bench.py (for rocprofv3 kernel traces): python tools/prof_scoring.py"""
import os
import sys
import argparse
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'crb-active-3ddet_amd'))
import torch  # noqa: E402
import bench  # noqa: E402

if __name__ == '__main__':
    a = argparse.Namespace(scoring_pool=96, scoring_repeats=2, points=20000)
    torch.cuda.set_device(0)
    print(bench.crb_scoring_bench(a, 0, 1, torch.device('cuda', 0)))
