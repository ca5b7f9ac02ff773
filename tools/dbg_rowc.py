import os, sys
os.environ['CRB_MEASURE_LIB'] = '1'
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'crb-active-3ddet_amd'))
import numpy as np, torch
from crbhip import sparse, voxel, lib
dev = torch.device('cuda', 0)
torch.manual_seed(0)
B = 1
coords = torch.unique(torch.cat([torch.zeros(3000, 1, dtype=torch.int32), torch.randint(0, 12, (3000, 3), dtype=torch.int32)], 1), dim=0).to(dev).contiguous()
rb = sparse.subm_rulebook(coords, [12, 12, 12], [3, 3, 3])
n, c = rb.n_out, 64
x = torch.randn(n, c, device=dev)
table = rb.table_for('nbr', c, c)
ref = {}
for c0 in range(64):
    w = torch.zeros(27, c, c, device=dev); w[:, c0, :] = 1.0
    lib.crb_sparse_conv_set_rowc(0)
    ref[c0] = sparse._conv_forward_raw(x, w, table, n)[:, 0].clone()
for c0 in (0, 1, 2, 4, 5, 16, 17, 21, 37, 63):
    w = torch.zeros(27, c, c, device=dev); w[:, c0, :] = 1.0
    lib.crb_sparse_conv_set_rowc(1)
    y = sparse._conv_forward_raw(x, w, table, n)[:, 0]
    match = [c1 for c1 in range(64) if float((y - ref[c1]).abs().max()) < 1e-4]
    # per-row: which channel matches each row?
    rows = []
    for rrow in range(8):
        rows.append([c1 for c1 in range(64) if abs(float(y[rrow] - ref[c1][rrow])) < 1e-5][:3])
    print('W row', c0, '-> output equals default with W row', match, 'per-row', rows)
