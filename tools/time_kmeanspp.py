import sys, time
sys.path.insert(0, 'crb-active-3ddet_amd')
import torch
import pcdet
from pcdet.query_strategies import scoring
X = torch.randn(500, 65536, device='cuda')
for _ in range(2):
    torch.cuda.synchronize(); t = time.time()
    sel = scoring.kmeans_plusplus_device(X, 300, random_state=0)
    torch.cuda.synchronize(); print('kmeans++ device: %.3f s' % (time.time() - t))
import gc
gc.collect(); gc.freeze()
torch.cuda.synchronize(); t = time.time()
sel = scoring.kmeans_plusplus_device(X, 300, random_state=0)
torch.cuda.synchronize(); print('after gc.freeze: %.3f s' % (time.time() - t))
