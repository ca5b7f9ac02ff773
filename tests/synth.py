"""seeded synthetic inputs — the generator itself lives in the package so bench.py / smoke() share it"""
from pcdet.datasets.synthetic import *  # noqa: F401,F403
from pcdet.datasets.synthetic import kitti_frame, kitti_batch, random_sparse_coords  # noqa: F401
