"""Dataset side of the hot path: the synthetic KITTI/Waymo-shaped dataset used for benchmarking and tests, the
collate layout of DatasetTemplate.collate_batch (pcdet/datasets/dataset.py:160-229) and the rank-strided eval sampler
of pcdet/datasets/__init__.py:26-46. Real-dataset readers / augmentors are out of scope."""
from .synthetic_dataset import SyntheticDataset, build_synthetic_dataloader  # noqa: F401
from .sampler import DistributedSampler  # noqa: F401
