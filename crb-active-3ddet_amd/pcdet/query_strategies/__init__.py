"""pcdet.query_strategies factory (pcdet/query_strategies/__init__.py:12-29). `llal` needs the loss-prediction module of
the LLAL detector variant (`pv_rcnn_llal`, not on the SECOND / PV-RCNN / CRB path) and is not provided."""
from .badge_sampling import BadgeSampling
from .bald_sampling import BALDSampling
from .confidence_sampling import ConfidenceSampling
from .coreset_sampling import CoresetSampling
from .crb_sampling import CRBSampling
from .entropy_sampling import EntropySampling
from .montecarlo_sampling import MonteCarloSampling
from .random_sampling import RandomSampling
from .strategy import Strategy  # noqa: F401

__factory = {
    'random': RandomSampling,
    'entropy': EntropySampling,
    'confidence': ConfidenceSampling,
    'bald': BALDSampling,
    'montecarlo': MonteCarloSampling,
    'coreset': CoresetSampling,
    'badge': BadgeSampling,
    'crb': CRBSampling,
}


def names():
    return sorted(__factory.keys())


def build_strategy(method, model, labelled_loader, unlabelled_loader, rank, active_label_dir, cfg):
    if method not in __factory:
        raise KeyError('Unknown query strategy: {} (available: {})'.format(method, names()))
    return __factory[method](model, labelled_loader, unlabelled_loader, rank, active_label_dir, cfg)
