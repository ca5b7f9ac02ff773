"""pcdet.query_strategies factory (pcdet/query_strategies/__init__.py:12-29). Only the strategies on the CRB hot path
are provided; the other baselines are out of scope."""
from .crb_sampling import CRBSampling
from .entropy_sampling import EntropySampling
from .random_sampling import RandomSampling
from .strategy import Strategy  # noqa: F401

__factory = {
    'random': RandomSampling,
    'entropy': EntropySampling,
    'crb': CRBSampling,
}


def names():
    return sorted(__factory.keys())


def build_strategy(method, model, labelled_loader, unlabelled_loader, rank, active_label_dir, cfg):
    if method not in __factory:
        raise KeyError('Unknown query strategy: {} (available: {})'.format(method, names()))
    return __factory[method](model, labelled_loader, unlabelled_loader, rank, active_label_dir, cfg)
