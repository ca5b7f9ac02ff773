"""Minimal attribute-dict config with the reference's loader entry points (pcdet/config.py:51-85:
cfg_from_yaml_file with _BASE_CONFIG_ includes, cfg_from_list overrides). easydict is not a dependency."""
from pathlib import Path

import yaml


class EasyDict(dict):
    """dict with attribute access, recursively applied (drop-in for easydict.EasyDict on the keys we use)"""

    def __init__(self, d=None, **kwargs):
        super().__init__()
        d = dict(d or {})
        d.update(kwargs)
        for k, v in d.items():
            self[k] = v

    @classmethod
    def _wrap(cls, v):
        if isinstance(v, dict) and not isinstance(v, EasyDict):
            return cls(v)
        if isinstance(v, (list, tuple)):
            return type(v)(cls._wrap(x) for x in v)
        return v

    def __setitem__(self, k, v):
        super().__setitem__(k, self._wrap(v))

    def __setattr__(self, k, v):
        self[k] = v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def update(self, *a, **kw):
        for k, v in dict(*a, **kw).items():
            self[k] = v


def merge_new_config(config, new_config):
    if '_BASE_CONFIG_' in new_config:
        with open(new_config['_BASE_CONFIG_'], 'r') as f:
            base = yaml.safe_load(f)
        config.update(EasyDict(base))
    for key, val in new_config.items():
        if not isinstance(val, dict):
            config[key] = val
            continue
        if key not in config:
            config[key] = EasyDict()
        merge_new_config(config[key], val)
    return config


def cfg_from_yaml_file(cfg_file, config):
    with open(cfg_file, 'r') as f:
        new_config = yaml.safe_load(f)
    merge_new_config(config=config, new_config=new_config)
    return config


def cfg_from_list(cfg_list, config):
    """--set KEY.SUB value ... overrides with literal_eval typing"""
    from ast import literal_eval
    assert len(cfg_list) % 2 == 0
    for k, v in zip(cfg_list[0::2], cfg_list[1::2]):
        keys = k.split('.')
        d = config
        for sub in keys[:-1]:
            assert sub in d, 'NotFoundKey: %s' % sub
            d = d[sub]
        assert keys[-1] in d, 'NotFoundKey: %s' % keys[-1]
        try:
            value = literal_eval(v)
        except Exception:
            value = v
        d[keys[-1]] = value


def log_config_to_file(cfg, pre='cfg', logger=None):
    for key, val in cfg.items():
        if isinstance(val, EasyDict):
            logger.info('\n%s.%s = edict()' % (pre, key))
            log_config_to_file(val, pre=pre + '.' + key, logger=logger)
        else:
            logger.info('%s.%s: %s' % (pre, key, val))


cfg = EasyDict()
cfg.ROOT_DIR = (Path(__file__).resolve().parent / '../').resolve()
cfg.LOCAL_RANK = 0
