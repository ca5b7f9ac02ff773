"""pcdet.models entry points with the reference's names (pcdet/models/__init__.py:16-51):
build_network, load_data_to_gpu, model_fn_decorator."""
from collections import namedtuple

import numpy as np
import torch

from .detectors import build_detector

_HOST_ONLY_KEYS = frozenset(('frame_id', 'metadata', 'calib'))       # stay numpy / python objects
_INT_KEYS = frozenset(('image_shape',))                             # uploaded as int32, everything else as float32
ModelReturn = namedtuple('ModelReturn', ['loss', 'tb_dict', 'disp_dict'])


def build_network(model_cfg, num_class, dataset):
    return build_detector(model_cfg=model_cfg, num_class=num_class, dataset=dataset)


def load_data_to_gpu(batch_dict):
    """in place: numpy arrays -> device tensors (float32; int32 for image_shape), host tensors -> device; entries that are
    already device tensors, the bookkeeping keys and non-array values are left as they are"""
    for key in list(batch_dict.keys()):
        val = batch_dict[key]
        if torch.is_tensor(val):
            if not val.is_cuda:
                batch_dict[key] = val.cuda(non_blocking=True)
        elif isinstance(val, np.ndarray) and key not in _HOST_ONLY_KEYS:
            host = torch.from_numpy(val)
            batch_dict[key] = (host.int() if key in _INT_KEYS else host.float()).cuda()


def model_fn_decorator():
    """-> model_func(model, batch_dict) = (mean loss, tb_dict, disp_dict), advancing the (possibly DDP-wrapped) model's step"""
    def model_func(model, batch_dict):
        load_data_to_gpu(batch_dict)
        ret_dict, tb_dict, disp_dict = model(batch_dict)
        getattr(model, 'module', model).update_global_step()
        return ModelReturn(ret_dict['loss'].mean(), tb_dict, disp_dict)

    return model_func
