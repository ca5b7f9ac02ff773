"""pcdet.models entry points (pcdet/models/__init__.py:16-51): build_network, load_data_to_gpu, model_fn_decorator."""
from collections import namedtuple

import numpy as np
import torch

from .detectors import build_detector


def build_network(model_cfg, num_class, dataset):
    return build_detector(model_cfg=model_cfg, num_class=num_class, dataset=dataset)


def load_data_to_gpu(batch_dict):
    """numpy -> cuda float (ints for image_shape); tensors already on the device are left alone
    (pcdet/models/__init__.py:23-34)"""
    for key, val in batch_dict.items():
        if isinstance(val, torch.Tensor):
            if not val.is_cuda:
                batch_dict[key] = val.cuda(non_blocking=True)
            continue
        if not isinstance(val, np.ndarray):
            continue
        if key in ['frame_id', 'metadata', 'calib']:
            continue
        if key in ['image_shape']:
            batch_dict[key] = torch.from_numpy(val).int().cuda()
        else:
            batch_dict[key] = torch.from_numpy(val).float().cuda()


def model_fn_decorator():
    ModelReturn = namedtuple('ModelReturn', ['loss', 'tb_dict', 'disp_dict'])

    def model_func(model, batch_dict):
        load_data_to_gpu(batch_dict)
        ret_dict, tb_dict, disp_dict = model(batch_dict)
        loss = ret_dict['loss'].mean()
        if hasattr(model, 'update_global_step'):
            model.update_global_step()
        else:
            model.module.update_global_step()
        return ModelReturn(loss, tb_dict, disp_dict)

    return model_func
