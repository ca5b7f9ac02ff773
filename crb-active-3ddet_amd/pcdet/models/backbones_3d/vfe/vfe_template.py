"""base class of the voxel feature encoders (pcdet/models/backbones_3d/vfe/vfe_template.py)"""
import torch.nn as nn


class VFETemplate(nn.Module):
    """subclasses provide get_output_feature_dim() and forward(batch_dict)"""

    def __init__(self, model_cfg, **kwargs):
        super().__init__()
        self.model_cfg = model_cfg

    def get_output_feature_dim(self):
        raise NotImplementedError(type(self).__name__ + '.get_output_feature_dim')

    def forward(self, **kwargs):
        raise NotImplementedError(type(self).__name__ + '.forward')
