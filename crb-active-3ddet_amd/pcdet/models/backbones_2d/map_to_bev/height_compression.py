"""HeightCompression (pcdet/models/backbones_2d/map_to_bev/height_compression.py:10-26): dense() scatter (HIP) + view.

MI355X: by default the scatter kernel writes the BEV tensor directly in channels_last memory (B, H, W, C*D) — the same
logical (B, C*D, H, W) tensor, same values — because MIOpen's fp32 NHWC igemm kernels then run without the
NCHW<->NHWC `batched_transpose` launches it otherwise wraps around them (measured: 99.1 -> 91.0 ms per SECOND bs=16 step).
Set `pcdet.models.backbones_2d.map_to_bev.height_compression.CHANNELS_LAST = False` for the NCHW layout."""
import torch.nn as nn

from crbhip import sparse as _sp

CHANNELS_LAST = True


class HeightCompression(nn.Module):
    def __init__(self, model_cfg, **kwargs):
        super().__init__()
        self.model_cfg = model_cfg
        self.num_bev_features = self.model_cfg.NUM_BEV_FEATURES

    def forward(self, batch_dict):
        sp = batch_dict['encoded_spconv_tensor']
        if CHANNELS_LAST and len(sp.spatial_shape) == 3:
            batch_dict['spatial_features'] = _sp.to_bev_channels_last(sp.features, sp.indices, sp.batch_size,
                                                                      sp.spatial_shape)
        else:
            x = sp.dense()
            N, C, D, H, W = x.shape
            batch_dict['spatial_features'] = x.view(N, C * D, H, W)
        batch_dict['spatial_features_stride'] = batch_dict['encoded_spconv_tensor_stride']
        return batch_dict
