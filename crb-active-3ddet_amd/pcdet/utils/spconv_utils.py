"""pcdet/utils/spconv_utils.py:3-34 surface: `spconv` (the gfx950-backed package), replace_feature, find_all_spconv_keys"""
from typing import Set

import torch.nn as nn

import spconv.pytorch as spconv  # noqa: F401  (resolves to crb-active-3ddet_amd/spconv)


def find_all_spconv_keys(model: nn.Module, prefix="") -> Set[str]:
    found = set()
    for name, child in model.named_children():
        new_prefix = f"{prefix}.{name}" if prefix != "" else name
        if isinstance(child, spconv.conv.SparseConvolution):
            found.add(f"{new_prefix}.weight")
        found.update(find_all_spconv_keys(child, prefix=new_prefix))
    return found


def replace_feature(out, new_features):
    return out.replace_feature(new_features)
