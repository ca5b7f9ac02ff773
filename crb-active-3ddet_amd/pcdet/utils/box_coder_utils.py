"""ResidualCoder (pcdet/utils/box_coder_utils.py:5-77): anchor-relative 7-DoF box encoding."""
import torch


class ResidualCoder(object):
    def __init__(self, code_size=7, encode_angle_by_sincos=False, **kwargs):
        super().__init__()
        self.code_size = code_size + (1 if encode_angle_by_sincos else 0)
        self.encode_angle_by_sincos = encode_angle_by_sincos

    def encode_torch(self, boxes, anchors):
        """boxes/anchors (..., 7+C) -> residuals. Sizes are clamped at 1e-5 (out of place: the reference clamps its
        inputs in place, box_coder_utils.py:22-23; callers here never rely on that side effect)."""
        xa, ya, za = anchors[..., 0:1], anchors[..., 1:2], anchors[..., 2:3]
        dxa, dya, dza = [torch.clamp_min(anchors[..., i:i + 1], 1e-5) for i in (3, 4, 5)]
        xg, yg, zg = boxes[..., 0:1], boxes[..., 1:2], boxes[..., 2:3]
        dxg, dyg, dzg = [torch.clamp_min(boxes[..., i:i + 1], 1e-5) for i in (3, 4, 5)]
        ra, rg = anchors[..., 6:7], boxes[..., 6:7]
        diagonal = torch.sqrt(dxa ** 2 + dya ** 2)
        parts = [(xg - xa) / diagonal, (yg - ya) / diagonal, (zg - za) / dza,
                 torch.log(dxg / dxa), torch.log(dyg / dya), torch.log(dzg / dza)]
        if self.encode_angle_by_sincos:
            parts += [torch.cos(rg) - torch.cos(ra), torch.sin(rg) - torch.sin(ra)]
        else:
            parts += [rg - ra]
        if boxes.shape[-1] > 7:
            parts.append(boxes[..., 7:] - anchors[..., 7:])
        return torch.cat(parts, dim=-1)

    def decode_torch(self, box_encodings, anchors):
        xa, ya, za, dxa, dya, dza, ra = [anchors[..., i:i + 1] for i in range(7)]
        e = box_encodings
        diagonal = torch.sqrt(dxa ** 2 + dya ** 2)
        xg = e[..., 0:1] * diagonal + xa
        yg = e[..., 1:2] * diagonal + ya
        zg = e[..., 2:3] * dza + za
        dxg = torch.exp(e[..., 3:4]) * dxa
        dyg = torch.exp(e[..., 4:5]) * dya
        dzg = torch.exp(e[..., 5:6]) * dza
        if self.encode_angle_by_sincos:
            rg = torch.atan2(e[..., 7:8] + torch.sin(ra), e[..., 6:7] + torch.cos(ra))
            rest_e, rest_a = e[..., 8:], anchors[..., 7:]
        else:
            rg = e[..., 6:7] + ra
            rest_e, rest_a = e[..., 7:], anchors[..., 7:]
        parts = [xg, yg, zg, dxg, dyg, dzg, rg]
        if rest_e.shape[-1] > 0:
            parts.append(rest_e + rest_a)
        return torch.cat(parts, dim=-1)
