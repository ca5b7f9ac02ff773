"""AnchorGenerator (pcdet/models/dense_heads/target_assigner/anchor_generator.py:4-60). Anchors are built on the CPU
with the same float32 arithmetic (torch.arange steps) and moved by the head; no .cuda() in constructors."""
import torch


class AnchorGenerator(object):
    def __init__(self, anchor_range, anchor_generator_config):
        super().__init__()
        self.anchor_generator_cfg = anchor_generator_config
        self.anchor_range = anchor_range
        self.anchor_sizes = [c['anchor_sizes'] for c in anchor_generator_config]
        self.anchor_rotations = [c['anchor_rotations'] for c in anchor_generator_config]
        self.anchor_heights = [c['anchor_bottom_heights'] for c in anchor_generator_config]
        self.align_center = [c.get('align_center', False) for c in anchor_generator_config]
        assert len(self.anchor_sizes) == len(self.anchor_rotations) == len(self.anchor_heights)
        self.num_of_anchor_sets = len(self.anchor_sizes)

    def generate_anchors(self, grid_sizes):
        """-> list of (nz, ny, nx, n_size, n_rot, 7) tensors, list of anchors per location"""
        assert len(grid_sizes) == self.num_of_anchor_sets
        r = self.anchor_range
        all_anchors, per_loc = [], []
        for gs, sizes, rots, heights, centered in zip(grid_sizes, self.anchor_sizes, self.anchor_rotations,
                                                      self.anchor_heights, self.align_center):
            per_loc.append(len(rots) * len(sizes) * len(heights))
            if centered:
                xs, ys = (r[3] - r[0]) / gs[0], (r[4] - r[1]) / gs[1]
                xo, yo = xs / 2, ys / 2
            else:
                xs, ys = (r[3] - r[0]) / (gs[0] - 1), (r[4] - r[1]) / (gs[1] - 1)
                xo, yo = 0, 0
            x = torch.arange(r[0] + xo, r[3] + 1e-5, step=xs, dtype=torch.float32)
            y = torch.arange(r[1] + yo, r[4] + 1e-5, step=ys, dtype=torch.float32)
            z = x.new_tensor(heights)
            sz = x.new_tensor(sizes)              # (S,3)
            rt = x.new_tensor(rots)               # (R)
            nx, ny, nz, S, R = len(x), len(y), len(z), sz.shape[0], rt.shape[0]
            a = x.new_zeros((nz, ny, nx, S, R, 7))
            a[..., 0] = x.view(1, 1, nx, 1, 1)
            a[..., 1] = y.view(1, ny, 1, 1, 1)
            a[..., 2] = z.view(nz, 1, 1, 1, 1)
            a[..., 3:6] = sz.view(1, 1, 1, S, 1, 3)
            a[..., 6] = rt.view(1, 1, 1, 1, R)
            a[..., 2] += a[..., 5] / 2           # bottom height -> box centre
            all_anchors.append(a)
        return all_anchors, per_loc
