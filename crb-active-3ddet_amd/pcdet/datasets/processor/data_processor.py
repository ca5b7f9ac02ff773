"""VoxelGeneratorWrapper (pcdet/datasets/processor/data_processor.py:15-60) over the gfx950 voxel generator."""
import numpy as np


class VoxelGeneratorWrapper():
    def __init__(self, vsize_xyz, coors_range_xyz, num_point_features, max_num_points_per_voxel, max_num_voxels):
        from spconv.utils import Point2VoxelCPU3d as VoxelGenerator
        self.spconv_ver = 2
        self._voxel_generator = VoxelGenerator(
            vsize_xyz=vsize_xyz, coors_range_xyz=coors_range_xyz, num_point_features=num_point_features,
            max_num_points_per_voxel=max_num_points_per_voxel, max_num_voxels=max_num_voxels)

    def generate(self, points):
        """points (n,C) float32 numpy -> voxels (M,T,C), coordinates (M,3)[z,y,x], num_points (M)"""
        import cumm.tensorview as tv
        tv_voxels, tv_coordinates, tv_num_points = self._voxel_generator.point_to_voxel(
            tv.from_numpy(np.ascontiguousarray(points, dtype=np.float32)))
        return tv_voxels.numpy(), tv_coordinates.numpy(), tv_num_points.numpy()
