__all__ = {}
try:
    from .voxel_set_abstraction import VoxelSetAbstraction
    __all__['VoxelSetAbstraction'] = VoxelSetAbstraction
except ImportError:
    pass
