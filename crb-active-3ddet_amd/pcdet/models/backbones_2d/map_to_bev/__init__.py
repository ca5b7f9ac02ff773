from .height_compression import HeightCompression

__all__ = {
    'HeightCompression': HeightCompression,
}
