from . import pointcloud  # noqa: F401
