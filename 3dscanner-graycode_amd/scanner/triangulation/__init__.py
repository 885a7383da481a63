from .triangulate import Triangulate, Triangulation  # noqa: F401
