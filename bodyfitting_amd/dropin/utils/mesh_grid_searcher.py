from bodyfitting_amd.mesh_grid_searcher import MeshGridSearcher  # noqa: F401
