from bodyfitting_amd.mesh_grid_searcher import MeshGridSearcher  # noqa: F401  (thirdparty/mesh_grid/test_mesh_grid.py:2 imports it top-level)
