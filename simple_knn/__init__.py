"""Drop-in for the reference's ``simple_knn`` package (submodules/simple-knn), backed by the gfx950
library: ``from simple_knn._C import distCUDA2`` (lib/models/gaussian_model.py:5) keeps working."""
