"""Drop-in module name for SplatFields' initialisation-time k-NN (reference scene/gaussian_model.py:25:
``from simple_knn._C import distCUDA2``), routed to the MI355X implementation in splatfields_amd."""
