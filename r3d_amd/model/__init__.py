"""Mirror of the reference's ``model`` package for the hot path: ``from model.futr_safuser_tokenfusion import FUTR``
(main_darai.py:24) becomes ``from r3d_amd.model.futr_safuser_tokenfusion import FUTR``."""
