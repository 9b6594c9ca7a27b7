"""tatt_amd — MI355X-native (gfx950) implementation of the TATT scene-text super-resolution hot path.

Public surface mirrors the reference modules:  tatt_amd.model.tsrn.TSRN_TL_TRANS, tatt_amd.loss.ImageLoss ...
"""
__version__ = "0.1.0"
