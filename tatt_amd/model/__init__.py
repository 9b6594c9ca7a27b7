from . import tsrn, transformer_v2, stn_head, tps_spatial_transformer  # noqa: F401
