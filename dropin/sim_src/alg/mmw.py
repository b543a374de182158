"""`from sim_src.alg.mmw import mmw` -> the MI355X solver (sig_sdp_mmw_amd.mmw.mmw)."""
from sig_sdp_mmw_amd.mmw import mmw  # noqa: F401
