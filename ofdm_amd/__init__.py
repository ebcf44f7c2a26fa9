"""ofdm_amd -- 802.11a/g/p OFDM transmit chain on AMD MI355X behind the reference's Python API.

  ofdm_amd.batch              batched plans / kernels (TxContext, TxPlan)
  ofdm_amd.ftw_packet_utils   drop-in for the reference's examples/ftw_packet_utils.py
  ofdm_amd.ftw_ofdm           drop-in for examples/ftw_ofdm.py (ofdm_mod, ftw_transmit_path)
  ofdm_amd.ftw, ofdm_amd.gr   block factories and the minimal flowgraph layer they plug into
"""
__version__ = "0.1.0"
