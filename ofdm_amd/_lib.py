"""ctypes binding of libofdmtx.so (the HIP library; see include/ofdmtx.h).

There is no CPU fallback: if the shared library is missing or a call fails, an exception is
raised.  The library is built in-tree by ofdm_amd/csrc/Makefile (`python -c "import
__graft_entry__ as g; g.build()"`).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("OFDMTX_LIB", os.path.join(_HERE, "libofdmtx.so"))

OK = 0
EINVAL, EREGIME, ELENGTH, EHIP, ENOMEM, ESHAPE, ETOOBIG, EBUSY, ECOMM = -1, -2, -3, -4, -5, -6, -7, -8, -9
F_PREAMBLE, F_ZEROGAP, F_SWAPIQ, F_PNC_A, F_PNC_B, F_RAW_PSDU = 1, 2, 4, 8, 16, 32
PILOT_REF_STREAM, PILOT_PER_PACKET, PILOT_PNC_A, PILOT_PNC_B, PILOT_PNC_ALL = 0, 1, 2, 3, 4
# status bits of the loop-back verifier (ofdmtx_rx_demod / ofdmtx_rx_decode)
RX_PILOTS, RX_NULLS, RX_SIGNAL, RX_LENGTH, RX_SERVICE, RX_TAIL, RX_FCS, RX_CODE = 1, 2, 4, 8, 16, 32, 64, 128


class OfdmTxError(RuntimeError):
    def __init__(self, code, msg):
        RuntimeError.__init__(self, "libofdmtx: %s (code %d)" % (msg, code))
        self.code = code


class PktInfo(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("n_cbps", "n_bpsc", "n_dbps", "n_sym", "n_data", "n_pad",
                                          "txtime", "length", "plcp_len", "msg_len", "n_rows")]


class PlanSizes(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("n_pkts", "payload_bytes", "plcp_bytes", "msg_bytes",
                                           "n_symbols", "n_rows", "n_samples")]


class RawSizes(C.Structure):
    _fields_ = ([(n, C.c_uint32) for n in ("fft_length", "cp_length", "data_tones", "n_pre")] +
                [(n, C.c_uint64) for n in ("n_sym", "n_frames", "n_rows", "n_samples")])


# name -> (restype, argtypes); must list every function include/ofdmtx.h declares
_VP = C.c_void_p
PROTOTYPES = {
    "ofdmtx_strerror": (C.c_char_p, [C.c_int]),
    "ofdmtx_last_hip_error": (C.c_int, []),
    "ofdmtx_create": (C.c_int, [C.c_int, C.POINTER(_VP)]),
    "ofdmtx_destroy": (C.c_int, [_VP]),
    "ofdmtx_set_constellation": (C.c_int, [_VP, C.c_int, _VP, C.c_int]),
    "ofdmtx_set_preamble": (C.c_int, [_VP, _VP, C.c_int]),
    "ofdmtx_set_gap": (C.c_int, [_VP, _VP, C.c_int]),
    "ofdmtx_get_table": (C.c_int, [C.c_int, _VP, C.c_int]),
    "ofdmtx_crc32": (C.c_uint32, [C.c_char_p, C.c_size_t]),
    "ofdmtx_update_crc32": (C.c_uint32, [C.c_uint32, C.c_char_p, C.c_size_t]),
    "ofdmtx_packet_info": (C.c_int, [C.c_int, C.c_int, C.c_double, C.c_uint, C.POINTER(PktInfo)]),
    "ofdmtx_plan_uniform": (C.c_int, [_VP, C.c_uint64, C.c_int, C.c_uint32, C.c_uint64, C.c_double,
                                      C.c_int, C.c_uint32, C.c_uint, C.c_float, C.POINTER(_VP)]),
    "ofdmtx_plan_ragged": (C.c_int, [_VP, C.c_uint64, _VP, _VP, _VP, C.c_double, C.c_int, C.c_uint32,
                                     C.c_uint, C.c_float, C.POINTER(_VP)]),
    "ofdmtx_plan_ragged_device": (C.c_int, [_VP, C.c_uint64, _VP, _VP, _VP, C.c_double, C.c_int, C.c_uint32,
                                            C.c_uint, C.c_float, _VP, C.POINTER(_VP)]),
    "ofdmtx_plan_destroy": (C.c_int, [_VP]),
    "ofdmtx_plan_get_sizes": (C.c_int, [_VP, C.POINTER(PlanSizes)]),
    "ofdmtx_plan_get_layout": (C.c_int, [_VP] * 9),
    "ofdmtx_encode_bits": (C.c_int, [_VP] * 8),
    "ofdmtx_modulate": (C.c_int, [_VP] * 6),
    "ofdmtx_modulate_taps": (C.c_int, [_VP] * 8),
    "ofdmtx_symbol_taps": (C.c_int, [_VP] * 8),
    "ofdmtx_tx": (C.c_int, [_VP] * 6),
    "ofdmtx_tx_ragged_static": (C.c_int, [_VP, C.c_uint64, _VP, _VP, _VP, C.c_double, C.c_int, C.c_uint32, C.c_uint,
                                          C.c_float, _VP, _VP, C.c_uint64, C.POINTER(C.c_uint64), _VP]),
    "ofdmtx_scramble": (C.c_int, [_VP, _VP, C.c_uint32, C.c_uint32, _VP, _VP]),
    "ofdmtx_conv_encode": (C.c_int, [_VP, _VP, C.c_uint32, C.c_int, C.c_uint32, _VP, _VP]),
    "ofdmtx_interleave": (C.c_int, [_VP, _VP, C.c_uint32, C.c_int, _VP, _VP]),
    "ofdmtx_scale_cc": (C.c_int, [_VP, _VP, C.c_uint64, C.c_float, C.c_float, _VP, _VP]),
    "ofdmtx_rx_demod": (C.c_int, [_VP] * 6),
    "ofdmtx_rx_decode": (C.c_int, [_VP] * 6),
    "ofdmtx_repeat": (C.c_int, [_VP, _VP, C.c_uint64, C.c_uint64, _VP, _VP]),
    "ofdmtx_raw_plan_frames": (C.c_int, [_VP, C.c_uint32, C.c_uint32, _VP, _VP, C.c_uint32, C.c_uint64,
                                         C.c_uint32, C.POINTER(_VP)]),
    "ofdmtx_raw_plan_flags": (C.c_int, [_VP, C.c_uint32, C.c_uint32, _VP, _VP, C.c_uint32, _VP, C.c_uint64,
                                        C.POINTER(_VP)]),
    "ofdmtx_raw_plan_destroy": (C.c_int, [_VP]),
    "ofdmtx_raw_plan_get_sizes": (C.c_int, [_VP, C.POINTER(RawSizes)]),
    "ofdmtx_raw_modulate": (C.c_int, [_VP] * 6),
    "ofdmtx_raw_taps": (C.c_int, [_VP] * 8),
    "ofdmtx_stream_alloc": (C.c_int, [_VP, C.c_uint64, C.POINTER(_VP), C.POINTER(C.c_int)]),
    "ofdmtx_stream_alloc_parts": (C.c_int, [_VP, C.c_uint64, C.c_int, C.POINTER(_VP), C.POINTER(C.c_int)]),
    "ofdmtx_stream_info": (C.c_int, [_VP, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint32),
                                     C.POINTER(C.c_uint32)]),
    "ofdmtx_stream_free": (C.c_int, [_VP, _VP]),
    "ofdmtx_comm_unique_id": (C.c_int, [_VP]),
    "ofdmtx_comm_create": (C.c_int, [_VP, C.c_int, C.c_int, _VP, C.POINTER(_VP)]),
    "ofdmtx_comm_destroy": (C.c_int, [_VP]),
    "ofdmtx_gather": (C.c_int, [_VP, _VP, C.c_uint64, _VP, _VP, C.c_int, _VP]),
    "ofdmtx_last_comm_error": (C.c_int, []),
    "ofdmtx_raw_framebytes": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_uint64, C.POINTER(C.c_uint32)]),
    "ofdmtx_raw_tx": (C.c_int, [_VP, _VP, C.c_int, C.c_int, C.c_int, _VP, _VP, _VP]),
    "ofdmtx_raw_qam_tx": (C.c_int, [_VP, C.c_int, C.c_int, C.c_int, C.c_uint32, C.c_uint32, _VP, C.c_uint64, _VP,
                                    _VP]),
}

_lib = None


def load():
    """Load the library once; raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise OfdmTxError(EINVAL, "%s not found -- build it with ofdm_amd/csrc/Makefile "
                                  "(there is no CPU fallback)" % LIB_PATH)
    # PyTorch ships its own HIP runtime (same soname); import it first so that this library
    # binds to the runtime torch's tensors and streams live in instead of loading a second one.
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc):
    if rc != OK:
        lib = load()
        msg = lib.ofdmtx_strerror(rc).decode()
        if rc == EHIP:
            msg += " (hipError %d)" % lib.ofdmtx_last_hip_error()
        if rc == ECOMM:
            msg += " (ncclResult %d)" % lib.ofdmtx_last_comm_error()
        if rc == ELENGTH:
            raise ValueError("MPDU-length must be in [0, 3168]")      # ftw_packet_utils.py:119-121
        raise OfdmTxError(rc, msg)


def packet_info(regime, payload_len, symbol_time=4.0, flags=F_PREAMBLE | F_ZEROGAP):
    info = PktInfo()
    check(load().ofdmtx_packet_info(int(regime), int(payload_len), float(symbol_time), int(flags),
                                    C.byref(info)))
    return info


def crc32(data):
    data = bytes(data)
    return load().ofdmtx_crc32(data, len(data))


def update_crc32(crc, data):
    data = bytes(data)
    return load().ofdmtx_update_crc32(int(crc) & 0xFFFFFFFF, data, len(data))


def get_table(which):
    """Static table `which` (TABLE_*) as a complex64 numpy array."""
    import numpy as np
    buf = np.zeros(640, dtype=np.float32)
    n = load().ofdmtx_get_table(int(which), buf.ctypes.data, len(buf))
    if n < 0:
        check(n)
    return buf[:n].copy().view(np.complex64)


STATIC_MAX_PKTS = 65536
TABLE_BPSK, TABLE_QPSK, TABLE_QAM16, TABLE_QAM64, TABLE_PREAMBLE = 1, 2, 4, 6, 100
