"""ctypes binding of libglc_hip.so (include/glc.h).  There is no fallback: if the library is
missing or a symbol cannot be bound, importing this module raises."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libglc_hip.so")

GLC_OK, GLC_EINVAL, GLC_EHIP, GLC_ENOMEM, GLC_EFORMAT, GLC_ENODEV, GLC_EIO = 0, -1, -2, -3, -4, -5, -6


class GlcInfo(C.Structure):
    _fields_ = [
        ("sample_rate", C.c_uint32),
        ("channels", C.c_uint16),
        ("reserved", C.c_uint16),
        ("total_samples", C.c_uint64),
        ("encoder_delay", C.c_uint32),
        ("padding", C.c_uint32),
        ("original_length", C.c_uint64),
        ("n_frames", C.c_uint64),
        ("n_raw_frames", C.c_uint64),
        ("total_nnz", C.c_uint64),
    ]


class GlcCompactInfo(C.Structure):
    _fields_ = [
        ("n_frames", C.c_uint64),
        ("n_pairs", C.c_uint64),
        ("n_raw_rows", C.c_uint64),
        ("bytes", C.c_uint64),
    ]


class GlcFramesView(C.Structure):
    """glc_frames_view (include/glc.h): EncodedAudio as flat arrays."""
    _fields_ = [
        ("sample_rate", C.c_uint32),
        ("channels", C.c_uint16),
        ("reserved", C.c_uint16),
        ("total_samples", C.c_uint64),
        ("encoder_delay", C.c_uint32),
        ("padding", C.c_uint32),
        ("original_length", C.c_uint64),
        ("n_frames", C.c_uint64),
        ("n_lists", C.c_uint64),
        ("n_pairs", C.c_uint64),
        ("n_scales", C.c_uint64),
        ("n_raw", C.c_uint64),
        ("list_begin", C.c_void_p),
        ("list_off", C.c_void_p),
        ("pairs", C.c_void_p),
        ("scale_begin", C.c_void_p),
        ("scales", C.c_void_p),
        ("raw_tag", C.c_void_p),
        ("raw_begin", C.c_void_p),
        ("raw", C.c_void_p),
    ]


class GlcFramesGather(C.Structure):
    """glc_frames_gather (include/glc.h): EncodedAudio by pointer per vector."""
    _fields_ = [
        ("sample_rate", C.c_uint32),
        ("channels", C.c_uint16),
        ("reserved", C.c_uint16),
        ("total_samples", C.c_uint64),
        ("encoder_delay", C.c_uint32),
        ("padding", C.c_uint32),
        ("original_length", C.c_uint64),
        ("n_frames", C.c_uint64),
        ("lists_per_frame", C.c_void_p),
        ("list_ptr", C.c_void_p),
        ("list_len", C.c_void_p),
        ("scales_per_frame", C.c_void_p),
        ("scale_ptr", C.c_void_p),
        ("raw_ptr", C.c_void_p),
        ("raw_len", C.c_void_p),
    ]


FRAMES_HOOK = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(GlcFramesView), C.c_uint64, C.c_uint64)


class GlcPlan(C.Structure):
    _fields_ = [
        ("n_frames", C.c_uint64),
        ("padded_len", C.c_uint64),
        ("per_channel", C.c_uint64),
        ("encoder_delay", C.c_uint32),
        ("padding", C.c_uint32),
    ]


_vp, _u8p = C.c_void_p, C.POINTER(C.c_uint8)

# name -> (restype, argtypes): every symbol include/glc.h declares
SIGNATURES = {
    "glc_ctx_create": (C.c_int, [C.c_int, C.c_uint32, C.POINTER(_vp)]),
    "glc_ctx_destroy": (None, [_vp]),
    "glc_last_error": (C.c_char_p, [_vp]),
    "glc_ctx_stream": (_vp, [_vp]),
    "glc_ctx_device": (C.c_int, [_vp]),
    "glc_ctx_set_stream": (C.c_int, [_vp, _vp]),
    "glc_ctx_synchronize": (C.c_int, [_vp]),
    "glc_ctx_timer_begin": (C.c_int, [_vp]),
    "glc_ctx_timer_end": (C.c_int, [_vp, C.POINTER(C.c_float)]),
    "glc_plan_encode": (C.c_int, [C.c_uint64, C.c_uint16, C.POINTER(GlcPlan)]),
    "glc_encode": (C.c_int, [_vp, _vp, C.c_uint64, C.c_uint16, C.POINTER(_vp)]),
    "glc_record_bytes": (C.c_uint64, [C.c_uint16]),
    "glc_encode_range_device": (C.c_int, [_vp, _vp, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint16,
                                          C.c_uint64, C.c_uint64, _vp, _vp]),
    "glc_mdct_forward_device": (C.c_int, [_vp, _vp, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint16,
                                          C.c_uint64, C.c_uint64, _vp]),
    "glc_frames_from_records": (C.c_int, [C.c_uint32, C.c_uint64, C.c_uint16, _vp, C.c_uint64,
                                          C.POINTER(_vp)]),
    "glc_frames_from_device_records": (C.c_int, [_vp, _vp, C.c_uint64, C.c_uint64, C.c_uint16, C.POINTER(_vp)]),
    "glc_compact_bound": (C.c_uint64, [C.c_uint16, C.c_uint64]),
    "glc_compact_device_records": (C.c_int, [_vp, _vp, C.c_uint64, C.c_uint16, _vp, C.c_uint64,
                                             C.POINTER(GlcCompactInfo)]),
    "glc_compact_records": (C.c_int, [_vp, C.c_uint64, C.c_uint16, _vp, C.c_uint64, C.POINTER(GlcCompactInfo)]),
    "glc_frames_from_compact": (C.c_int, [C.c_uint32, C.c_uint64, C.c_uint16, C.POINTER(_vp),
                                          C.POINTER(C.c_uint64), C.c_uint32, C.POINTER(_vp)]),
    "glc_decoded_len": (C.c_uint64, [_vp]),
    "glc_decode": (C.c_int, [_vp, _vp, _vp, C.c_uint64, C.POINTER(C.c_uint64)]),
    "glc_decode_device": (C.c_int, [_vp, _vp, _vp, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "glc_decode_range_device": (C.c_int, [_vp, _vp, C.c_uint64, C.c_uint64, _vp, C.c_uint64]),
    "glc_imdct_device": (C.c_int, [_vp, _vp, C.c_uint64, C.c_uint64, _vp]),
    "glc_decode_stream_begin": (C.c_int, [_vp, _vp]),
    "glc_decode_stream_next": (C.c_int, [_vp, _vp, C.c_uint64, C.POINTER(C.c_uint64),
                                         C.POINTER(C.c_int)]),
    "glc_serialized_size": (C.c_uint64, [_vp]),
    "glc_serialize": (C.c_int, [_vp, _vp, C.c_uint64, C.POINTER(C.c_uint64)]),
    "glc_deserialize": (C.c_int, [_vp, C.c_uint64, C.POINTER(_vp)]),
    "glc_save": (C.c_int, [_vp, C.c_char_p]),
    "glc_load": (C.c_int, [C.c_char_p, C.POINTER(_vp)]),
    "glc_frames_free": (None, [_vp]),
    "glc_frames_info": (C.c_int, [_vp, C.POINTER(GlcInfo)]),
    "glc_frame_is_raw": (C.c_int, [_vp, C.c_uint64]),
    "glc_frame_sparse": (C.c_int, [_vp, C.c_uint64, C.c_uint32, _vp, _vp, C.c_uint32,
                                   C.POINTER(C.c_uint32)]),
    "glc_frame_scale": (C.c_int, [_vp, C.c_uint64, C.c_uint32, C.POINTER(C.c_float)]),
    "glc_frame_raw": (C.c_int, [_vp, C.c_uint64, _vp, C.c_uint64, C.POINTER(C.c_uint64)]),
    "glc_frames_get_view": (C.c_int, [_vp, C.POINTER(GlcFramesView)]),
    "glc_frames_from_parts": (C.c_int, [C.POINTER(GlcFramesView), C.c_uint64, C.POINTER(_vp)]),
    "glc_frames_from_gather": (C.c_int, [C.POINTER(GlcFramesGather), C.c_uint64, C.POINTER(_vp)]),
    "glc_frames_stream_id": (C.c_uint64, [_vp]),
    "glc_ctx_resident_stream": (C.c_uint64, [_vp]),
    "glc_decode_resident": (C.c_int, [_vp, C.c_uint64, _vp, C.c_uint64, C.POINTER(C.c_uint64)]),
    "glc_encode_hooked": (C.c_int, [_vp, _vp, C.c_uint64, C.c_uint16, FRAMES_HOOK, _vp, C.POINTER(_vp)]),
    "glc_ctx_tables": (C.c_int, [_vp, _vp, _vp, C.POINTER(C.c_float), _vp, _vp,
                                 C.POINTER(C.c_uint32)]),
    "glc_wav_load": (C.c_int, [C.c_char_p, C.POINTER(_vp), C.POINTER(C.c_uint64), C.POINTER(C.c_uint32),
                               C.POINTER(C.c_uint16)]),
    "glc_wav_save16": (C.c_int, [C.c_char_p, _vp, C.c_uint64, C.c_uint32, C.c_uint16]),
    "glc_free": (None, [_vp]),
    "glc_flac_encode": (C.c_int, [_vp, C.c_uint64, C.c_uint32, C.c_uint16, C.c_uint8, C.POINTER(_vp),
                                  C.POINTER(C.c_uint64)]),
    "glc_flac_save": (C.c_int, [C.c_char_p, _vp, C.c_uint64, C.c_uint32, C.c_uint16, C.c_uint8]),
    "glc_flac_load": (C.c_int, [C.c_char_p, C.POINTER(_vp), C.POINTER(C.c_uint64), C.POINTER(C.c_uint32),
                                C.POINTER(C.c_uint16)]),
    "glc_flac_decode": (C.c_int, [_vp, C.c_uint64, C.POINTER(_vp), C.POINTER(C.c_uint64),
                                  C.POINTER(C.c_uint32), C.POINTER(C.c_uint16)]),
    "glc_version": (C.c_char_p, []),
}

if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
        "(or `make -C gapless-lossy-codec_amd/csrc`).  There is no CPU fallback.")

# PyTorch's ROCm wheel bundles its own libamdhip64.so / libhsa-runtime64.so, with the SAME SONAMEs
# (libamdhip64.so.7, libhsa-runtime64.so.1) as the system ROCm that libglc_hip.so links.  If torch is
# going to live in this process it must be loaded BEFORE libglc_hip.so: the dynamic loader then
# satisfies the library's NEEDED entries with the runtime torch already mapped, and the process has
# ONE HIP runtime - stream and event handles mean the same thing on both sides (glc_ctx_set_stream
# with a torch stream is sound).  The other order maps two runtimes (torch finds its own by RPATH):
# the second one to initialise sees no device, and handles must never cross.  hip_runtimes_mapped()
# reports which world this process is in (tests/test_host.py, tests/test_gpu_parity.py assert one).
try:  # torch is plumbing (device memory, torch.distributed), not a dependency of the codec
    import torch as _torch  # noqa: F401
except Exception:  # pragma: no cover - torch absent: the system ROCm runtime alone is fine
    _torch = None

lib = C.CDLL(LIB_PATH)
for _name, (_res, _args) in SIGNATURES.items():
    _fn = getattr(lib, _name)  # AttributeError if the ABI is incomplete
    _fn.restype = _res
    _fn.argtypes = _args


def hip_runtimes_mapped():
    """Paths of every libamdhip64 mapped into this process (/proc/self/maps).  One entry: torch and
    libglc_hip.so share a runtime and may exchange hipStream_t handles.  Two: they must not."""
    import re
    found = set()
    with open("/proc/self/maps") as fh:
        for ln in fh:
            m = re.search(r"(/\S*libamdhip64[^/\s]*)", ln)
            if m:
                found.add(os.path.realpath(m.group(1)))
    return sorted(found)


class GlcError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"glc error {code}: {msg}")
        self.code = code


def check(rc: int, ctx=None) -> None:
    if rc != GLC_OK:
        msg = lib.glc_last_error(ctx) or lib.glc_last_error(None) or b""
        raise GlcError(rc, msg.decode("utf-8", "replace"))
