"""MI355X-native MDCT / quantiser hot path of the gapless lossy codec (reference:
ajcm474/gapless-lossy-codec v0.5.0, src/codec.rs) behind the reference's Encoder / Decoder API.

The directory name carries a hyphen (it mirrors the reference's crate name), so import it through
the top-level alias module: `import glc_amd`.
"""
from .codec import (FRAME_SIZE, FRAMES_PER_CHUNK, HOP_SIZE, AudioChunk, AudioHeader, Decoder,
                    EncodedAudio, EncodedFrame, Encoder, GaplessInfo, decode_flac, encode_flac,
                    encode_flac_with_level, export_to_flac, export_to_flac_with_level, export_to_wav,
                    load_audio_file_lossless, load_encoded, load_flac, load_wav, plan_encode, save_encoded,
                    compact_bound, compact_records)
from ._lib import GlcError, LIB_PATH, SIGNATURES, lib
from . import shard

__all__ = ["Encoder", "Decoder", "EncodedAudio", "EncodedFrame", "AudioHeader", "GaplessInfo",
           "AudioChunk", "save_encoded", "load_encoded", "plan_encode", "load_wav", "export_to_wav",
           "encode_flac", "encode_flac_with_level", "export_to_flac", "export_to_flac_with_level", "load_flac",
           "decode_flac", "load_audio_file_lossless", "GlcError", "shard", "compact_bound", "compact_records",
           "FRAME_SIZE", "HOP_SIZE", "FRAMES_PER_CHUNK", "LIB_PATH", "SIGNATURES", "lib"]
