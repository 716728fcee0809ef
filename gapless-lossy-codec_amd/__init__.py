"""MI355X-native MDCT / quantiser hot path of the gapless lossy codec (reference:
ajcm474/gapless-lossy-codec v0.5.0, src/codec.rs) behind the reference's Encoder / Decoder API.

The directory name carries a hyphen (it mirrors the reference's crate name), so import it through
the top-level alias module: `import glc_amd`.
"""
from .codec import (FRAME_SIZE, FRAMES_PER_CHUNK, HOP_SIZE, AudioChunk, AudioHeader, Decoder,
                    EncodedAudio, EncodedFrame, Encoder, GaplessInfo, export_to_wav, load_encoded,
                    load_wav, plan_encode, save_encoded)
from ._lib import GlcError, LIB_PATH, SIGNATURES, lib
from . import shard

__all__ = ["Encoder", "Decoder", "EncodedAudio", "EncodedFrame", "AudioHeader", "GaplessInfo",
           "AudioChunk", "save_encoded", "load_encoded", "plan_encode", "load_wav", "export_to_wav", "GlcError", "shard",
           "FRAME_SIZE", "HOP_SIZE", "FRAMES_PER_CHUNK", "LIB_PATH", "SIGNATURES", "lib"]
