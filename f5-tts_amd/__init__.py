"""f5-tts_amd — MI355X-native (gfx950) inference hot path for F5-TTS:
``CFM.sample`` + DiT forward + Vocos mel front-end + Vocos decode as hand-written HIP kernels
behind a C-ABI library (``csrc/libf5hip.so``, header ``include/f5hip.h``), with a thin host-side
mirror of the reference's ``load_model`` / ``load_vocoder`` / ``CFM.sample`` surface."""
from .config import (DIT_TINY, DIT_TINY_V0, E2TTS_BASE, F5TTS_BASE, F5TTS_V1_BASE, PRESETS, UNETT_TINY, VOCOS_MEL_24K, VOCOS_TINY,
                     DiTConfig, VocosConfig)

__all__ = ["DiTConfig", "VocosConfig", "F5TTS_V1_BASE", "F5TTS_BASE", "DIT_TINY", "DIT_TINY_V0", "VOCOS_MEL_24K",
           "VOCOS_TINY", "PRESETS", "E2TTS_BASE", "UNETT_TINY"]
