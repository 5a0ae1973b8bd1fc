"""Architecture constants of the hot path (host side).

Mirrors the ``model.arch`` / ``model.mel_spec`` blocks of the reference's Hydra configs
(reference ``src/f5_tts/configs/F5TTS_v1_Base.yaml:20-47``, ``F5TTS_Base.yaml:20-46``,
``E2TTS_Base.yaml:20-41``) and the keyword arguments of ``DiT.__init__``
(reference ``src/f5_tts/model/backbones/dit.py:171-192``).
"""
from __future__ import annotations

import math
from dataclasses import asdict, dataclass, replace
from typing import Optional, Tuple

# mel front-end constants: reference src/f5_tts/infer/utils_infer.py:52-58
TARGET_SAMPLE_RATE = 24000
N_MEL_CHANNELS = 100
HOP_LENGTH = 256
WIN_LENGTH = 1024
N_FFT = 1024


@dataclass(frozen=True)
class DiTConfig:
    dim: int = 1024
    depth: int = 22
    heads: int = 16
    dim_head: int = 64
    ff_mult: int = 2
    mel_dim: int = N_MEL_CHANNELS
    text_num_embeds: int = 2545
    text_dim: int = 512
    text_mask_padding: bool = True
    conv_layers: int = 4
    pe_attn_head: Optional[int] = None  # None = rope on all heads, k = first k heads only
    attn_mask_enabled: bool = False
    conv_pos_kernel: int = 31
    conv_pos_groups: int = 16
    backbone: str = "DiT"
    # optional DiT variants no shipped yaml turns on, but the constructor accepts (reference dit.py:181-189)
    qk_norm: Optional[str] = None                    # "rms_norm": per-head RMSNorm(eps 1e-6) on q and k before rope (modules.py:402-409,493-496)
    long_skip_connection: bool = False               # Linear(2D -> D, no bias) over cat(x_out, x_in) after the blocks (dit.py:228,354-365)
    text_embedding_average_upsampling: bool = False  # zipvoice-style late upsampling of the text tokens over the frames (dit.py:55-84,131-137)
    skip_connect_type: str = "concat"                # UNetT only: "concat" | "add" | "none" (unett.py:127,289-295)

    @property
    def ff_inner(self) -> int:
        return int(self.dim * self.ff_mult)

    def arch_kwargs(self) -> dict:
        """kwargs accepted by the reference ``DiT(**model_cfg, text_num_embeds=..., mel_dim=...)`` / ``UNetT(...)``."""
        if self.backbone == "MMDiT":  # mmdit.py:94-109: text_dim == dim, no conv text blocks, rope on all heads
            return dict(dim=self.dim, depth=self.depth, heads=self.heads, dim_head=self.dim_head, ff_mult=self.ff_mult,
                        text_mask_padding=self.text_mask_padding, qk_norm=self.qk_norm, attn_mask_enabled=self.attn_mask_enabled,
                        mel_dim=self.mel_dim, text_num_embeds=self.text_num_embeds)
        if self.backbone == "UNetT":
            return dict(dim=self.dim, depth=self.depth, heads=self.heads, dim_head=self.dim_head, ff_mult=self.ff_mult,
                        text_dim=self.text_dim, text_mask_padding=self.text_mask_padding, conv_layers=self.conv_layers,
                        pe_attn_head=self.pe_attn_head, attn_mask_enabled=self.attn_mask_enabled, mel_dim=self.mel_dim,
                        text_num_embeds=self.text_num_embeds, qk_norm=self.qk_norm, skip_connect_type=self.skip_connect_type)
        return dict(dim=self.dim, depth=self.depth, heads=self.heads, dim_head=self.dim_head,
                    ff_mult=self.ff_mult, text_dim=self.text_dim, text_mask_padding=self.text_mask_padding,
                    conv_layers=self.conv_layers, pe_attn_head=self.pe_attn_head,
                    attn_mask_enabled=self.attn_mask_enabled, mel_dim=self.mel_dim,
                    text_num_embeds=self.text_num_embeds, qk_norm=self.qk_norm,
                    long_skip_connection=self.long_skip_connection,
                    text_embedding_average_upsampling=self.text_embedding_average_upsampling)

    def to_dict(self) -> dict:
        return asdict(self)


@dataclass(frozen=True)
class VocosConfig:
    """charactr/vocos-mel-24khz ``config.yaml`` (loader: reference src/f5_tts/infer/utils_infer.py:106-129)."""
    input_channels: int = N_MEL_CHANNELS
    dim: int = 512
    intermediate_dim: int = 1536
    num_layers: int = 8
    n_fft: int = N_FFT
    hop_length: int = HOP_LENGTH


@dataclass(frozen=True)
class BigVGANConfig:
    """``config.json`` of nvidia/bigvgan_v2_24khz_100band_256x, the generator the reference pairs with ``mel_spec_type="bigvgan"``
    (reference src/f5_tts/infer/utils_infer.py:130-144).  The generator source is an un-vendored submodule of the reference
    (``.gitmodules:1-3``); field names are upstream's (NVIDIA/BigVGAN ``bigvgan.py`` / ``config.json``)."""
    num_mels: int = N_MEL_CHANNELS
    upsample_rates: Tuple[int, ...] = (4, 4, 2, 2, 2, 2)
    upsample_kernel_sizes: Tuple[int, ...] = (8, 8, 4, 4, 4, 4)
    upsample_initial_channel: int = 1536
    resblock: str = "1"                                  # "1": AMPBlock1 (convs1 + convs2), "2": AMPBlock2
    resblock_kernel_sizes: Tuple[int, ...] = (3, 7, 11)
    resblock_dilation_sizes: Tuple[Tuple[int, ...], ...] = ((1, 3, 5), (1, 3, 5), (1, 3, 5))
    activation: str = "snakebeta"                        # "snake" | "snakebeta"
    snake_logscale: bool = True
    use_tanh_at_final: bool = False
    use_bias_at_final: bool = False

    @property
    def hop(self) -> int:
        return math.prod(self.upsample_rates)

    def channels(self, stage: int) -> int:
        """channels after upsampling stage `stage` (0-based)"""
        return self.upsample_initial_channel // (2 ** (stage + 1))


F5TTS_V1_BASE = DiTConfig()  # api/cli default (reference src/f5_tts/api.py:26)
# E2-TTS: flat U-Net transformer (reference src/f5_tts/model/backbones/unett.py:108-186, configs/E2TTS_Base.yaml:25-31):
# text_dim defaults to mel_dim, no ConvNeXt text blocks, ff_mult 4, rope on head 0 only, concat skip connections
E2TTS_BASE = DiTConfig(dim=1024, depth=24, heads=16, dim_head=64, ff_mult=4, text_dim=N_MEL_CHANNELS, conv_layers=0,
                       text_mask_padding=False, pe_attn_head=1, backbone="UNetT")
F5TTS_BASE = replace(F5TTS_V1_BASE, text_mask_padding=False, pe_attn_head=1)
# the Small models (configs/F5TTS_v1_Small.yaml:25-36, F5TTS_Small.yaml:25-35, E2TTS_Small.yaml:25-32): dim 768 = 12 heads x 64,
# i.e. 48 channels per group in the grouped conv-position embedding
F5TTS_V1_SMALL = DiTConfig(dim=768, depth=18, heads=12, dim_head=64, ff_mult=2, text_dim=512, conv_layers=4)
F5TTS_SMALL = replace(F5TTS_V1_SMALL, text_mask_padding=False, pe_attn_head=1)
E2TTS_SMALL = DiTConfig(dim=768, depth=20, heads=12, dim_head=64, ff_mult=4, text_dim=N_MEL_CHANNELS, conv_layers=0,
                        text_mask_padding=False, pe_attn_head=1, backbone="UNetT")
# reduced sizes used by the parity tests (same code path, seconds on the CPU oracle)
DIT_TINY = DiTConfig(dim=256, depth=2, heads=4, dim_head=64, ff_mult=2, text_dim=128, conv_layers=2,
                     text_num_embeds=255)
DIT_TINY_V0 = replace(DIT_TINY, text_mask_padding=False, pe_attn_head=1)
UNETT_TINY = DiTConfig(dim=256, depth=4, heads=4, dim_head=64, ff_mult=4, text_dim=N_MEL_CHANNELS, conv_layers=0,
                       text_mask_padding=False, pe_attn_head=1, text_num_embeds=255, backbone="UNetT")
# every optional DiT switch at once (qk RMSNorm, long skip, average upsampling, key-padding mask)
DIT_TINY_FLAGS = replace(DIT_TINY, qk_norm="rms_norm", long_skip_connection=True, text_embedding_average_upsampling=True,
                         attn_mask_enabled=True)
# MMDiT (reference backbones/mmdit.py; no yaml ships, count_params_gflops.py:19 sketches dim 512 / depth 16 / heads 16 / ff_mult 2):
# a second token stream for the text, joint attention, text_dim == dim
MMDIT_TINY = DiTConfig(dim=256, depth=3, heads=4, dim_head=64, ff_mult=2, text_dim=256, conv_layers=0, text_num_embeds=255,
                       backbone="MMDiT")
MMDIT_SMALL = DiTConfig(dim=512, depth=16, heads=16, dim_head=32, ff_mult=2, text_dim=512, conv_layers=0, text_num_embeds=2545,
                        backbone="MMDiT")
VOCOS_MEL_24K = VocosConfig()
VOCOS_TINY = VocosConfig(dim=128, intermediate_dim=384, num_layers=2)
BIGVGAN_V2_24K_100B_256X = BigVGANConfig()
# reduced sizes for the parity tests: two stages (x4, x2), 64 -> 32 -> 16 channels, and a three-stage one that ends in 24 / 12
# channels (rows that are not a multiple of the 32-element operand block) with AMPBlock2 / plain snake / tanh / final bias
BIGVGAN_TINY = BigVGANConfig(num_mels=20, upsample_rates=(4, 2), upsample_kernel_sizes=(8, 4), upsample_initial_channel=64,
                             resblock_kernel_sizes=(3, 7), resblock_dilation_sizes=((1, 3), (1, 3)))
BIGVGAN_TINY2 = BigVGANConfig(num_mels=20, upsample_rates=(2, 2, 2), upsample_kernel_sizes=(4, 4, 4), upsample_initial_channel=96,
                              resblock="2", resblock_kernel_sizes=(3, 11), resblock_dilation_sizes=((1, 5), (1, 3, 5)),
                              activation="snake", snake_logscale=False, use_tanh_at_final=True, use_bias_at_final=True)

PRESETS = {
    "F5TTS_v1_Base": F5TTS_V1_BASE,
    "F5TTS_Base": F5TTS_BASE,
    "tiny": DIT_TINY,
    "tiny_v0": DIT_TINY_V0,
    "E2TTS_Base": E2TTS_BASE,
    "tiny_unett": UNETT_TINY,
    "F5TTS_v1_Small": F5TTS_V1_SMALL,
    "F5TTS_Small": F5TTS_SMALL,
    "E2TTS_Small": E2TTS_SMALL,
    "tiny48": replace(DIT_TINY, dim=768, heads=12),  # 48 channels per conv group (768 / 16), like the Small models, at depth 2
    "tiny1024": replace(DIT_TINY, dim=1024, heads=16),  # 64 channels per conv group (1024 / 16), like the Base models, at depth 2: the MX conv-position kernel
    "tiny_inner512": replace(DIT_TINY, heads=8),  # attention width heads*dim_head = 512 != dim = 256 (modules.py:397-400)
    "tiny_flags": DIT_TINY_FLAGS,
    "tiny_mask": replace(DIT_TINY, attn_mask_enabled=True),  # key-padding mask alone: what the packed-row path (option "packed_rows") needs
    "F5TTS_v1_Small_mask": replace(F5TTS_V1_SMALL, attn_mask_enabled=True),
    "tiny_qknorm": replace(DIT_TINY, qk_norm="rms_norm"),
    "tiny_longskip": replace(DIT_TINY, long_skip_connection=True),
    "tiny_avgup": replace(DIT_TINY, text_embedding_average_upsampling=True),
    "tiny_unett_add": replace(UNETT_TINY, skip_connect_type="add", qk_norm="rms_norm", attn_mask_enabled=True),
    "tiny_unett_noskip": replace(UNETT_TINY, skip_connect_type="none"),
    "tiny_mmdit": MMDIT_TINY,
    "tiny_mmdit_mask": replace(MMDIT_TINY, attn_mask_enabled=True, qk_norm="rms_norm"),
    "tiny_mmdit_nopad": replace(MMDIT_TINY, text_mask_padding=False),
    # full-width variants of the optional paths, for cost measurements only (what the attention default costs where the q|k|v
    # launch cannot pack the score corrections): the v1 Base DiT with qk RMSNorm; an MMDiT at the Base width (no yaml of it ships)
    "F5TTS_v1_Base_qknorm": replace(F5TTS_V1_BASE, qk_norm="rms_norm"),
    "MMDiT_1024": DiTConfig(dim=1024, depth=22, heads=16, dim_head=64, ff_mult=2, text_dim=1024, conv_layers=0, text_num_embeds=2545, backbone="MMDiT"),
}
