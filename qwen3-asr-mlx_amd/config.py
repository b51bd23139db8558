"""Model configuration (mirror of reference config.py:14-150)."""
from __future__ import annotations

import json
from dataclasses import dataclass, field
from pathlib import Path
from typing import Any


def _pick(d: dict, cls, name: str, *alts: str):
    for key in (name, *alts):
        if key in d:
            return d[key]
    return getattr(cls, name)


@dataclass
class AudioEncoderConfig:
    """Audio encoder hyper-parameters (reference config.py:14-58)."""
    d_model: int = 1024
    encoder_layers: int = 24
    encoder_attention_heads: int = 16
    encoder_ffn_dim: int = 4096
    num_mel_bins: int = 128
    max_source_positions: int = 1500
    output_dim: int = 2048
    n_window: int = 50
    n_window_infer: int = 800
    conv_chunksize: int = 500
    activation_function: str = "gelu"
    downsample_hidden_size: int = 480

    @classmethod
    def from_dict(cls, d: dict[str, Any]) -> "AudioEncoderConfig":
        a = d.get("audio_encoder_config", d)          # nested or flat (config.py:33)
        return cls(
            d_model=_pick(a, cls, "d_model"),
            encoder_layers=_pick(a, cls, "encoder_layers", "num_hidden_layers"),
            encoder_attention_heads=_pick(a, cls, "encoder_attention_heads"),
            encoder_ffn_dim=_pick(a, cls, "encoder_ffn_dim"),
            num_mel_bins=_pick(a, cls, "num_mel_bins"),
            max_source_positions=_pick(a, cls, "max_source_positions"),
            output_dim=_pick(a, cls, "output_dim"),
            n_window=_pick(a, cls, "n_window"),
            n_window_infer=_pick(a, cls, "n_window_infer"),
            conv_chunksize=_pick(a, cls, "conv_chunksize"),
            activation_function=_pick(a, cls, "activation_function"),
            downsample_hidden_size=_pick(a, cls, "downsample_hidden_size"),
        )


@dataclass
class TextDecoderConfig:
    """Text decoder hyper-parameters (reference config.py:61-101); flat keys of config.json."""
    hidden_size: int = 2048
    num_hidden_layers: int = 28
    num_attention_heads: int = 16
    num_key_value_heads: int = 8
    head_dim: int = 128
    intermediate_size: int = 6144
    hidden_act: str = "silu"
    vocab_size: int = 151936
    max_position_embeddings: int = 65536
    rms_norm_eps: float = 1e-6
    rope_theta: float = 1_000_000.0
    mrope_section: list[int] = field(default_factory=lambda: [24, 20, 20])
    rope_interleaved: bool = True

    @classmethod
    def from_dict(cls, d: dict[str, Any]) -> "TextDecoderConfig":
        kw = {}
        for name in ("hidden_size", "num_hidden_layers", "num_attention_heads", "num_key_value_heads", "head_dim",
                     "intermediate_size", "hidden_act", "vocab_size", "max_position_embeddings", "rms_norm_eps",
                     "rope_theta", "rope_interleaved"):
            kw[name] = d.get(name, getattr(cls, name))
        kw["mrope_section"] = d.get("mrope_section", [24, 20, 20])
        return cls(**kw)


@dataclass
class ModelConfig:
    """Top-level config (reference config.py:104-150)."""
    audio_encoder: AudioEncoderConfig = field(default_factory=AudioEncoderConfig)
    text_decoder: TextDecoderConfig = field(default_factory=TextDecoderConfig)
    audio_token_id: int = 151676
    audio_start_token_id: int = 151669
    audio_end_token_id: int = 151670

    @classmethod
    def from_dict(cls, d: dict[str, Any]) -> "ModelConfig":
        return cls(
            audio_encoder=AudioEncoderConfig.from_dict(d),
            text_decoder=TextDecoderConfig.from_dict(d),
            audio_token_id=d.get("audio_token_id", 151676),
            audio_start_token_id=d.get("audio_start_token_id", 151669),
            audio_end_token_id=d.get("audio_end_token_id", 151670),
        )

    @classmethod
    def from_pretrained(cls, model_path: str | Path) -> "ModelConfig":
        path = Path(model_path)
        if path.is_dir():
            cfg = path / "config.json"
        else:                                           # by-name download (config.py:143-147); needs network
            from huggingface_hub import hf_hub_download
            cfg = Path(hf_hub_download(repo_id=str(model_path), filename="config.json"))
        return cls.from_dict(json.loads(cfg.read_text(encoding="utf-8")))
