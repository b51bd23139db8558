"""Model hyper-parameters: the three config objects of the reference (config.py:14-150: AudioEncoderConfig,
TextDecoderConfig, ModelConfig) generated from field tables, so names, defaults and the accepted config.json spellings sit
in one place each.  Dataclass semantics (keyword construction, equality, repr) and `from_dict` / `from_pretrained` are
the reference's."""
from __future__ import annotations

import dataclasses
import json
from pathlib import Path
from typing import Any

# (field, type, default, alternative config.json keys)
_AUDIO_FIELDS = (
    ("d_model", int, 1024, ()), ("encoder_layers", int, 24, ("num_hidden_layers",)), ("encoder_attention_heads", int, 16, ()),
    ("encoder_ffn_dim", int, 4096, ()), ("num_mel_bins", int, 128, ()), ("max_source_positions", int, 1500, ()),
    ("output_dim", int, 2048, ()), ("n_window", int, 50, ()), ("n_window_infer", int, 800, ()), ("conv_chunksize", int, 500, ()),
    ("activation_function", str, "gelu", ()), ("downsample_hidden_size", int, 480, ()),
)
_TEXT_FIELDS = (
    ("hidden_size", int, 2048, ()), ("num_hidden_layers", int, 28, ()), ("num_attention_heads", int, 16, ()),
    ("num_key_value_heads", int, 8, ()), ("head_dim", int, 128, ()), ("intermediate_size", int, 6144, ()),
    ("hidden_act", str, "silu", ()), ("vocab_size", int, 151936, ()), ("max_position_embeddings", int, 65536, ()),
    ("rms_norm_eps", float, 1e-6, ()), ("rope_theta", float, 1_000_000.0, ()),
    ("mrope_section", list, [24, 20, 20], ()),          # dead config: the text side is plain half-split RoPE
    ("rope_interleaved", bool, True, ()),
)
_TOKEN_IDS = (("audio_token_id", 151676), ("audio_start_token_id", 151669), ("audio_end_token_id", 151670))


def _spec(name, typ, default):
    if isinstance(default, list):
        return (name, typ, dataclasses.field(default_factory=lambda d=default: list(d)))
    return (name, typ, dataclasses.field(default=default))


def _read(table, src: dict) -> dict:
    out = {}
    for name, _typ, default, alts in table:
        for key in (name, *alts):
            if key in src:
                out[name] = src[key]
                break
        else:
            out[name] = list(default) if isinstance(default, list) else default
    return out


def _audio_from_dict(cls, d: dict[str, Any]):
    """Accepts the nested `audio_encoder_config` block or flat keys (reference config.py:33)."""
    return cls(**_read(_AUDIO_FIELDS, d.get("audio_encoder_config", d)))


def _text_from_dict(cls, d: dict[str, Any]):
    """Flat keys of config.json (reference config.py:84-101)."""
    return cls(**_read(_TEXT_FIELDS, d))


AudioEncoderConfig = dataclasses.make_dataclass(
    "AudioEncoderConfig", [_spec(n, t, v) for n, t, v, _ in _AUDIO_FIELDS],
    namespace={"from_dict": classmethod(_audio_from_dict), "__doc__": "Audio encoder hyper-parameters (reference config.py:14-58)."})
TextDecoderConfig = dataclasses.make_dataclass(
    "TextDecoderConfig", [_spec(n, t, v) for n, t, v, _ in _TEXT_FIELDS],
    namespace={"from_dict": classmethod(_text_from_dict), "__doc__": "Text decoder hyper-parameters (reference config.py:61-101)."})


def _model_from_dict(cls, d: dict[str, Any]):
    ids = {name: d.get(name, default) for name, default in _TOKEN_IDS}
    return cls(audio_encoder=AudioEncoderConfig.from_dict(d), text_decoder=TextDecoderConfig.from_dict(d), **ids)


def _model_from_pretrained(cls, model_path):
    """A local directory holding config.json, or a hub id (reference config.py:143-147; the latter needs network)."""
    where = Path(model_path)
    if where.is_dir():
        cfg_file = where / "config.json"
    else:
        from huggingface_hub import hf_hub_download
        cfg_file = Path(hf_hub_download(repo_id=str(model_path), filename="config.json"))
    return cls.from_dict(json.loads(cfg_file.read_text(encoding="utf-8")))


ModelConfig = dataclasses.make_dataclass(
    "ModelConfig",
    [("audio_encoder", AudioEncoderConfig, dataclasses.field(default_factory=AudioEncoderConfig)),
     ("text_decoder", TextDecoderConfig, dataclasses.field(default_factory=TextDecoderConfig))]
    + [(name, int, dataclasses.field(default=default)) for name, default in _TOKEN_IDS],
    namespace={"from_dict": classmethod(_model_from_dict), "from_pretrained": classmethod(_model_from_pretrained),
               "__doc__": "Top-level config (reference config.py:104-150)."})
for _c in (AudioEncoderConfig, TextDecoderConfig, ModelConfig):
    _c.__module__ = __name__
