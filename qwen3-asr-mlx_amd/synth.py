"""Synthetic checkpoints in the reference's layout (no real weights exist offline): seeded
N(0, 0.02^2) Linear/Conv weights, unit norm weights, zero biases, bf16 -- for tests and bench."""
from __future__ import annotations

import json
from pathlib import Path

from .config import AudioEncoderConfig, ModelConfig, TextDecoderConfig


def small_config() -> ModelConfig:
    """Reduced model that still satisfies every kernel's shape contract (head dims 64/128)."""
    return ModelConfig(
        audio_encoder=AudioEncoderConfig(d_model=256, encoder_layers=2, encoder_attention_heads=4, encoder_ffn_dim=512,
                                         output_dim=512, downsample_hidden_size=32),
        text_decoder=TextDecoderConfig(hidden_size=512, num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=2,
                                       head_dim=128, intermediate_size=1024, vocab_size=151936),
    )


def random_state(cfg: ModelConfig, seed: int = 0, device="cpu", bias_std: float = 0.0, norm_jitter: float = 0.0,
                 embed_std: float = 0.02) -> dict:
    """Checkpoint-keyed tensors (bf16).  bias_std / norm_jitter > 0 make biases and norm weights
    non-trivial so tests exercise them; embed_std widens the logit spread of the tied head
    (logit std = embed_std * sqrt(hidden_size) for a unit-RMS final hidden state)."""
    import torch
    g = torch.Generator(device=device).manual_seed(seed)
    a, t = cfg.audio_encoder, cfg.text_decoder
    st = {}

    def lin(name, out_f, in_f, bias):
        st[name + ".weight"] = (torch.randn(out_f, in_f, generator=g, device=device) * 0.02).to(torch.bfloat16)
        if bias:
            st[name + ".bias"] = (torch.randn(out_f, generator=g, device=device) * bias_std).to(torch.bfloat16)

    def norm(name, n, bias):
        st[name + ".weight"] = (1.0 + torch.randn(n, generator=g, device=device) * norm_jitter).to(torch.bfloat16)
        if bias:
            st[name + ".bias"] = (torch.randn(n, generator=g, device=device) * bias_std).to(torch.bfloat16)

    C1 = a.downsample_hidden_size
    for i, cin in ((1, 1), (2, C1), (3, C1)):
        st[f"audio_tower.conv2d{i}.weight"] = (torch.randn(C1, 3, 3, cin, generator=g, device=device) * (0.25 if i == 1 else 0.02)).to(torch.bfloat16)
        st[f"audio_tower.conv2d{i}.bias"] = (torch.randn(C1, generator=g, device=device) * bias_std).to(torch.bfloat16)
    fq = a.num_mel_bins
    for _ in range(3):
        fq = (fq - 1) // 2 + 1
    lin("audio_tower.conv_out", a.d_model, C1 * fq, False)
    for i in range(a.encoder_layers):
        p = f"audio_tower.layers.{i}."
        norm(p + "self_attn_layer_norm", a.d_model, True)
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            lin(p + "self_attn." + n, a.d_model, a.d_model, True)
        norm(p + "final_layer_norm", a.d_model, True)
        lin(p + "fc1", a.encoder_ffn_dim, a.d_model, True)
        lin(p + "fc2", a.d_model, a.encoder_ffn_dim, True)
    norm("audio_tower.ln_post", a.d_model, True)
    lin("audio_tower.proj1", a.d_model, a.d_model, True)
    lin("audio_tower.proj2", a.output_dim, a.d_model, True)

    st["model.embed_tokens.weight"] = (torch.randn(t.vocab_size, t.hidden_size, generator=g, device=device) * embed_std).to(torch.bfloat16)
    qd, kd = t.num_attention_heads * t.head_dim, t.num_key_value_heads * t.head_dim
    for i in range(t.num_hidden_layers):
        p = f"model.layers.{i}."
        norm(p + "input_layernorm", t.hidden_size, False)
        lin(p + "self_attn.q_proj", qd, t.hidden_size, False)
        lin(p + "self_attn.k_proj", kd, t.hidden_size, False)
        lin(p + "self_attn.v_proj", kd, t.hidden_size, False)
        lin(p + "self_attn.o_proj", t.hidden_size, qd, False)
        norm(p + "self_attn.q_norm", t.head_dim, False)
        norm(p + "self_attn.k_norm", t.head_dim, False)
        norm(p + "post_attention_layernorm", t.hidden_size, False)
        lin(p + "mlp.gate_proj", t.intermediate_size, t.hidden_size, False)
        lin(p + "mlp.up_proj", t.intermediate_size, t.hidden_size, False)
        lin(p + "mlp.down_proj", t.hidden_size, t.intermediate_size, False)
    norm("model.norm", t.hidden_size, False)
    return st


def config_dict(cfg: ModelConfig) -> dict:
    """config.json in the shape reference config.py:118-128 parses (nested audio, flat text)."""
    a, t = cfg.audio_encoder, cfg.text_decoder
    d = {k: getattr(t, k) for k in ("hidden_size", "num_hidden_layers", "num_attention_heads", "num_key_value_heads",
                                    "head_dim", "intermediate_size", "hidden_act", "vocab_size", "max_position_embeddings",
                                    "rms_norm_eps", "rope_theta", "mrope_section", "rope_interleaved")}
    d["audio_encoder_config"] = {k: getattr(a, k) for k in (
        "d_model", "encoder_layers", "encoder_attention_heads", "encoder_ffn_dim", "num_mel_bins", "max_source_positions",
        "output_dim", "n_window", "n_window_infer", "conv_chunksize", "activation_function", "downsample_hidden_size")}
    d.update(audio_token_id=cfg.audio_token_id, audio_start_token_id=cfg.audio_start_token_id,
             audio_end_token_id=cfg.audio_end_token_id)
    return d


class SyntheticTokenizer:
    """Stand-in for the Qwen BPE files (absent offline): byte-level ids 0..255, everything else is
    rendered as <id>.  " English" encodes to its real single id 6364."""
    _LANG = {" English": [6364]}

    def encode(self, text: str) -> list[int]:
        return list(self._LANG.get(text, text.encode("utf-8")))

    def decode(self, token_ids, skip_special_tokens: bool = True) -> str:
        out = []
        for t in token_ids:
            if t < 256:
                out.append(chr(t) if 32 <= t < 127 else "?")
            elif t >= 151643 and skip_special_tokens:
                continue
            else:
                out.append(f"<{t}>")
        return "".join(out)

    def build_prompt(self, n_audio_tokens: int, language: str = "English") -> list[int]:
        from .tokenizer import build_prompt
        return build_prompt(n_audio_tokens, self.encode(f" {language}"))


def write_model_dir(path, cfg: ModelConfig, seed: int = 0, **kw) -> Path:
    """config.json + model.safetensors + a byte-level tokenizer.json, loadable by from_pretrained."""
    from safetensors.torch import save_file
    from tokenizers import Tokenizer as HFTokenizer, decoders, models, pre_tokenizers
    path = Path(path)
    path.mkdir(parents=True, exist_ok=True)
    (path / "config.json").write_text(json.dumps(config_dict(cfg)))
    save_file({k: v.cpu().contiguous() for k, v in random_state(cfg, seed, **kw).items()}, str(path / "model.safetensors"))
    alphabet = pre_tokenizers.ByteLevel.alphabet()
    vocab = {ch: i for i, ch in enumerate(sorted(alphabet))}
    tok = HFTokenizer(models.BPE(vocab=vocab, merges=[]))
    tok.pre_tokenizer = pre_tokenizers.ByteLevel(add_prefix_space=False)
    tok.decoder = decoders.ByteLevel()
    tok.save(str(path / "tokenizer.json"))
    return path
