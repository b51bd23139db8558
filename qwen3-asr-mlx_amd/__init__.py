"""qwen3-asr on AMD MI355X: the Qwen3ASR.from_pretrained()/transcribe() surface of
gabrimatic/qwen3-asr-mlx over hand-written HIP kernels (libq3asr_hip.so).

Import name: ``qwen3_asr_mlx_amd`` (this directory is ``qwen3-asr-mlx_amd``; the repo-root module
``qwen3_asr_mlx_amd.py`` maps one onto the other).  The public names are the reference's 20 exports
(reference __init__.py:16-37), in its order, each bound from the module that implements it here.
"""
from . import _lib, audio, config, decoder, encoder, generate as _generate, model, ops, tokenizer  # noqa: F401

__version__ = "0.1.0"

_PUBLIC = (
    (audio, ("load_audio", "log_mel_spectrogram")),
    (config, ("AudioEncoderConfig", "TextDecoderConfig", "ModelConfig")),
    (encoder, ("AudioEncoder", "load_encoder_weights")),
    (decoder, ("KVCache", "TextDecoder", "load_decoder_weights")),
    (_generate, ("generate", "prepare_inputs", "sample")),
    (tokenizer, ("Tokenizer", "build_prompt", "parse_output")),
    (model, ("Qwen3ASR", "TranscriptionResult", "LANGUAGE_MAP")),
)
__all__ = ["__version__"]
for _mod, _names in _PUBLIC:
    for _n in _names:
        globals()[_n] = getattr(_mod, _n)
        __all__.append(_n)
generate_batch = _generate.generate_batch          # batch API (new; not part of the reference surface)
generate_continuous = _generate.generate_continuous
from . import serve  # noqa: E402
ContinuousTranscriber = serve.ContinuousTranscriber   # queue front door over refilled decode rows (new)
del _mod, _names, _n
