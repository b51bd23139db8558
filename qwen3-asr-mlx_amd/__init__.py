"""qwen3-asr on AMD MI355X: the Qwen3ASR.from_pretrained()/transcribe() surface of
gabrimatic/qwen3-asr-mlx over hand-written HIP kernels (libq3asr_hip.so).

Import name: ``qwen3_asr_mlx_amd`` (this directory is ``qwen3-asr-mlx_amd``; the repo-root module
``qwen3_asr_mlx_amd.py`` maps one onto the other).  Same 20 exports as reference __init__.py:16-37.
"""
__version__ = "0.1.0"

from .audio import load_audio, log_mel_spectrogram
from .config import AudioEncoderConfig, ModelConfig, TextDecoderConfig
from .decoder import KVCache, TextDecoder, load_decoder_weights
from .encoder import AudioEncoder, load_encoder_weights
from .generate import generate, generate_batch, prepare_inputs, sample
from .model import LANGUAGE_MAP, Qwen3ASR, TranscriptionResult
from .tokenizer import Tokenizer, build_prompt, parse_output

__all__ = [
    "__version__", "load_audio", "log_mel_spectrogram", "AudioEncoderConfig", "TextDecoderConfig", "ModelConfig",
    "AudioEncoder", "load_encoder_weights", "KVCache", "TextDecoder", "load_decoder_weights", "generate",
    "prepare_inputs", "sample", "Tokenizer", "build_prompt", "parse_output", "Qwen3ASR", "TranscriptionResult",
    "LANGUAGE_MAP",
]
