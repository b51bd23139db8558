"""Import shim: the product package lives in the directory ``qwen3-asr-mlx_amd`` (not a valid
Python identifier); ``import qwen3_asr_mlx_amd`` loads it under this name."""
import importlib.util
import sys
from pathlib import Path

_pkg = Path(__file__).resolve().parent / "qwen3-asr-mlx_amd"
_spec = importlib.util.spec_from_file_location(__name__, _pkg / "__init__.py", submodule_search_locations=[str(_pkg)])
_mod = importlib.util.module_from_spec(_spec)
sys.modules[__name__] = _mod
_spec.loader.exec_module(_mod)
