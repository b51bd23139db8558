"""Seeded test signals shared by tests, bench and the golden generator (same as tools/gen_golden.signal)."""
import numpy as np


def signal(kind: str, n: int, seed: int = None) -> np.ndarray:
    if kind == "silence":
        return np.zeros(n, dtype=np.float32)
    if kind == "tone":
        t = np.arange(n, dtype=np.float64) / 16000.0
        return (0.5 * np.sin(2 * np.pi * 440.0 * t)).astype(np.float32)
    if kind == "noise":
        return (np.random.default_rng(0 if seed is None else seed).standard_normal(n) * 0.1).astype(np.float32)
    if kind == "speechlike":
        rng = np.random.default_rng(1234 if seed is None else seed)
        t = np.arange(n, dtype=np.float64) / 16000.0
        env = 0.5 * (1.0 - np.cos(2 * np.pi * 3.0 * t))
        gap = ((t % 6.0) > 5.6).astype(np.float64)
        return (0.1 * rng.standard_normal(n) * env * (1.0 - 0.999 * gap)).astype(np.float32)
    raise ValueError(kind)
