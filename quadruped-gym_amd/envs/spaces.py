"""Gymnasium spaces when gymnasium is installed, else a minimal duck-typed ``Box`` stand-in
(the reference imports gymnasium at ``src/envs/quadruped.py:5-6``; it is absent from the build image)."""
from __future__ import annotations

import numpy as np

try:  # pragma: no cover - depends on the host
    import gymnasium as gym
    from gymnasium import spaces as _spaces
    Box = _spaces.Box
    EnvBase = gym.Env
    HAVE_GYMNASIUM = True
except Exception:  # gymnasium not installed
    HAVE_GYMNASIUM = False

    class Box:
        def __init__(self, low, high, shape, dtype=np.float32):
            self.shape = tuple(shape)
            self.dtype = np.dtype(dtype)
            self.low = np.full(self.shape, low, dtype=self.dtype)
            self.high = np.full(self.shape, high, dtype=self.dtype)
            self._rng = np.random.default_rng()

        def seed(self, seed=None):
            self._rng = np.random.default_rng(seed)
            return [seed]

        def sample(self):
            lo = np.where(np.isfinite(self.low), self.low, -1.0)
            hi = np.where(np.isfinite(self.high), self.high, 1.0)
            return self._rng.uniform(lo, hi).astype(self.dtype)

        def contains(self, x):
            x = np.asarray(x)
            return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))

        def __repr__(self):
            return f"Box({self.low.min()}, {self.high.max()}, {self.shape}, {self.dtype})"

    class EnvBase:
        metadata = {}

        def __init__(self):
            pass
