"""`Box` space. The reference uses gym.spaces.Box (KAMMA.py:5, 158-169); gym / gymnasium are
not part of this image, so a minimal stand-in with the same attributes is used when neither
imports. Only .low/.high/.shape/.dtype/.sample()/.contains() are provided."""
from __future__ import annotations

import numpy as np

try:  # pragma: no cover - not installed in the build image
    from gymnasium.spaces import Box  # type: ignore
except Exception:  # noqa: BLE001
    try:  # pragma: no cover
        from gym.spaces import Box  # type: ignore
    except Exception:  # noqa: BLE001

        class Box:  # type: ignore[no-redef]
            def __init__(self, low, high, shape=None, dtype=np.float32):
                self.dtype = np.dtype(dtype)
                self.shape = tuple(shape) if shape is not None else np.shape(low)
                self.low = np.full(self.shape, low, dtype=self.dtype)
                self.high = np.full(self.shape, high, dtype=self.dtype)
                self._rng = np.random.default_rng()

            def seed(self, seed=None):
                self._rng = np.random.default_rng(seed)

            def sample(self):
                return self._rng.uniform(self.low, self.high).astype(self.dtype)

            def contains(self, x) -> bool:
                x = np.asarray(x)
                return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))

            def __repr__(self):
                return f"Box({self.low.min()}, {self.high.max()}, {self.shape}, {self.dtype})"
