"""Seeded synthetic panoramas (no datasets are reachable): the two distributions SURVEY.md 8(d) names.

  kind "S": band-limited -- per channel 127 + sum of 4 separable sinusoids (spatial frequency
            <= width/64 cycles) + a linear ramp; neighbouring pixels differ by at most ~12 levels,
            so a 1/32-pixel coordinate difference moves a channel by less than half a level.
  kind "N": uniform noise in 0..255 -- the stress input for the bit-exact integer path.
Panorama i of a batch uses seed 1000 + i.
"""
import numpy as np


def synth_pano(pw, ph, seed=1000, kind="S"):
    rng = np.random.default_rng(seed)
    if kind == "N":
        return rng.integers(0, 256, size=(ph, pw, 3), dtype=np.uint8)
    if kind != "S":
        raise ValueError("kind must be 'S' or 'N'")
    x = np.arange(pw, dtype=np.float32)
    y = np.arange(ph, dtype=np.float32)
    out = np.empty((ph, pw, 3), dtype=np.uint8)
    kmax = max(1, pw // 64)
    for c in range(3):
        acc = np.full((ph, pw), 127.0, dtype=np.float32)
        for _ in range(4):
            kx = int(rng.integers(1, kmax + 1))
            ky = int(rng.integers(1, kmax + 1))
            amp = np.float32(rng.uniform(8.0, 24.0))
            phx, phy = rng.uniform(0.0, 2 * np.pi, size=2)
            sx = np.sin(2 * np.pi * kx * x / pw + phx).astype(np.float32)
            cy = (amp * np.cos(np.pi * ky * y / ph + phy)).astype(np.float32)
            acc += cy[:, None] * sx[None, :]
        acc += (20.0 * (x / pw - 0.5)).astype(np.float32)[None, :]
        acc += (10.0 * (y / ph - 0.5)).astype(np.float32)[:, None]
        np.rint(acc, out=acc)
        np.clip(acc, 0, 255, out=acc)
        out[:, :, c] = acc.astype(np.uint8)
    return out
