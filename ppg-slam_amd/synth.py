"""Deterministic synthetic 8-bit test images for the PPG front-end.

The reference ships no images (only misc/SystemOverview.png), so parity and
throughput runs use seeded rectangle/line scenes.  A *scene* is a canvas larger
than the frame; frame ``t`` of a stream is the crop of that canvas displaced by
``(3*t, 1*t)`` pixels, so consecutive frames have a known, non-trivial match.

The PRNG is splitmix64 so the same seed gives the same bytes on every host.
"""
import numpy as np

_MASK = (1 << 64) - 1


class SplitMix64:
    def __init__(self, seed: int):
        self.s = seed & _MASK

    def next(self) -> int:
        self.s = (self.s + 0x9E3779B97F4A7C15) & _MASK
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _MASK
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _MASK
        return z ^ (z >> 31)

    def below(self, n: int) -> int:
        """Uniform integer in [0, n)."""
        return self.next() % n

    def between(self, lo: int, hi: int) -> int:
        """Uniform integer in [lo, hi]."""
        return lo + self.below(hi - lo + 1)


def _draw_line(img, x0, y0, x1, y1, g):
    """1-px Bresenham segment, clipped to the canvas."""
    h, w = img.shape
    dx, dy = abs(x1 - x0), -abs(y1 - y0)
    sx = 1 if x0 < x1 else -1
    sy = 1 if y0 < y1 else -1
    err = dx + dy
    while True:
        if 0 <= x0 < w and 0 <= y0 < h:
            img[y0, x0] = g
        if x0 == x1 and y0 == y1:
            break
        e2 = 2 * err
        if e2 >= dy:
            err += dy
            x0 += sx
        if e2 <= dx:
            err += dx
            y0 += sy


MARGIN_X = 96   # canvas margin so that up to 32 displaced frames fit
MARGIN_Y = 32


def make_scene(height: int, width: int, seed: int, kind: str = "sparse") -> np.ndarray:
    """Canvas of shape (height+MARGIN_Y, width+MARGIN_X), uint8."""
    ch, cw = height + MARGIN_Y, width + MARGIN_X
    rng = SplitMix64(seed)
    img = np.full((ch, cw), 128, dtype=np.uint8)
    if kind == "sparse":
        for _ in range(60):
            rw = rng.between(20, max(21, cw // 3))
            rh = rng.between(20, max(21, ch // 3))
            x = rng.between(-rw // 2, cw - rw // 2)
            y = rng.between(-rh // 2, ch - rh // 2)
            g = rng.below(256)
            img[max(0, y):max(0, y + rh), max(0, x):max(0, x + rw)] = g
    elif kind == "dense":
        # small rectangles on a jittered lattice: many isolated corners
        step = 20
        for gy in range(0, ch, step):
            for gx in range(0, cw, step):
                rw = rng.between(7, 13)
                rh = rng.between(7, 13)
                x = gx + rng.between(0, 5)
                y = gy + rng.between(0, 5)
                g = rng.below(256)
                img[y:y + rh, x:x + rw] = g
    else:
        raise ValueError(kind)
    for _ in range(40):
        _draw_line(img, rng.below(cw), rng.below(ch), rng.below(cw), rng.below(ch), rng.below(256))
    return img


def frame_from_scene(scene: np.ndarray, height: int, width: int, t: int) -> np.ndarray:
    """Frame ``t`` (0 <= t < 32): crop displaced by (3t, t) px, contiguous uint8."""
    t = t % 32
    ox, oy = 3 * t, t
    return np.ascontiguousarray(scene[oy:oy + height, ox:ox + width])


def make_frame(height: int, width: int, seed: int, t: int = 0, kind: str = "sparse") -> np.ndarray:
    return frame_from_scene(make_scene(height, width, seed, kind), height, width, t)


def random_map_graph(rng, desc, pos, m, n_medges, key_edges=None):
    """A synthetic local map for Matcher::ExtendMapMatches: m map points sampled from a frame's key points (descriptor
    and projection perturbed) and n_medges random map edges between them, as the flattened arrays of ppg_mapgraph.
    With `key_edges` (the frame's [E][2] key edges) every key edge whose two ends were sampled becomes a map edge as
    well: the map then repeats the frame's own graph, and the seed growing finds consistent neighbours to walk to."""
    n = desc.shape[0]
    src = rng.choice(n, size=m, replace=False)
    mdesc = desc[src] + 0.15 * rng.normal(size=(m, 256)).astype(np.float32) / 16
    mdesc = (mdesc / np.linalg.norm(mdesc, axis=1, keepdims=True)).astype(np.float32)
    inc = [[] for _ in range(m)]
    for eid in range(n_medges):
        a, b = rng.choice(m, size=2, replace=False)
        inc[a].append((eid, int(b)))
        inc[b].append((eid, int(a)))
    if key_edges is not None:
        where = np.full(n, -1, np.int64)
        where[src] = np.arange(m)
        eid = n_medges
        for s, e in np.asarray(key_edges).reshape(-1, 2):
            a, b = int(where[s]), int(where[e])
            if a >= 0 and b >= 0:
                inc[a].append((eid, b))
                inc[b].append((eid, a))
                eid += 1
    ne = sum(len(i) for i in inc)
    other = np.array([o for i in inc for _, o in i], np.int32)
    other[rng.random(ne) < 0.05] = -1
    return dict(desc=mdesc, proj=(pos[src] + rng.uniform(-4, 4, (m, 2))).astype(np.float32),
                viewcos=rng.uniform(0.99, 1.0, m).astype(np.float32), in_view=(rng.random(m) < 0.8).astype(np.uint8),
                bad=(rng.random(m) < 0.05).astype(np.uint8), nobs=rng.integers(0, 3, m).astype(np.int32),
                edge_off=np.cumsum([0] + [len(i) for i in inc]).astype(np.int32),
                edge_id=np.array([e for i in inc for e, _ in i], np.int32), edge_other=other,
                edge_ok=(rng.random(ne) < 0.9).astype(np.uint8))
