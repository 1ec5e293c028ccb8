"""Height-field terrain for the arenas (host side, numpy).

What the reference builds with ``Terrain`` (src/my_utils/origin_terrain_generator.py:37-183) and
the sub-terrain generators of src/my_utils/origin_terrain_utils.py:39-347: ONE int16 height
field shared by all robots — ``numLevels x numTerrains`` tiles of ``mapLength x mapWidth``
metres inside a flat border — plus the tile origins the robots are placed on. Every call is
commented out in the reference's task (KAMMA.py:296-349), its runs are on the flat plane; here
the field feeds the step kernel's wheel contact and ``get_heights`` (include/ois.h
``ois_terrain_attach``).

The families are restated as closed forms over the pixel grid (no Python loops over pixels or
rings); the raw values are the reference's integers: heights in units of ``vertical_scale``,
pixels of ``horizontal_scale``, ``int()`` truncation and ``astype(int16)`` exactly where the
reference has them. Families that draw random numbers take a ``numpy.random.RandomState`` and
call it in the reference's order (the reference uses numpy's global state, which it never
seeds: its fields are not reproducible, so nothing here can be pinned to one).
"""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np


class SubTerrain:
    """One tile: ``height_field_raw`` int16 [width, length] (origin_terrain_utils.py:437-446)."""

    def __init__(self, terrain_name: str = "terrain", width: int = 256, length: int = 256,
                 vertical_scale: float = 1.0, horizontal_scale: float = 1.0):
        self.terrain_name = terrain_name
        self.vertical_scale = vertical_scale
        self.horizontal_scale = horizontal_scale
        self.width = int(width)
        self.length = int(length)
        self.height_field_raw = np.zeros((self.width, self.length), dtype=np.int16)


def _ix(t: SubTerrain) -> np.ndarray:
    return np.arange(t.width).reshape(t.width, 1)


def _iy(t: SubTerrain) -> np.ndarray:
    return np.arange(t.length).reshape(1, t.length)


def flat_terrain(terrain: SubTerrain, height: float = 0.0) -> SubTerrain:
    """origin_terrain_utils.py:86-111: the whole tile raised by int(height / vertical_scale)."""
    terrain.height_field_raw += np.int16(int(height / terrain.vertical_scale))
    return terrain


def sloped_terrain(terrain: SubTerrain, slope: float = 1) -> SubTerrain:
    """:114-133: a ramp along x, raw = trunc(max_height * ix / width)."""
    max_height = int(slope * (terrain.horizontal_scale / terrain.vertical_scale) * terrain.width)
    ramp = (max_height * _ix(terrain) / terrain.width).astype(terrain.height_field_raw.dtype)
    terrain.height_field_raw += ramp  # broadcast over the columns
    return terrain


def pyramid_sloped_terrain(terrain: SubTerrain, slope: float = 1, platform_size: float = 1.0) -> SubTerrain:
    """:136-168: product of two tents, flattened at the height the tent has at the platform corner."""
    cx, cy = int(terrain.width / 2), int(terrain.length / 2)
    tent_x = (cx - np.abs(cx - _ix(terrain))) / cx
    tent_y = (cy - np.abs(cy - _iy(terrain))) / cy
    max_height = int(slope * (terrain.horizontal_scale / terrain.vertical_scale) * (terrain.width / 2))
    terrain.height_field_raw += (max_height * tent_x * tent_y).astype(terrain.height_field_raw.dtype)
    half = int(platform_size / terrain.horizontal_scale / 2)
    corner = terrain.height_field_raw[terrain.width // 2 - half, terrain.length // 2 - half]
    terrain.height_field_raw = np.clip(terrain.height_field_raw, min(corner, 0), max(corner, 0))
    return terrain


def stairs_terrain(terrain: SubTerrain, step_width: float, step_height: float) -> SubTerrain:
    """:235-255: step i (of width // step_width) along x is raised by (i + 1) * step_height."""
    sw = int(step_width / terrain.horizontal_scale)
    sh = int(step_height / terrain.vertical_scale)
    level = _ix(terrain) // sw
    rise = np.where(level < terrain.width // sw, (level + 1) * sh, 0)
    terrain.height_field_raw += rise.astype(np.int16)
    return terrain


def pyramid_stairs_terrain(terrain: SubTerrain, step_width: float, step_height: float,
                           platform_size: float = 1.0) -> SubTerrain:
    """:258-287: concentric square rings, ring k (k >= 1) SET to k * step_height. A pixel's ring
    is how many step widths it is away from the nearest tile edge, capped by the number of
    rings the reference's while-loop draws before the inner square is no larger than the
    platform."""
    sw = int(step_width / terrain.horizontal_scale)
    sh = int(step_height / terrain.vertical_scale)
    platform = int(platform_size / terrain.horizontal_scale)
    W, L = terrain.width, terrain.length
    rings = 0  # iterations of `while (stop_x - start_x) > platform and (stop_y - start_y) > platform`
    while (W - 2 * rings * sw) > platform and (L - 2 * rings * sw) > platform:
        rings += 1
    ix, iy = _ix(terrain), _iy(terrain)
    ring = np.minimum(np.minimum(ix // sw, (W - 1 - ix) // sw), np.minimum(iy // sw, (L - 1 - iy) // sw))
    ring = np.minimum(ring, rings)
    terrain.height_field_raw = np.where(ring >= 1, ring * sh, terrain.height_field_raw).astype(np.int16)
    return terrain


def discrete_obstacles_terrain(terrain: SubTerrain, max_height: float, min_size: float, max_size: float,
                               num_rects: int, platform_size: float = 1.0,
                               rng: Optional[np.random.RandomState] = None) -> SubTerrain:
    """:171-208: num_rects axis-aligned boxes of height in {-H, -H//2, H//2, H}, later boxes
    overwrite earlier ones, the central platform is cleared. Draw order per box: width, length,
    start_i, start_j, height."""
    rng = rng or np.random
    H = int(max_height / terrain.vertical_scale)
    lo = int(min_size / terrain.horizontal_scale)
    hi = int(max_size / terrain.horizontal_scale)
    platform = int(platform_size / terrain.horizontal_scale)
    rows, cols = terrain.height_field_raw.shape
    heights = [-H, -H // 2, H // 2, H]
    sizes = range(lo, hi, 4)
    for _ in range(num_rects):
        w = rng.choice(sizes)
        l = rng.choice(sizes)
        i0 = rng.choice(range(0, rows - w, 4))
        j0 = rng.choice(range(0, cols - l, 4))
        terrain.height_field_raw[i0:i0 + w, j0:j0 + l] = rng.choice(heights)
    x1, x2 = (terrain.width - platform) // 2, (terrain.width + platform) // 2
    y1, y2 = (terrain.length - platform) // 2, (terrain.length + platform) // 2
    terrain.height_field_raw[x1:x2, y1:y2] = 0
    return terrain


def random_uniform_terrain(terrain: SubTerrain, min_height: float, max_height: float, step: float = 1,
                           downsampled_scale: Optional[float] = None,
                           rng: Optional[np.random.RandomState] = None) -> SubTerrain:
    """:39-83: noise drawn on a coarse grid, spline-interpolated to the pixel grid, rounded."""
    from scipy import interpolate
    rng = rng or np.random
    if downsampled_scale is None:
        downsampled_scale = terrain.horizontal_scale
    lo = int(min_height / terrain.vertical_scale)
    hi = int(max_height / terrain.vertical_scale)
    st = int(step / terrain.vertical_scale)
    levels = np.arange(lo, hi + st, st)
    extent_w = terrain.width * terrain.horizontal_scale
    extent_l = terrain.length * terrain.horizontal_scale
    coarse = rng.choice(levels, (int(extent_w / downsampled_scale), int(extent_l / downsampled_scale)))
    xs = np.linspace(0, extent_w, coarse.shape[0])
    ys = np.linspace(0, extent_l, coarse.shape[1])
    spline = interpolate.RectBivariateSpline(ys, xs, coarse)
    fine = np.rint(spline(np.linspace(0, extent_l, terrain.length), np.linspace(0, extent_w, terrain.width)))
    terrain.height_field_raw += fine.astype(np.int16)
    return terrain


def wave_terrain(terrain: SubTerrain, num_waves: int = 1, amplitude: float = 1.0) -> SubTerrain:
    """:211-232: A cos(iy / div) + A sin(ix / div), div = length / (2 pi num_waves)."""
    A = int(0.5 * amplitude / terrain.vertical_scale)
    if num_waves > 0:
        div = terrain.length / (num_waves * np.pi * 2)
        field = A * np.cos(_iy(terrain) / div) + A * np.sin(_ix(terrain) / div)
        terrain.height_field_raw += field.astype(terrain.height_field_raw.dtype)
    return terrain


def stepping_stones_terrain(terrain: SubTerrain, stone_size: float, stone_distance: float, max_height: float,
                            platform_size: float = 1.0, depth: float = -10,
                            rng: Optional[np.random.RandomState] = None) -> SubTerrain:
    """:290-346: rows (or columns, for wide tiles) of square stones over a pit, each row shifted
    by a random offset, each stone at a random height; central platform at 0."""
    rng = rng or np.random
    size = int(stone_size / terrain.horizontal_scale)
    gap = int(stone_distance / terrain.horizontal_scale)
    H = int(max_height / terrain.vertical_scale)
    platform = int(platform_size / terrain.horizontal_scale)
    levels = np.arange(-H - 1, H, step=1)
    f = terrain.height_field_raw
    f[:, :] = int(depth / terrain.vertical_scale)
    along_y = terrain.length >= terrain.width
    outer_n, inner_n = (terrain.length, terrain.width) if along_y else (terrain.width, terrain.length)

    def put(o0, o1, i0, i1, value):  # outer = y when the strips run along x
        if along_y:
            f[i0:i1, o0:o1] = value
        else:
            f[o0:o1, i0:i1] = value

    o = 0
    while o < outer_n:
        o_end = min(outer_n, o + size)
        i = rng.randint(0, size)
        put(o, o_end, 0, max(0, i - gap), rng.choice(levels))  # partial stone before the first gap
        while i < inner_n:
            put(o, o_end, i, min(inner_n, i + size), rng.choice(levels))
            i += size + gap
        o += size + gap
    x1, x2 = (terrain.width - platform) // 2, (terrain.width + platform) // 2
    y1, y2 = (terrain.length - platform) // 2, (terrain.length + platform) // 2
    f[x1:x2, y1:y2] = 0
    return terrain


DEFAULT_TERRAIN_CFG: Dict[str, object] = {
    # TASK_CFG["task"]["env"]["terrain"] (KAMMA.py:51-66)
    "staticFriction": 0.85, "dynamicFriction": 0.85, "restitution": 0.0, "curriculum": False,
    "maxInitMapLevel": 0, "mapLength": 10.0, "mapWidth": 10.0, "numLevels": 6, "numTerrains": 2,
    "slopeTreshold": 0.5,
    # read by Terrain.__init__ (:46) but absent from the reference's TASK_CFG
    "terrainProportions": [0.1, 0.5, 0.4],
    # build-defined, not a key of the reference: ois_terrain.footprint_stride (include/ois.h; docs/SPEC.md section 3.8)
    "footprintStride": 1,
}


class Terrain:
    """The global height field and the tile origins (origin_terrain_generator.py:37-183).

    Attributes as there: ``horizontal_scale`` 0.02, ``vertical_scale`` 0.006, ``border_size`` 20,
    ``env_rows`` / ``env_cols``, ``height_field_raw`` = ``heightsamples`` int16
    [tot_rows, tot_cols], ``env_origins`` float [env_rows, env_cols, 3]. The triangle mesh the
    reference hands to PhysX is not built: the step kernel samples the field itself.
    """

    def __init__(self, cfg: Optional[Dict[str, object]] = None, num_robots: int = 1, seed: Optional[int] = None):
        cfg = {**DEFAULT_TERRAIN_CFG, **(cfg or {})}
        self.cfg = cfg
        self.rng = np.random.RandomState(seed)
        self.horizontal_scale = 0.02
        self.vertical_scale = 0.006
        self.border_size = 20
        self.env_length = cfg["mapLength"]
        self.env_width = cfg["mapWidth"]
        props = list(cfg["terrainProportions"])
        self.proportions = [float(np.sum(props[: i + 1])) for i in range(len(props))]
        self.slopeThr = cfg["slopeTreshold"]
        self.env_rows = int(cfg["numLevels"])
        self.env_cols = int(cfg["numTerrains"])
        self.num_maps = self.env_rows * self.env_cols
        self.num_per_env = int(num_robots / self.num_maps)
        self.env_origins = np.zeros((self.env_rows, self.env_cols, 3))
        self.width_per_env_pixels = int(self.env_width / self.horizontal_scale)
        self.length_per_env_pixels = int(self.env_length / self.horizontal_scale)
        self.border = int(self.border_size / self.horizontal_scale)
        self.tot_cols = int(self.env_cols * self.width_per_env_pixels) + 2 * self.border
        self.tot_rows = int(self.env_rows * self.length_per_env_pixels) + 2 * self.border
        self.height_field_raw = np.zeros((self.tot_rows, self.tot_cols), dtype=np.int16)
        if cfg["curriculum"]:
            self.curiculum(num_robots, num_terrains=self.env_cols, num_levels=self.env_rows)
        else:
            self.randomized_terrain()
        self.heightsamples = self.height_field_raw

    # -- tile placement ---------------------------------------------------------------------
    def _new_tile(self) -> SubTerrain:
        # the reference passes width_per_env_pixels for both sides (:81-87, :116-122)
        return SubTerrain("terrain", width=self.width_per_env_pixels, length=self.width_per_env_pixels,
                          vertical_scale=self.vertical_scale, horizontal_scale=self.horizontal_scale)

    def _place(self, i: int, j: int, tile: SubTerrain) -> None:
        x0 = self.border + i * self.length_per_env_pixels
        y0 = self.border + j * self.width_per_env_pixels
        self.height_field_raw[x0:x0 + self.length_per_env_pixels, y0:y0 + self.width_per_env_pixels] = tile.height_field_raw
        # origin: tile centre, at the highest point of the central 2 m x 2 m (:104-110, :170-176)
        x1 = int((self.env_length / 2.0 - 1) / self.horizontal_scale)
        x2 = int((self.env_length / 2.0 + 1) / self.horizontal_scale)
        y1 = int((self.env_width / 2.0 - 1) / self.horizontal_scale)
        y2 = int((self.env_width / 2.0 + 1) / self.horizontal_scale)
        z = np.max(tile.height_field_raw[x1:x2, y1:y2]) * self.vertical_scale
        self.env_origins[i, j] = [(i + 0.5) * self.env_length, (j + 0.5) * self.env_width, z]

    def randomized_terrain(self) -> None:
        """:71-110: every tile draws its family: 10 % pyramid slope (half of them roughened),
        50 % pyramid stairs of +-0.15 m, 40 % discrete obstacles."""
        rng = self.rng
        for k in range(self.num_maps):
            i, j = np.unravel_index(k, (self.env_rows, self.env_cols))
            tile = self._new_tile()
            choice = rng.uniform(0, 1)
            if choice < 0.1:
                rough = rng.choice([0, 1])
                pyramid_sloped_terrain(tile, rng.choice([-0.3, -0.2, 0, 0.2, 0.3]))
                if rough:
                    random_uniform_terrain(tile, min_height=-0.1, max_height=0.1, step=0.05, downsampled_scale=0.2, rng=rng)
            elif choice < 0.6:
                pyramid_stairs_terrain(tile, step_width=0.31, step_height=rng.choice([-0.15, 0.15]), platform_size=3.0)
            else:
                discrete_obstacles_terrain(tile, 0.15, 1.0, 2.0, 40, platform_size=3.0, rng=rng)
            self._place(int(i), int(j), tile)

    def curiculum(self, num_robots: int, num_terrains: int, num_levels: int) -> None:
        """:112-176: level i = row, type j = column. Level 0 is flat for both types; from level 1
        on type 0 is descending pyramid stairs (1 m steps), type 1 a descending pyramid slope,
        both scaled by difficulty = i / num_levels. Only levels 0-3 of types 0-1 are drawn
        (the reference's if-chain), further tiles stay flat."""
        for j in range(num_terrains):
            for i in range(num_levels):
                tile = self._new_tile()
                difficulty = i / num_levels
                slope = -difficulty * 0.4
                step_height = -(0.1175 * num_levels / (num_levels - 1)) * difficulty
                if i == 0 and j in (0, 1):
                    flat_terrain(tile, height=0.0)
                elif i in (1, 2, 3) and j == 0:
                    pyramid_stairs_terrain(tile, step_width=1.0, step_height=step_height, platform_size=3.0)
                elif i in (1, 2, 3) and j == 1:
                    pyramid_sloped_terrain(tile, slope=slope, platform_size=3.0)
                self._place(i, j, tile)

    # -- hand-over to the environment ---------------------------------------------------------
    def sample_levels_types(self, num_envs: int, max_init_level: Optional[int] = None,
                            rng: Optional[np.random.RandomState] = None) -> np.ndarray:
        """terrain_levels ~ randint(0, maxInitMapLevel + 1), terrain_types ~ randint(0, numTerrains)
        (KAMMA.py:340-347; without the curriculum maxInitMapLevel = numLevels - 1). int32 [N, 2]."""
        rng = rng or self.rng
        if max_init_level is None:
            max_init_level = int(self.cfg["maxInitMapLevel"]) if self.cfg["curriculum"] else self.env_rows - 1
        lv = rng.randint(0, max_init_level + 1, size=num_envs)
        ty = rng.randint(0, self.env_cols, size=num_envs)
        return np.stack([lv, ty], axis=1).astype(np.int32)


def height_at(field: np.ndarray, x, y, horizontal_scale: float = 0.02, vertical_scale: float = 0.006,
              border_size: float = 20.0) -> np.ndarray:
    """Ground height (metres) at world (x, y) by the rule of get_heights (KAMMA.py:882-893)."""
    px = np.clip(((np.asarray(x, np.float32) + np.float32(border_size)) / np.float32(horizontal_scale)).astype(np.int64),
                 0, field.shape[0] - 2)
    py = np.clip(((np.asarray(y, np.float32) + np.float32(border_size)) / np.float32(horizontal_scale)).astype(np.int64),
                 0, field.shape[1] - 2)
    return np.minimum(field[px, py], field[px + 1, py + 1]).astype(np.float32) * np.float32(vertical_scale)
