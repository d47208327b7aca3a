"""Second, independent restatement of Map::GenerateHeightBitmap (/root/reference/src/map/Map.cpp:144-262) -- TEST
INFRASTRUCTURE: only tests/ may import it; the product's generator is vrc_scene_diamond_square (csrc/svo_builder.cpp), with
which this file shares no code (not even the C++ <random> library: the Mersenne Twister, generate_canonical and the
distribution are written out here from their published definitions).

Parity unpinned against the reference itself: Map.cpp needs SFML (absent from the image), so the function cannot be compiled
here; what this file pins is the product against a second reading of the same source.  Known answers it is checked
against (tests/test_oracle_cpu.py): the C++ standard's requirement that the 10000th output of a default-constructed
std::mt19937 is 4123659995, and glibc's first rand() of an unseeded process (1804289383 -> corner seed 58, Map.cpp:160).

What the source does, line by line:
  :146-148  std::mt19937 gen (default seed 5489); std::uniform_real_distribution<double> dis(-1, 1)
  :157      DATA_SIZE = dimensions.x + 1 -- but Sample/SetSample (:265-271) wrap with `& (dimensions - 1)`
  :160      SEED = rand() % 10 + 55 (unseeded rand(): 58)
  :163-166  the four "corners" -- with the wrap all four are sample (0, 0)
  :168-180  for sideLength = DATA_SIZE - 1; sideLength >= 2; sideLength /= 2, h /= 2  (h starts at 20)
  :187-203  squares: centre = mean of the 4 corners + (f_rand() * 2 * h) - h, x outer, y inner
  :210-241  diamonds: x steps by halfSide, y starts at (x + halfSide) % sideLength and steps by sideLength; the neighbours are
            taken `% DATA_SIZE` (NOT the array size: the left neighbour of x = 0 is sample DATA_SIZE - halfSide, which the
            `&` then maps to an odd column that has not been written yet -- reproduced as written); the two edge copies
            (:238-239) land on the sample itself
  :248      height = uint8(min(max(v, 0), dimensions.z))
"""
import numpy as np


class MT19937:
    """MT19937 (Matsumoto & Nishimura 1998) as std::mt19937 instantiates it: w = 32, n = 624, m = 397, r = 31,
    a = 0x9908b0df, tempering (11, 0xffffffff), (7, 0x9d2c5680), (15, 0xefc60000), 18, seeding multiplier 1812433253."""

    def __init__(self, seed: int = 5489):
        mt = np.zeros(624, dtype=np.uint64)
        mt[0] = seed & 0xFFFFFFFF
        for i in range(1, 624):
            prev = int(mt[i - 1])
            mt[i] = (1812433253 * (prev ^ (prev >> 30)) + i) & 0xFFFFFFFF
        self.mt = mt.astype(np.uint32)
        self.out = np.zeros(0, dtype=np.uint32)
        self.pos = 0

    def _twist(self):
        mt = self.mt
        # the recurrence reads words written earlier in the same pass: three vectorisable spans
        def span(lo, hi, off):
            y = (mt[lo:hi] & np.uint32(0x80000000)) | (mt[lo + 1:hi + 1] & np.uint32(0x7FFFFFFF))
            mag = np.where(y & np.uint32(1), np.uint32(0x9908B0DF), np.uint32(0))
            mt[lo:hi] = mt[lo + off:hi + off] ^ (y >> np.uint32(1)) ^ mag
        span(0, 227, 397)                   # k + 397 < 624
        span(227, 454, -227)                # k + 397 - 624 in [0, 227): already new
        span(454, 623, -227)                # ... in [227, 396): already new
        y = (mt[623] & np.uint32(0x80000000)) | (mt[0] & np.uint32(0x7FFFFFFF))
        mt[623] = mt[396] ^ (y >> np.uint32(1)) ^ (np.uint32(0x9908B0DF) if (y & np.uint32(1)) else np.uint32(0))
        y = mt.copy()
        y ^= y >> np.uint32(11)
        y ^= (y << np.uint32(7)) & np.uint32(0x9D2C5680)
        y ^= (y << np.uint32(15)) & np.uint32(0xEFC60000)
        y ^= y >> np.uint32(18)
        self.out, self.pos = y, 0

    def words(self, count: int) -> np.ndarray:
        """The next `count` 32-bit outputs."""
        parts, need = [], count
        while need > 0:
            if self.pos >= self.out.size:
                self._twist()
            take = min(need, self.out.size - self.pos)
            parts.append(self.out[self.pos:self.pos + take])
            self.pos += take
            need -= take
        return np.concatenate(parts) if parts else np.zeros(0, dtype=np.uint32)


def uniform_minus1_1(gen: MT19937, count: int) -> np.ndarray:
    """std::uniform_real_distribution<double>(-1, 1) over std::mt19937, `count` draws: generate_canonical<double, 53> takes
    two 32-bit words (low word first), sum = w0 + w1 * 2^32 in double arithmetic (rounds to 53 bits), canonical = sum / 2^64,
    clamped below 1; the distribution returns canonical * (b - a) + a."""
    w = gen.words(2 * count).astype(np.float64).reshape(count, 2)
    s = w[:, 0] + w[:, 1] * 4294967296.0
    c = s / 18446744073709551616.0
    c = np.where(c >= 1.0, np.nextafter(1.0, 0.0), c)
    return c * 2.0 + (-1.0)


def height_field(dim: int, corner_seed: float = 58.0) -> np.ndarray:
    """The double field of Map::GenerateHeightBitmap for dimensions = (dim, dim, dim), indexed [y, x] (x + y * dim)."""
    n, size = dim, dim + 1
    hm = np.zeros(n * n, dtype=np.float64)
    mask = n - 1
    gen = MT19937()

    def idx(x, y):
        return (x & mask) + (y & mask) * n

    for cx, cy in ((0, 0), (0, n), (n, 0), (n, n)):
        hm[idx(cx, cy)] = corner_seed
    h = 20.0
    side = size - 1
    while side >= 2:
        half = side // 2
        # squares: the draws are consumed x outer, y inner
        xs = range(0, size - 1, side)
        r = uniform_minus1_1(gen, len(xs) * len(xs))
        k = 0
        for x in xs:
            for y in range(0, size - 1, side):
                avg = hm[idx(x, y)] + hm[idx(x + side, y)] + hm[idx(x, y + side)] + hm[idx(x + side, y + side)]
                avg /= 4.0
                hm[idx(x + half, y + half)] = avg + (r[k] * 2 * h) - h
                k += 1
        # diamonds
        cells = [(x, y) for x in range(0, size - 1, half) for y in range((x + half) % side, size - 1, side)]
        r = uniform_minus1_1(gen, len(cells))
        for k, (x, y) in enumerate(cells):
            avg = (hm[idx((x - half + size) % size, y)] + hm[idx((x + half) % size, y)]
                   + hm[idx(x, (y + half) % size)] + hm[idx(x, (y - half + size) % size)])
            avg /= 4.0
            avg = avg + (r[k] * 2 * h) - h
            hm[idx(x, y)] = avg
            if x == 0:
                hm[idx(size - 1, y)] = avg
            if y == 0:
                hm[idx(x, size - 1)] = avg
        side //= 2
        h /= 2.0
    return hm.reshape(n, n)


def height_bytes(dim: int, corner_seed: float = 58.0) -> np.ndarray:
    """:248 -- the uint8 the reference stores per column: min(max(v, 0), dimensions.z) truncated (dimensions.z capped at 255: a
    double above 255 has no defined uint8 value)."""
    v = np.minimum(np.maximum(height_field(dim, corner_seed), 0.0), float(min(dim, 255)))
    return v.astype(np.uint8)
