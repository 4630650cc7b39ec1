"""Minimal stand-in for the three shapely classes the reference imports (utils.py:14-16).

TEST INFRASTRUCTURE, container-only.  shapely/GEOS are third-party, un-vendored and un-pinned in
the reference tree, so this shim IS the definition of their behaviour for the oracle
("parity unpinned" at this boundary):

  * Point(x, y).buffer(r).boundary  -> regular 64-gon ring (GEOS default 16 segments/quadrant),
    vertex k at (x + r*cos(-k*pi/32), y + r*sin(-k*pi/32))
  * ring.intersection(LineString([a, b])) -> EMPTY / POINT / MULTIPOINT; the empty result prints as
    'LINESTRING EMPTY' (GEOS >= 3.9 typed empties, the literal utils.py:279,306 compare against) or, with
    UNTYPED_EMPTY set, as 'GEOMETRYCOLLECTION EMPTY' (GEOS <= 3.8, the reference's Python-2.7 platform)
  * Polygon(4 corners): intersection/union areas for axis-aligned boxes (utils.py:440-443,
    455-458) and contains() (utils.py:197-209, dead code in the reference)

The arithmetic order below is mirrored operation-for-operation in oracle/cn_oracle.c
(ring_segment, cno_iou) so that the C restatement can be compared bit-for-bit.
"""
import math

_PC = [math.cos(-k * math.pi / 32.0) for k in range(64)]
_PS = [math.sin(-k * math.pi / 32.0) for k in range(64)]


class _Coords(list):
    pass


class Point(object):
    def __init__(self, *args):
        if len(args) == 1:
            args = tuple(args[0])
        self.x = float(args[0])
        self.y = float(args[1])
        self.coords = _Coords([(self.x, self.y)])

    def buffer(self, r):
        return _Disc(self.x, self.y, float(r))

    def __str__(self):
        return "POINT (%r %r)" % (self.x, self.y)


class _Disc(object):
    def __init__(self, cx, cy, r):
        self.boundary = _Ring(cx, cy, r)


# GEOS version switch (cn_config.geos_untyped_empty).  False: GEOS >= 3.9 typed empties -- an empty ring/line intersection
# prints as 'LINESTRING EMPTY', the literal utils.py:279,306 compare against.  True: GEOS <= 3.8 (what shapely <= 1.7, the last
# Python-2.7 release, links): the empty result is an untyped collection and prints as 'GEOMETRYCOLLECTION EMPTY'; its .geoms is
# empty, so utils.py:281 `i.geoms[0]` raises IndexError and utils.py:308 `i.x` raises AttributeError.
UNTYPED_EMPTY = False


class _Empty(object):
    geoms = []

    def __str__(self):
        return "GEOMETRYCOLLECTION EMPTY" if UNTYPED_EMPTY else "LINESTRING EMPTY"


class _MultiPoint(object):
    def __init__(self, pts):
        self.geoms = [Point(p[0], p[1]) for p in pts]

    def __str__(self):
        return "MULTIPOINT (%s)" % ", ".join("%r %r" % (g.x, g.y) for g in self.geoms)


class LineString(object):
    def __init__(self, coords):
        self.coords = _Coords([(float(c[0]), float(c[1])) for c in coords])


class _Ring(object):
    def __init__(self, cx, cy, r):
        self.cx, self.cy, self.r = cx, cy, r

    def intersection(self, line):
        (ax, ay), (bx, by) = line.coords[0], line.coords[1]
        cx, cy, r = self.cx, self.cy, self.r
        rx = bx - ax
        ry = by - ay
        hits = []
        for k in range(64):
            k2 = (k + 1) & 63
            c0x = cx + r * _PC[k]
            c0y = cy + r * _PS[k]
            c1x = cx + r * _PC[k2]
            c1y = cy + r * _PS[k2]
            sx = c1x - c0x
            sy = c1y - c0y
            den = rx * sy - ry * sx
            if den == 0.0:
                continue
            qx = c0x - ax
            qy = c0y - ay
            t = (qx * sy - qy * sx) / den
            u = (qx * ry - qy * rx) / den
            if t >= 0.0 and t <= 1.0 and u >= 0.0 and u < 1.0:
                hits.append((ax + t * rx, ay + t * ry))
        if not hits:
            return _Empty()
        if len(hits) == 1:
            return Point(hits[0][0], hits[0][1])
        return _MultiPoint(hits)


class _Area(object):
    def __init__(self, a):
        self.area = a


class Polygon(object):
    def __init__(self, corners):
        self.pts = [(float(c[0]), float(c[1])) for c in corners]
        xs = [p[0] for p in self.pts]
        ys = [p[1] for p in self.pts]
        self.xp, self.xm, self.yp, self.ym = max(xs), min(xs), max(ys), min(ys)

    @property
    def area(self):
        return (self.xp - self.xm) * (self.yp - self.ym)

    def _inter(self, o):
        ix = min(self.xp, o.xp) - max(self.xm, o.xm)
        iy = min(self.yp, o.yp) - max(self.ym, o.ym)
        return ix * iy if (ix > 0.0 and iy > 0.0) else 0.0

    def intersection(self, o):
        return _Area(self._inter(o))

    def union(self, o):
        return _Area(self.area + o.area - self._inter(o))

    def contains(self, pt):
        # even-odd rule; only reached from the reference's dead code (result never read)
        x, y = pt.x, pt.y
        inside = False
        n = len(self.pts)
        for i in range(n):
            x0, y0 = self.pts[i]
            x1, y1 = self.pts[(i + 1) % n]
            if (y0 > y) != (y1 > y):
                xi = x0 + (y - y0) * (x1 - x0) / (y1 - y0)
                if x < xi:
                    inside = not inside
        return inside
