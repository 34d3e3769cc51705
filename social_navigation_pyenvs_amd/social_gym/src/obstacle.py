"""Polygon wall (reference: social_gym/src/obstacle.py:7-66, geometry only -- no sprite)."""
import numpy as np

from .utils import PRECISION


class Obstacle:
    def __init__(self, game, vertices):
        if len(vertices) < 3:
            raise Exception("Obstacle has to have at least 3 vertices")
        self.vertices = np.array(vertices, dtype=PRECISION)
        # segment k joins vertex k and k+1 (closing the polygon), endpoints in Python list order (:31-32)
        self.segments = {}
        nv = len(vertices)
        for k in range(nv):
            a, b = vertices[k], vertices[(k + 1) % nv]
            self.segments[k] = [min(a, b), max(a, b)]

    def get_segments(self):
        return self.segments

    def get_closest_point(self, point):
        """Closest point over the segments; the LAST minimum wins on ties (serial path, :62-65)."""
        best_d, best = 10000, np.array([0.0, 0.0], dtype=PRECISION)
        for seg in self.segments.values():
            a = np.array(min(seg[0], seg[1]), dtype=PRECISION)
            b = np.array(max(seg[0], seg[1]), dtype=PRECISION)
            t = np.dot(point - a, b - a) / (np.linalg.norm(b - a) ** 2)
            h = a + min(max(0, t), 1) * (b - a)
            d = np.linalg.norm(h - point)
            if d <= best_d:
                best, best_d = h, d
        return best, best_d
