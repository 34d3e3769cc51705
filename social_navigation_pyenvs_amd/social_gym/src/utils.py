"""Host-side scalar helpers with the reference's semantics (social_gym/src/utils.py:5-40)."""
import math

import numpy as np

PRECISION = np.float64


def bound_angle(angle):
    """Wrap into [-pi, pi]; multiples beyond +-2 pi are folded with the sign-of-divisor modulo first."""
    two_pi = 2 * math.pi
    if angle >= two_pi:
        angle %= two_pi
    if angle <= -two_pi:
        angle %= -two_pi
    if angle > math.pi:
        angle -= two_pi
    if angle < -math.pi:
        angle += two_pi
    return angle


def point_to_segment_dist(x1, y1, x2, y2, x3, y3):
    """Distance from (x3, y3) to the segment (x1, y1)-(x2, y2)."""
    px, py = x2 - x1, y2 - y1
    if px == 0 and py == 0:
        return math.hypot(x3 - x1, y3 - y1)
    u = ((x3 - x1) * px + (y3 - y1) * py) / (px * px + py * py)
    u = 1 if u > 1 else (0 if u < 0 else u)
    return math.hypot(x1 + u * px - x3, y1 + u * py - y3)


def is_multiple(number, dividend, tolerance=1e-7):
    mod = number % dividend
    return (abs(mod) <= tolerance) or (abs(dividend - mod) <= tolerance)


def round_time(time):
    if time < 10.0:
        return round(time, 3)
    if time < 100.0:
        return round(time, 2)
    if time < 1000.0:
        return round(time, 1)
    return round(time)
