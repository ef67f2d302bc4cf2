"""Angle helper (reference `confrez/control/utils.py:28-29`)."""
from math import pi


def pi_2_pi(angle):
    """Wrap to [-pi, pi)."""
    return (angle + pi) % (2 * pi) - pi
