"""approx-crate style comparisons used by the reference's tests (approx 0.5.1)."""
import sys

EPS = sys.float_info.epsilon


def rel_eq(a, b, max_relative=EPS, epsilon=EPS):
    if a == b:
        return True
    d = abs(a - b)
    if d <= epsilon:
        return True
    return d <= max(abs(a), abs(b)) * max_relative


def assert_rel(a, b, max_relative=EPS):
    if isinstance(a, (int, float)):
        assert rel_eq(a, b, max_relative), (a, b)
        return
    a, b = list(a), list(b)
    assert len(a) == len(b), (a, b)
    for x, y in zip(a, b):
        assert rel_eq(x, y, max_relative), (a, b)
