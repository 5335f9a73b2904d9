"""mitsuba.core subset: Transform4f, Thread().file_resolver(), Bitmap."""
import numpy as np

from .._host import Bitmap  # noqa: F401
from ...scenedesc import Transform4f as _T


class Transform4f(_T):
    """include/mitsuba/core/transform.h — accepts a nested list / ndarray too."""

    def __init__(self, matrix=None, inverse=None):
        if matrix is not None and not isinstance(matrix, np.ndarray):
            matrix = np.array(matrix, dtype=np.float32)
        super().__init__(matrix, inverse)

    def __mul__(self, other):
        r = _T.__mul__(self, other)
        return Transform4f(r.matrix, r.inv)

    def transform_vector(self, v):
        return (self.matrix[:3, :3] @ np.asarray(v, dtype=np.float32)).astype(np.float32)

    def transform_point(self, p):
        q = self.matrix @ np.append(np.asarray(p, dtype=np.float32), np.float32(1))
        return (q[:3] / q[3]).astype(np.float32)

    @staticmethod
    def translate(v):
        r = _T.translate(v)
        return Transform4f(r.matrix, r.inv)

    @staticmethod
    def scale(v):
        if np.isscalar(v):
            v = [v, v, v]
        r = _T.scale(v)
        return Transform4f(r.matrix, r.inv)

    @staticmethod
    def rotate(axis, angle):
        r = _T.rotate(axis, angle)
        return Transform4f(r.matrix, r.inv)

    @staticmethod
    def look_at(origin, target, up):
        r = _T.look_at(origin, target, up)
        return Transform4f(r.matrix, r.inv)


class _FileResolver:
    def __init__(self):
        self.paths = []

    def append(self, p):
        self.paths.append(p)


class Thread:
    _fr = _FileResolver()

    @staticmethod
    def thread():
        return Thread

    @staticmethod
    def file_resolver():
        return Thread._fr
