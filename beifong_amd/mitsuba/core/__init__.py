"""mitsuba.core subset: Vector3f / Point3f, Transform4f (= ScalarTransform4f), Thread().file_resolver(), Bitmap, Struct."""
import numpy as np

from .._host import Bitmap, Struct  # noqa: F401
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


ScalarTransform4f = Transform4f      # the scalar variants' Transform4f IS ScalarTransform4f (python_scripts/Render.py:30)


class _Vec3(np.ndarray):
    """Vector3f / Point3f of the scalar variants: three float32 components (x, y, z attributes, numpy arithmetic)."""

    def __new__(cls, x=0.0, y=None, z=None):
        if y is None:
            v = np.broadcast_to(np.asarray(x, dtype=np.float32), (3,)).copy()
        else:
            v = np.array([x, y, z], dtype=np.float32)
        return v.view(cls)

    x = property(lambda self: float(self[0]))
    y = property(lambda self: float(self[1]))
    z = property(lambda self: float(self[2]))


class Vector3f(_Vec3):
    pass


class Point3f(_Vec3):
    pass


ScalarVector3f, ScalarPoint3f = Vector3f, Point3f
