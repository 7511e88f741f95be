"""Host-side mirror of the reference's Vector (src/vector.rs:9-37) and DistanceMetric enum
(src/distance.rs:9-16).  Only storage and shape live here; every distance is computed on the GPU."""
import enum

import numpy as np


class DistanceMetric(enum.IntEnum):
    Euclidean = 0
    Cosine = 1
    DotProduct = 2


class Vector:
    __slots__ = ("data",)

    def __init__(self, data):                     # Vector::new  vector.rs:15-17
        self.data = np.ascontiguousarray(data, dtype=np.float32).reshape(-1)

    def dimension(self):                          # vector.rs:20
        return int(self.data.size)

    def as_slice(self):                           # vector.rs:25
        return self.data

    def has_same_dimension(self, other):          # vector.rs:30
        return self.dimension() == other.dimension()

    def __eq__(self, other):
        return isinstance(other, Vector) and np.array_equal(self.data, other.data)

    def __repr__(self):
        return f"Vector({self.data.tolist()})"
