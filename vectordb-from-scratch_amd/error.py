"""Error vocabulary of the reference (src/error.rs:10-31), as Python exceptions."""


class VectorDbError(Exception):
    pass


class DimensionMismatch(VectorDbError):          # error.rs:12-13
    def __init__(self, expected, actual):
        super().__init__(f"Dimension mismatch: expected {expected}, got {actual}")
        self.expected, self.actual = expected, actual


class VectorNotFound(VectorDbError):             # error.rs:15-16
    def __init__(self, id):
        super().__init__(f"Vector not found: {id}")
        self.id = id


class InvalidVector(VectorDbError):              # error.rs:18-19
    def __init__(self, reason):
        super().__init__(f"Invalid vector: {reason}")
        self.reason = reason


class IndexError_(VectorDbError):                # error.rs:30  IndexError(String)
    def __init__(self, msg):
        super().__init__(f"Index error: {msg}")


class NanDistance(IndexError_):
    """The reference panics on a NaN distance (flat_index.rs:62); a panic cannot cross the C ABI."""
