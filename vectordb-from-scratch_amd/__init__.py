"""MI355X-native brute-force kNN engine behind the `Index` trait of
Ricoledan/vectordb-from-scratch (FlatIndex hot path only).

csrc/   hand-written HIP kernels for gfx950 + the C ABI of include/vdb_flat.h and include/vdb_hnsw.h
*.py    host-side mirror of the reference interface for this path (Vector, DistanceMetric,
        Index, VectorStore, MetadataFilter, errors) driving the C ABI through ctypes
"""
from . import _ffi
from .build import build
from .error import (DimensionMismatch, IndexError_, InvalidVector, NanDistance, VectorDbError,
                    VectorNotFound)
from .index import GpuFlatIndex, Index
from .hnsw import GpuHnswIndex, HnswParams
from .storage import BatchInsertItem, Metadata, MetadataFilter, SearchResult, VectorStore
from .vector import DistanceMetric, Vector

__all__ = ["build", "GpuFlatIndex", "GpuHnswIndex", "HnswParams", "Index", "VectorStore", "Vector", "DistanceMetric", "Metadata",
           "MetadataFilter", "SearchResult", "BatchInsertItem", "VectorDbError", "DimensionMismatch",
           "InvalidVector", "VectorNotFound", "IndexError_", "NanDistance"]
