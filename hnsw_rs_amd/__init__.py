"""hnsw_rs_amd -- MI355X-native HNSW search engine behind the API of the Rust `hnsw` crate of
Gumo-A/hnsw_rs.  The product is libhnsw_mi355x.so (HIP kernels for gfx950 + host index, C ABI in
include/hnsw_mi355x.h); this package is its thin host-side mirror of the reference's interface.
"""
from ._lib import (VEC_F32, VEC_QUANT8, UINT32_MAX, HnswError, lib)  # noqa: F401
from .hnsw import HNSW, Graph, Point, device_count, draw_levels, synth_rows  # noqa: F401
