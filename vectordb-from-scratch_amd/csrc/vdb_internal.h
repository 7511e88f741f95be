// vdb_internal.h -- hooks of the device row store (vdb_flat.cpp) for the other host-side indexes of this library
// (vdb_hnsw.cpp).  Not part of the C ABI.
#pragma once
#include <stddef.h>
#include <stdint.h>

struct vdb_flat_index;

namespace vdb_internal {

// Marker the pair kernels write instead of a distance when a Cosine pair has a zero norm on either side
// (distance.rs:51-55 InvalidVector); a plain NaN distance stays a plain NaN.
constexpr uint32_t ZERO_NORM_MARK = 0x7fc0deadu;

// Uploads pending rows, then pads the nq queries and computes their exact-order norms on the device.  The prepared
// block stays valid until the next search / pairs_begin on the handle.
int pairs_begin(vdb_flat_index* ix, const float* queries, size_t nq, size_t dim);
// out[i] = DistanceMetric::distance(prepared query pair_q[i], device row pair_row[i])   (distance.rs:20-33)
int pairs_eval(vdb_flat_index* ix, const uint32_t* pair_q, const uint32_t* pair_row, size_t n, float* out);
// out[i] = distance(device row row_a[i], device row row_b[i])
int rows_eval(vdb_flat_index* ix, const uint32_t* row_a, const uint32_t* row_b, size_t n, float* out);
// out[r] = distance(prepared query q, device row r) for r < n_rows
int query_vs_rows(vdb_flat_index* ix, uint32_t q, uint32_t n_rows, float* out);
// device row of a stored id (0xffffffff: absent, or a row of another dimension kept host-side)
uint32_t row_of(vdb_flat_index* ix, uint64_t id);
uint32_t n_rows(vdb_flat_index* ix);
// device pointers of the row store and of the query block prepared by pairs_begin (valid until the next call on the handle)
struct DeviceView { const float* rows; uint32_t ld, dim; const float* nd; const float* qp; const float* qnorm; int metric; void* stream; uint32_t* status; };
int device_view(vdb_flat_index* ix, DeviceView* out);
// thread-local last-error state shared by every entry point of the library (vdb_last_error)
int set_error(int code, const char* msg);
int set_dim_error(size_t expected, size_t actual);

}  // namespace vdb_internal
