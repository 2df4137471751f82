// KV scatter and device-resident block-table maintenance (integer / byte work, HBM-bound).
#include "nvx_dev.h"

namespace NVX_NS {

// One 16-byte chunk per thread.  chunk id -> (token, K|V, chunk within row).  Consecutive
// lanes copy consecutive 16-B chunks of one row, so both the gather from the (strided) source
// and the scatter into the slot row are fully coalesced 1-KiB wave accesses.
__global__ __launch_bounds__(256) void store_kv_kernel(
    const uint4* __restrict__ k, const uint4* __restrict__ v, int64_t k_stride16, int64_t v_stride16,
    uint4* __restrict__ k_cache, uint4* __restrict__ v_cache, const int64_t* __restrict__ slots,
    int64_t n_tokens, int chunks_per_row, int64_t n_slots) {
    const int64_t total = n_tokens * chunks_per_row * 2;
    for (int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; id < total;
         id += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(id % chunks_per_row);
        const int64_t rest = id / chunks_per_row;
        const int which = (int)(rest & 1);
        const int64_t tok = rest >> 1;
        const int64_t slot = slots[tok];
        if (slot < 0 || slot >= n_slots) continue;  // -1 = padded row of a captured graph
        if (which == 0) k_cache[slot * chunks_per_row + c] = k[tok * k_stride16 + c];
        else            v_cache[slot * chunks_per_row + c] = v[tok * v_stride16 + c];
    }
}

extern "C" int nvx_store_kv(const void* k, const void* v, int64_t k_row_stride, int64_t v_row_stride,
                            void* k_cache, void* v_cache, const int64_t* slots, int64_t n_tokens,
                            int64_t row_elems, int64_t n_slots, nvx_stream_t stream) {
    NVX_REQUIRE(n_tokens >= 0 && row_elems > 0 && n_slots >= 0, NVX_EINVAL, "nvx_store_kv: negative size");
    if (n_tokens == 0) return NVX_OK;
    NVX_REQUIRE(k && v && k_cache && v_cache && slots, NVX_EINVAL, "nvx_store_kv: null pointer");
    NVX_REQUIRE(row_elems % 8 == 0 && k_row_stride % 8 == 0 && v_row_stride % 8 == 0, NVX_EUNSUPPORTED,
                "nvx_store_kv: row_elems and row strides must be multiples of 8 elements");
    NVX_REQUIRE(k_row_stride >= row_elems && v_row_stride >= row_elems, NVX_EINVAL,
                "nvx_store_kv: row stride smaller than the row");
    NVX_REQUIRE(nvx_aligned16(k) && nvx_aligned16(v) && nvx_aligned16(k_cache) && nvx_aligned16(v_cache),
                NVX_EINVAL, "nvx_store_kv: pointers must be 16-byte aligned");
    const int chunks = (int)(row_elems / 8);
    const int64_t total = n_tokens * chunks * 2;
    int64_t blocks = (total + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipLaunchKernelGGL(store_kv_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                       (const uint4*)k, (const uint4*)v, k_row_stride / 8, v_row_stride / 8,
                       (uint4*)k_cache, (uint4*)v_cache, slots, n_tokens, chunks, n_slots);
    return nvx_check_launch("nvx_store_kv");
}

// ---- block table ------------------------------------------------------------------------
__global__ __launch_bounds__(256) void table_write_rows_kernel(
    const int32_t* __restrict__ kv_indices, const int32_t* __restrict__ kv_indptr,
    const int32_t* __restrict__ rows, int32_t* __restrict__ table, int64_t n_rows, int64_t cap) {
    const int b = blockIdx.x;
    const int32_t row = rows[b];
    if (row < 0 || row >= n_rows) return;
    const int32_t s0 = kv_indptr[b];
    int32_t n = kv_indptr[b + 1] - s0;
    if (n > cap) n = (int32_t)cap;
    int32_t* dst = table + (int64_t)row * cap;
    for (int i = threadIdx.x; i < n; i += blockDim.x) dst[i] = kv_indices[s0 + i];
}

extern "C" int nvx_block_table_write_rows(const int32_t* kv_indices, const int32_t* kv_indptr,
                                          const int32_t* rows, int64_t batch, int32_t* table,
                                          int64_t n_rows, int64_t row_capacity, nvx_stream_t stream) {
    NVX_REQUIRE(batch >= 0 && n_rows > 0 && row_capacity > 0, NVX_EINVAL, "nvx_block_table_write_rows: bad size");
    if (batch == 0) return NVX_OK;
    NVX_REQUIRE(kv_indices && kv_indptr && rows && table, NVX_EINVAL, "nvx_block_table_write_rows: null pointer");
    hipLaunchKernelGGL(table_write_rows_kernel, dim3((unsigned)batch), dim3(256), 0, (hipStream_t)stream,
                       kv_indices, kv_indptr, rows, table, n_rows, row_capacity);
    return nvx_check_launch("nvx_block_table_write_rows");
}

__global__ __launch_bounds__(256) void decode_step_setup_kernel(
    const int32_t* __restrict__ step, int64_t stride, int64_t padded, int32_t* __restrict__ table,
    int64_t n_rows, int64_t cap, int64_t pad_start, int64_t* __restrict__ kv_start,
    int32_t* __restrict__ kv_len, int64_t* __restrict__ out_slots, int64_t* __restrict__ positions,
    int64_t* __restrict__ input_ids, const int64_t* __restrict__ prev_ids) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= padded) return;
    int32_t batch = step[0];
    if (batch > padded) batch = (int32_t)padded;
    if (i < batch) {
        const int32_t row = step[4 + i];
        const int32_t pos = step[4 + stride + i];
        const int32_t slot = step[4 + 2 * stride + i];
        int64_t tok = step[4 + 3 * stride + i];
        // token -1-j: the id sampled for row j of the step before, still on the device
        if (tok < 0) tok = (prev_ids && -1 - tok < stride) ? prev_ids[-1 - tok] : 0;
        const bool ok = row >= 0 && row < n_rows && pos >= 0 && pos < cap;
        if (ok) table[(int64_t)row * cap + pos] = slot;
        kv_start[i] = ok ? (int64_t)row * cap : pad_start;
        kv_len[i] = ok ? pos + 1 : 1;
        out_slots[i] = ok ? (int64_t)slot : -1;
        positions[i] = ok ? pos : 0;
        input_ids[i] = tok;
    } else {
        kv_start[i] = pad_start;
        kv_len[i] = 1;
        out_slots[i] = -1;
        positions[i] = 0;
        input_ids[i] = 0;
    }
}

extern "C" int nvx_decode_step_setup(const int32_t* step, int64_t stride, int64_t padded_batch,
                                     int32_t* table, int64_t n_rows, int64_t row_capacity,
                                     int64_t pad_start, int64_t* kv_start, int32_t* kv_len,
                                     int64_t* out_slots, int64_t* positions, int64_t* input_ids,
                                     const int64_t* prev_ids, nvx_stream_t stream) {
    NVX_REQUIRE(padded_batch > 0 && stride >= padded_batch && n_rows > 0 && row_capacity > 0, NVX_EINVAL,
                "nvx_decode_step_setup: bad size");
    NVX_REQUIRE(pad_start >= 0 && pad_start < n_rows * row_capacity, NVX_EINVAL,
                "nvx_decode_step_setup: pad_start outside the table");
    NVX_REQUIRE(step && table && kv_start && kv_len && out_slots && positions && input_ids, NVX_EINVAL,
                "nvx_decode_step_setup: null pointer");
    const unsigned blocks = (unsigned)((padded_batch + 255) / 256);
    hipLaunchKernelGGL(decode_step_setup_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, step, stride,
                       padded_batch, table, n_rows, row_capacity, pad_start, kv_start, kv_len, out_slots,
                       positions, input_ids, prev_ids);
    return nvx_check_launch("nvx_decode_step_setup");
}

}  // namespace NVX_NS
