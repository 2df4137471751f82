// All-reduce of a decode step's row-parallel projection outputs across the ranks of a tensor-parallel group,
// written straight into the peers' memory over xGMI and fused with the RMSNorm that consumes the sum.
//
// Every rank owns a landing buffer in uncached device memory that all peers have mapped (hipIpc*).  A call is one
// launch of one 256-thread workgroup per row.  The workgroup of row i on rank r
//   1. pushes its row, as 8-byte granules {two 16-bit values, 32-bit tag}, into slot (parity, r, i) of every
//      peer's landing buffer (posted writes, no round trip);
//   2. polls slot (parity, p, i) of its OWN buffer for every peer p until each granule carries the tag of this
//      exchange (local reads; a granule is written by one 8-byte store, so data and tag arrive together and no
//      fence or flag is needed);
//   3. adds the tp contributions in rank order in fp32 -- the same order on every rank, so all ranks hold the
//      same bits -- rounds the sum to the 16-bit format (what an all-reduce of 16-bit tensors hands over) and
//      goes on as rmsnorm_kernel does: + residual, sum of squares, scale.
// The tag of row i is a per-row exchange count kept in device memory and advanced by the kernel itself, so a
// captured graph replays correctly; the parity of the count picks one of two slots, which is enough: a rank can
// overwrite the slot of exchange e only at e + 2, which it reaches only after every peer has pushed row i of
// e + 1, i.e. after that peer's launch for e (which read the slot) has completed.  Rows are independent of each
// other: no grid-wide barrier, no counter shared between workgroups.
// Every wait is bounded by a wall-clock budget; a wait that runs out raises the group's status word (pinned host
// memory) and the kernel finishes with whatever it has, so no wave can spin forever.
#include "nvx_dev.h"
#include "peer_comm.h"

namespace NVX_NS {

__device__ __forceinline__ void peer_store(uint64_t* p, uint64_t v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);     // global_store_dwordx2 ... sc0 sc1
}
__device__ __forceinline__ uint64_t peer_load(const uint64_t* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);  // global_load_dwordx2 ... sc0 sc1
}

template <int SLABS, bool HAS_RES, bool NORM>
__global__ __launch_bounds__(256) void peer_allreduce_kernel(
    nvx_peer_args_t a, const uint16_t* x, int64_t x_stride, const uint16_t* residual, const uint16_t* __restrict__ weight,
    uint16_t* y, uint16_t* residual_out, int hidden, float eps) {
    __shared__ float red[4];
    const int tid = threadIdx.x;
    const int row = blockIdx.x;
    const uint32_t e = a.epoch[row] + 1u;
    const uint64_t tag = (uint64_t)e << 32;
    const int64_t half = hidden >> 1;                                   // granules per row
    const int64_t slot0 = (int64_t)(e & 1u) * a.tp * a.max_rows;        // first slot of this exchange's parity
    const int64_t mine = ((slot0 + (int64_t)a.rank * a.max_rows + row) * half);
    float v[SLABS][4];
    uint2 own[SLABS], res[SLABS], wv[SLABS];
    // own row: load, push to every peer
#pragma unroll
    for (int it = 0; it < SLABS; ++it) {
        const int col = it * 1024 + tid * 4;
        if (col < hidden) {
            own[it] = *reinterpret_cast<const uint2*>(x + (int64_t)row * x_stride + col);
            if (HAS_RES) res[it] = *reinterpret_cast<const uint2*>(residual + (int64_t)row * hidden + col);
            if (NORM) wv[it] = *reinterpret_cast<const uint2*>(weight + col);
        }
    }
#pragma unroll
    for (int it = 0; it < SLABS; ++it) {
        const int col = it * 1024 + tid * 4;
        if (col < hidden) {
#pragma unroll
            for (int p = 0; p < NVX_PEER_MAX_RANKS; ++p) {
                if (p < a.tp && p != a.rank) {
                    uint64_t* dst = a.recv[p] + mine + (col >> 1);
                    peer_store(dst, tag | own[it].x);
                    peer_store(dst + 1, tag | own[it].y);
                }
            }
        }
    }
    // peers' rows: poll the local landing slots, add in rank order
    uint64_t t0 = 0;
    unsigned spins = 0;
    bool gave_up = false;
    float ss = 0.f;
#pragma unroll
    for (int it = 0; it < SLABS; ++it) {
        const int col = it * 1024 + tid * 4;
        if (col < hidden) {
            uint64_t g[NVX_PEER_MAX_RANKS][2];
            for (;;) {
                bool ok = true;
#pragma unroll
                for (int p = 0; p < NVX_PEER_MAX_RANKS; ++p) {
                    if (p < a.tp && p != a.rank) {
                        const uint64_t* src = a.recv[a.rank] + (slot0 + (int64_t)p * a.max_rows + row) * half + (col >> 1);
                        g[p][0] = peer_load(src);
                        g[p][1] = peer_load(src + 1);
                    }
                }
#pragma unroll
                for (int p = 0; p < NVX_PEER_MAX_RANKS; ++p)
                    if (p < a.tp && p != a.rank) ok = ok && (uint32_t)(g[p][0] >> 32) == e && (uint32_t)(g[p][1] >> 32) == e;
                if (ok || gave_up) break;
                __builtin_amdgcn_s_sleep(4);
                if ((++spins & 31u) == 0u) {
                    const uint64_t now = wall_clock64();
                    if (t0 == 0) t0 = now;
                    else if (now - t0 > a.timeout_ticks) {
                        gave_up = true;
                        __hip_atomic_store(a.status, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    }
                }
            }
            float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int p = 0; p < NVX_PEER_MAX_RANKS; ++p) {
                if (p < a.tp) {
                    const uint32_t w0 = p == a.rank ? own[it].x : (uint32_t)g[p][0];
                    const uint32_t w1 = p == a.rank ? own[it].y : (uint32_t)g[p][1];
                    if (p == 0) {
                        acc[0] = h16lo_to_f32(w0); acc[1] = h16hi_to_f32(w0); acc[2] = h16lo_to_f32(w1); acc[3] = h16hi_to_f32(w1);
                    } else {
                        acc[0] += h16lo_to_f32(w0); acc[1] += h16hi_to_f32(w0); acc[2] += h16lo_to_f32(w1); acc[3] += h16hi_to_f32(w1);
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) v[it][j] = round_h16(acc[j]);
            if (!NORM) {
                *reinterpret_cast<uint2*>(y + (int64_t)row * hidden + col) =
                    make_uint2(pack_h16x2(v[it][0], v[it][1]), pack_h16x2(v[it][2], v[it][3]));
            } else {
                if (HAS_RES) {
                    v[it][0] += h16lo_to_f32(res[it].x); v[it][1] += h16hi_to_f32(res[it].x);
                    v[it][2] += h16lo_to_f32(res[it].y); v[it][3] += h16hi_to_f32(res[it].y);
                    *reinterpret_cast<uint2*>(residual_out + (int64_t)row * hidden + col) =
                        make_uint2(pack_h16x2(v[it][0], v[it][1]), pack_h16x2(v[it][2], v[it][3]));
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) ss += v[it][j] * v[it][j];
            }
        }
    }
    if (NORM) {
        ss = wave_sum(ss);
        if ((tid & 63) == 0) red[tid >> 6] = ss;
    }
    __syncthreads();                      // also: every thread has read epoch[row] before thread 0 advances it
    if (tid == 0) a.epoch[row] = e;
    if (NORM) {
        ss = red[0] + red[1] + red[2] + red[3];
        const float inv = 1.0f / sqrtf(ss / (float)hidden + eps);
#pragma unroll
        for (int it = 0; it < SLABS; ++it) {
            const int col = it * 1024 + tid * 4;
            if (col < hidden) {
                *reinterpret_cast<uint2*>(y + (int64_t)row * hidden + col) =
                    make_uint2(pack_h16x2((v[it][0] * inv) * h16lo_to_f32(wv[it].x), (v[it][1] * inv) * h16hi_to_f32(wv[it].x)),
                               pack_h16x2((v[it][2] * inv) * h16lo_to_f32(wv[it].y), (v[it][3] * inv) * h16hi_to_f32(wv[it].y)));
            }
        }
    }
}

template <int SLABS>
static void launch_peer(const nvx_peer_args_t& a, const void* x, int64_t xs, const void* res, const void* w, void* y, void* ro,
                        int rows, int hidden, float eps, hipStream_t st) {
    const dim3 grid((unsigned)rows), block(256);
#define NVX_PEER(HR, NM) hipLaunchKernelGGL((peer_allreduce_kernel<SLABS, HR, NM>), grid, block, 0, st, a, (const uint16_t*)x, xs, \
        (const uint16_t*)res, (const uint16_t*)w, (uint16_t*)y, (uint16_t*)ro, hidden, eps)
    if (!w) NVX_PEER(false, false);
    else if (res) NVX_PEER(true, true);
    else NVX_PEER(false, true);
#undef NVX_PEER
}

}  // namespace NVX_NS

extern "C" int NVX_ENTRY(nvx_peer_allreduce_norm)(void* comm, const void* x, int64_t x_row_stride, const void* residual,
                                                  const void* weight, void* y, void* residual_out, int64_t rows, int64_t hidden,
                                                  float eps, nvx_stream_t stream) {
    using namespace NVX_NS;
    const nvx_peer_comm_t* c = (const nvx_peer_comm_t*)comm;
    NVX_REQUIRE(c && c->magic == NVX_PEER_MAGIC, NVX_EINVAL, "nvx_peer_allreduce_norm: not a peer group");
    NVX_REQUIRE(rows >= 0 && rows <= c->max_rows, NVX_ENOSPACE, "nvx_peer_allreduce_norm: %lld rows, the group was made for %lld",
                (long long)rows, (long long)c->max_rows);
    NVX_REQUIRE(hidden == c->hidden, NVX_EINVAL, "nvx_peer_allreduce_norm: hidden %lld, the group was made for %lld",
                (long long)hidden, (long long)c->hidden);
    if (rows == 0) return NVX_OK;
    NVX_REQUIRE(x && y, NVX_EINVAL, "nvx_peer_allreduce_norm: null pointer");
    NVX_REQUIRE(weight || !residual, NVX_EINVAL, "nvx_peer_allreduce_norm: a residual needs the norm (weight)");
    NVX_REQUIRE(!residual || residual_out, NVX_EINVAL, "nvx_peer_allreduce_norm: residual without residual_out");
    NVX_REQUIRE(x_row_stride >= hidden && x_row_stride % 4 == 0, NVX_EINVAL, "nvx_peer_allreduce_norm: row stride");
    NVX_REQUIRE((((uintptr_t)x | (uintptr_t)y | (uintptr_t)weight | (uintptr_t)residual | (uintptr_t)residual_out) & 7u) == 0, NVX_EINVAL,
                "nvx_peer_allreduce_norm: pointers must be 8-byte aligned");
    for (int p = 0; p < c->tp; ++p)
        NVX_REQUIRE(c->recv[p], NVX_EINVAL, "nvx_peer_allreduce_norm: rank %d's buffer has not been imported", p);
    nvx_peer_args_t a;
    for (int p = 0; p < NVX_PEER_MAX_RANKS; ++p) a.recv[p] = c->recv[p];
    a.epoch = c->epoch; a.status = c->status; a.tp = c->tp; a.rank = c->rank; a.max_rows = c->max_rows; a.timeout_ticks = c->timeout_ticks;
    hipStream_t st = (hipStream_t)stream;
    const int h = (int)hidden, r = (int)rows;
    if (h <= 1024) launch_peer<1>(a, x, x_row_stride, residual, weight, y, residual_out, r, h, eps, st);
    else if (h <= 2048) launch_peer<2>(a, x, x_row_stride, residual, weight, y, residual_out, r, h, eps, st);
    else if (h <= 4096) launch_peer<4>(a, x, x_row_stride, residual, weight, y, residual_out, r, h, eps, st);
    else if (h <= 5120) launch_peer<5>(a, x, x_row_stride, residual, weight, y, residual_out, r, h, eps, st);
    else launch_peer<8>(a, x, x_row_stride, residual, weight, y, residual_out, r, h, eps, st);
    return nvx_check_launch("nvx_peer_allreduce_norm");
}

// ---- the group itself: format-independent, built once ---------------------------------------------------------
#ifndef NVX_F16
#include <stdlib.h>
#include <string.h>

static_assert(sizeof(hipIpcMemHandle_t) == 64, "the header promises 64-byte handles");

extern "C" int64_t nvx_peer_exchange_bytes(int tp_size, int64_t max_rows, int64_t hidden) {
    if (tp_size < 1 || tp_size > NVX_PEER_MAX_RANKS || max_rows < 1 || hidden < 4) return 0;
    return 2 * (int64_t)tp_size * max_rows * (hidden / 2) * 8;
}

extern "C" int nvx_peer_create(int tp_size, int rank, int64_t max_rows, int64_t hidden, int64_t timeout_ms, void** comm_out) {
    NVX_REQUIRE(comm_out, NVX_EINVAL, "nvx_peer_create: null pointer");
    *comm_out = nullptr;
    NVX_REQUIRE(tp_size >= 1 && tp_size <= NVX_PEER_MAX_RANKS && rank >= 0 && rank < tp_size, NVX_EUNSUPPORTED,
                "nvx_peer_create: %d ranks (at most %d)", tp_size, NVX_PEER_MAX_RANKS);
    NVX_REQUIRE(max_rows >= 1 && max_rows <= 512, NVX_EUNSUPPORTED, "nvx_peer_create: max_rows must be 1..512 (a decode batch; larger "
                "tensors belong on the library's bandwidth-bound all-reduce)");
    NVX_REQUIRE(hidden >= 4 && hidden % 4 == 0 && hidden <= 8192, NVX_EUNSUPPORTED, "nvx_peer_create: hidden must be a multiple of 4, <= 8192");
    nvx_peer_comm_t* c = (nvx_peer_comm_t*)calloc(1, sizeof(nvx_peer_comm_t));
    NVX_REQUIRE(c, NVX_ENOSPACE, "nvx_peer_create: out of host memory");
    c->tp = tp_size; c->rank = rank; c->max_rows = max_rows; c->hidden = hidden;
    c->timeout_ticks = (uint64_t)(timeout_ms > 0 ? timeout_ms : 10000) * 100000ull;   // 100 MHz clock
    const size_t bytes = (size_t)nvx_peer_exchange_bytes(tp_size, max_rows, hidden);
    void* buf = nullptr;
    hipError_t e = hipGetDevice(&c->device);
    // Uncached device memory: peers' writes arrive beside this GPU's L2, so a cached line would go stale; fine-grained
    // memory is the second choice.  Ordinary (coarse-grained) memory is not coherent while a kernel runs and is refused.
    if (e == hipSuccess) {
        e = hipExtMallocWithFlags(&buf, bytes, hipDeviceMallocUncached);
        if (e != hipSuccess) { (void)hipGetLastError(); e = hipExtMallocWithFlags(&buf, bytes, hipDeviceMallocFinegrained); }
    }
    if (e == hipSuccess) e = hipMemset(buf, 0, bytes);
    if (e == hipSuccess) e = hipMalloc((void**)&c->epoch, (size_t)max_rows * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMemset(c->epoch, 0, (size_t)max_rows * sizeof(uint32_t));
    if (e == hipSuccess) e = hipHostMalloc((void**)&c->status, 64, hipHostMallocMapped);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) {
        (void)hipGetLastError();
        if (buf) (void)hipFree(buf);
        if (c->epoch) (void)hipFree(c->epoch);
        if (c->status) (void)hipHostFree(c->status);
        free(c);
        return nvx_fail(NVX_ENOSPACE, "nvx_peer_create: %s", hipGetErrorString(e));
    }
    *c->status = 0;
    c->recv[rank] = (uint64_t*)buf;
    c->magic = NVX_PEER_MAGIC;
    *comm_out = c;
    return NVX_OK;
}

extern "C" int nvx_peer_export(void* comm, unsigned char* handle_out) {
    nvx_peer_comm_t* c = (nvx_peer_comm_t*)comm;
    NVX_REQUIRE(c && c->magic == NVX_PEER_MAGIC && handle_out, NVX_EINVAL, "nvx_peer_export: bad argument");
    hipIpcMemHandle_t h;
    hipError_t e = hipIpcGetMemHandle(&h, c->recv[c->rank]);
    if (e != hipSuccess) { (void)hipGetLastError(); return nvx_fail(NVX_ELAUNCH, "nvx_peer_export: hipIpcGetMemHandle: %s", hipGetErrorString(e)); }
    memcpy(handle_out, &h, 64);
    return NVX_OK;
}

extern "C" int nvx_peer_import(void* comm, int peer_rank, const unsigned char* handle) {
    nvx_peer_comm_t* c = (nvx_peer_comm_t*)comm;
    NVX_REQUIRE(c && c->magic == NVX_PEER_MAGIC && handle, NVX_EINVAL, "nvx_peer_import: bad argument");
    NVX_REQUIRE(peer_rank >= 0 && peer_rank < c->tp && peer_rank != c->rank, NVX_EINVAL, "nvx_peer_import: rank %d is not a peer", peer_rank);
    NVX_REQUIRE(!c->recv[peer_rank], NVX_EINVAL, "nvx_peer_import: rank %d was imported before", peer_rank);
    hipIpcMemHandle_t h;
    memcpy(&h, handle, 64);
    void* p = nullptr;
    hipError_t e = hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess);
    if (e != hipSuccess) { (void)hipGetLastError(); return nvx_fail(NVX_ELAUNCH, "nvx_peer_import: hipIpcOpenMemHandle: %s", hipGetErrorString(e)); }
    c->recv[peer_rank] = (uint64_t*)p;
    c->opened[peer_rank] = true;
    return NVX_OK;
}

extern "C" int nvx_peer_status(void* comm) {
    nvx_peer_comm_t* c = (nvx_peer_comm_t*)comm;
    if (!c || c->magic != NVX_PEER_MAGIC) return NVX_EINVAL;
    return __atomic_load_n(c->status, __ATOMIC_RELAXED);
}

extern "C" int nvx_peer_destroy(void* comm) {
    nvx_peer_comm_t* c = (nvx_peer_comm_t*)comm;
    NVX_REQUIRE(c && c->magic == NVX_PEER_MAGIC, NVX_EINVAL, "nvx_peer_destroy: not a peer group");
    (void)hipDeviceSynchronize();
    for (int p = 0; p < c->tp; ++p)
        if (c->opened[p]) (void)hipIpcCloseMemHandle(c->recv[p]);
    (void)hipFree(c->recv[c->rank]);
    (void)hipFree(c->epoch);
    (void)hipHostFree(c->status);
    (void)hipGetLastError();
    c->magic = 0;
    free(c);
    return NVX_OK;
}
#endif
