// Microbenchmark: what HBM rate does the decode-attention GATHER pattern reach by itself, and does it depend on how the
// 2 KB row of a token (8 kv heads x 256 B) is cut between workgroups?
//
//   mode A (the shipped kernel's pattern): a workgroup owns (chunk of 128 tokens, ONE kv head); a 16-lane quarter-wave
//           loads the 256-byte piece of its head from one token's K row and V row: a wave instruction touches four
//           separate 256-byte segments, and the eight pieces of a row are fetched by eight workgroups.
//   mode B: a workgroup owns a chunk of tokens for ALL heads; a wave instruction loads 1 KB contiguous (four heads of one
//           token), two waves cover the row.
// No arithmetic beyond an xor of the loaded words; TPQ rows of K and of V in flight per wave before the first use
// (as in the shipped kernel), DEPTH tiles in flight.
//
// build: hipcc -O3 --offload-arch=gfx950 tools/gather_probe.hip -o tools/gather_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
#include <random>

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ u32x4 ntload(const char* p) { return __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p)); }

// mode A: unit = chunk * 8 + head; chunk = 128 tokens
template <int TPQ>
__global__ __launch_bounds__(256, 4) void gather_a(const char* __restrict__ kc, const char* __restrict__ vc, const int* __restrict__ idx,
                                                    int n_tokens, int chunk, unsigned* __restrict__ sink) {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, quarter = lane >> 4, piece = lane & 15;
    const int n_units = ((n_tokens + chunk - 1) / chunk) * 8;
    unsigned acc = 0;
    for (int unit = blockIdx.x; unit < n_units; unit += gridDim.x) {
        const int c = unit >> 3, head = unit & 7;
        const int lo = c * chunk, n = min(chunk, n_tokens - lo);
        const char* kb = kc + head * 256 + piece * 16;
        const char* vb = vc + head * 256 + piece * 16;
        constexpr int TILE = 4 * TPQ, STEP = 4 * TILE;
        int slots[TPQ];
#pragma unroll
        for (int u = 0; u < TPQ; ++u) { const int t = wave * TILE + u * 4 + quarter; slots[u] = t < n ? idx[lo + t] : 0; }
        for (int tile = wave * TILE; tile < n; tile += STEP) {
            u32x4 k[TPQ], v[TPQ];
#pragma unroll
            for (int u = 0; u < TPQ; ++u) {
                const unsigned long long off = (unsigned long long)(unsigned)slots[u] * 2048ull;
                k[u] = ntload(kb + off); v[u] = ntload(vb + off);
            }
#pragma unroll
            for (int u = 0; u < TPQ; ++u) { const int t = tile + STEP + u * 4 + quarter; slots[u] = t < n ? idx[lo + t] : 0; }
#pragma unroll
            for (int u = 0; u < TPQ; ++u) acc ^= k[u].x ^ k[u].y ^ k[u].z ^ k[u].w ^ v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
        }
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

// mode A': as mode A, but every kv head has a pool of its own ([head][slot][256 B] instead of [slot][head][256 B]): with the
// slots of a sequence in a row (a prompt's are), a workgroup then reads one contiguous run of 256-byte rows.
template <int TPQ>
__global__ __launch_bounds__(256, 4) void gather_a_headpools(const char* __restrict__ kc, const char* __restrict__ vc, const int* __restrict__ idx,
                                                              int n_tokens, int chunk, size_t pool_bytes, unsigned* __restrict__ sink) {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, quarter = lane >> 4, piece = lane & 15;
    const int n_units = ((n_tokens + chunk - 1) / chunk) * 8;
    unsigned acc = 0;
    for (int unit = blockIdx.x; unit < n_units; unit += gridDim.x) {
        const int c = unit >> 3, head = unit & 7;
        const int lo = c * chunk, n = min(chunk, n_tokens - lo);
        const char* kb = kc + (size_t)head * pool_bytes + piece * 16;
        const char* vb = vc + (size_t)head * pool_bytes + piece * 16;
        constexpr int TILE = 4 * TPQ, STEP = 4 * TILE;
        int slots[TPQ];
#pragma unroll
        for (int u = 0; u < TPQ; ++u) { const int t = wave * TILE + u * 4 + quarter; slots[u] = t < n ? idx[lo + t] : 0; }
        for (int tile = wave * TILE; tile < n; tile += STEP) {
            u32x4 k[TPQ], v[TPQ];
#pragma unroll
            for (int u = 0; u < TPQ; ++u) {
                const unsigned long long off = (unsigned long long)(unsigned)slots[u] * 256ull;
                k[u] = ntload(kb + off); v[u] = ntload(vb + off);
            }
#pragma unroll
            for (int u = 0; u < TPQ; ++u) { const int t = tile + STEP + u * 4 + quarter; slots[u] = t < n ? idx[lo + t] : 0; }
#pragma unroll
            for (int u = 0; u < TPQ; ++u) acc ^= k[u].x ^ k[u].y ^ k[u].z ^ k[u].w ^ v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
        }
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

// mode A-rot: mode A, but the workgroup of head h walks its chunk starting `rot * h` tokens in (wrapping round), so that the
// eight workgroups of a chunk -- launched side by side, one per XCD -- are not on the same 2 KB rows at the same time.
template <int TPQ>
__global__ __launch_bounds__(256, 4) void gather_a_rot(const char* __restrict__ kc, const char* __restrict__ vc, const int* __restrict__ idx,
                                                        int n_tokens, int chunk, int rot, unsigned* __restrict__ sink) {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, quarter = lane >> 4, piece = lane & 15;
    const int n_units = ((n_tokens + chunk - 1) / chunk) * 8;
    unsigned acc = 0;
    for (int unit = blockIdx.x; unit < n_units; unit += gridDim.x) {
        const int c = unit >> 3, head = unit & 7;
        const int lo = c * chunk, n = min(chunk, n_tokens - lo);
        const int shift = (head * rot) % (n > 0 ? n : 1);
        const char* kb = kc + head * 256 + piece * 16;
        const char* vb = vc + head * 256 + piece * 16;
        constexpr int TILE = 4 * TPQ, STEP = 4 * TILE;
        auto at = [&](int t) { int r = t + shift; r = r >= n ? r - n : r; return idx[lo + r]; };
        int slots[TPQ];
#pragma unroll
        for (int u = 0; u < TPQ; ++u) { const int t = wave * TILE + u * 4 + quarter; slots[u] = t < n ? at(t) : 0; }
        for (int tile = wave * TILE; tile < n; tile += STEP) {
            u32x4 k[TPQ], v[TPQ];
#pragma unroll
            for (int u = 0; u < TPQ; ++u) {
                const unsigned long long off = (unsigned long long)(unsigned)slots[u] * 2048ull;
                k[u] = ntload(kb + off); v[u] = ntload(vb + off);
            }
#pragma unroll
            for (int u = 0; u < TPQ; ++u) { const int t = tile + STEP + u * 4 + quarter; slots[u] = t < n ? at(t) : 0; }
#pragma unroll
            for (int u = 0; u < TPQ; ++u) acc ^= k[u].x ^ k[u].y ^ k[u].z ^ k[u].w ^ v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
        }
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

// mode B: unit = chunk of `chunk` tokens, all heads; wave w: heads 4 (w & 1) .. +3 (1 KB of the row), tokens of parity w >> 1
template <int TPQ>
__global__ __launch_bounds__(256, 4) void gather_b(const char* __restrict__ kc, const char* __restrict__ vc, const int* __restrict__ idx,
                                                    int n_tokens, int chunk, unsigned* __restrict__ sink) {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int n_units = (n_tokens + chunk - 1) / chunk;
    unsigned acc = 0;
    const int half = wave & 1, par = wave >> 1;
    for (int unit = blockIdx.x; unit < n_units; unit += gridDim.x) {
        const int lo = unit * chunk, n = min(chunk, n_tokens - lo);
        const char* kb = kc + half * 1024 + lane * 16;
        const char* vb = vc + half * 1024 + lane * 16;
        constexpr int STEP = 2 * TPQ;      // two token parities interleave tiles of TPQ tokens
        int slots[TPQ];
#pragma unroll
        for (int u = 0; u < TPQ; ++u) { const int t = par * TPQ + u; slots[u] = t < n ? idx[lo + t] : 0; }   // wave-uniform
        for (int tile = par * TPQ; tile < n; tile += STEP) {
            u32x4 k[TPQ], v[TPQ];
#pragma unroll
            for (int u = 0; u < TPQ; ++u) {
                const unsigned long long off = (unsigned long long)(unsigned)slots[u] * 2048ull;
                k[u] = ntload(kb + off); v[u] = ntload(vb + off);
            }
#pragma unroll
            for (int u = 0; u < TPQ; ++u) { const int t = tile + STEP + u; slots[u] = t < n ? idx[lo + t] : 0; }
#pragma unroll
            for (int u = 0; u < TPQ; ++u) acc ^= k[u].x ^ k[u].y ^ k[u].z ^ k[u].w ^ v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
        }
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

// mode C: unit = chunk of `chunk` tokens, all heads; ONE wave loads a token's whole 2 KB K row (32 bytes per lane, two
// 16-byte loads) and its whole V row; the four waves interleave tiles of TPQ tokens
template <int TPQ>
__global__ __launch_bounds__(256, 4) void gather_c(const char* __restrict__ kc, const char* __restrict__ vc, const int* __restrict__ idx,
                                                    int n_tokens, int chunk, unsigned* __restrict__ sink) {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int n_units = (n_tokens + chunk - 1) / chunk;
    unsigned acc = 0;
    for (int unit = blockIdx.x; unit < n_units; unit += gridDim.x) {
        const int lo = unit * chunk, n = min(chunk, n_tokens - lo);
        const char* kb = kc + lane * 32;
        const char* vb = vc + lane * 32;
        constexpr int STEP = 4 * TPQ;
        int slots[TPQ];
#pragma unroll
        for (int u = 0; u < TPQ; ++u) { const int t = wave * TPQ + u; slots[u] = t < n ? idx[lo + t] : 0; }
        for (int tile = wave * TPQ; tile < n; tile += STEP) {
            u32x4 k[TPQ][2], v[TPQ][2];
#pragma unroll
            for (int u = 0; u < TPQ; ++u) {
                const unsigned long long off = (unsigned long long)(unsigned)slots[u] * 2048ull;
                k[u][0] = ntload(kb + off); k[u][1] = ntload(kb + off + 16);
                v[u][0] = ntload(vb + off); v[u][1] = ntload(vb + off + 16);
            }
#pragma unroll
            for (int u = 0; u < TPQ; ++u) { const int t = tile + STEP + u; slots[u] = t < n ? idx[lo + t] : 0; }
#pragma unroll
            for (int u = 0; u < TPQ; ++u) acc ^= k[u][0].x ^ k[u][1].y ^ k[u][0].z ^ k[u][1].w ^ v[u][0].x ^ v[u][1].y ^ v[u][0].z ^ v[u][1].w;
        }
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

// plain streaming read of the same number of bytes (the ceiling a copy-like access reaches)
__global__ __launch_bounds__(256, 4) void stream_read(const char* __restrict__ kc, size_t bytes, unsigned* __restrict__ sink) {
    unsigned acc = 0;
    const size_t n16 = bytes / 16;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256 * 4) {
        u32x4 a = ntload(kc + i * 16);
        u32x4 b = i + (size_t)gridDim.x * 256 < n16 ? ntload(kc + (i + (size_t)gridDim.x * 256) * 16) : a;
        u32x4 c = i + (size_t)gridDim.x * 512 < n16 ? ntload(kc + (i + (size_t)gridDim.x * 512) * 16) : a;
        u32x4 d = i + (size_t)gridDim.x * 768 < n16 ? ntload(kc + (i + (size_t)gridDim.x * 768) * 16) : a;
        acc ^= a.x ^ b.y ^ c.z ^ d.w;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

int main(int argc, char** argv) {
    const int n_slots = 1 << 20;                       // 2 GB of K + 2 GB of V
    const int tokens = argc > 1 ? atoi(argv[1]) : 25000;   // per launch (the bench's mean decode batch: 24.8k rows)
    const int reps = 16;
    char *kc, *vc; int* idx; unsigned* sink;
    CHECK(hipMalloc(&kc, (size_t)n_slots * 2048)); CHECK(hipMalloc(&vc, (size_t)n_slots * 2048));
    CHECK(hipMemset(kc, 1, (size_t)n_slots * 2048)); CHECK(hipMemset(vc, 2, (size_t)n_slots * 2048));
    std::vector<int> perm(n_slots);
    for (int i = 0; i < n_slots; ++i) perm[i] = i;
    std::mt19937 rng(0);
    std::shuffle(perm.begin(), perm.end(), rng);
    CHECK(hipMalloc(&idx, (size_t)n_slots * 4)); CHECK(hipMemcpy(idx, perm.data(), (size_t)n_slots * 4, hipMemcpyHostToDevice));
    CHECK(hipMalloc(&sink, 64));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const double bytes = (double)tokens * 4096.0;
    auto timeit = [&](const char* name, auto launch) {
        for (int r = 0; r < 3; ++r) launch(r);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        for (int r = 0; r < reps; ++r) launch(3 + r);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms * 1e3 / reps;
        printf("%-44s %8.2f us  %7.1f GB/s\n", name, us, bytes / us / 1e3);
    };
    const int windows = n_slots / tokens;
    for (int grid : {1024, 2048}) {
        char name[128];
#define RUN_A(T) snprintf(name, sizeof name, "A per-head 256B pieces  TPQ=%d grid=%d", T, grid); \
        timeit(name, [&](int r) { hipLaunchKernelGGL(gather_a<T>, dim3(grid), dim3(256), 0, 0, kc, vc, idx + (size_t)(r % windows) * tokens, tokens, 128, sink); });
#define RUN_B(T, C) snprintf(name, sizeof name, "B all heads, 1KB/wave    TPQ=%d chunk=%d grid=%d", T, C, grid); \
        timeit(name, [&](int r) { hipLaunchKernelGGL(gather_b<T>, dim3(grid), dim3(256), 0, 0, kc, vc, idx + (size_t)(r % windows) * tokens, tokens, C, sink); });
#define RUN_C(T, C) snprintf(name, sizeof name, "C all heads, 2KB/wave    TPQ=%d chunk=%d grid=%d", T, C, grid); \
        timeit(name, [&](int r) { hipLaunchKernelGGL(gather_c<T>, dim3(grid), dim3(256), 0, 0, kc, vc, idx + (size_t)(r % windows) * tokens, tokens, C, sink); });
        RUN_C(1, 16) RUN_C(2, 16) RUN_C(1, 32) RUN_C(2, 32) RUN_C(4, 32) RUN_C(2, 64) RUN_C(4, 64)
        RUN_A(2) RUN_A(4) RUN_A(8)
        RUN_B(2, 16) RUN_B(4, 16) RUN_B(8, 16) RUN_B(2, 32) RUN_B(4, 32) RUN_B(8, 32) RUN_B(4, 64) RUN_B(8, 64)
    }
    // How the slots of a sequence lie matters as much as how a row is cut.  Three slot lists of the same length: a random
    // permutation (above), runs of 558 consecutive slots (prompts as the allocator hands them out) starting at random
    // places, and those runs with every fourth token replaced by a far slot (tokens appended one per decode step).
    {
        std::vector<int> runs(n_slots), mixed(n_slots);
        std::uniform_int_distribution<int> start(0, n_slots - 600);
        for (int i = 0; i < n_slots; i += 558) {
            const int s0 = start(rng);
            for (int j = 0; j < 558 && i + j < n_slots; ++j) { runs[i + j] = s0 + j; mixed[i + j] = (j & 3) == 3 ? perm[i + j] : s0 + j; }
        }
        int *idx_runs, *idx_mixed;
        CHECK(hipMalloc(&idx_runs, (size_t)n_slots * 4)); CHECK(hipMemcpy(idx_runs, runs.data(), (size_t)n_slots * 4, hipMemcpyHostToDevice));
        CHECK(hipMalloc(&idx_mixed, (size_t)n_slots * 4)); CHECK(hipMemcpy(idx_mixed, mixed.data(), (size_t)n_slots * 4, hipMemcpyHostToDevice));
        const size_t pool_bytes = (size_t)n_slots * 256;
        struct { const char* what; const int* list; } lists[] = {{"random slots", idx}, {"runs of 558", idx_runs}, {"runs, every 4th far", idx_mixed}};
        for (auto& l : lists) {
            char name[160];
            const int* base = l.list;
            snprintf(name, sizeof name, "A  [slot][head] rows, %s, grid=2048", l.what);
            timeit(name, [&](int r) { hipLaunchKernelGGL(gather_a<2>, dim3(2048), dim3(256), 0, 0, kc, vc, base + (size_t)(r % windows) * tokens, tokens, 128, sink); });
            for (int rot : {8, 16, 48}) {
                snprintf(name, sizeof name, "A  rows, head h starts %d*h tokens in, %s", rot, l.what);
                timeit(name, [&](int r) { hipLaunchKernelGGL(gather_a_rot<2>, dim3(2048), dim3(256), 0, 0, kc, vc, base + (size_t)(r % windows) * tokens, tokens, 128, rot, sink); });
            }
            snprintf(name, sizeof name, "A' [head][slot] pools, %s, grid=2048", l.what);
            timeit(name, [&](int r) { hipLaunchKernelGGL(gather_a_headpools<2>, dim3(2048), dim3(256), 0, 0, kc, vc, base + (size_t)(r % windows) * tokens, tokens, 128, pool_bytes, sink); });
        }
    }
    timeit("stream read of the same bytes, grid=2048", [&](int r) {
        const int w = (int)std::max<size_t>(1, ((size_t)n_slots * 2048) / (size_t)bytes);
        hipLaunchKernelGGL(stream_read, dim3(2048), dim3(256), 0, 0, kc + (size_t)(r % w) * (size_t)bytes, (size_t)bytes, sink); });
    return 0;
}
