// State of one tensor-parallel group's direct-peer exchange (host side), shared by the two format builds of
// peer_allreduce.hip: the group functions (create / export / import / status / destroy) exist once, the kernels
// and their launcher once per 16-bit format.
#pragma once
#include <stdint.h>

#define NVX_PEER_MAX_RANKS 8
#define NVX_PEER_MAGIC 0x50454552u   // "PEER"

struct nvx_peer_comm_t {
    uint32_t magic;
    int tp, rank, device;
    int64_t max_rows, hidden;
    uint64_t* recv[NVX_PEER_MAX_RANKS];   // rank p's landing buffer as this process addresses it ([rank] = the local one)
    bool opened[NVX_PEER_MAX_RANKS];      // mapped with hipIpcOpenMemHandle (to be closed again)
    uint32_t* epoch;                      // device, max_rows: how many exchanges row i has been through
    int* status;                          // pinned host memory the kernels can write: 0, or 1 after a wait ran out
    uint64_t timeout_ticks;               // of the 100 MHz wall clock
};

// what a kernel needs of it, by value
struct nvx_peer_args_t {
    uint64_t* recv[NVX_PEER_MAX_RANKS];
    uint32_t* epoch;
    int* status;
    int tp, rank;
    int64_t max_rows;
    uint64_t timeout_ticks;
};
