// vdb_rccl.h -- the RCCL entry points this library calls, resolved at run time (dlopen: an RCCL already in the process --
// PyTorch ships one -- is reused, otherwise librccl.so.1 of the ROCm installation is loaded).  Shared by vdb_shard.cpp (one
// process per GPU) and vdb_multi.cpp (one process, several GPUs).  rccl.h: ncclResult_t = int, ncclComm_t = opaque pointer,
// ncclUniqueId = 128 opaque bytes passed BY VALUE, ncclDataType_t ncclInt8 = 0.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

#include "../../include/vdb_shard.h"

namespace vdb_rccl {

struct NcclId { char internal[VDB_SHARD_UNIQUE_ID_BYTES]; };
typedef int (*fn_get_unique_id)(NcclId*);
typedef int (*fn_comm_init_rank)(void**, int, NcclId, int);
typedef int (*fn_comm_init_all)(void**, int, const int*);
typedef int (*fn_comm_destroy)(void*);
typedef int (*fn_comm_count)(void*, int*);
typedef int (*fn_all_gather)(const void*, void*, size_t, int, void*, hipStream_t);
typedef int (*fn_group)(void);
typedef const char* (*fn_error_string)(int);
struct Rccl {
    void* lib = nullptr;
    fn_get_unique_id get_unique_id = nullptr;
    fn_comm_init_rank comm_init_rank = nullptr;
    fn_comm_init_all comm_init_all = nullptr;          // single-process, one communicator per listed device
    fn_comm_destroy comm_destroy = nullptr;
    fn_comm_count comm_count = nullptr;
    fn_all_gather all_gather = nullptr;
    fn_group group_start = nullptr, group_end = nullptr;
    fn_error_string error_string = nullptr;
    char why[256] = {0};
};
// null when librccl could not be loaded: why() then says why
const Rccl* rccl();
const char* why();

}  // namespace vdb_rccl
