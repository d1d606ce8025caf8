/* TEST INFRASTRUCTURE ONLY: the slice of <rccl/rccl.h> that summerset_amd/csrc/comm.hip uses, for the emulator build
 * (tests/hostsim).  The shipped library links the real librccl; here the same comm.hip source is compiled against this
 * header and tests/hostsim/rccl_sim.cpp, which move the bytes between PROCESSES of one host through POSIX shared memory --
 * so that smr_comm_exchange's N > 1 path (who posts which receive and which send, in which order, with which sizes) runs
 * in the CPU suite with two gloo-launched ranks.  Semantics kept from NCCL: a communicator is (id, rank, world); init
 * blocks until every rank has joined; between a pair of ranks sends and receives match in posting order and their byte
 * counts must agree (the stand-in FAILS where RCCL would hang or overrun); the operations of a group complete together at
 * ncclGroupEnd; outside a group a send blocks until its receive has taken it.  Every wait has a deadline and ends in an
 * error, never in a hang. */
#ifndef SMR_HOSTSIM_RCCL_H
#define SMR_HOSTSIM_RCCL_H
#include <stddef.h>
#include <stdint.h>

#include <hip/hip_runtime.h>

#define NCCL_UNIQUE_ID_BYTES 128
typedef struct { char internal[NCCL_UNIQUE_ID_BYTES]; } ncclUniqueId;
typedef struct ncclSimComm *ncclComm_t;
typedef enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3, ncclInvalidArgument = 4,
               ncclInvalidUsage = 5 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclUint8 = 1, ncclInt32 = 2, ncclUint32 = 3, ncclInt64 = 4, ncclUint64 = 5 } ncclDataType_t;
typedef enum { ncclSum = 0, ncclProd = 1, ncclMax = 2, ncclMin = 3 } ncclRedOp_t;

extern "C" {
const char *ncclGetErrorString(ncclResult_t r);
ncclResult_t ncclGetUniqueId(ncclUniqueId *id);
ncclResult_t ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId id, int rank);
ncclResult_t ncclCommDestroy(ncclComm_t comm);
ncclResult_t ncclGroupStart(void);
ncclResult_t ncclGroupEnd(void);
ncclResult_t ncclSend(const void *buf, size_t count, ncclDataType_t dt, int peer, ncclComm_t comm, hipStream_t stream);
ncclResult_t ncclRecv(void *buf, size_t count, ncclDataType_t dt, int peer, ncclComm_t comm, hipStream_t stream);
ncclResult_t ncclAllReduce(const void *send, void *recv, size_t count, ncclDataType_t dt, ncclRedOp_t op, ncclComm_t comm, hipStream_t stream);
/* test hooks (not NCCL's): the posting log of this process' last completed group, "R<peer>:<bytes> S<peer>:<bytes> ..." */
const char *ncclSimLastGroupLog(void);
}
#endif
