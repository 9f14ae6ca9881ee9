// TEST INFRASTRUCTURE: the handful of NCCL/RCCL declarations csrc/comm.hip needs, for the x86 build of that file (tests/hostsim/build_sim.py) -- the simulator
// has no ROCm headers on its include path.  Matches the public NCCL ABI for these items (ncclUniqueId = 128 opaque bytes; ncclFloat32 = 7, ncclBfloat16 = 9,
// ncclSum = 0): the product build uses the real <rccl/rccl.h>, and the test-only libfakerccl.so (tests/hostsim/fakerccl.cpp) implements exactly this subset.
#pragma once
#include <stddef.h>
#define NCCL_UNIQUE_ID_BYTES 128
typedef struct { char internal[NCCL_UNIQUE_ID_BYTES]; } ncclUniqueId;
typedef struct ncclComm* ncclComm_t;
typedef enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3, ncclInvalidArgument = 4, ncclInvalidUsage = 5 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclUint8 = 1, ncclInt32 = 2, ncclUint32 = 3, ncclInt64 = 4, ncclUint64 = 5, ncclFloat16 = 6, ncclFloat32 = 7, ncclFloat64 = 8,
               ncclBfloat16 = 9 } ncclDataType_t;
typedef enum { ncclSum = 0, ncclProd = 1, ncclMax = 2, ncclMin = 3, ncclAvg = 4 } ncclRedOp_t;
