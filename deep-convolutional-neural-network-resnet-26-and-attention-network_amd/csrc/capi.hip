// ABI bookkeeping for libmil_hip.so (see include/mil_hip.h).
#include "common.cuh"

extern "C" int mil_abi_version(void) { return 1; }
