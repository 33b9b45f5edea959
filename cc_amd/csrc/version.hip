#include "../../include/ccengine.h"
// (ABI revision << 32) | first 32 bits of the SHA-1 over every kernel source (cc_amd/build.py passes -DCC_SRC_HASH):
// measurement artefacts that depend on the kernels (profiles/pmc_traffic.json) record it and are refused on a mismatch
#ifndef CC_SRC_HASH
#define CC_SRC_HASH 0
#endif
extern "C" size_t cc_version(void) { return ((size_t)2 << 32) | (size_t)(CC_SRC_HASH); }
