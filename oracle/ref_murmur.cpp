// ref_murmur.cpp -- C-ABI door into the reference's own MurmurHash3.cpp (compiled where it lies under /root/reference by
// oracle/build_ref.sh; output only in oracle/_ref/).  TEST INFRASTRUCTURE: validates the oracle's restatement and the device
// primitive on random inputs.
#include "MurmurHash3.h"

extern "C" void ref_murmur3_x64_128(const void* key, int len, unsigned seed, void* out) { MurmurHash3_x64_128(key, len, seed, out); }
extern "C" void ref_murmur3_x86_32(const void* key, int len, unsigned seed, void* out) { MurmurHash3_x86_32(key, len, seed, out); }
