// Library identification for libdeer_hip.so (include/deer_hip.h).
#include "common.h"
extern "C" const char* deer_hip_arch(void) { return "gfx950"; }
extern "C" int deer_hip_abi_version(void) { return 1; }
