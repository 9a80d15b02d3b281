// dam_api.hip -- library identification entry points of include/dam_hip.h.
#include "dam_common.h"

extern "C" const char* dam_arch(void) { return "gfx950"; }
extern "C" int dam_abi_version(void) { return DAM_ABI_VERSION; }
