#include "../../include/crb_hip.h"
extern "C" int crb_abi_version(void) { return 1; }
