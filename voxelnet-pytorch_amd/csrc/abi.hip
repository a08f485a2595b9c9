#include "common.h"
extern "C" int vn_abi_version(void) { return 3; }
extern "C" const char *vn_build_info(void) { return "libvoxelnet_hip gfx950 (CDNA4) " __DATE__ " " __TIME__; }
