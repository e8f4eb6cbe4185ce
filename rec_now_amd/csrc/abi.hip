#include "common.hpp"
extern "C" int recnow_abi_version(void) { return 1; }
