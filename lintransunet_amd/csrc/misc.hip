#include "common.h"

extern "C" int ltu_version(void) { return 1; }
