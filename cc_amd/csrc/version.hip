#include "../../include/ccengine.h"
extern "C" int cc_version(void) { return 1; }
