#include "../../include/ccengine.h"
extern "C" size_t cc_version(void) { return 1; }
