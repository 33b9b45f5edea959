// Kernel-selection switches and per-kernel timing exist only in the TOOLS build of the library
// (tools/_bin/libccengine_tools.so, compiled with -DCC_TOOLS by cc_amd/build.py): bench.py's instrumented eager step and the
// A/B scripts under tools/ load that build.  The product library cc_amd/libccengine.so reads no environment variable and keeps
// no state between calls: every switch below collapses to its default at compile time.
#pragma once
#include <stdlib.h>

namespace cctools {
#ifdef CC_TOOLS
inline int env_flag(const char* name) {
    const char* v = getenv(name);
    return (v && v[0] == '1') ? 1 : 0;
}
inline int env_int(const char* name, int dflt) {
    const char* v = getenv(name);
    return v ? atoi(v) : dflt;
}
#else
constexpr int env_flag(const char*) { return 0; }
constexpr int env_int(const char*, int dflt) { return dflt; }
#endif
}  // namespace cctools
