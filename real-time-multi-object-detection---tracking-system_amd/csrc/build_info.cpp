// build_info.cpp -- what this binary was built from: the digest of csrc/ the Makefile computed (tools/kernel_digest.py: every *.hip, *.h and the
// Makefile itself) and whether the diagnostic switches (make DIAG=1) are compiled in.  Rebuilt whenever any of those files changes.
#include "../../include/rtmodt.h"

#ifndef RTMODT_SRC_DIGEST
#define RTMODT_SRC_DIGEST "unknown"
#endif

extern "C" const char *rtmodt_build_info(void) {
#if defined(RTMODT_DIAG)
    return "csrc_sha256=" RTMODT_SRC_DIGEST " diag=1";
#else
    return "csrc_sha256=" RTMODT_SRC_DIGEST " diag=0";
#endif
}
