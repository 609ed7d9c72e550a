// Error plumbing of libzkmi355x.so: no HIP dependency, so that the host-only translation units (pairing_host.hip) also build with plain g++
// under AddressSanitizer / UBSan (make asan-host).
#pragma once
#include "../../include/zkmi355x.h"

namespace zk {
int set_error(int code, const char* what, const char* file, int line);
}

#define ZK_FAIL(code, what) return ::zk::set_error((code), (what), __FILE__, __LINE__)
#define ZKCHK(expr)                 \
    do {                            \
        int _rc = (expr);           \
        if (_rc != ZK_OK) return _rc; \
    } while (0)
