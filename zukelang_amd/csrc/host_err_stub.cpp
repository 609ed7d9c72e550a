// Error reporting of the HOST-ONLY sanitizer build (make asan-host): the product's set_error lives in zk_api.hip next to the HIP context.
#include "zk_err.h"

#include <stdio.h>
#include <string>

static std::string g_last;
namespace zk {
int set_error(int code, const char* what, const char* file, int line) {
    char buf[512];
    snprintf(buf, sizeof buf, "%s (%s:%d)", what ? what : "", file, line);
    g_last = buf;
    return code;
}
}  // namespace zk
extern "C" __attribute__((visibility("default"))) const char* zk_last_error(void) { return g_last.c_str(); }
extern "C" __attribute__((visibility("default"))) const char* zk_strerror(int code) { return code == 0 ? "ok" : "error (host-only sanitizer build: see zk_last_error)"; }
