// Error reporting and version entry points of libyolo_hip.so.
#include "common.h"

namespace yolo {

char* last_error_buf() {
    static thread_local char buf[512] = {0};
    return buf;
}

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(last_error_buf(), 512, fmt, ap);
    va_end(ap);
    return code;
}

}  // namespace yolo

extern "C" const char* yolo_last_error(void) { return yolo::last_error_buf(); }
extern "C" int yolo_abi_version(void) { return 6; }
