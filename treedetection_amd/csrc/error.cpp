// Error channel of the C ABI (include/treedet.h: td_last_error): one message buffer per calling thread.
// Its own translation unit because the host-only sanitizer build (make asan: contours / epilogue / region / geometry /
// tiffcodec with -fsanitize=address,undefined, no HIP objects) needs it without api.cpp's device entry points.
#include <cstdarg>
#include <cstdio>

static thread_local char g_err[1024] = "";

void td_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* td_last_error(void) { return g_err; }
