// lab: the error sink the product sources expect (csrc/api.hip owns it in the library)
#include <stdarg.h>
#include <stdio.h>
extern "C" __attribute__((visibility("hidden"))) void lafs_set_error(const char* fmt, ...) {
  va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr);
}
