// internal helpers shared by the translation units of libescat.so (not part of the ABI)
#ifndef ESCAT_INTERNAL_H
#define ESCAT_INTERNAL_H
__attribute__((visibility("hidden"), format(printf, 2, 3))) int escat_fail(int code, const char *fmt, ...);
#endif
