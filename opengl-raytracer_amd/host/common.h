// common.h -- logging / fatal-error conventions of the reference (src/core/common.h:53-105):
// Info / Warn print and continue, FatalError prints to stderr and abort()s.
#pragma once
#include <cstdio>
#include <cstdlib>

#define GLRT_Info(...) do { std::printf("[INFO] "); std::printf(__VA_ARGS__); std::printf("\n"); } while (0)
#define GLRT_Warn(...) do { std::fprintf(stderr, "[WARNING] "); std::fprintf(stderr, __VA_ARGS__); std::fprintf(stderr, "\n"); } while (0)
#define GLRT_FatalError(...)                                                                  \
    do {                                                                                      \
        std::fflush(stdout);                                                                  \
        std::fprintf(stderr, "[ERROR] %s:%d: ", __FILE__, __LINE__);                          \
        std::fprintf(stderr, __VA_ARGS__);                                                    \
        std::fprintf(stderr, "\n");                                                           \
        std::abort();                                                                         \
    } while (0)

#if defined(_WIN32)
#define GLRT_API __declspec(dllexport)
#else
#define GLRT_API __attribute__((visibility("default")))
#endif
