// Shim for the meson-generated jetstream/config.hh (include/jetstream/config.hh.in): a release, shared, Linux, CPU-only
// configuration -- what the reference's default build defines for the headers oracle/ref_helpers.cc includes.
#pragma once
#define JETSTREAM_VERSION_STR "1.9.1"
#define JETSTREAM_VERSION_MAJOR 1
#define JETSTREAM_VERSION_MINOR 9
#define JETSTREAM_VERSION_PATCH 1
#define JST_IS_SHARED
#define JST_RELEASE_MODE
#define JST_OS_LINUX
#define JETSTREAM_BACKEND_CPU_AVAILABLE
