// Shim (test infrastructure): the reference's patched fmt lives in namespace jst::fmt and arrives through a meson wrap
// that is not available here; torch ships a header-only fmt.  Only jetstream/logger.hh's declarations need it for the
// reference's header-inline helpers (ApproxLog10, the waterfall ring plan) to compile IN PLACE -- no logging is called.
#pragma once
#ifndef FMT_HEADER_ONLY
#define FMT_HEADER_ONLY
#endif
#include <fmt/ostream.h>
namespace jst { namespace fmt = ::fmt; }
