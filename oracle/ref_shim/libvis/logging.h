// Stand-in for <libvis/logging.h> -- TEST INFRASTRUCTURE ONLY (oracle/_ref build).  The real header wraps loguru; the reference
// files compiled into oracle/_ref use its CHECK_* macros (abort on failure, as loguru does) and LOG(severity).
#ifndef CBA_REF_SHIM_LIBVIS_LOGGING_
#define CBA_REF_SHIM_LIBVIS_LOGGING_
#include <cstdlib>
#include <iostream>
#include "libvis/libvis.h"
// loguru's CHECK macros take a streamed message (CHECK(x) << "why"): a temporary that aborts in its destructor when the test failed
#include <sstream>
namespace cba_ref_shim {
struct CheckStream {
  bool fail; std::ostringstream os;
  CheckStream(bool f, const char* what) : fail(f) { if (fail) os << "CHECK failed: " << what << " "; }
  ~CheckStream() { if (fail) { std::cerr << os.str() << std::endl; std::abort(); } }
  template <class T> CheckStream& operator<<(const T& v) { if (fail) os << v; return *this; }
};
}
#define CBA_REF_CHECK_OP(a, op, b) ::cba_ref_shim::CheckStream(!((a) op (b)), #a " " #op " " #b)
#ifndef CHECK_GE
#define CHECK(a) ::cba_ref_shim::CheckStream(!(a), #a)
#define CHECK_EQ(a, b) CBA_REF_CHECK_OP(a, ==, b)
#define CHECK_NE(a, b) CBA_REF_CHECK_OP(a, !=, b)
#define CHECK_GE(a, b) CBA_REF_CHECK_OP(a, >=, b)
#define CHECK_GT(a, b) CBA_REF_CHECK_OP(a, >, b)
#define CHECK_LE(a, b) CBA_REF_CHECK_OP(a, <=, b)
#define CHECK_LT(a, b) CBA_REF_CHECK_OP(a, <, b)
#endif
#endif
