// Stand-in for <libvis/logging.h> -- TEST INFRASTRUCTURE ONLY (oracle/_ref build).  The real header wraps loguru; the reference
// files compiled into oracle/_ref use its CHECK_* macros (abort on failure, as loguru does) and LOG(severity).
#ifndef CBA_REF_SHIM_LIBVIS_LOGGING_
#define CBA_REF_SHIM_LIBVIS_LOGGING_
#include <cstdlib>
#include <iostream>
#include "libvis/libvis.h"
#define CBA_REF_CHECK_OP(a, op, b) do { if (!((a) op (b))) { std::cerr << "CHECK failed: " #a " " #op " " #b << std::endl; std::abort(); } } while (0)
#ifndef CHECK_GE
#define CHECK(a) do { if (!(a)) { std::cerr << "CHECK failed: " #a << std::endl; std::abort(); } } while (0)
#define CHECK_EQ(a, b) CBA_REF_CHECK_OP(a, ==, b)
#define CHECK_NE(a, b) CBA_REF_CHECK_OP(a, !=, b)
#define CHECK_GE(a, b) CBA_REF_CHECK_OP(a, >=, b)
#define CHECK_GT(a, b) CBA_REF_CHECK_OP(a, >, b)
#define CHECK_LE(a, b) CBA_REF_CHECK_OP(a, <=, b)
#define CHECK_LT(a, b) CBA_REF_CHECK_OP(a, <, b)
#endif
#endif
