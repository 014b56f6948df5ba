// Stand-in for <libvis/libvis.h> -- TEST INFRASTRUCTURE ONLY (oracle/_ref build).  The reference header declares the
// Qt application wrapper and basic integer typedefs; the files compiled into oracle/_ref need only the typedefs.
#ifndef CBA_REF_SHIM_LIBVIS_
#define CBA_REF_SHIM_LIBVIS_
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <iostream>
#include <memory>
#include <string>
#include <unordered_map>
#include <vector>
// LOG(severity) << ...: the real header pulls in the loguru-based logging (LV/logging.h); messages go to stderr here.
#ifndef LOG
#define LOG(severity) std::cerr
#endif
namespace vis {
typedef std::size_t usize;
typedef uint8_t u8; typedef uint16_t u16; typedef uint32_t u32; typedef uint64_t u64;
typedef int8_t i8; typedef int16_t i16; typedef int32_t i32; typedef int64_t i64;
// The real header imports std into vis (LV/libvis.h:39).  This decides overload resolution in the reference code --
// e.g. sin(float) in ApplyLocalUpdateToQuaternion resolves to std::sin(float), evaluated in fp32 -- so it is mirrored.
using namespace std;
}
// Eigen's class-level operator new for aligned members: nothing to align in the stand-in
#ifndef EIGEN_MAKE_ALIGNED_OPERATOR_NEW
#define EIGEN_MAKE_ALIGNED_OPERATOR_NEW
#endif
#include "libvis/logging.h"   // the real libvis.h makes the CHECK_* macros available to every includer
#endif
