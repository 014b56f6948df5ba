// LOG(severity) << ... goes nowhere -- TEST INFRASTRUCTURE ONLY (oracle/_ref build; forced into every translation unit of
// libcalibref_f14.so / libcalibref_ba.so with -include so that the reference's progress messages do not end up in the test output)
#ifndef CBA_REF_SHIM_LM_QUIET_LOG_
#define CBA_REF_SHIM_LM_QUIET_LOG_
#include <ostream>
namespace cba_ref_shim {
struct NullLog {
  template <class T> NullLog& operator<<(const T&) { return *this; }
  NullLog& operator<<(std::ostream& (*)(std::ostream&)) { return *this; }
};
}
#define LOG(severity) ::cba_ref_shim::NullLog()
#endif
