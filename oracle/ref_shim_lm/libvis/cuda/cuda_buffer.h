// Stand-in for <libvis/cuda/cuda_buffer.h> -- TEST INFRASTRUCTURE ONLY (oracle/_ref build): nothing is uploaded anywhere
#ifndef CBA_REF_SHIM_LM_CUDA_BUFFER_
#define CBA_REF_SHIM_LM_CUDA_BUFFER_
#include <cuda_runtime.h>
#include "libvis/image.h"
struct float3 { float x, y, z; };
inline float3 make_float3(float x, float y, float z) { return float3{x, y, z}; }
namespace vis {
template <class T> struct CUDABuffer_ {};
template <class T>
class CUDABuffer {
 public:
  CUDABuffer(int, int) {}
  template <class S> void UploadAsync(S, const Image<T>&) {}
  CUDABuffer_<T> ToCUDA() const { return CUDABuffer_<T>(); }
};
}
#endif
