// Stand-in for APP/models/cuda_central_generic_model.cuh -- TEST INFRASTRUCTURE ONLY (oracle/_ref build): the members that
// CentralGenericModel::CreateCUDACameraModel fills in (never called by the tests; the reference's CUDA path is out of scope).
#ifndef CBA_REF_SHIM_LM_CUDA_CENTRAL_GENERIC_MODEL_
#define CBA_REF_SHIM_LM_CUDA_CENTRAL_GENERIC_MODEL_
#include "libvis/cuda/cuda_buffer.h"
namespace vis {
class CUDACameraModel { public: virtual ~CUDACameraModel() {} };
class CUDACentralGenericModel : public CUDACameraModel {
 public:
  int m_width, m_height, m_calibration_min_x, m_calibration_min_y, m_calibration_max_x, m_calibration_max_y;
  CUDABuffer_<float3> m_grid;
};
}
#endif
