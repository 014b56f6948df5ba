// Stand-in for APP/models/all_models.h -- TEST INFRASTRUCTURE ONLY (oracle/_ref build).
// The reference's header pulls in the three parametric models (OpenCV, radial, thin-prism fisheye), which are out of this project's
// scope (the generic models only) and whose sources need the real Sophus.  This stand-in keeps the dispatch macros' names and
// calling convention (a statement block that sees `_<object>` as a reference to the concrete model and `_<object>_type` as its type)
// for the two generic models, and declares the parametric classes as never-instantiated types so that the dynamic_casts which pick a
// regularisation branch (APP/bundle_adjustment/joint_optimization.cc:842-848) still compile.
#pragma once
#include <cstdlib>
#include <libvis/libvis.h>
#include "camera_calibration/models/central_generic.h"
#include "camera_calibration/models/noncentral_generic.h"
namespace vis {
// (parameters() / use_equidistant_projection(): named by SaveCameraModel's branches for these models, APP/io/calibration_io.cc:565-608, which
// no test can reach because no object of these classes can exist)
struct CbaNoParameters { double operator[](int) const { std::abort(); } int size() const { std::abort(); } };
class CentralOpenCVModel : public CameraModel { CentralOpenCVModel() = delete; public: CbaNoParameters parameters() const { std::abort(); } };
class CentralRadialModel : public CameraModel { CentralRadialModel() = delete; public: CbaNoParameters parameters() const { std::abort(); } };
class CentralThinPrismFisheyeModel : public CameraModel {
  CentralThinPrismFisheyeModel() = delete;
 public:
  CbaNoParameters parameters() const { std::abort(); }
  bool use_equidistant_projection() const { std::abort(); }
};

#define CBA_REF_MODEL_BRANCH(object, qualifier, Model, ...)                                                         \
  {                                                                                                                 \
    typedef Model _##object##_type;                                                                                 \
    qualifier _##object##_type& _##object = static_cast<qualifier _##object##_type&>(object);                       \
    (void)_##object;                                                                                                \
    __VA_ARGS__;                                                                                                    \
  }
#define CBA_REF_MODEL_DISPATCH(object, qualifier, ...)                                                              \
  {                                                                                                                 \
    if ((object).type() == CameraModel::Type::CentralGeneric) CBA_REF_MODEL_BRANCH(object, qualifier, CentralGenericModel, __VA_ARGS__) \
    else if ((object).type() == CameraModel::Type::NoncentralGeneric) CBA_REF_MODEL_BRANCH(object, qualifier, NoncentralGenericModel, __VA_ARGS__) \
    else std::abort();                                                                                              \
  }
#define IDENTIFY_CAMERA_MODEL(object, ...) CBA_REF_MODEL_DISPATCH(object, , __VA_ARGS__)
#define IDENTIFY_CONST_CAMERA_MODEL(object, ...) CBA_REF_MODEL_DISPATCH(object, const, __VA_ARGS__)
#define IDENTIFY_CAMERA_MODEL_TYPE(type, ...)                                                                       \
  {                                                                                                                 \
    if ((type) == CameraModel::Type::CentralGeneric) { typedef CentralGenericModel _##type; __VA_ARGS__; }          \
    else if ((type) == CameraModel::Type::NoncentralGeneric) { typedef NoncentralGenericModel _##type; __VA_ARGS__; } \
    else std::abort();                                                                                              \
  }
}
