// Stand-in for APP/models/all_models.h -- TEST INFRASTRUCTURE ONLY (oracle/_ref build).
// The reference's header pulls in the three parametric models (OpenCV, radial, thin-prism fisheye), which are out of this project's
// scope (the generic models only) and whose sources need the real Sophus.  This stand-in keeps the dispatch macros' names and
// calling convention (a statement block that sees `_<object>` as a reference to the concrete model and `_<object>_type` as its type)
// for the two generic models, and declares the parametric classes as aborting stand-ins so that the code which names them (the dynamic_casts
// that pick a regularisation branch, APP/bundle_adjustment/joint_optimization.cc:842-848; SaveCameraModel; ResampleModel) still compiles.
#pragma once
#include <cstdlib>
#include <libvis/libvis.h>
#include "camera_calibration/models/central_generic.h"
#include "camera_calibration/models/noncentral_generic.h"
namespace vis {
// The parametric classes: what SaveCameraModel's and ResampleModel's branches for them name (APP/io/calibration_io.cc:565-608,
// APP/calibration.cc:477-527) -- constructors, parameters(), FitToDenseModel -- declared so that those functions compile WHOLE; every
// member aborts: no test takes those branches (the generic models only).
struct CbaNoParameters { double operator[](int) const { std::abort(); } int size() const { std::abort(); } };
class CbaUnbuiltModel : public CameraModel {
 public:
  CbaUnbuiltModel(int width, int height, Type type) : CameraModel(width, height, 0, 0, width - 1, height - 1, type) {}
  CameraModel* duplicate() override { std::abort(); }
  bool Project(const Vec3d&, Vec2d*) const override { std::abort(); }
  bool ProjectWithInitialEstimate(const Vec3d&, Vec2d*) const override { std::abort(); }
  bool Unproject(double, double, Line3d*) const override { std::abort(); }
  int update_parameter_count() const override { std::abort(); }
  CbaNoParameters parameters() const { std::abort(); }
};
class CentralOpenCVModel : public CbaUnbuiltModel {
 public:
  CentralOpenCVModel(int width, int height) : CbaUnbuiltModel(width, height, Type::CentralOpenCV) {}
  bool FitToDenseModel(const Image<Vec3d>&, Mat3d*, int, int) { std::abort(); }
};
class CentralRadialModel : public CbaUnbuiltModel {
 public:
  CentralRadialModel(int width, int height, int) : CbaUnbuiltModel(width, height, Type::CentralRadial) {}
  bool FitToDenseModel(const Image<Vec3d>&, int, int) { std::abort(); }
};
class CentralThinPrismFisheyeModel : public CbaUnbuiltModel {
 public:
  CentralThinPrismFisheyeModel(int width, int height, bool) : CbaUnbuiltModel(width, height, Type::CentralThinPrismFisheye) {}
  bool FitToDenseModel(const Image<Vec3d>&, Mat3d*, int, int) { std::abort(); }
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
