// Stand-in for <libvis/eigen.h> next to ref_shim_lm/Eigen -- TEST INFRASTRUCTURE ONLY (oracle/_ref build)
#ifndef CBA_REF_SHIM_LM_LIBVIS_EIGEN_
#define CBA_REF_SHIM_LM_LIBVIS_EIGEN_
#include <Eigen/Dense>
#include <Eigen/Geometry>
#include "libvis/libvis.h"
namespace vis {
using namespace Eigen;
typedef Matrix<double, 2, 1> Vec2d; typedef Matrix<double, 3, 1> Vec3d; typedef Matrix<double, 4, 1> Vec4d;
typedef Matrix<float, 2, 1> Vec2f;  typedef Matrix<float, 3, 1> Vec3f;
typedef Matrix<int, 2, 1> Vec2i;
typedef Matrix<u8, 3, 1> Vec3u8;
typedef Matrix<double, 2, 2> Mat2d; typedef Matrix<double, 3, 3> Mat3d;
typedef ParametrizedLine<double, 3> Line3d;
}
#endif
