// Stand-in for <libvis/eigen.h> -- TEST INFRASTRUCTURE ONLY (oracle/_ref build): vis::Vec*/Mat* aliases over the
// Eigen stand-in of this directory, and `using namespace Eigen` inside vis as the reference header does.
#ifndef CBA_REF_SHIM_LIBVIS_EIGEN_
#define CBA_REF_SHIM_LIBVIS_EIGEN_
#include <Eigen/Dense>
#include "libvis/libvis.h"
namespace vis {
using namespace Eigen;
typedef Matrix<double, 2, 1> Vec2d; typedef Matrix<double, 3, 1> Vec3d; typedef Matrix<double, 4, 1> Vec4d;
typedef Matrix<float, 2, 1> Vec2f;  typedef Matrix<float, 3, 1> Vec3f;
typedef Matrix<int, 2, 1> Vec2i;
typedef Matrix<double, 2, 2> Mat2d; typedef Matrix<double, 3, 3> Mat3d;
}
#endif
