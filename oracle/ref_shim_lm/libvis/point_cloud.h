// Stand-in for <libvis/point_cloud.h> -- TEST INFRASTRUCTURE ONLY (oracle/_ref build): APP/util.h names the type in one declaration
#ifndef CBA_REF_SHIM_LM_POINT_CLOUD_
#define CBA_REF_SHIM_LM_POINT_CLOUD_
#include "libvis/eigen.h"
namespace vis {
template <class T> class PointCloud {};
struct Point3fC3u8 {};
typedef PointCloud<Point3fC3u8> Point3fC3u8Cloud;
typedef PointCloud<Vec3f> Point3fCloud;
}
#endif
