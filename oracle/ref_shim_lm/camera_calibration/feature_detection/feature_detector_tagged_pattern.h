// Stand-in for the reference's feature detector header (Qt / OpenCV behind it) -- TEST INFRASTRUCTURE ONLY: APP/dataset.cc is compiled
// whole into oracle/_ref/libcalibref_f14.so and only Dataset::ExtractKnownGeometries touches this class (never called there).
#pragma once
#include <unordered_map>
#include <libvis/eigen.h>
namespace vis {
class FeatureDetectorTaggedPattern {
 public:
  int GetPatternCount() const { return 0; }
  float GetCellLengthInMeters() const { return 0.f; }
  void GetCorners(int, std::unordered_map<int, Vec2i>*) const {}
};
}
