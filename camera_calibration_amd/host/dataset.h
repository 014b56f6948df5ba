// Host mirror of PointFeature / Imageset / Dataset (APP/dataset.h:57-212 in the reference tree) and
// BAState (APP/bundle_adjustment/ba_state.h:46-97, ba_state.cc:78-91).
#pragma once
#include <string>
#include <unordered_map>
#include "camera_model.h"

namespace vis {

// APP/dataset.h:49-55
struct KnownGeometry {
  std::unordered_map<int, Vec2i> feature_id_to_position;   // feature id -> integer position on the pattern
  float cell_length_in_meters = 0.f;
};

struct PointFeature {
  PointFeature() = default;
  PointFeature(const Vec2f& xy_, int id_) : xy(xy_), id(id_) {}
  Vec2f xy;                                 // measurement, pixel-corner convention
  int id = -1;                              // pattern feature id
  int index = -1;                           // index into BAState::points
  Vec2d last_projection = Vec2d::Zero();    // warm start cache (joint_optimization.cc:325-343)
};

class Imageset {
 public:
  explicit Imageset(int num_cameras) : m_features(num_cameras) {}
  const std::vector<PointFeature>& FeaturesOfCamera(int c) const { return m_features[c]; }
  std::vector<PointFeature>& FeaturesOfCamera(int c) { return m_features[c]; }
  bool CameraHasFeatures(int c) const { return !m_features[c].empty(); }
  const std::string& GetFilename() const { return filename; }
  void SetFilename(const std::string& f) { filename = f; }
 private:
  std::vector<std::vector<PointFeature>> m_features;
  std::string filename;
};

class Dataset {
 public:
  Dataset() : m_num_cameras(0) {}
  explicit Dataset(int num_cameras) { Reset(num_cameras); }
  void Reset(int num_cameras) { m_num_cameras = num_cameras; image_sizes.assign(num_cameras, Vec2i()); m_imagesets.clear(); }
  void SetImageSize(int c, const Vec2i& s) { image_sizes[c] = s; }
  const Vec2i& GetImageSize(int c) const { return image_sizes[c]; }
  std::shared_ptr<Imageset> NewImageset() { m_imagesets.emplace_back(new Imageset(m_num_cameras)); return m_imagesets.back(); }
  void DeleteLastImageset() { m_imagesets.pop_back(); }
  std::shared_ptr<const Imageset> GetImageset(int i) const { return m_imagesets[i]; }
  std::shared_ptr<Imageset> GetImageset(int i) { return m_imagesets[i]; }
  int ImagesetCount() const { return (int)m_imagesets.size(); }
  int num_cameras() const { return m_num_cameras; }
  void SetKnownGeometriesCount(int n) { m_known_geometries.resize(n); }
  int KnownGeometriesCount() const { return (int)m_known_geometries.size(); }
  KnownGeometry& GetKnownGeometry(int i) { return m_known_geometries[i]; }
  const KnownGeometry& GetKnownGeometry(int i) const { return m_known_geometries[i]; }
 private:
  std::vector<KnownGeometry> m_known_geometries;
  int m_num_cameras;
  std::vector<Vec2i> image_sizes;
  std::vector<std::shared_ptr<Imageset>> m_imagesets;
};

struct BAState {
  int num_cameras() const { return (int)intrinsics.size(); }
  int num_imagesets() const { return (int)image_used.size(); }
  SE3d image_tr_global(int camera_index, int imageset_index) const { return camera_tr_rig[camera_index] * rig_tr_global[imageset_index]; }
  // ba_state.cc:78-91
  void ComputeFeatureIdToPointsIndex(Dataset* dataset) {
    for (int i = 0; i < dataset->ImagesetCount(); ++i)
      for (int c = 0; c < dataset->num_cameras(); ++c)
        for (PointFeature& f : dataset->GetImageset(i)->FeaturesOfCamera(c)) {
          auto it = feature_id_to_points_index.find(f.id);
          f.index = (it == feature_id_to_points_index.end()) ? -1 : it->second;
        }
  }
  std::vector<bool> image_used;
  std::unordered_map<int, int> feature_id_to_points_index;
  std::vector<SE3d> camera_tr_rig;
  std::vector<SE3d> rig_tr_global;
  std::vector<std::shared_ptr<CameraModel>> intrinsics;
  std::vector<Vec3d> points;
};

}  // namespace vis
