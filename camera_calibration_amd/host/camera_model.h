// Host mirror of the reference's CameraModel hierarchy for the two generic models
// (APP/models/camera_model.h:42-204, central_generic.h, noncentral_generic.h, central_grid.h; APP =
// applications/camera_calibration/src/camera_calibration in the reference tree).  The objects hold the
// grids; every projection / unprojection call goes to the HIP engine through the C-ABI (include/cba.h),
// there is no CPU implementation behind them.
#pragma once
#include "../../include/cba.h"
#include "vis_types.h"

namespace vis {

class CameraModel {
 public:
  enum class Type { CentralGeneric = 0, CentralThinPrismFisheye = 1, CentralOpenCV = 2, CentralRadial = 3,
                    NoncentralGeneric = 4, InvalidType = 5, NumTypes = 5 };
  CameraModel(int width, int height, int min_x, int min_y, int max_x, int max_y, Type type)
      : m_width(width), m_height(height), m_calibration_min_x(min_x), m_calibration_min_y(min_y),
        m_calibration_max_x(max_x), m_calibration_max_y(max_y), m_type(type) {}
  virtual ~CameraModel() { release_device_model(); }
  // the device-resident copy used by Project / Unproject is per object: copies start without one
  CameraModel(const CameraModel& o)
      : device(o.device), m_width(o.m_width), m_height(o.m_height), m_calibration_min_x(o.m_calibration_min_x),
        m_calibration_min_y(o.m_calibration_min_y), m_calibration_max_x(o.m_calibration_max_x),
        m_calibration_max_y(o.m_calibration_max_y), m_type(o.m_type) {}
  CameraModel& operator=(const CameraModel& o) {
    if (this != &o) {
      device = o.device; m_width = o.m_width; m_height = o.m_height; m_calibration_min_x = o.m_calibration_min_x;
      m_calibration_min_y = o.m_calibration_min_y; m_calibration_max_x = o.m_calibration_max_x;
      m_calibration_max_y = o.m_calibration_max_y; m_type = o.m_type;
      grid_changed();
    }
    return *this;
  }
  virtual CameraModel* duplicate() = 0;
  virtual int update_parameter_count() const = 0;
  virtual bool GetGridResolution(int* rx, int* ry) const = 0;
  // CameraModel::Project / ProjectWithInitialEstimate / Unproject (camera_model.h:84-120) -> cba_project / cba_unproject
  bool Project(const Vec3d& local_point, Vec2d* result) const;
  bool ProjectWithInitialEstimate(const Vec3d& local_point, Vec2d* result) const;
  bool Unproject(double x, double y, Line3d* result) const;
  inline bool IsInCalibratedArea(double x, double y) const {  // camera_model.h:159-162
    return x >= m_calibration_min_x && y >= m_calibration_min_y && x < m_calibration_max_x + 1 && y < m_calibration_max_y + 1;
  }
  inline Vec2d CenterOfCalibratedArea() const {
    return Vec2d(0.5 * (m_calibration_min_x + m_calibration_max_x + 1), 0.5 * (m_calibration_min_y + m_calibration_max_y + 1));
  }
  static bool IsCentral(Type t) { return t != Type::NoncentralGeneric; }
  int width() const { return m_width; } int height() const { return m_height; }
  int calibration_min_x() const { return m_calibration_min_x; } int calibration_min_y() const { return m_calibration_min_y; }
  int calibration_max_x() const { return m_calibration_max_x; } int calibration_max_y() const { return m_calibration_max_y; }
  Type type() const { return m_type; }
  // packed view for the C-ABI
  virtual cba_camera abi_camera() const = 0;
  virtual std::vector<double> abi_grid() const = 0;
  virtual void set_abi_grid(const double* g) = 0;
  int device = 0;  // HIP device used by the model-level calls
  /// Project / Unproject run on a device-resident copy of the model (cba_model, include/cba.h) that is created on first
  /// use and refreshed when the grid may have changed (any non-const grid access marks it stale), so a caller that
  /// projects feature by feature (APP/calibration_report.cc:101-148) pays one small kernel launch per call.
  void grid_changed() const { m_dev_stale = true; }
 protected:
  cba_model* device_model(int device_ordinal) const;   // joint_optimization_hip.cc
  void release_device_model() const;
  mutable cba_model* m_dev = nullptr;
  mutable int m_dev_device = -1;
  mutable bool m_dev_stale = true;
  mutable cba_camera m_dev_camera = {};   // what m_dev was created for; a model whose description changed gets a new handle
  // NOTE: a reference returned by the non-const grid() accessors must not be kept across Project / Unproject calls -- the
  // accessor marks the device copy stale when it is CALLED, later writes through the kept reference are not seen.
  int m_width, m_height, m_calibration_min_x, m_calibration_min_y, m_calibration_max_x, m_calibration_max_y;
  Type m_type;
};

class CentralGenericModel : public CameraModel {
 public:
  static constexpr int IntrinsicsJacobianSize = 2 * 16;
  CentralGenericModel(int grid_resolution_x, int grid_resolution_y, int min_x, int min_y, int max_x, int max_y, int width, int height)
      : CameraModel(width, height, min_x, min_y, max_x, max_y, Type::CentralGeneric) { m_grid.SetSize(grid_resolution_x, grid_resolution_y); }
  CameraModel* duplicate() override { return new CentralGenericModel(*this); }
  int update_parameter_count() const override { return 2 * m_grid.width() * m_grid.height(); }
  bool GetGridResolution(int* rx, int* ry) const override { *rx = m_grid.width(); *ry = m_grid.height(); return true; }
  // Grid-only fitting (APP/models/central_generic.cc:267-431, 551-568), LM on the GPU through
  // cba_fit_grid_to_directions.  Defined in host/central_generic_fit_hip.cc.
  bool FitToDenseModel(const Image<Vec3d>& dense_model, int subsample_step, int max_iteration_count = 10);
  void FitToPixelDirections(const std::vector<Vec2d>& pixels, const std::vector<Vec3d>& directions, int max_iteration_count);
  // central_grid.h:127-131 (evaluated in float) and :150-154
  Vec2d GridPointToPixelCornerConv(int x, int y) const {
    return Vec2d(m_calibration_min_x + ((x - 1.f) / (m_grid.width() - 3.f)) * (m_calibration_max_x + 1 - m_calibration_min_x),
                 m_calibration_min_y + ((y - 1.f) / (m_grid.height() - 3.f)) * (m_calibration_max_y + 1 - m_calibration_min_y));
  }
  Vec2d PixelCornerConvToGridPoint(double x, double y) const {
    return Vec2d(1.f + (m_grid.width() - 3.f) * (x - m_calibration_min_x) / (m_calibration_max_x + 1 - m_calibration_min_x),
                 1.f + (m_grid.height() - 3.f) * (y - m_calibration_min_y) / (m_calibration_max_y + 1 - m_calibration_min_y));
  }
  void SetGrid(const Image<Vec3d>& g) { m_grid = g; grid_changed(); }
  const Image<Vec3d>& grid() const { return m_grid; }
  Image<Vec3d>& grid() { grid_changed(); return m_grid; }
  static int exterior_cells_per_side() { return 1; }
  cba_camera abi_camera() const override {
    return cba_camera{CBA_CENTRAL_GENERIC, m_width, m_height, m_calibration_min_x, m_calibration_min_y, m_calibration_max_x,
                      m_calibration_max_y, (int)m_grid.width(), (int)m_grid.height()};
  }
  std::vector<double> abi_grid() const override {
    size_t G = (size_t)m_grid.width() * m_grid.height();
    std::vector<double> g(3 * G);
    for (size_t i = 0; i < G; ++i) for (int k = 0; k < 3; ++k) g[3 * i + k] = m_grid.data()[i].v[k];
    return g;
  }
  void set_abi_grid(const double* g) override {
    size_t G = (size_t)m_grid.width() * m_grid.height();
    for (size_t i = 0; i < G; ++i) for (int k = 0; k < 3; ++k) m_grid.data()[i].v[k] = g[3 * i + k];
    grid_changed();
  }
 private:
  Image<Vec3d> m_grid;
};

class NoncentralGenericModel : public CameraModel {
 public:
  static constexpr int IntrinsicsJacobianSize = 5 * 16;
  NoncentralGenericModel(int grid_resolution_x, int grid_resolution_y, int min_x, int min_y, int max_x, int max_y, int width, int height)
      : CameraModel(width, height, min_x, min_y, max_x, max_y, Type::NoncentralGeneric) {
    m_point_grid.SetSize(grid_resolution_x, grid_resolution_y); m_direction_grid.SetSize(grid_resolution_x, grid_resolution_y);
  }
  CameraModel* duplicate() override { return new NoncentralGenericModel(*this); }
  int update_parameter_count() const override { return 5 * m_direction_grid.width() * m_direction_grid.height(); }
  bool GetGridResolution(int* rx, int* ry) const override { *rx = m_point_grid.width(); *ry = m_point_grid.height(); return true; }
  void SetPointGrid(const Image<Vec3d>& g) { m_point_grid = g; grid_changed(); }
  void SetDirectionGrid(const Image<Vec3d>& g) { m_direction_grid = g; grid_changed(); }
  const Image<Vec3d>& point_grid() const { return m_point_grid; }
  const Image<Vec3d>& direction_grid() const { return m_direction_grid; }
  Image<Vec3d>& point_grid() { grid_changed(); return m_point_grid; }
  Image<Vec3d>& direction_grid() { grid_changed(); return m_direction_grid; }
  // noncentral_generic.cc:136-146: the central model's direction grid, a zero point grid, its rectangle and size
  void InitializeFromCentralGenericModel(const CentralGenericModel& other) {
    m_direction_grid = other.grid();
    m_point_grid.SetSize(m_direction_grid.width(), m_direction_grid.height());
    for (size_t i = 0; i < (size_t)m_point_grid.width() * m_point_grid.height(); ++i) m_point_grid.data()[i] = Vec3d(0, 0, 0);
    m_calibration_min_x = other.calibration_min_x(); m_calibration_min_y = other.calibration_min_y();
    m_calibration_max_x = other.calibration_max_x(); m_calibration_max_y = other.calibration_max_y();
    m_width = other.width(); m_height = other.height();
    grid_changed();
  }
  // noncentral_generic.cc:148-154
  void Scale(double factor) {
    for (size_t i = 0; i < (size_t)m_point_grid.width() * m_point_grid.height(); ++i)
      for (int k = 0; k < 3; ++k) m_point_grid.data()[i].v[k] *= factor;
    grid_changed();
  }
  // same conversion as the central model (noncentral_generic.h, central_grid.h:150-154)
  Vec2d PixelCornerConvToGridPoint(double x, double y) const {
    return Vec2d(1.f + (m_point_grid.width() - 3.f) * (x - m_calibration_min_x) / (m_calibration_max_x + 1 - m_calibration_min_x),
                 1.f + (m_point_grid.height() - 3.f) * (y - m_calibration_min_y) / (m_calibration_max_y + 1 - m_calibration_min_y));
  }
  cba_camera abi_camera() const override {
    return cba_camera{CBA_NONCENTRAL_GENERIC, m_width, m_height, m_calibration_min_x, m_calibration_min_y, m_calibration_max_x,
                      m_calibration_max_y, (int)m_point_grid.width(), (int)m_point_grid.height()};
  }
  std::vector<double> abi_grid() const override {
    size_t G = (size_t)m_point_grid.width() * m_point_grid.height();
    std::vector<double> g(6 * G);
    for (size_t i = 0; i < G; ++i) for (int k = 0; k < 3; ++k) { g[3 * i + k] = m_direction_grid.data()[i].v[k]; g[3 * G + 3 * i + k] = m_point_grid.data()[i].v[k]; }
    return g;
  }
  void set_abi_grid(const double* g) override {
    size_t G = (size_t)m_point_grid.width() * m_point_grid.height();
    for (size_t i = 0; i < G; ++i) for (int k = 0; k < 3; ++k) { m_direction_grid.data()[i].v[k] = g[3 * i + k]; m_point_grid.data()[i].v[k] = g[3 * G + 3 * i + k]; }
    grid_changed();
  }
 private:
  Image<Vec3d> m_point_grid, m_direction_grid;
};

}  // namespace vis
