// On-disk formats of the calibration pipeline (see calibration_io.h).
#include "calibration_io.h"

#include <arpa/inet.h>
#include <sys/stat.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iomanip>
#include <map>
#include <sstream>

namespace vis {
namespace {

void MakeDirs(const std::string& dir) {
  if (dir.empty()) return;
  std::string cur;
  for (size_t i = 0; i <= dir.size(); ++i) {
    if (i == dir.size() || dir[i] == '/') {
      if (!cur.empty()) mkdir(cur.c_str(), 0777);
    }
    if (i < dir.size()) cur.push_back(dir[i]);
  }
}
std::string DirOf(const std::string& path) {
  size_t p = path.rfind('/');
  return p == std::string::npos ? std::string() : path.substr(0, p);
}
std::string Join(const std::string& a, const std::string& b) { return (!a.empty() && a.back() == '/') ? a + b : a + "/" + b; }

// integers through htonl, floats raw: APP/io/io_util.h:37-120
void write_u32(u32 v, FILE* f) { u32 t = htonl(v); fwrite(&t, 4, 1, f); }
void write_i32(int v, FILE* f) { u32 t = htonl((u32)v); fwrite(&t, 4, 1, f); }
void write_f32(float v, FILE* f) { fwrite(&v, 4, 1, f); }
bool read_u32(u32* v, FILE* f) { u32 t; if (fread(&t, 4, 1, f) != 1) return false; *v = ntohl(t); return true; }
bool read_i32(int* v, FILE* f) { u32 t; if (fread(&t, 4, 1, f) != 1) return false; *v = (int)ntohl(t); return true; }
bool read_f32(float* v, FILE* f) { return fread(v, 4, 1, f) == 1; }

// ---- reader for the YAML subset the reference writes -------------------------------------------------------
// top level:  "key : scalar" | "key : [v, v, ...]" | "key:" followed by a block sequence of flat maps
struct YamlDoc {
  std::map<std::string, std::string> scalars;
  std::map<std::string, std::vector<double>> lists;
  std::map<std::string, std::vector<std::map<std::string, std::string>>> seqs;
  bool ok = false;
};
std::string Trim(const std::string& s) {
  size_t a = s.find_first_not_of(" \t\r\n"), b = s.find_last_not_of(" \t\r\n");
  return a == std::string::npos ? std::string() : s.substr(a, b - a + 1);
}
YamlDoc ParseYaml(const char* path) {
  YamlDoc doc;
  std::ifstream in(path);
  if (!in) return doc;
  std::string line, cur_seq;
  while (std::getline(in, line)) {
    if (Trim(line).empty() || Trim(line)[0] == '#') continue;
    const bool indented = line[0] == ' ';
    std::string t = Trim(line);
    if (indented && !cur_seq.empty()) {
      bool new_item = false;
      if (t.compare(0, 2, "- ") == 0) { new_item = true; t = Trim(t.substr(2)); }
      size_t c = t.find(':');
      if (c == std::string::npos) return doc;
      if (new_item) doc.seqs[cur_seq].emplace_back();
      if (doc.seqs[cur_seq].empty()) return doc;
      doc.seqs[cur_seq].back()[Trim(t.substr(0, c))] = Trim(t.substr(c + 1));
      continue;
    }
    size_t c = t.find(':');
    if (c == std::string::npos) return doc;
    const std::string key = Trim(t.substr(0, c)), val = Trim(t.substr(c + 1));
    cur_seq.clear();
    if (val.empty()) { cur_seq = key; doc.seqs[key]; }
    else if (val[0] == '[') {
      std::vector<double>& out = doc.lists[key];
      const char* p = val.c_str() + 1;
      while (*p && *p != ']') {
        char* end = nullptr;
        double v = std::strtod(p, &end);
        if (end == p) break;
        out.push_back(v);
        p = end;
        while (*p == ',' || *p == ' ') ++p;
      }
    } else doc.scalars[key] = val;
  }
  doc.ok = true;
  return doc;
}

void SaveGrid(const Image<Vec3d>& grid, std::ofstream& stream) {
  stream << "[";
  for (u32 y = 0; y < grid.height(); ++y)
    for (u32 x = 0; x < grid.width(); ++x) {
      if (x != 0 || y != 0) stream << ", ";
      const Vec3d& e = grid.data()[x + (size_t)y * grid.width()];
      stream << e.x() << ", " << e.y() << ", " << e.z();
    }
  stream << "]" << std::endl;
}
bool LoadGrid(const YamlDoc& doc, const char* key, int gw, int gh, bool normalized, Image<Vec3d>* grid) {
  auto it = doc.lists.find(key);
  if (it == doc.lists.end() || (int)it->second.size() != 3 * gw * gh) {
    std::fprintf(stderr, "LoadCameraModel: expected %d entries in '%s'\n", 3 * gw * gh, key);
    return false;
  }
  grid->SetSize(gw, gh);
  for (int i = 0; i < gw * gh; ++i) {
    Vec3d v(it->second[3 * i], it->second[3 * i + 1], it->second[3 * i + 2]);
    if (normalized) { double n = v.norm(); v = Vec3d(v.x() / n, v.y() / n, v.z() / n); }   // :667-670
    grid->data()[i] = v;
  }
  return true;
}

}  // namespace

bool SaveDataset(const char* path, const Dataset& dataset) {
  MakeDirs(DirOf(path));
  FILE* file = fopen(path, "wb");
  if (!file) return false;
  fwrite("calib_data", 1, 10, file);
  write_u32(0, file);   // version
  write_u32((u32)dataset.num_cameras(), file);
  for (int c = 0; c < dataset.num_cameras(); ++c) { write_u32((u32)dataset.GetImageSize(c).x(), file); write_u32((u32)dataset.GetImageSize(c).y(), file); }
  write_u32((u32)dataset.ImagesetCount(), file);
  for (int i = 0; i < dataset.ImagesetCount(); ++i) {
    std::shared_ptr<const Imageset> imageset = dataset.GetImageset(i);
    const std::string& filename = imageset->GetFilename();
    write_u32((u32)filename.size(), file);
    fwrite(filename.data(), 1, filename.size(), file);
    for (int c = 0; c < dataset.num_cameras(); ++c) {
      const std::vector<PointFeature>& features = imageset->FeaturesOfCamera(c);
      write_u32((u32)features.size(), file);
      for (const PointFeature& f : features) { write_f32(f.xy.x(), file); write_f32(f.xy.y(), file); write_i32(f.id, file); }
    }
  }
  write_u32((u32)dataset.KnownGeometriesCount(), file);
  for (int g = 0; g < dataset.KnownGeometriesCount(); ++g) {
    const KnownGeometry& geometry = dataset.GetKnownGeometry(g);
    write_f32(geometry.cell_length_in_meters, file);
    write_u32((u32)geometry.feature_id_to_position.size(), file);
    for (const auto& item : geometry.feature_id_to_position) { write_i32(item.first, file); write_i32(item.second.x(), file); write_i32(item.second.y(), file); }
  }
  fclose(file);
  return true;
}

bool LoadDataset(const char* path, Dataset* dataset) {
  FILE* file = fopen(path, "rb");
  if (!file) { std::fprintf(stderr, "Cannot read file: %s\n", path); return false; }
  auto fail = [&](const char* why) { std::fprintf(stderr, "Cannot parse file: %s (%s)\n", path, why); fclose(file); return false; };
  char header[10];
  if (fread(header, 1, 10, file) != 10 || std::memcmp(header, "calib_data", 10) != 0) return fail("invalid file header");
  u32 version;
  if (!read_u32(&version, file) || version != 0) return fail("unsupported file format version");
  u32 num_cameras;
  if (!read_u32(&num_cameras, file)) return fail("unexpected end of file");
  dataset->Reset((int)num_cameras);
  for (u32 c = 0; c < num_cameras; ++c) {
    u32 w, h;
    if (!read_u32(&w, file) || !read_u32(&h, file)) return fail("unexpected end of file");
    dataset->SetImageSize((int)c, Vec2i((int)w, (int)h));
  }
  u32 num_imagesets;
  if (!read_u32(&num_imagesets, file)) return fail("unexpected end of file");
  for (u32 i = 0; i < num_imagesets; ++i) {
    std::shared_ptr<Imageset> imageset = dataset->NewImageset();
    u32 len;
    if (!read_u32(&len, file)) return fail("unexpected end of file");
    std::string filename(len, '\0');
    if (len && fread(&filename[0], 1, len, file) != len) return fail("unexpected end of file");
    imageset->SetFilename(filename);
    for (u32 c = 0; c < num_cameras; ++c) {
      u32 n;
      if (!read_u32(&n, file)) return fail("unexpected end of file");
      std::vector<PointFeature>& features = imageset->FeaturesOfCamera((int)c);
      features.resize(n);
      for (PointFeature& f : features) {
        float x, y; int id;
        if (!read_f32(&x, file) || !read_f32(&y, file) || !read_i32(&id, file)) return fail("unexpected end of file");
        f.xy = Vec2f(x, y); f.id = id;
      }
    }
  }
  u32 num_geometries;
  if (!read_u32(&num_geometries, file)) return fail("unexpected end of file");
  dataset->SetKnownGeometriesCount((int)num_geometries);
  for (u32 g = 0; g < num_geometries; ++g) {
    KnownGeometry& geometry = dataset->GetKnownGeometry((int)g);
    u32 n;
    if (!read_f32(&geometry.cell_length_in_meters, file) || !read_u32(&n, file)) return fail("unexpected end of file");
    for (u32 k = 0; k < n; ++k) {
      int id, x, y;
      if (!read_i32(&id, file) || !read_i32(&x, file) || !read_i32(&y, file)) return fail("unexpected end of file");
      geometry.feature_id_to_position[id] = Vec2i(x, y);
    }
  }
  fclose(file);
  return true;
}

bool SavePoses(const std::vector<bool>& image_used, const std::vector<SE3d>& image_tr_pattern, const char* path) {
  if (image_used.size() != image_tr_pattern.size()) return false;
  MakeDirs(DirOf(path));
  std::ofstream stream(path, std::ios::out);
  if (!stream) return false;
  stream << std::setprecision(14);
  stream << "# Each pose gives the B_tr_A transformation (i.e., A to B with right-multiplication), where the spaces A and B are defined by the filename. Quaternions are written as used by the Eigen library." << std::endl;
  stream << "pose_count: " << image_used.size() << std::endl;
  stream << "poses:" << std::endl;
  for (usize i = 0; i < image_used.size(); ++i) {
    if (!image_used[i]) continue;
    const SE3d& pose = image_tr_pattern[i];
    stream << "  - index: " << i << std::endl;
    stream << "    tx: " << pose.translation().x() << std::endl;
    stream << "    ty: " << pose.translation().y() << std::endl;
    stream << "    tz: " << pose.translation().z() << std::endl;
    stream << "    qx: " << pose.unit_quaternion().x() << std::endl;
    stream << "    qy: " << pose.unit_quaternion().y() << std::endl;
    stream << "    qz: " << pose.unit_quaternion().z() << std::endl;
    stream << "    qw: " << pose.unit_quaternion().w() << std::endl;
  }
  // the reference's convenience file next to it: the camera centres of the used poses as red vertices (:817-836)
  std::ofstream obj_stream((std::string(path) + ".obj").c_str(), std::ios::out);
  if (!obj_stream) return false;
  obj_stream << std::setprecision(14);
  for (usize i = 0; i < image_used.size(); ++i) {
    if (!image_used[i]) continue;
    const Vec3d camera_position = image_tr_pattern[i].inverse().translation();
    obj_stream << "v " << camera_position.x() << " " << camera_position.y() << " " << camera_position.z() << " 1 0 0" << std::endl;
  }
  return true;
}

bool LoadPoses(std::vector<bool>* image_used, std::vector<SE3d>* image_tr_pattern, const char* path) {
  YamlDoc doc = ParseYaml(path);
  if (!doc.ok || !doc.scalars.count("pose_count")) { std::fprintf(stderr, "Cannot read file: %s\n", path); return false; }
  const int pose_count = std::atoi(doc.scalars["pose_count"].c_str());
  image_used->clear();
  image_used->resize(pose_count, false);
  image_tr_pattern->assign(pose_count, SE3d());
  for (auto& node : doc.seqs["poses"]) {
    const int index = std::atoi(node["index"].c_str());
    if (index < 0 || index >= pose_count) { std::fprintf(stderr, "Error while parsing file: %s\n", path); return false; }
    (*image_used)[index] = true;
    auto num = [&](const char* k) { return std::strtod(node[k].c_str(), nullptr); };
    // SE3d(q, t) normalises the quaternion like Sophus' setQuaternion
    (*image_tr_pattern)[index] = SE3d(Quaterniond(num("qw"), num("qx"), num("qy"), num("qz")), Vec3d(num("tx"), num("ty"), num("tz")));
  }
  return true;
}

bool SaveCameraModel(const CameraModel& model, const char* path) {
  MakeDirs(DirOf(path));
  const auto* central = dynamic_cast<const CentralGenericModel*>(&model);
  const auto* noncentral = dynamic_cast<const NoncentralGenericModel*>(&model);
  if (!central && !noncentral) { std::fprintf(stderr, "SaveCameraModel() is not implemented for this camera model type.\n"); return false; }
  std::ofstream stream(path, std::ios::out);
  if (!stream) return false;
  stream << std::setprecision(14);
  stream << "type : " << (central ? "CentralGenericModel" : "NoncentralGenericModel") << std::endl;
  stream << "width : " << model.width() << std::endl;
  stream << "height : " << model.height() << std::endl;
  stream << "calibration_min_x : " << model.calibration_min_x() << std::endl;
  stream << "calibration_min_y : " << model.calibration_min_y() << std::endl;
  stream << "calibration_max_x : " << model.calibration_max_x() << std::endl;
  stream << "calibration_max_y : " << model.calibration_max_y() << std::endl;
  if (central) {
    stream << "grid_width : " << central->grid().width() << std::endl;
    stream << "grid_height : " << central->grid().height() << std::endl;
    stream << "# The grid is stored in row-major order, top to bottom. Each row is stored left to right. Each grid point is stored as x, y, z." << std::endl;
    stream << "grid : ";
    SaveGrid(central->grid(), stream);
  } else {
    stream << "grid_width : " << noncentral->point_grid().width() << std::endl;
    stream << "grid_height : " << noncentral->point_grid().height() << std::endl;
    stream << "# The grids are stored in row-major order, top to bottom. Each row is stored left to right. Each grid point is stored as x, y, z." << std::endl;
    stream << "point_grid : ";
    SaveGrid(noncentral->point_grid(), stream);
    stream << "direction_grid : ";
    SaveGrid(noncentral->direction_grid(), stream);
  }
  return true;
}

std::shared_ptr<CameraModel> LoadCameraModel(const char* path) {
  YamlDoc doc = ParseYaml(path);
  if (!doc.ok) { std::fprintf(stderr, "Cannot read file: %s\n", path); return nullptr; }
  auto geti = [&](const char* k) { return std::atoi(doc.scalars[k].c_str()); };
  const int width = geti("width"), height = geti("height");
  if (width < 1 || height < 1) { std::fprintf(stderr, "Cannot parse file: %s (invalid image dimensions)\n", path); return nullptr; }
  const std::string type = doc.scalars["type"];
  const int gw = geti("grid_width"), gh = geti("grid_height");
  if (gw < 4 || gh < 4) { std::fprintf(stderr, "Cannot parse file: %s (invalid grid dimensions)\n", path); return nullptr; }
  const int min_x = geti("calibration_min_x"), min_y = geti("calibration_min_y"), max_x = geti("calibration_max_x"), max_y = geti("calibration_max_y");
  if (type == "CentralGenericModel" || type == "CentralGenericBSplineModel") {
    Image<Vec3d> grid;
    if (!LoadGrid(doc, "grid", gw, gh, /*normalized*/ true, &grid)) return nullptr;
    auto* model = new CentralGenericModel(gw, gh, min_x, min_y, max_x, max_y, width, height);
    model->SetGrid(grid);
    return std::shared_ptr<CameraModel>(model);
  }
  if (type == "NoncentralGenericModel" || type == "NoncentralGenericBSplineModel") {
    Image<Vec3d> point_grid, direction_grid;
    if (!LoadGrid(doc, "point_grid", gw, gh, /*normalized*/ false, &point_grid)) return nullptr;
    if (!LoadGrid(doc, "direction_grid", gw, gh, /*normalized*/ true, &direction_grid)) return nullptr;
    auto* model = new NoncentralGenericModel(gw, gh, min_x, min_y, max_x, max_y, width, height);
    model->SetPointGrid(point_grid);
    model->SetDirectionGrid(direction_grid);
    return std::shared_ptr<CameraModel>(model);
  }
  std::fprintf(stderr, "Cannot load camera model type: %s\n", type.c_str());
  return nullptr;
}

bool SavePointsAndIndexMapping(const BAState& calibration, const char* path) {
  MakeDirs(DirOf(path));
  std::ofstream stream(path, std::ios::out);
  if (!stream) return false;
  stream << std::setprecision(14);
  stream << "# Each point is stored as x, y, z." << std::endl;
  stream << "points : [";
  for (usize i = 0; i < calibration.points.size(); ++i) {
    const Vec3d& p = calibration.points[i];
    stream << p.x() << ", " << p.y() << ", " << p.z();
    if (i + 1 < calibration.points.size()) stream << ", ";
  }
  stream << "]" << std::endl;
  stream << "feature_id_to_point_index:" << std::endl;
  for (const auto& item : calibration.feature_id_to_points_index) {
    stream << "  - feature_id: " << item.first << std::endl;
    stream << "    point_index: " << item.second << std::endl;
  }
  stream.close();
  // the pattern points as blue vertices (:923-935)
  std::ofstream obj_stream((std::string(path) + ".obj").c_str(), std::ios::out);
  if (!obj_stream) return false;
  obj_stream << std::setprecision(14);
  for (usize i = 0; i < calibration.points.size(); ++i) {
    const Vec3d& p = calibration.points[i];
    obj_stream << "v " << p.x() << " " << p.y() << " " << p.z() << " 0 0 1" << std::endl;
  }
  return true;
}

bool LoadPointsAndIndexMapping(std::vector<Vec3d>* optimized_geometry, std::unordered_map<int, int>* feature_id_to_points_index,
                               const char* path) {
  YamlDoc doc = ParseYaml(path);
  if (!doc.ok || !doc.lists.count("points")) { std::fprintf(stderr, "Cannot read file: %s\n", path); return false; }
  const std::vector<double>& pts = doc.lists["points"];
  if (pts.size() % 3 != 0) { std::fprintf(stderr, "Cannot parse file: %s (points node size is not an integer multiple of 3)\n", path); return false; }
  optimized_geometry->resize(pts.size() / 3);
  for (size_t i = 0; i < pts.size() / 3; ++i) (*optimized_geometry)[i] = Vec3d(pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]);
  feature_id_to_points_index->clear();
  for (auto& node : doc.seqs["feature_id_to_point_index"])
    feature_id_to_points_index->insert(std::make_pair(std::atoi(node["feature_id"].c_str()), std::atoi(node["point_index"].c_str())));
  return true;
}

bool SaveBAState(const char* base_path, const BAState& state) {
  MakeDirs(base_path);
  if (!SavePoses(state.image_used, state.rig_tr_global, Join(base_path, "rig_tr_global.yaml").c_str())) return false;
  std::vector<bool> dummy(state.camera_tr_rig.size(), true);
  if (!SavePoses(dummy, state.camera_tr_rig, Join(base_path, "camera_tr_rig.yaml").c_str())) return false;
  for (int c = 0; c < state.num_cameras(); ++c) {
    std::ostringstream filename;
    filename << "intrinsics" << c << ".yaml";
    if (!SaveCameraModel(*state.intrinsics[c], Join(base_path, filename.str()).c_str())) return false;
  }
  return SavePointsAndIndexMapping(state, Join(base_path, "points.yaml").c_str());
}

bool LoadBAState(const char* base_path, BAState* state, Dataset* dataset) {
  if (!LoadPoses(&state->image_used, &state->rig_tr_global, Join(base_path, "rig_tr_global.yaml").c_str())) return false;
  std::vector<bool> dummy;
  if (!LoadPoses(&dummy, &state->camera_tr_rig, Join(base_path, "camera_tr_rig.yaml").c_str())) return false;
  state->intrinsics.clear();
  for (int c = 0;; ++c) {
    std::ostringstream filename;
    filename << "intrinsics" << c << ".yaml";
    const std::string model_path = Join(base_path, filename.str());
    struct stat sb;
    if (stat(model_path.c_str(), &sb) != 0) {
      if (c == 0) { std::fprintf(stderr, "No intrinsics file found since %s does not exist.\n", model_path.c_str()); return false; }
      break;
    }
    std::shared_ptr<CameraModel> model = LoadCameraModel(model_path.c_str());
    if (!model) return false;
    state->intrinsics.push_back(model);
  }
  if (!LoadPointsAndIndexMapping(&state->points, &state->feature_id_to_points_index, Join(base_path, "points.yaml").c_str())) return false;
  if (dataset) state->ComputeFeatureIdToPointsIndex(dataset);
  return true;
}

}  // namespace vis
