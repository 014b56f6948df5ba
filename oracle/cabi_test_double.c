/* oracle/cabi_test_double.c -- TEST INFRASTRUCTURE ONLY: a CPU test double of the part of include/cba.h that the reference-side adapter calls
 * (integration/reference.patch: joint_optimization_hip.cc), backed by the oracle (cba_oracle.c).
 *
 * Purpose: to EXECUTE the adapter's marshalling on a machine without a GPU.  `make -C oracle patched_double` builds the PATCHED reference's
 * bundle-adjustment path (the reference's own OptimizeJointly with SchurMode::HIP, its Dataset / BAState / generic models, the adapter) with
 * this file in place of camera_calibration_amd/libcalib_ba_hip.so; tests/test_integration_patch.py then compares SchurMode::HIP with
 * SchurMode::Dense of that library.  What is under test is the adapter (observation order, sequential imageset indices, pose packing, the
 * GetGridForHIP / SetGridFromHIP hooks, the read-back incl. the warm-start cache, lambda and the accepted flag) -- NOT the engine: nothing here is
 * HIP code, and nothing in the product (camera_calibration_amd/, include/) links, loads or knows this file.  The object is compiled with hidden
 * visibility INTO oracle/_ref/patched_double/libcalibref_ba.so: its cba_* functions exist inside that one test library only and can neither
 * shadow nor be shadowed by the product's library when both are loaded into one process.
 */
#include <stdlib.h>
#include <string.h>

#include "../include/cba.h"
#include "cba_oracle.h"

struct cba_problem {
  cba_config cfg;
  orc_camera* cams;
  orc_problem pb;
  orc_state st;
  int64_t n_obs;
  float* xy; int32_t *pt, *im, *cm; double* lp;
  double** grids; size_t* grid_doubles;
};

static const char* g_error = "";
const char* cba_last_error(void) { return g_error; }

static size_t grid_doubles_of(const cba_camera* c) { return (size_t)(c->model_type == CBA_CENTRAL_GENERIC ? 3 : 6) * c->grid_w * c->grid_h; }

int cba_create(const cba_config* config, cba_problem** out) {
  if (!config || !out || config->n_cameras < 1) { g_error = "cba_create (test double): bad arguments"; return CBA_ERR_ARG; }
  cba_problem* p = (cba_problem*)calloc(1, sizeof(cba_problem));
  p->cfg = *config;
  p->cams = (orc_camera*)calloc(config->n_cameras, sizeof(orc_camera));
  p->grids = (double**)calloc(config->n_cameras, sizeof(double*));
  p->grid_doubles = (size_t*)calloc(config->n_cameras, sizeof(size_t));
  for (int c = 0; c < config->n_cameras; ++c) {
    const cba_camera* k = &config->cameras[c];
    p->cams[c].model_type = k->model_type == CBA_CENTRAL_GENERIC ? ORC_CENTRAL_GENERIC : ORC_NONCENTRAL_GENERIC;
    p->cams[c].width = k->width; p->cams[c].height = k->height;
    p->cams[c].calib_min_x = k->calib_min_x; p->cams[c].calib_min_y = k->calib_min_y;
    p->cams[c].calib_max_x = k->calib_max_x; p->cams[c].calib_max_y = k->calib_max_y;
    p->cams[c].grid_w = k->grid_w; p->cams[c].grid_h = k->grid_h;
    p->grid_doubles[c] = grid_doubles_of(k);
    p->grids[c] = (double*)calloc(p->grid_doubles[c], sizeof(double));
  }
  p->st.rig_tr_global = (double*)calloc(7 * (size_t)(config->n_images > 0 ? config->n_images : 1), sizeof(double));
  p->st.camera_tr_rig = (double*)calloc(7 * (size_t)config->n_cameras, sizeof(double));
  p->st.points = (double*)calloc(3 * (size_t)(config->n_points > 0 ? config->n_points : 1), sizeof(double));
  p->st.grids = p->grids;
  *out = p;
  return CBA_OK;
}

void cba_destroy(cba_problem* p) {
  if (!p) return;
  for (int c = 0; c < p->cfg.n_cameras; ++c) free(p->grids[c]);
  free(p->grids); free(p->grid_doubles); free(p->cams);
  free(p->st.rig_tr_global); free(p->st.camera_tr_rig); free(p->st.points);
  free(p->xy); free(p->pt); free(p->im); free(p->cm); free(p->lp);
  free(p);
}

int cba_set_observations(cba_problem* p, int64_t n, const float* xy, const int32_t* point_index, const int32_t* image_index,
                         const int32_t* camera_index, const double* last_projection) {
  free(p->xy); free(p->pt); free(p->im); free(p->cm); free(p->lp);
  const size_t m = (size_t)(n > 0 ? n : 1);
  p->n_obs = n;
  p->xy = (float*)malloc(2 * m * sizeof(float)); p->pt = (int32_t*)malloc(m * sizeof(int32_t));
  p->im = (int32_t*)malloc(m * sizeof(int32_t)); p->cm = (int32_t*)malloc(m * sizeof(int32_t));
  p->lp = (double*)calloc(2 * m, sizeof(double));
  memcpy(p->xy, xy, 2 * (size_t)n * sizeof(float)); memcpy(p->pt, point_index, (size_t)n * sizeof(int32_t));
  memcpy(p->im, image_index, (size_t)n * sizeof(int32_t)); memcpy(p->cm, camera_index, (size_t)n * sizeof(int32_t));
  if (last_projection) memcpy(p->lp, last_projection, 2 * (size_t)n * sizeof(double));
  orc_problem pb = {p->cfg.n_cameras, p->cfg.n_images, p->cfg.n_points, n, p->cams, p->xy, p->pt, p->im, p->cm, p->lp,
                    p->cfg.numerical_diff_delta, p->cfg.localize_only, p->cfg.eliminate_points};
  p->pb = pb;
  return CBA_OK;
}

int cba_set_state(cba_problem* p, const double* rig_tr_global, const double* camera_tr_rig, const double* points, const double* const* grids) {
  memcpy(p->st.rig_tr_global, rig_tr_global, 7 * (size_t)p->cfg.n_images * sizeof(double));
  memcpy(p->st.camera_tr_rig, camera_tr_rig, 7 * (size_t)p->cfg.n_cameras * sizeof(double));
  memcpy(p->st.points, points, 3 * (size_t)p->cfg.n_points * sizeof(double));
  for (int c = 0; c < p->cfg.n_cameras; ++c) memcpy(p->grids[c], grids[c], p->grid_doubles[c] * sizeof(double));
  return CBA_OK;
}

int cba_get_state(cba_problem* p, double* rig_tr_global, double* camera_tr_rig, double* points, double* const* grids) {
  memcpy(rig_tr_global, p->st.rig_tr_global, 7 * (size_t)p->cfg.n_images * sizeof(double));
  memcpy(camera_tr_rig, p->st.camera_tr_rig, 7 * (size_t)p->cfg.n_cameras * sizeof(double));
  memcpy(points, p->st.points, 3 * (size_t)p->cfg.n_points * sizeof(double));
  for (int c = 0; c < p->cfg.n_cameras; ++c) memcpy(grids[c], p->grids[c], p->grid_doubles[c] * sizeof(double));
  return CBA_OK;
}

int cba_get_last_projection(cba_problem* p, double* out) {
  memcpy(out, p->lp, 2 * (size_t)p->n_obs * sizeof(double));
  return CBA_OK;
}

int cba_cost(cba_problem* p, double* cost, int64_t* n_valid, double* cost_vector) {
  double* v = cost_vector ? cost_vector : (double*)malloc((size_t)(p->n_obs > 0 ? p->n_obs : 1) * sizeof(double));
  const double c = orc_cost_pass(&p->pb, &p->st, v);
  if (n_valid) { int64_t k = 0; for (int64_t i = 0; i < p->n_obs; ++i) k += v[i] >= 0; *n_valid = k; }
  if (!cost_vector) free(v);
  if (cost) *cost = c;
  return CBA_OK;
}

int cba_step(cba_problem* p, double init_lambda, int32_t max_lm_attempts, double init_lambda_factor, cba_report* report) {
  (void)max_lm_attempts; (void)init_lambda_factor;       /* the oracle's loop has the reference's 50 / 0.00001 built in */
  double lambda = 0; int32_t performed = 0, attempts = 0;
  double initial = 0;
  {   /* report.initial_cost: a cost pass that leaves the warm-start cache as it found it (the real engine's Jacobian pass reports it for free) */
    const size_t m = (size_t)(p->n_obs > 0 ? p->n_obs : 1);
    double* v = (double*)malloc(m * sizeof(double));
    double* keep = (double*)malloc(2 * m * sizeof(double));
    memcpy(keep, p->lp, 2 * m * sizeof(double));
    initial = orc_cost_pass(&p->pb, &p->st, v);
    memcpy(p->lp, keep, 2 * m * sizeof(double));
    free(keep); free(v);
  }
  const double final_cost = orc_optimize_jointly(&p->pb, &p->st, 1, init_lambda, &lambda, &performed, NULL, &attempts);
  if (report) {
    memset(report, 0, sizeof(*report));
    report->initial_cost = initial; report->final_cost = final_cost; report->lambda = lambda; report->accepted = performed; report->lm_attempts = attempts;
  }
  return CBA_OK;
}
