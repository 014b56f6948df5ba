// Grid-only LM of the central-generic model: CentralGenericModel::FitToPixelDirections.
// Reference: APP/models/central_generic.cc:44-83 (state), :86-150 (residual and Jacobian w.r.t. the local grid
// updates), :153-225 (cost function), :551-568 (driver), APP = applications/camera_calibration/src/camera_calibration.
//
//   k_fit_pass<true>    one lane per (grid point, direction) sample: r = normalize(sum_c w_c P_c) - measurement with
//                       the generated code's 15-digit weight literals, J = w_c (I - d d^T)/|v| [t1 t2] (3 x 32),
//                       cost 0.5 r^2 per scalar residual, the sample's 4x4 patch origin as bucket key
//   k_fit_pass<false>   cost-only pass through UnprojectFromGrid (exact-fraction weights)
//   k_fit_key_*         counting sort of the samples by patch origin
//   k_fit_accumulate    one wavefront per patch origin sums the 32x32 (upper) block of J^T J and the 32 entries of
//                       J^T r of its bucket in registers, then ONE atomic per entry -- patches of neighbouring
//                       origins overlap, samples of the same origin (tens to hundreds for dense fits) do not collide
#include <hip/hip_runtime.h>

#include "cba_internal.h"
#include "model.hip.h"

namespace cba {

constexpr int kFitRec = 3 + 96;   // r[3], J[3][32]

template <bool JAC>
__global__ void __launch_bounds__(256) k_fit_pass(int gw, int gh, const double* __restrict__ grid, const double* __restrict__ tang,
                                                  int64_t n, const double* __restrict__ gp, const double* __restrict__ dirs,
                                                  double* __restrict__ cost_vec, double* __restrict__ rec, int* __restrict__ keys,
                                                  int* __restrict__ status) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double gx = gp[2 * i] + 2, gy = gp[2 * i + 1] + 2;
  const int ix = JAC ? (int)floor(gx) : (int)gx, iy = JAC ? (int)floor(gy) : (int)gy;   // :94-95 vs b_spline.h:73-74
  if (ix - 3 < 0 || iy - 3 < 0 || ix >= gw || iy >= gh) {   // CHECK() in the reference: the sample lies outside the grid
    atomicExch(status, 3);
    cost_vec[3 * i] = cost_vec[3 * i + 1] = cost_vec[3 * i + 2] = -1.0;
    if (JAC) keys[i] = -1;
    return;
  }
  double wx[4], wy[4];
  if (JAC) { double dwx[4], dwy[4]; weights_jac(gx - (ix - 3), wx, dwx); weights_jac(gy - (iy - 3), wy, dwy); }
  else { weights_value(gx - (ix - 3), wx); weights_value(gy - (iy - 3), wy); }
  double v[3] = {0, 0, 0};
#pragma unroll
  for (int y = 0; y < 4; ++y) {
    double row[3] = {0, 0, 0};
    const double* P = grid + 3 * (size_t)((ix - 3) + (iy - 3 + y) * gw);
#pragma unroll
    for (int x = 0; x < 4; ++x) { row[0] += wx[x] * P[3 * x]; row[1] += wx[x] * P[3 * x + 1]; row[2] += wx[x] * P[3 * x + 2]; }
    v[0] += wy[y] * row[0]; v[1] += wy[y] * row[1]; v[2] += wy[y] * row[2];
  }
  double d[3], r[3];
  if (JAC) {
    const double inv = 1.0 / sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    d[0] = v[0] * inv; d[1] = v[1] * inv; d[2] = v[2] * inv;
    double* o = rec + (size_t)i * kFitRec;
#pragma unroll
    for (int a = 0; a < 3; ++a) { r[a] = d[a] - dirs[3 * i + a]; o[a] = r[a]; }
    for (int c = 0; c < 16; ++c) {
      const int seq = (ix - 3 + (c & 3)) + (iy - 3 + (c >> 2)) * gw;
      const double* t = tang + 6 * (size_t)seq;
      const double s = wx[c & 3] * wy[c >> 2] * inv;
      const double dt1 = d[0] * t[0] + d[1] * t[1] + d[2] * t[2], dt2 = d[0] * t[3] + d[1] * t[4] + d[2] * t[5];
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        o[3 + a * 32 + 2 * c] = s * (t[a] - d[a] * dt1);
        o[3 + a * 32 + 2 * c + 1] = s * (t[3 + a] - d[a] * dt2);
      }
    }
    keys[i] = (ix - 3) + (iy - 3) * gw;
  } else {
    normalize3(v[0], v[1], v[2]);
#pragma unroll
    for (int a = 0; a < 3; ++a) r[a] = v[a] - dirs[3 * i + a];
  }
#pragma unroll
  for (int a = 0; a < 3; ++a) cost_vec[3 * i + a] = 0.5 * r[a] * r[a];
}

__global__ void __launch_bounds__(256) k_fit_key_count(const int* __restrict__ keys, int64_t n, int* __restrict__ count) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && keys[i] >= 0) atomicAdd(count + keys[i], 1);
}
__global__ void __launch_bounds__(1024) k_fit_key_scan(const int* __restrict__ count, int n, int* __restrict__ start) {
  __shared__ int sh[1024];
  __shared__ int carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int base = 0; base < n; base += 1024) {
    const int i = base + threadIdx.x;
    const int v = (i < n) ? count[i] : 0;
    sh[threadIdx.x] = v;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
      const int t = ((int)threadIdx.x >= off) ? sh[threadIdx.x - off] : 0;
      __syncthreads();
      sh[threadIdx.x] += t;
      __syncthreads();
    }
    if (i < n) start[i] = carry + sh[threadIdx.x] - v;
    __syncthreads();
    if (threadIdx.x == 1023) carry += sh[1023];
    __syncthreads();
  }
  if (threadIdx.x == 0) start[n] = carry;
}
__global__ void __launch_bounds__(256) k_fit_key_fill(const int* __restrict__ keys, int64_t n, const int* __restrict__ start,
                                                      int* __restrict__ fill, int* __restrict__ order) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && keys[i] >= 0) order[start[keys[i]] + atomicAdd(fill + keys[i], 1)] = (int)i;
}

__global__ void __launch_bounds__(256) k_fit_accumulate(int gw, int n_keys, const double* __restrict__ rec, const int* __restrict__ start,
                                                        const int* __restrict__ order, double* __restrict__ H, int ld,
                                                        double* __restrict__ b) {
  constexpr int KG = 32, NPAIR = KG * (KG + 1) / 2, NE = (NPAIR + 63) / 64;
  __shared__ double sJ[4][3][KG];
  __shared__ double sR[4][4];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int key = blockIdx.x * 4 + wv;
  if (key >= n_keys) return;
  const int o_begin = start[key], o_end = start[key + 1];
  if (o_begin == o_end) return;
  unsigned short pi[NE], pk[NE];
#pragma unroll
  for (int t = 0; t < NE; ++t) {
    const int e = lane + 64 * t;
    int i = 0;
    if (e < NPAIR) { int rem = e; while (rem >= KG - i) { rem -= KG - i; ++i; } pi[t] = (unsigned short)i; pk[t] = (unsigned short)(i + rem); }
    else { pi[t] = 0; pk[t] = 0; }
  }
  double acc[NE];
#pragma unroll
  for (int t = 0; t < NE; ++t) acc[t] = 0.0;
  double bacc = 0.0;
  for (int idx = o_begin; idx < o_end; ++idx) {
    const double* r = rec + (size_t)order[idx] * kFitRec;
    __builtin_amdgcn_wave_barrier();
    for (int k = lane; k < 3 * KG; k += 64) sJ[wv][k / KG][k % KG] = r[3 + k];
    if (lane < 3) sR[wv][lane] = r[lane];
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
#pragma unroll
    for (int t = 0; t < NE; ++t) {
      const int i = pi[t], k = pk[t];
      acc[t] += sJ[wv][0][i] * sJ[wv][0][k] + sJ[wv][1][i] * sJ[wv][1][k] + sJ[wv][2][i] * sJ[wv][2][k];
    }
    if (lane < KG) bacc += sJ[wv][0][lane] * sR[wv][0] + sJ[wv][1][lane] * sR[wv][1] + sJ[wv][2][lane] * sR[wv][2];
  }
  const int cy0 = key / gw, cx0 = key - cy0 * gw;
  auto column = [&](int p) { const int c = p >> 1; return 2 * ((cx0 + (c & 3)) + (cy0 + (c >> 2)) * gw) + (p & 1); };
#pragma unroll
  for (int t = 0; t < NE; ++t) {
    if (lane + 64 * t >= NPAIR) continue;
    unsafeAtomicAdd(H + (size_t)column(pi[t]) * ld + column(pk[t]), acc[t]);   // patch order is ascending: row <= col
  }
  if (lane < KG) unsafeAtomicAdd(b + column(lane), bacc);
}

__global__ void __launch_bounds__(256) k_fit_set_rhs(double* __restrict__ S, int ld, const double* __restrict__ b, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) S[(size_t)i * ld + (ld - 1)] = b[i];
}
__global__ void __launch_bounds__(256) k_fit_diag_sum(const double* __restrict__ H, int ld, int n, double* __restrict__ out) {
  __shared__ double sh[256];
  double acc = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) acc += H[(size_t)i * ld + i];
  sh[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) { if ((int)threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s]; __syncthreads(); }
  if (threadIdx.x == 0) out[0] = sh[0];
}

int launch_fit_pass(bool jac, int gw, int gh, const double* grid, const double* tang, int64_t n, const double* gp,
                    const double* dirs, double* cost_vec, double* rec, int* keys, int* status, hipStream_t s) {
  if (n == 0) return CBA_OK;
  dim3 grid_dim((unsigned)((n + 255) / 256)), block(256);
  if (jac) hipLaunchKernelGGL(k_fit_pass<true>, grid_dim, block, 0, s, gw, gh, grid, tang, n, gp, dirs, cost_vec, rec, keys, status);
  else hipLaunchKernelGGL(k_fit_pass<false>, grid_dim, block, 0, s, gw, gh, grid, tang, n, gp, dirs, cost_vec, rec, keys, status);
  CBA_HIP(hipGetLastError());
  return CBA_OK;
}
int launch_fit_accumulate(int gw, int gh, int64_t n, const double* rec, const int* keys, int* count, int* start, int* fill,
                          int* order, double* H, int ld, double* b, hipStream_t s) {
  const int n_keys = gw * gh;
  CBA_HIP(hipMemsetAsync(count, 0, sizeof(int) * (size_t)n_keys, s));
  CBA_HIP(hipMemsetAsync(fill, 0, sizeof(int) * (size_t)n_keys, s));
  if (n == 0) return CBA_OK;
  dim3 g((unsigned)((n + 255) / 256)), block(256);
  hipLaunchKernelGGL(k_fit_key_count, g, block, 0, s, keys, n, count);
  hipLaunchKernelGGL(k_fit_key_scan, dim3(1), dim3(1024), 0, s, count, n_keys, start);
  hipLaunchKernelGGL(k_fit_key_fill, g, block, 0, s, keys, n, start, fill, order);
  hipLaunchKernelGGL(k_fit_accumulate, dim3((unsigned)((n_keys + 3) / 4)), block, 0, s, gw, n_keys, rec, start, order, H, ld, b);
  CBA_HIP(hipGetLastError());
  return CBA_OK;
}
int launch_fit_set_rhs(double* S, int ld, const double* b, int n, hipStream_t s) {
  hipLaunchKernelGGL(k_fit_set_rhs, dim3((n + 255) / 256), dim3(256), 0, s, S, ld, b, n);
  CBA_HIP(hipGetLastError());
  return CBA_OK;
}
int launch_fit_diag_sum(const double* H, int ld, int n, double* out, hipStream_t s) {
  hipLaunchKernelGGL(k_fit_diag_sum, dim3(1), dim3(256), 0, s, H, ld, n, out);
  CBA_HIP(hipGetLastError());
  return CBA_OK;
}

}  // namespace cba
