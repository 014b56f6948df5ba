// Reference point (not shipped, not linked into the engine): rocBLAS DGEMM / DSYRK-shaped product of the
// Schur-complement size, to know what a tuned library reaches on this chip for C = A^T B, fp64.
// hipcc --offload-arch=gfx950 -O2 tools/rocblas_dgemm_ref.cpp -lrocblas -o /tmp/dgemm_ref
#include <hip/hip_runtime.h>
#include <rocblas/rocblas.h>
#include <cstdio>
#include <vector>
int main(int argc, char** argv) {
  int n = argc > 1 ? atoi(argv[1]) : 12544, k = argc > 2 ? atoi(argv[2]) : 3008;
  rocblas_handle h; rocblas_create_handle(&h);
  double *A, *B, *C;
  hipMalloc(&A, sizeof(double) * (size_t)n * k); hipMalloc(&B, sizeof(double) * (size_t)n * k); hipMalloc(&C, sizeof(double) * (size_t)n * n);
  std::vector<double> hA((size_t)n * k);
  for (size_t i = 0; i < hA.size(); ++i) hA[i] = (double)((i * 2654435761u) % 1000) / 1000.0 - 0.5;
  hipMemcpy(A, hA.data(), hA.size() * 8, hipMemcpyHostToDevice); hipMemcpy(B, hA.data(), hA.size() * 8, hipMemcpyHostToDevice);
  hipMemset(C, 0, sizeof(double) * (size_t)n * n);
  double alpha = -1.0, beta = 1.0;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  // row-major K x n operands == column-major n x K:  C(n x n) = A * B^T in column-major terms
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    rocblas_dgemm(h, rocblas_operation_none, rocblas_operation_transpose, n, n, k, &alpha, A, n, B, n, &beta, C, n);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("rocblas_dgemm n=%d k=%d: %.3f ms  %.2f TFLOP/s (full square)\n", n, k, ms, 2.0 * n * n * k / ms / 1e9);
  }
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    rocblas_dsyrk(h, rocblas_fill_lower, rocblas_operation_none, n, k, &alpha, A, n, &beta, C, n);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("rocblas_dsyrk n=%d k=%d: %.3f ms  %.2f TFLOP/s (triangle flops)\n", n, k, ms, 1.0 * n * n * k / ms / 1e9);
  }
  return 0;
}
