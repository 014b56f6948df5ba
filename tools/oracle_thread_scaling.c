/* Developer harness (not shipped): thread scaling of the CPU oracle's dense solver on the host it runs on.
 * gcc -O3 -march=x86-64-v3 -fopenmp -std=c11 -D_POSIX_C_SOURCE=199309L tools/oracle_thread_scaling.c oracle/cba_oracle.c -lm -o tools/bin/oracle_thread_scaling */
#include <stdio.h>
#include <stdlib.h>
#include <time.h>
#include "../oracle/cba_oracle.h"
static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
int main(int argc, char** argv) {
  int n = argc > 1 ? atoi(argv[1]) : 6144;
  double* A = malloc((size_t)n * n * 8), *b = malloc(n * 8), *x = malloc(n * 8);
  srand(1);
  for (size_t i = 0; i < (size_t)n * n; i++) A[i] = 0;
  for (int i = 0; i < n; i++) { for (int j = i; j < n; j++) A[(size_t)i * n + j] = (rand() / (double)RAND_MAX - 0.5) * 0.01; A[(size_t)i * n + i] = 1.0 + rand() / (double)RAND_MAX; b[i] = 1; }
  for (int a = 2; a < argc; ++a) {
    int nt = atoi(argv[a]);
    orc_set_num_threads(nt);
    double t = now();
    orc_ldlt_solve_upper(A, n, b, x);
    t = now() - t;
    printf("ldlt n=%d threads=%d %.3f s  %.2f GFLOP/s\n", n, nt, t, (double)n * n * n / 3 / t / 1e9);
    fflush(stdout);
  }
  return 0;
}
