// Developer experiment: where do the workgroups of a GEMM-shaped launch land (XCD / shader engine / CU), on the null stream
// and on a CU-masked stream?  The bulk GEMM's tile order assumes "workgroup b runs on XCD b % 8"; if a masked queue
// dispatches differently, the L2 sharing of the operand panels is lost -- a candidate for the 20 % the mask costs.
//   hipcc --offload-arch=gfx950 -O2 tools/dispatch_probe.hip -o tools/bin/dispatch_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <map>
#include <vector>
__global__ void __launch_bounds__(256) k_probe(unsigned* out, unsigned long long* t, int spin_us) {
  __shared__ volatile double pad[9000];     // 72 KB: two workgroups per CU, like the 128 x 128 GEMM
  pad[threadIdx.x * 35] = 1.0;
  if (threadIdx.x == 0) {
    const unsigned xcc = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 20);     // HW_REG_XCC_ID[3:0]
    const unsigned hw = __builtin_amdgcn_s_getreg((32 - 1) << 11 | 0 << 6 | 4);      // HW_REG_HW_ID
    out[2 * blockIdx.x] = xcc; out[2 * blockIdx.x + 1] = hw;
    const unsigned long long t0 = wall_clock64();
    t[2 * blockIdx.x] = t0;
    while (wall_clock64() - t0 < (unsigned long long)spin_us * 100) {}
    t[2 * blockIdx.x + 1] = wall_clock64();
  }
  __syncthreads();
  if (pad[(threadIdx.x * 35 + 1) % 9000] == 3.0) out[0] = 0;
}
static void report(const char* name, hipStream_t s, int blocks, int spin_us, unsigned* d_out, unsigned long long* d_t) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k_probe, dim3(blocks), dim3(256), 0, s, d_out, d_t, spin_us);
  hipEventRecord(e0, s);
  hipLaunchKernelGGL(k_probe, dim3(blocks), dim3(256), 0, s, d_out, d_t, spin_us);
  hipEventRecord(e1, s);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned> h(2 * blocks); std::vector<unsigned long long> ht(2 * blocks);
  hipMemcpy(h.data(), d_out, sizeof(unsigned) * 2 * blocks, hipMemcpyDeviceToHost);
  hipMemcpy(ht.data(), d_t, sizeof(unsigned long long) * 2 * blocks, hipMemcpyDeviceToHost);
  int match = 0; int per_xcc[8] = {0}; int rot[8] = {0}; int rot_changes = 0, last_rot = -1;
  std::map<unsigned, int> per_cu;            // key: xcc << 16 | se << 8 | cu
  std::map<unsigned, int> per_se;
  for (int b = 0; b < blocks; ++b) {
    const unsigned xcc = h[2 * b] & 15, hw = h[2 * b + 1];
    const unsigned cu = (hw >> 8) & 15, se = (hw >> 13) & 7;
    if ((int)xcc == b % 8) ++match;
    const int r = ((int)xcc - b % 8 + 8) % 8; rot[r]++;
    if (r != last_rot) { if (last_rot >= 0) ++rot_changes; last_rot = r; }
    per_xcc[xcc & 7]++;
    per_cu[xcc << 16 | se << 8 | cu]++;
    per_se[xcc << 16 | se]++;
  }
  int cmin = 1 << 30, cmax = 0; for (auto& kv : per_cu) { cmin = std::min(cmin, kv.second); cmax = std::max(cmax, kv.second); }
  int smin = 1 << 30, smax = 0; for (auto& kv : per_se) { smin = std::min(smin, kv.second); smax = std::max(smax, kv.second); }
  // first wave of the launch: the first (2 x CUs) blocks -- do consecutive blocks alternate XCDs?
  int first_match = 0; const int first = std::min(blocks, 512);
  for (int b = 0; b < first; ++b) if ((int)(h[2 * b] & 15) == b % 8) ++first_match;
  unsigned long long tmin = ~0ull, tmax = 0; for (int b = 0; b < blocks; ++b) { tmin = std::min(tmin, ht[2 * b]); tmax = std::max(tmax, ht[2 * b + 1]); }
  printf("%-34s %d blocks x %d us: %.3f ms (device span %.3f ms); xcc == b %% 8 for %.1f %% (first 512: %.1f %%); per XCD", name, blocks, spin_us, ms,
         (tmax - tmin) / 1e5, 100.0 * match / blocks, 100.0 * first_match / first);
  for (int x = 0; x < 8; ++x) printf(" %d", per_xcc[x]);
  printf("; (xcc - b) mod 8 histogram");
  for (int x = 0; x < 8; ++x) printf(" %d", rot[x]);
  printf(", changes along b: %d", rot_changes);
  printf("; CUs used %zu, blocks per CU %d..%d; SEs used %zu, blocks per SE %d..%d\n", per_cu.size(), cmin, cmax, per_se.size(), smin, smax);
}
int main() {
  setvbuf(stdout, nullptr, _IONBF, 0);
  const int blocks = 3655, spin = 100;
  unsigned* d_out; unsigned long long* d_t;
  hipMalloc(&d_out, sizeof(unsigned) * 2 * blocks); hipMalloc(&d_t, sizeof(unsigned long long) * 2 * blocks);
  uint32_t m[8]; for (int i = 0; i < 8; ++i) m[i] = 0xffffffffu;
  hipStream_t s_all, s_m8, s_plain;
  hipStreamCreateWithFlags(&s_plain, hipStreamNonBlocking);
  hipExtStreamCreateWithCUMask(&s_all, 8, m);
  m[0] = 0xffffff00u;
  hipExtStreamCreateWithCUMask(&s_m8, 8, m);
  report("null stream", nullptr, blocks, spin, d_out, d_t);
  report("non-blocking stream", s_plain, blocks, spin, d_out, d_t);
  report("CU mask: all 256 bits set", s_all, blocks, spin, d_out, d_t);
  report("CU mask: bits 0-7 cleared", s_m8, blocks, spin, d_out, d_t);
  report("CU mask: bits 0-7 cleared, again", s_m8, blocks, spin, d_out, d_t);
  report("null stream again", nullptr, blocks, spin, d_out, d_t);
  return 0;
}
