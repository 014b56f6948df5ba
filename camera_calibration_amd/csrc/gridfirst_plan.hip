// Host-side plan of the grid-first elimination order: see gridfirst_plan.h.  No device code in this file.
#include "gridfirst_plan.h"

#include <algorithm>
#include <cmath>

namespace cba {

namespace {

inline int round_up_i(int v, int m) { return (v + m - 1) / m * m; }

struct Bits {
  int words = 0;
  std::vector<uint64_t> w;
  void init(int n_rows, int n_bits) { words = (n_bits + 63) / 64; w.assign((size_t)n_rows * words, 0ull); }
  uint64_t* row(int r) { return w.data() + (size_t)r * words; }
  const uint64_t* row(int r) const { return w.data() + (size_t)r * words; }
  void set(int r, int c) { row(r)[c >> 6] |= 1ull << (c & 63); }
  bool get(int r, int c) const { return (row(r)[c >> 6] >> (c & 63)) & 1ull; }
};

// automatic number of strips: two.  A banded factorisation is ONE chain of dependent pivots (158 blocks of 64 at BASELINE configs[1],
// 17 us each: 2.7 ms for 52 GFLOP); two strips eliminated towards the separator between them are two chains of half the length
// and cost nothing extra (no strip is eliminated away from a separator: 52 GFLOP).  Every further strip lies between two separators
// and fills one of them through its whole length: 3 / 4 strips cost 68 / 78 GFLOP, and measured on MI355X the launch is then bound
// by the tile tasks, not by the chains (4 strips: 2.5 ms).
int auto_strips(int n_lines) { return n_lines >= 16 ? 2 : 1; }

}  // namespace

void gf_flop_model(const cba_camera* cams, int C, int N, int P, double* pose_first, double* grid_first) {
  const double rp = 3.0 * P + (C > 1 ? 6.0 * C : 0.0);
  const double A = rp + 6.0 * N;
  double G = 0, band = 0, strip = 0;
  for (int c = 0; c < C; ++c) {
    const int ppg = cams[c].model_type == CBA_CENTRAL_GENERIC ? 2 : 5;
    const double g = (double)ppg * cams[c].grid_w * cams[c].grid_h;
    const double hb = (3.0 * std::min(cams[c].grid_w, cams[c].grid_h) + 3.0) * ppg + ppg - 1;
    G += g;
    band += g * hb * hb;              // banded factor
    strip += 2.0 * hb * g * A;        // its row strip: every border column through the band
  }
  const double D = rp + G;
  // pose-first: block-sparse Schur product (about a third of the dense upper product is executed at the BASELINE configurations,
  // DESIGN.md section 2) + the dense D x D factorisation
  if (pose_first) *pose_first = 0.33 * D * D * 6.0 * N + D * D * D / 3.0;
  // grid-first: banded factor + row strip + border update (upper, K = G) + border factorisation
  if (grid_first) *grid_first = band + strip + A * A * G + A * A * A / 3.0;
}

void gf_order_imagesets(const GfPlan& pl, const std::vector<uint64_t>& touched, int N, int first_col, std::vector<int>* slot_of_image) {
  const int W = pl.grid_words, nbg = pl.nbg;
  slot_of_image->assign(N, 0);
  if (N == 0) return;
  // closure of every imageset's rows under the fill of the grid factor
  std::vector<uint64_t> act((size_t)N * W, 0ull);
  std::vector<int> first(N, nbg), bits(N, 0);
  for (int i = 0; i < N; ++i) {
    uint64_t* a = &act[(size_t)i * W];
    for (int w = 0; w < W; ++w) a[w] = touched[(size_t)i * W + w];
    for (int r = 0; r < nbg; ++r)
      if ((a[r >> 6] >> (r & 63)) & 1ull) {
        if (first[i] == nbg) first[i] = r;
        for (int w = r >> 6; w < W; ++w) a[w] |= pl.gridrow[(size_t)r * W + w];
      }
    for (int w = 0; w < W; ++w) bits[i] += __builtin_popcountll(a[w]);
  }
  std::vector<char> placed(N, 0);
  std::vector<uint64_t> uni(W);
  int slot = 0;
  while (slot < N) {
    // imagesets of the 128-column tile that slot's first column falls into: up to the first slot that starts in the next tile
    const int tile = (first_col + 6 * slot) >> 7;
    int count = 0;
    while (slot + count < N && ((first_col + 6 * (slot + count)) >> 7) == tile) ++count;
    int seed = -1;
    for (int i = 0; i < N; ++i)
      if (!placed[i] && (seed < 0 || first[i] < first[seed] || (first[i] == first[seed] && bits[i] < bits[seed]))) seed = i;
    for (int w = 0; w < W; ++w) uni[w] = act[(size_t)seed * W + w];
    (*slot_of_image)[seed] = slot; placed[seed] = 1;
    for (int k = 1; k < count; ++k) {
      int best = -1, best_grow = 0x7fffffff, best_dist = 0x7fffffff;
      for (int i = 0; i < N; ++i) {
        if (placed[i]) continue;
        int grow = 0, dist = 0;
        for (int w = 0; w < W; ++w) {
          const uint64_t a = act[(size_t)i * W + w];
          grow += __builtin_popcountll(a & ~uni[w]);
          dist += __builtin_popcountll(a ^ uni[w]);
        }
        if (grow < best_grow || (grow == best_grow && dist < best_dist)) { best = i; best_grow = grow; best_dist = dist; }
      }
      for (int w = 0; w < W; ++w) uni[w] |= act[(size_t)best * W + w];
      (*slot_of_image)[best] = slot + k; placed[best] = 1;
    }
    slot += count;
  }
}

int gf_build_plan(const cba_camera* cams, int C, int N, int P, int strips_override, int single_tile_tasks, GfPlan* out) {
  if (!cams || !out || C < 1 || C > 16 || N < 0 || P < 0) return CBA_ERR_ARG;
  GfPlan& pl = *out;
  pl = GfPlan();
  pl.n_cameras = C; pl.n_images = N; pl.n_points = P;
  pl.gperm.resize(C);
  // ---- elimination order of the grid unknowns, rows of F ----
  struct Group { int f0, f1; };                       // rows [f0, f1) of F, f0 a multiple of 64 (f1 padded up by the next group's start)
  std::vector<std::vector<Group>> groups(C);
  std::vector<int> cam_first(C + 1, 0);
  for (int c = 0; c < C; ++c) {
    const int ppg = cams[c].model_type == CBA_CENTRAL_GENERIC ? 2 : 5;
    if (cams[c].grid_w < 4 || cams[c].grid_h < 4) return CBA_ERR_ARG;
    cam_first[c + 1] = cam_first[c] + ppg * cams[c].grid_w * cams[c].grid_h;
  }
  pl.G = cam_first[C];
  pl.f_of_grid.assign(pl.G, -1);
  int f = 0;
  for (int c = 0; c < C; ++c) {
    const int gw = cams[c].grid_w, gh = cams[c].grid_h;
    const int ppg = cams[c].model_type == CBA_CENTRAL_GENERIC ? 2 : 5;
    const bool long_is_x = gw >= gh;
    const int nl = long_is_x ? gw : gh, ns = long_is_x ? gh : gw;
    int S = strips_override > 0 ? strips_override : auto_strips(nl);
    while (S > 1 && nl - 3 * (S - 1) < S) --S;          // every strip keeps at least one line
    pl.strips[c] = S;
    // lines of the strips and of the separators
    std::vector<std::pair<int, int>> strip_lines, sep_lines;
    {
      const int interior = nl - 3 * (S - 1);
      const int base = interior / S, extra = interior % S;
      int pos = 0;
      for (int s = 0; s < S; ++s) {
        const int len = base + (s < extra ? 1 : 0);
        strip_lines.push_back({pos, pos + len});
        pos += len;
        if (s + 1 < S) { sep_lines.push_back({pos, pos + 3}); pos += 3; }
      }
    }
    pl.gperm[c].assign((size_t)gw * gh, -1);
    int rank = 0;
    auto place_lines = [&](int l0, int l1, bool backward = false) {
      for (int li = l0; li < l1; ++li) {
        const int l = backward ? l1 - 1 - (li - l0) : li;
        for (int t = 0; t < ns; ++t) {
          const int gx = long_is_x ? l : t, gy = long_is_x ? t : l;
          pl.gperm[c][gx + (size_t)gy * gw] = rank;
          for (int d = 0; d < ppg; ++d) pl.f_of_grid[cam_first[c] + ppg * rank + d] = f++;
          ++rank;
        }
      }
    };
    for (int s = 0; s < S; ++s) {
      Group g{f, 0};
      // A strip eliminated towards a separator meets it with its last band only; one eliminated AWAY from a separator fills that
      // separator's rows of the factor through the whole strip (a "spike": at BASELINE configs[1] 23 GFLOP per spike against 43
      // for the band itself).  The first strip runs forward, the last one backward (from the image border towards its separator):
      // two strips have no spike at all, S strips have S - 2.
      place_lines(strip_lines[s].first, strip_lines[s].second, S > 1 && s == S - 1);
      f = round_up_i(f, 64);
      g.f1 = f;
      groups[c].push_back(g);
    }
    if (S > 1) {
      Group g{f, 0};
      for (auto& sl : sep_lines) place_lines(sl.first, sl.second);
      f = round_up_i(f, 64);
      g.f1 = f;
      groups[c].push_back(g);
    }
  }
  pl.Gf = round_up_i(f, 128);
  if (pl.Gf > f) groups[C - 1].back().f1 = pl.Gf;        // a last block of identity rows joins the last chain
  pl.grid_of_f.assign(pl.Gf, -1);
  for (int e = 0; e < pl.G; ++e) pl.grid_of_f[pl.f_of_grid[e]] = e;
  pl.n_rp = (C > 1 ? 6 * C : 0) + 3 * P;
  pl.n_border = pl.n_rp + 6 * N;
  {
    const int nF = pl.Gf + pl.n_border;
    pl.n_fact = round_up_i(nF, 64);
    pl.n_pad = round_up_i(nF + 1, 128);
    if (pl.n_fact >= pl.n_pad) pl.n_pad += 128;        // the right-hand side column (n_pad - 1) stays outside the factored rows
  }
  pl.nbg = pl.Gf / 64; pl.nbf = pl.n_fact / 64; pl.ntc = pl.n_pad / 64;
  const int nbg = pl.nbg;
  for (int c = 0; c < C; ++c)
    for (auto& g : groups[c]) pl.chains.push_back(GfChain{g.f0 / 64, g.f1 / 64, 0, 0});

  // ---- block structure of the grid x grid part: two unknowns couple iff their control points are <= 3 apart in both directions ----
  Bits up;                                              // up.row(r): bit c >= r set = tile (r, c) non-zero
  up.init(nbg, nbg);
  for (int r = 0; r < nbg; ++r) up.set(r, r);
  for (int c = 0; c < C; ++c) {
    const int gw = cams[c].grid_w, gh = cams[c].grid_h;
    const int ppg = cams[c].model_type == CBA_CENTRAL_GENERIC ? 2 : 5;
    for (int y = 0; y < gh; ++y)
      for (int x = 0; x < gw; ++x) {
        const int fa = pl.f_of_grid[cam_first[c] + ppg * pl.gperm[c][x + (size_t)y * gw]];
        for (int y2 = std::max(0, y - 3); y2 <= std::min(gh - 1, y + 3); ++y2)
          for (int x2 = std::max(0, x - 3); x2 <= std::min(gw - 1, x + 3); ++x2) {
            const int fb = pl.f_of_grid[cam_first[c] + ppg * pl.gperm[c][x2 + (size_t)y2 * gw]];
            for (int d = 0; d < ppg; ++d)
              for (int d2 = 0; d2 < ppg; ++d2) {
                const int i = fa + d, j = fb + d2;
                if (i <= j) up.set(i >> 6, j >> 6);
              }
          }
      }
  }
  // consecutive blocks of one chain: the chain reads and rewrites tile (r, r + 1) whatever the geometry says (a block of identity
  // padding rows at the end of the grid part couples with nothing) -- part of the structure, so that it is formed for every attempt
  for (auto& ch : pl.chains)
    for (int r = ch.r0; r + 1 < ch.r1; ++r) up.set(r, r + 1);
  // half-bandwidth inside the strips (diagnostic)
  for (int c = 0; c < C; ++c) {
    const int ppg = cams[c].model_type == CBA_CENTRAL_GENERIC ? 2 : 5;
    pl.half_bandwidth = std::max(pl.half_bandwidth, (3 * std::min(cams[c].grid_w, cams[c].grid_h) + 3) * ppg + ppg - 1);
  }
  // ---- symbolic factorisation on blocks: eliminating block row k couples every pair of its columns ----
  for (int k = 0; k < nbg; ++k) {
    const uint64_t* rk = up.row(k);
    for (int i = k + 1; i < nbg; ++i) {
      if (!((rk[i >> 6] >> (i & 63)) & 1ull)) continue;
      uint64_t* ri = up.row(i);
      for (int w = i >> 6; w < up.words; ++w) {
        uint64_t m = rk[w];
        if (w == (i >> 6)) m &= ~0ull << (i & 63);
        ri[w] |= m;
      }
    }
  }
  // transposed: col.row(c): bit k < c set = L_kc non-zero
  Bits col;
  col.init(nbg, nbg);
  for (int r = 0; r < nbg; ++r)
    for (int c = r + 1; c < nbg; ++c)
      if (up.get(r, c)) col.set(c, r);
  for (auto& ch : pl.chains) {
    bool dep = false;
    for (int w = 0; w < col.words; ++w) dep = dep || col.row(ch.r0)[w] != 0;
    ch.dep = dep ? 1 : 0;
  }
  std::vector<int> chain_of(nbg, -1);
  for (size_t i = 0; i < pl.chains.size(); ++i)
    for (int r = pl.chains[i].r0; r < pl.chains[i].r1; ++r) chain_of[r] = (int)i;

  // ---- tasks ----
  // Rows are handed out in WAVE order: the i-th block row of every independent chain (strips of all cameras), i = 0, 1, ..., then
  // the same for the dependent chains (separators) -- row-major order over F would serve one strip after the other and the other
  // chains would starve.  Two ticket lists: list 0 = what the pivot chains wait for (PRE, PART, PARTFULL and the grid x grid tiles),
  // list 1 = the border tiles of the row strips (they feed the border update, not the chains).  Within a list a task only waits for
  // tasks in front of it or for a chain; list-0 tasks never wait for list-1 tasks, and the launch reserves workgroups that serve
  // list 0 first, so the chains cannot starve behind border tiles waiting for them (k_ldlt_sparse).
  std::vector<int> row_order;
  for (int phase = 0; phase < 2; ++phase) {
    int longest = 0;
    for (auto& ch : pl.chains) if (ch.dep == phase) longest = std::max(longest, ch.r1 - ch.r0);
    for (int i = 0; i < longest; ++i)
      for (auto& ch : pl.chains)
        if (ch.dep == phase && ch.r0 + i < ch.r1) row_order.push_back(ch.r0 + i);
  }
  std::vector<GfTask> list[2];
  auto add_task = [&](int which, int kind, int r, int c, const uint64_t* a, const uint64_t* b, int kend) {
    // K set = {k < kend : bit k of a (and of b, if given)} as intervals
    GfTask t{0, r, c, (int)pl.ivals.size()};
    int n = 0, run0 = -1;
    double rows = 0;
    for (int k = 0; k <= kend; ++k) {
      const bool on = k < kend && ((a[k >> 6] >> (k & 63)) & 1ull) && (!b || ((b[k >> 6] >> (k & 63)) & 1ull));
      if (on && run0 < 0) run0 = k;
      if (!on && run0 >= 0) { pl.ivals.push_back(GfIval{run0, k}); rows += k - run0; run0 = -1; ++n; }
    }
    t.kind_n = kind | (n << 8);
    list[which].push_back(t);
    pl.flops_grid += (kind == 4 ? 2.0 : 1.0) * (2.0 * 64 * 64 * 64 * rows + (kind == 2 || kind == 4 ? 2.0 * 64 * 64 * 64 : 0.0));
  };
  for (int r : row_order) {
    const GfChain& mine = pl.chains[chain_of[r]];
    // first block of a chain that depends on earlier rows: its diagonal tile less ALL rows above it (every one of them belongs to
    // a chain of the first phase or to an earlier block of this phase: in front of this task in list 0)
    if (mine.r0 == r && mine.dep) add_task(0, 3, r - 1, r, col.row(r), nullptr, r);
    const bool pre = r + 1 < mine.r1;
    if (pre) {
      add_task(0, 0, r, r + 1, col.row(r), col.row(r + 1), r);
      add_task(0, 1, r, r + 1, col.row(r + 1), nullptr, r);
    }
    for (int c = r + 1; c < nbg; ++c) {
      if (!up.get(r, c) || (pre && c == r + 1)) continue;
      add_task(0, 2, r, c, col.row(r), col.row(c), r);
    }
    // border columns (dense) and the block column of the right-hand side; padding-only block columns are skipped
    for (int c = nbg; c < pl.ntc; ++c) {
      if (c >= pl.nbf && c != pl.ntc - 1) continue;
      // the two column blocks of a 128-column tile (the border starts at a multiple of 128) in one task, unless asked otherwise
      const bool pair = !single_tile_tasks && ((c - nbg) & 1) == 0 && c + 1 < pl.nbf;
      add_task(1, pair ? 4 : 2, r, c, col.row(r), nullptr, r);
      if (pair) ++c;
    }
  }
  pl.n_tasks0 = (int)list[0].size();
  pl.tasks = list[0];
  pl.tasks.insert(pl.tasks.end(), list[1].begin(), list[1].end());

  // ---- structure of every factored block row (back substitution; forming kernel) ----
  pl.mask_words = (pl.nbf + 63) / 64;
  pl.rowmask.assign((size_t)pl.nbf * pl.mask_words, 0ull);
  auto rm_set = [&](int r, int c) { pl.rowmask[(size_t)r * pl.mask_words + (c >> 6)] |= 1ull << (c & 63); };
  for (int r = 0; r < pl.nbf; ++r)
    for (int c = r + 1; c < pl.nbf; ++c)
      if (r >= nbg || c >= nbg || up.get(r, c)) rm_set(r, c);
  for (int r = 0; r < nbg; ++r)
    for (int c = r; c < nbg; ++c)
      if (up.get(r, c)) { pl.grid_tiles.push_back(r); pl.grid_tiles.push_back(c); }
  pl.grid_words = (nbg + 63) / 64;
  pl.gridrow.assign((size_t)nbg * pl.grid_words, 0ull);
  for (int r = 0; r < nbg; ++r)
    for (int c = r + 1; c < nbg; ++c)
      if (up.get(r, c)) pl.gridrow[(size_t)r * pl.grid_words + (c >> 6)] |= 1ull << (c & 63);
  {
    const double A = (double)(pl.n_pad - pl.Gf);
    pl.flops_update = A * A * (double)pl.Gf;             // upper triangle, 2 flops per multiply-add
    const double Ab = (double)(pl.n_fact - pl.Gf);
    pl.flops_border = Ab * Ab * Ab / 3.0;
  }
  return CBA_OK;
}

}  // namespace cba
