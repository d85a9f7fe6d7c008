// Correspondence search entry points (mopt_icp_*): the step upstream of the sweep, i.e. what a
// registration model's update(x) does before every linearization (model.h:24-26,
// levenberg_marquadt_dyn.cpp:54).  Grid construction and search run on the GPU (icp_grid.hip,
// icpMatchKernel in sweep_kernels.hip); this file is their host side.
#include "cost_state.hpp"

#include <cmath>
#include <cstring>
#include <limits>

#include "moptimizer_amd/so3.hpp"

using namespace mopt_detail;

namespace {
// Grid over the targets and cell-ordered copy of the sources, built on the device (icp_grid.hip);
// the host only chooses the resolution.  On return `d_src_sorted` holds the n sources in cell
// order (packed xyz) and matcher->d_order (device memory) the original index of each.
template <typename S>
int buildIcpGrid(const S *src, long long n, const S *tgt, long long m, double max_distance,
                 unsigned flags, hipStream_t s, std::unique_ptr<IcpMatcher> &out_matcher,
                 DeviceScratch &d_src_sorted) {
  auto mt = std::make_unique<IcpMatcher>();
  mt->max_dist = max_distance;
  mt->num_targets = m;
  DeviceScratch d_tgt(s), d_src(s), d_perm_t(s), d_perm_s(s);
  MOPT_HIP_TRY(d_tgt.alloc(size_t(m) * 3 * sizeof(S)));
  MOPT_HIP_TRY(d_src.alloc(size_t(n) * 3 * sizeof(S)));
  // (clouds already in device memory are copied too: 24 B a point at HBM speed, and the caller's
  // arrays are free again when this returns, as with host arrays)
  const hipMemcpyKind kind = (flags & MOPT_INPUT_DEVICE) ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
  if (m > 0) MOPT_HIP_TRY(hipMemcpyAsync(d_tgt.p, tgt, size_t(m) * 3 * sizeof(S), kind, s));
  if (n > 0) MOPT_HIP_TRY(hipMemcpyAsync(d_src.p, src, size_t(n) * 3 * sizeof(S), kind, s));
  double lo[3], hi[3];
  MOPT_HIP_TRY(mopt::icpBoundingBox<S>(d_tgt.as<S>(), m, lo, hi, s));
  // Cell edge: a hair above the search radius over `reach`, so that the (2 reach + 1)^3 cells around
  // a query hold every target within the radius.  reach = 1 while that leaves about one target per
  // cell; where the radius spans many targets the cells are made finer (up to 8 to the radius, and
  // never more than ~64 M cells: a 256 MB offset table): the search's first round looks at 2 x 2 x 2 cells whatever the
  // radius, and what it costs goes with the targets in them.  Enlarged instead when even cells of
  // the radius's size would be more than that.
  // 2^26 cells (a 256 MB offset table at most) since round 5: a scanned surface fills a thin sheet of
  // its bounding box, and with 2^24 a million-point scan searched within 4 point spacings stayed at one
  // cell to the radius, ~12 targets per occupied cell — whole ICP solves 1 M x 1 M, radius 2 / 4 / 8
  // spacings: 1.29 / 2.16 / 1.90 ms -> 1.21 / 1.30 / 1.46 (profiles/r5_icp_cells_cap.txt; 2^28 gains
  // nothing more: finer cells speed the first round up and slow the second down)
  static const int cells_log2 = envInt("MOPT_ICP_MAX_CELLS_LOG2", 26);  // (measurements: 20 ... 28)
  const double kMostCells = double(1ll << (cells_log2 < 20 ? 20 : cells_log2 > 28 ? 28 : cells_log2));
  auto cellsAt = [&](double edge) {
    double cells = 1.0;
    for (int a = 0; a < 3; ++a) cells *= std::floor((hi[a] - lo[a]) / edge) + 1.0;
    return cells;
  };
  const double radius_cell = max_distance * 1.001;
  auto mostReach = [&](int wanted) {  // the largest reach <= wanted the cell budget allows
    while (wanted > 1 && cellsAt(radius_cell / wanted) > kMostCells) --wanted;
    return wanted;
  };
  int reach = 1;
  static const int forced_reach = envInt("MOPT_ICP_REACH", 0);  // tests and measurements
  // (a box thinner than a cell along some axis — a wall, a line, coincident points — says nothing
  // about the targets per cell by its volume: there the resolution comes from the occupied cells
  // below alone)
  bool thin_box = false;
  for (int a = 0; a < 3; ++a) thin_box = thin_box || !(hi[a] - lo[a] >= radius_cell);
  if (forced_reach > 0) {
    reach = mostReach(forced_reach > 8 ? 8 : forced_reach);
  } else if (!thin_box) {
    while (reach < 8 && double(m) > 1.5 * cellsAt(radius_cell / reach) && mostReach(reach + 1) > reach)
      ++reach;
  }
  out_matcher = std::move(mt);  // from here on the caller frees the matcher's device arrays
  IcpMatcher &g = *out_matcher;
  MOPT_HIP_TRY(d_perm_t.alloc(size_t(m) * sizeof(int)));
  // The box's volume says little about how the targets fill it (a scanned surface occupies a thin
  // sheet of the cells): after the first binning the cells that hold anything are counted, and while
  // they hold more than three targets each on average the grid is made finer and the targets are
  // binned again (at most twice more; a binning is ~0.25 ms per million).  Finer cells that do not
  // thin the targets out (coincident points; fewer than 1.5 times fewer per occupied cell) only add
  // empty rows to every search: the grid then goes back to the resolution before, and stays.
  int reach_before = 0;
  double per_cell_before = 0.0;
  bool settled = false;
  for (int attempt = 0;; ++attempt) {
    double cell = radius_cell / reach;
    while (cellsAt(cell) > kMostCells) cell *= 1.26;  // (only ever with reach == 1)
    g.reach = reach;
    g.cell = cell;
    long long ncells = 1;
    for (int a = 0; a < 3; ++a) {
      g.origin[a] = lo[a];
      g.dims[a] = int(std::floor((hi[a] - lo[a]) / cell)) + 1;
      ncells *= g.dims[a];
    }
    if (g.d_cell_start) deviceRelease(g.d_cell_start);
    g.d_cell_start = nullptr;
    MOPT_HIP_TRY(deviceAlloc(reinterpret_cast<void **>(&g.d_cell_start), size_t(ncells + 1) * sizeof(int)));
    MOPT_HIP_TRY(mopt::icpSortByCell<S>(d_tgt.as<S>(), m, g.origin, g.cell, g.dims, d_perm_t.as<int>(),
                                        g.d_cell_start, s));
    if (settled || forced_reach > 0 || m == 0 || attempt == 3) break;
    if (reach == 8 && reach_before == 0) break;  // by the volume rule, nothing finer to try
    long long occupied = 0;
    MOPT_HIP_TRY(mopt::icpCountOccupiedCells(g.d_cell_start, ncells, &occupied, s));
    const double per_cell = double(m) / double(occupied > 0 ? occupied : 1);
    if (reach_before > 0 && per_cell * 1.5 > per_cell_before) {
      reach = reach_before;
      settled = true;
      continue;
    }
    if (per_cell <= 3.0 || attempt == 2) break;
    const int wanted = int(std::ceil(reach * std::sqrt(per_cell / 1.5)));
    const int finer = mostReach(std::min(8, std::max(reach + 1, wanted)));
    if (finer <= reach) break;
    reach_before = reach;
    per_cell_before = per_cell;
    reach = finer;
  }
  // (one matched count per wave of the search: whole tiles of sources, 64 to a wave)
  constexpr long long kTile = mopt::TileShape<S>::kPoints;
  if (m > 0) {
    MOPT_HIP_TRY(deviceAlloc(&g.d_sorted, size_t(m) * 4 * sizeof(S)));
    MOPT_HIP_TRY(mopt::icpGatherPoints<S>(d_tgt.as<S>(), d_perm_t.as<int>(), m,
                                          static_cast<S *>(g.d_sorted), true, s));
  }
  // the sources in the cell order of their un-warped position; those with a NaN or infinite
  // coordinate sort behind the others and are left out of the cost (they can have no target, and
  // their arithmetic would turn every sum into NaN): the cost holds the first `kept` of the order
  MOPT_HIP_TRY(d_perm_s.alloc(size_t(n) * sizeof(int)));
  long long parked = 0;
  MOPT_HIP_TRY(mopt::icpSortByCell<S>(d_src.as<S>(), n, g.origin, g.cell, g.dims, d_perm_s.as<int>(),
                                      nullptr, s, &parked));
  const long long kept = n - parked;
  g.num_sources = n;
  g.num_waves = (kept + kTile - 1) / kTile * kTile / 64;
  MOPT_HIP_TRY(deviceAlloc(reinterpret_cast<void **>(&g.d_matched),
                           size_t(g.num_waves > 0 ? g.num_waves : 1) * sizeof(unsigned int)));
  MOPT_HIP_TRY(d_src_sorted.alloc(size_t(kept) * 3 * sizeof(S)));
  MOPT_HIP_TRY(mopt::icpGatherPoints<S>(d_src.as<S>(), d_perm_s.as<int>(), kept, d_src_sorted.as<S>(),
                                        false, s));
  MOPT_HIP_TRY(hipStreamSynchronize(s));
  g.d_order = d_perm_s.as<int>();  // the matcher keeps the order (its first `kept` entries) on the device
  d_perm_s.p = nullptr;
  g.kept = kept;
  return MOPT_OK;
}

}  // namespace

namespace mopt_detail {
// everything of a search that does not depend on the pose
template <typename S>
void fillIcpArgs(const mopt_cost *c, mopt::IcpMatchArgs<S> &a) {
  const IcpMatcher &mt = *c->matcher;
  a.tiles = static_cast<S *>(c->d_tiles);
  a.count = c->count;
  a.num_tiles = c->num_tiles;
  a.sorted = static_cast<const S *>(mt.d_sorted);
  a.cell_start = mt.d_cell_start;
  for (int k = 0; k < 3; ++k) {
    a.origin[k] = S(mt.origin[k]);
    a.dims[k] = mt.dims[k];
  }
  a.inv_cell = S(1.0 / mt.cell);
  a.cell = S(mt.cell);
  a.reach = mt.reach;
  a.max_dist2 = S(mt.max_dist * mt.max_dist);
  for (int k = 0; k < 12; ++k) a.T[k] = S(k % 5 == 0 ? 1 : 0);
  a.matched = nullptr;
  // The inner nine rows in lock step, the rings beyond row by row under the running bound: with every
  // ring in lock step (MOPT_ICP_LOCK_RINGS=8) reach 3 took 0.154 / 0.178 / 0.229 ms per 1 M at 0.4 / 0.7 /
  // 1.5 radii apart against 0.139 / 0.163 / 0.214, reach 8 on a surface 1.09 against 0.97, reach 2 the
  // same (profiles/r5_icp_lock_rings.txt)
  static const int lock_rings = envInt("MOPT_ICP_LOCK_RINGS", 1);
  a.lock_rings = lock_rings;
}
template void fillIcpArgs<float>(const mopt_cost *, mopt::IcpMatchArgs<float> &);
template void fillIcpArgs<double>(const mopt_cost *, mopt::IcpMatchArgs<double> &);
}  // namespace mopt_detail

namespace {
template <typename S>
int icpUpdate(mopt_cost *c, const S *x, int64_t *num_matched) {
  const IcpMatcher &mt = *c->matcher;
  mopt::IcpMatchArgs<S> a;
  fillIcpArgs<S>(c, a);
  const auto T = moptimizer::so3::rigidFrom6DOF<S>(x);
  std::memcpy(a.T, T.m, sizeof T.m);
  a.matched = num_matched ? mt.d_matched : nullptr;
  MOPT_HIP_TRY(mopt::launchIcpMatch<S>(a, c->stream));
  c->cache.valid = false;
  c->state_version += 1;
  if (num_matched) {
    // the count comes back like a sweep result: a one-wave kernel stores it into mapped host
    // memory and releases a sequence word — no memset, no copy, no stream synchronisation
    const int slot = mopt_detail::kResultSlots - 1;  // beyond any n*n + n + 1 <= 73
    const mopt::HostPublish pub = nextHostPublish(c, slot);
    MOPT_HIP_TRY(mopt::launchPublishCounter(mt.d_matched, mt.num_waves, pub, c->stream));
    const int rc = waitHostPublished(c, pub.sequence);
    if (rc != MOPT_OK) return rc;
    *num_matched = int64_t(c->h_result[slot]);
  }
  return MOPT_OK;
}
}  // namespace

extern "C" {

int mopt_icp_create(mopt_cost **out, int device, int scalar_bytes, const void *src_xyz,
                    int64_t num_src, const void *tgt_xyz, int64_t num_tgt, double max_distance) {
  return mopt_icp_create_from(out, device, scalar_bytes, src_xyz, num_src, tgt_xyz, num_tgt,
                              max_distance, MOPT_INPUT_HOST);
}

int mopt_icp_create_from(mopt_cost **out, int device, int scalar_bytes, const void *src_xyz,
                         int64_t num_src, const void *tgt_xyz, int64_t num_tgt, double max_distance,
                         unsigned flags) {
  if (!out) return fail(MOPT_ERR_INVALID_ARGUMENT, "out is NULL");
  *out = nullptr;
  if (num_src < 0 || num_tgt < 0 || (num_src > 0 && !src_xyz) || (num_tgt > 0 && !tgt_xyz) ||
      num_tgt > std::numeric_limits<int>::max() || num_src > std::numeric_limits<int>::max())
    return fail(MOPT_ERR_INVALID_ARGUMENT, "bad clouds");
  if (!(max_distance > 0.0)) return fail(MOPT_ERR_INVALID_ARGUMENT, "max_distance must be > 0");
  // (before any copy is sized by it: a 2-byte scalar type would be read as fp32, past the caller's buffer)
  if (scalar_bytes != 4 && scalar_bytes != 8)
    return fail(MOPT_ERR_INVALID_ARGUMENT, "scalar_bytes must be 4 (float) or 8 (double)");
  if (flags & ~unsigned(MOPT_INPUT_DEVICE))
    return fail(MOPT_ERR_INVALID_ARGUMENT, "unknown flag bits (MOPT_INPUT_HOST or MOPT_INPUT_DEVICE)");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(MOPT_ERR_NO_DEVICE, "no HIP device is visible to this process");
  if (device < 0 || device >= ndev) return fail(MOPT_ERR_INVALID_ARGUMENT, "device index out of range");
  MOPT_HIP_TRY(hipSetDevice(device));
  std::unique_ptr<IcpMatcher> matcher;
  auto freeMatcher = [&]() {
    if (!matcher) return;
    deviceRelease(matcher->d_sorted);
    deviceRelease(matcher->d_cell_start);
    deviceRelease(matcher->d_matched);
    deviceRelease(matcher->d_order);
  };
  hipStream_t build_stream = nullptr;
  MOPT_HIP_TRY(acquireStream(device, &build_stream));
  DeviceScratch d_src_sorted(build_stream);  // (read once more by the cost's own stream, which the create call synchronises)
  int rc = scalar_bytes == 8
               ? buildIcpGrid<double>(static_cast<const double *>(src_xyz), num_src,
                                      static_cast<const double *>(tgt_xyz), num_tgt, max_distance,
                                      flags, build_stream, matcher, d_src_sorted)
               : buildIcpGrid<float>(static_cast<const float *>(src_xyz), num_src,
                                     static_cast<const float *>(tgt_xyz), num_tgt, max_distance,
                                     flags, build_stream, matcher, d_src_sorted);
  (void)hipStreamSynchronize(build_stream);
  releaseStream(device, build_stream);
  // the sources go into the resident tiles in cell order; the target planes are filled by the
  // first search
  mopt_cost *raw = nullptr;
  if (rc == MOPT_OK) {
    g_creating_for_search = true;  // (its update(x) is queued on the stream: never on the direct path)
    rc = mopt_point2point_create(&raw, device, scalar_bytes, d_src_sorted.p, d_src_sorted.p,
                                 int64_t(matcher->kept), MOPT_INPUT_DEVICE);
    g_creating_for_search = false;
  }
  if (rc != MOPT_OK) {
    freeMatcher();
    return rc;
  }
  std::unique_ptr<mopt_cost, void (*)(mopt_cost *)> c(raw, destroyCost);
  c->matcher = std::move(matcher);
  const double zero8[6] = {0, 0, 0, 0, 0, 0};
  const float zero4[6] = {0, 0, 0, 0, 0, 0};
  rc = mopt_icp_update(c.get(), scalar_bytes == 8 ? static_cast<const void *>(zero8)
                                                   : static_cast<const void *>(zero4), nullptr);
  if (rc != MOPT_OK) return rc;
  *out = c.release();
  return MOPT_OK;
}

int mopt_icp_update(mopt_cost *c, const void *x, int64_t *num_matched) {
  if (!c || !x) return fail(MOPT_ERR_INVALID_ARGUMENT, "NULL argument");
  if (!c->matcher) return fail(MOPT_ERR_INVALID_ARGUMENT, "not a cost made by mopt_icp_create");
  MOPT_HIP_TRY(hipSetDevice(c->device));
  return c->scalar_bytes == 8 ? icpUpdate<double>(c, static_cast<const double *>(x), num_matched)
                              : icpUpdate<float>(c, static_cast<const float *>(x), num_matched);
}

int mopt_icp_grid(const mopt_cost *c, double *cell_edge, int *reach, int dims[3], double origin[3]) {
  if (!c || !c->matcher) return fail(MOPT_ERR_INVALID_ARGUMENT, "not an ICP cost");
  const IcpMatcher &mt = *c->matcher;
  if (cell_edge) *cell_edge = mt.cell;
  if (reach) *reach = mt.reach;
  for (int a = 0; a < 3; ++a) {
    if (dims) dims[a] = mt.dims[a];
    if (origin) origin[a] = mt.origin[a];
  }
  return MOPT_OK;
}

int mopt_icp_get_matches(mopt_cost *c, void *tgt_out_xyz) {
  if (!c || !tgt_out_xyz) return fail(MOPT_ERR_INVALID_ARGUMENT, "NULL argument");
  if (c->model != kModelPoint2Point) return fail(MOPT_ERR_INVALID_ARGUMENT, "not a point2point cost");
  MOPT_HIP_TRY(hipSetDevice(c->device));
  // An ICP cost keeps its sources in cell order and leaves out those with non-finite coordinates:
  // the triples go to the caller's positions on the device (NaN — all bits set — where nothing is
  // written), then one copy.
  const int *order = c->matcher ? c->matcher->d_order : nullptr;
  const long long rows = c->matcher ? c->matcher->num_sources : c->count;
  if (rows == 0) return MOPT_OK;
  const size_t bytes = size_t(rows) * 3 * c->scalar_bytes;
  void *d_tmp = nullptr;
  MOPT_HIP_TRY(deviceAlloc(&d_tmp, bytes));
  hipError_t e = order ? hipMemsetAsync(d_tmp, 0xff, bytes, c->stream) : hipSuccess;
  if (e == hipSuccess)
    e = c->scalar_bytes == 8
            ? mopt::launchGatherTargets<double>(static_cast<const double *>(c->d_tiles), c->count, order,
                                                static_cast<double *>(d_tmp), c->stream)
            : mopt::launchGatherTargets<float>(static_cast<const float *>(c->d_tiles), c->count, order,
                                               static_cast<float *>(d_tmp), c->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(tgt_out_xyz, d_tmp, bytes, hipMemcpyDeviceToHost, c->stream);
  const hipError_t synced = hipStreamSynchronize(c->stream);
  if (e == hipSuccess) e = synced;
  deviceRelease(d_tmp);
  if (e != hipSuccess) return fail(MOPT_ERR_HIP, std::string("gather: ") + hipGetErrorString(e));
  return MOPT_OK;
}

}  // extern "C"
