// Where do the microseconds of a config-5 reprojection sweep go?  A diagnostic build of the library's
// own kernel source with in-kernel stamps of the 100 MHz wall clock (MOPT_STAMP, sweep_device.hpp):
// the sweep of one cost (40 k or 60 k elements: 157 / 235 tiles of 256, one per workgroup) between two marker
// kernels on one stream, stamps read back after the last of many launches.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -Iinclude -Imoptimizer_0_amd/csrc \
//         -mllvm -amdgpu-kernarg-preload-count=4 scripts/probes/reproj_stamps.cpp -o /tmp/reproj_stamps
//   /tmp/reproj_stamps [elements]
// Stamp k of a workgroup (wave 0):  0 entry | 1 tile + constants arrived | 2 residual at x |
// 3 six perturbed residuals, twelve quotients | 4 accumulated | 5 (= 4 for one tile) | 6 row stored
#define MOPT_STAMP_WORDS 8
#include "../../moptimizer_0_amd/csrc/sweep_kernels.hip"
#include "../../moptimizer_0_amd/csrc/fd_kernels.hip"  // (only to resolve what the first refers to)

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(e)                                                  \
  do {                                                            \
    hipError_t r_ = (e);                                          \
    if (r_ != hipSuccess) {                                       \
      std::printf("%s: %s\n", #e, hipGetErrorString(r_));         \
      return 1;                                                   \
    }                                                             \
  } while (0)

__global__ void markerKernel(unsigned long long *out) { *out = __builtin_amdgcn_s_memrealtime(); }

// the pieces of the sweep's skeleton alone, for their dispatch time (events): an empty workgroup,
// the workgroup reduction of NACC values, the tile's loads with the LDS copy of the constants
template <int THREADS>
__global__ __launch_bounds__(THREADS) void emptyKernel(double *out) {
  if (out == nullptr) out[0] = 1.0;
}
template <int NACC, int THREADS>
__global__ __launch_bounds__(THREADS) void reduceOnlyKernel(double *partials, double seed) {
  double acc[NACC];
#pragma unroll
  for (int k = 0; k < NACC; ++k) acc[k] = seed * double(threadIdx.x + k);
  mopt::blockReduceStore<NACC, THREADS>(acc, partials + size_t(blockIdx.x) * NACC);
}
__global__ __launch_bounds__(256) void loadOnlyKernel(const mopt::ReprojSweepArgs A) {
  using namespace mopt;
  __shared__ double Mlds[6][12];
  const unsigned char *tb = A.tiles + size_t(blockIdx.x) * kReprojTileBytes;
  const double *planes = reinterpret_cast<const double *>(tb) + threadIdx.x;
  double P[4];
#pragma unroll
  for (int pl = 0; pl < 4; ++pl) P[pl] = planes[pl * kReprojTilePoints];
  const int2 px = reinterpret_cast<const int2 *>(tb + size_t(4) * kReprojTilePoints * 8)[threadIdx.x];
  if (threadIdx.x < 72) (&Mlds[0][0])[threadIdx.x] = (&A.M[1][0])[threadIdx.x];
  __syncthreads();
  double v = Mlds[threadIdx.x % 6][threadIdx.x % 12] + double(px.x + px.y);
#pragma unroll
  for (int pl = 0; pl < 4; ++pl) v += P[pl];
  if (v == 12345.678) A.partials[blockIdx.x] = v;
}

template <typename Launch>
double dispatchMicroseconds(Launch &&launch, hipEvent_t e0, hipEvent_t e1) {
  double sum = 0;
  const int reps = 200;
  for (int it = 0; it < reps; ++it) {
    launch(e0, e1);
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    if (it >= reps / 2) sum += ms;
  }
  return sum / (reps - reps / 2) * 1e3;
}

int main(int argc, char **argv) {
  using namespace mopt;
  const long long count = argc > 1 ? atoll(argv[1]) : 40000;
  const int tiles = int((count + kReprojTilePoints - 1) / kReprojTilePoints);
  std::vector<double> pts(size_t(count) * 4);
  std::vector<int> pix(size_t(count) * 2);
  unsigned long long s = 88172645463325252ull;
  auto uniform = [&] {
    s ^= s << 13; s ^= s >> 7; s ^= s << 17;
    return double(s >> 11) / 9007199254740992.0;
  };
  ReprojSweepArgs args{};
  const double K[12] = {500, 0, 320, 0, 0, 500, 240, 0, 0, 0, 1, 0};
  for (int j = 0; j < 7; ++j)
    for (int k = 0; k < 12; ++k) args.M[j][k] = K[k] * (1.0 + 1e-8 * j * (k + 1));
  for (int j = 0; j < 6; ++j) args.inv_h[j] = 1.0 / 1.4901161193847656e-8;
  args.cov[0] = args.cov[3] = 1.0;
  args.loss_kind = kLossGemanMcClure;
  args.loss_param = 100.0;
  for (long long i = 0; i < count; ++i) {
    const double X = uniform() * 2 - 1, Y = uniform() * 2 - 1, Z = 2 + 3 * uniform();
    pts[4 * i + 0] = X; pts[4 * i + 1] = Y; pts[4 * i + 2] = Z; pts[4 * i + 3] = 1.0;
    pix[2 * i + 0] = int(500 * X / Z + 320 + uniform());
    pix[2 * i + 1] = int(500 * Y / Z + 240 + uniform());
  }
  double *d_pts; int *d_pix; unsigned char *d_tiles; double *d_partials;
  unsigned long long *d_stamps, *d_marks;
  CHECK(hipMalloc(&d_pts, pts.size() * 8));
  CHECK(hipMalloc(&d_pix, pix.size() * 4));
  CHECK(hipMalloc(&d_tiles, size_t(tiles) * kReprojTileBytes));
  CHECK(hipMalloc(&d_partials, size_t(tiles) * kAccFull * 8));
  CHECK(hipMalloc(&d_stamps, size_t(tiles) * MOPT_STAMP_WORDS * 8));
  CHECK(hipMalloc(&d_marks, 2 * 8));
  CHECK(hipMemcpy(d_pts, pts.data(), pts.size() * 8, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(d_pix, pix.data(), pix.size() * 4, hipMemcpyHostToDevice));
  CHECK(hipMemset(d_stamps, 0, size_t(tiles) * MOPT_STAMP_WORDS * 8));
  CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_stamp_buffer), &d_stamps, sizeof d_stamps));
  CHECK(launchRelayoutReproj(d_pts, d_pix, count, d_tiles, tiles, nullptr));
  args.tiles = d_tiles;
  args.count = count;
  args.num_tiles = tiles;
  args.partials = d_partials;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  float ms_sum = 0;
  const int reps = 200;
  for (int it = 0; it < reps; ++it) {
    hipLaunchKernelGGL(markerKernel, dim3(1), dim3(1), 0, nullptr, d_marks);
    hipExtLaunchKernelGGL((reprojKernel<kCovIdentity, false>), dim3(tiles), dim3(256), 0, nullptr, e0, e1, 0, args);
    hipLaunchKernelGGL(markerKernel, dim3(1), dim3(1), 0, nullptr, d_marks + 1);
    CHECK(hipDeviceSynchronize());
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (it >= reps / 2) ms_sum += ms;
  }
  std::vector<unsigned long long> st(size_t(tiles) * MOPT_STAMP_WORDS), marks(2);
  CHECK(hipMemcpy(st.data(), d_stamps, st.size() * 8, hipMemcpyDeviceToHost));
  CHECK(hipMemcpy(marks.data(), d_marks, 16, hipMemcpyDeviceToHost));
  {
    const dim3 g(tiles);
    std::printf("dispatch time alone [us]: empty<256> %.2f  empty<512> %.2f  tile loads + constants to LDS %.2f\n",
                dispatchMicroseconds([&](hipEvent_t a, hipEvent_t b) { hipExtLaunchKernelGGL(emptyKernel<256>, g, dim3(256), 0, nullptr, a, b, 0, d_partials); }, e0, e1),
                dispatchMicroseconds([&](hipEvent_t a, hipEvent_t b) { hipExtLaunchKernelGGL(emptyKernel<512>, g, dim3(512), 0, nullptr, a, b, 0, d_partials); }, e0, e1),
                dispatchMicroseconds([&](hipEvent_t a, hipEvent_t b) { hipExtLaunchKernelGGL(loadOnlyKernel, g, dim3(256), 0, nullptr, a, b, 0, args); }, e0, e1));
    std::printf("  workgroup reduction alone: 1 value %.2f  23 values %.2f  28 values %.2f  43 values %.2f  28 values, 512 threads %.2f\n",
                dispatchMicroseconds([&](hipEvent_t a, hipEvent_t b) { hipExtLaunchKernelGGL((reduceOnlyKernel<1, 256>), g, dim3(256), 0, nullptr, a, b, 0, d_partials, 1.5); }, e0, e1),
                dispatchMicroseconds([&](hipEvent_t a, hipEvent_t b) { hipExtLaunchKernelGGL((reduceOnlyKernel<23, 256>), g, dim3(256), 0, nullptr, a, b, 0, d_partials, 1.5); }, e0, e1),
                dispatchMicroseconds([&](hipEvent_t a, hipEvent_t b) { hipExtLaunchKernelGGL((reduceOnlyKernel<28, 256>), g, dim3(256), 0, nullptr, a, b, 0, d_partials, 1.5); }, e0, e1),
                dispatchMicroseconds([&](hipEvent_t a, hipEvent_t b) { hipExtLaunchKernelGGL((reduceOnlyKernel<43, 256>), g, dim3(256), 0, nullptr, a, b, 0, d_partials, 1.5); }, e0, e1),
                dispatchMicroseconds([&](hipEvent_t a, hipEvent_t b) { hipExtLaunchKernelGGL((reduceOnlyKernel<28, 512>), g, dim3(512), 0, nullptr, a, b, 0, d_partials, 1.5); }, e0, e1));
    std::printf("  cost-only sweep %.2f\n",
                dispatchMicroseconds([&](hipEvent_t a, hipEvent_t b) { hipExtLaunchKernelGGL((reprojKernel<kCovIdentity, true>), g, dim3(256), 0, nullptr, a, b, 0, args); }, e0, e1));
  }
  std::printf("%lld elements, %d tiles: dispatch (events) %.2f us; marker before -> marker after %.2f us\n",
              count, tiles, ms_sum / (reps - reps / 2) * 1e3, (marks[1] - marks[0]) * 0.01);
  std::printf("stamp  min / median / max  [us after the marker before]      median step [us]\n");
  std::vector<double> prev_med;
  double last_med = 0;
  for (int k = 0; k < 7; ++k) {
    std::vector<double> v;
    for (int b = 0; b < tiles; ++b) v.push_back((double(st[size_t(b) * MOPT_STAMP_WORDS + k]) - double(marks[0])) * 0.01);
    std::sort(v.begin(), v.end());
    // per-workgroup step: stamp k - stamp k-1
    std::vector<double> d;
    if (k > 0)
      for (int b = 0; b < tiles; ++b)
        d.push_back((double(st[size_t(b) * MOPT_STAMP_WORDS + k]) - double(st[size_t(b) * MOPT_STAMP_WORDS + k - 1])) * 0.01);
    std::sort(d.begin(), d.end());
    std::printf("  %d    %6.2f / %6.2f / %6.2f                                   %6.2f\n", k, v.front(),
                v[v.size() / 2], v.back(), k ? d[d.size() / 2] : 0.0);
    last_med = v[v.size() / 2];
  }
  (void)last_med;
  return 0;
}
