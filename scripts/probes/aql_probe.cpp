// What does a kernel launch cost on the blocking call's critical path, through the HIP runtime and
// through AQL packets this process writes itself into an HSA queue of its own?
//   Each iteration = what a blocking sweep does: a 256-workgroup kernel, a one-workgroup kernel behind
//   it that publishes into mapped host memory, the host polling the sequence word.
//   hip   hipModuleLaunchKernel x 2 on one stream
//   aql   two dispatch packets (barrier bit, no completion signal) and one doorbell write; kernel
//         arguments in (a) the CPU agent's kernarg pool, (b) device memory the host can write
// Build: hipcc --genco --offload-arch=gfx950 scripts/probes/aql_probe_kernels.hip -o build/probes/aql_probe.hsaco
//        hipcc -O3 scripts/probes/aql_probe.cpp -o build/probes/aql_probe -lhsa-runtime64
// Run:   build/probes/aql_probe build/probes/aql_probe.hsaco
#include <hip/hip_runtime.h>
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>
#include <immintrin.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <fcntl.h>
#include <unistd.h>
#include <vector>

#define HIPCHECK(e)                                                                       \
  do {                                                                                    \
    hipError_t r_ = (e);                                                                  \
    if (r_ != hipSuccess) { std::printf("%s: %s\n", #e, hipGetErrorString(r_)); return 1; } \
  } while (0)
#define HSACHECK(e)                                                                       \
  do {                                                                                    \
    hsa_status_t r_ = (e);                                                                \
    if (r_ != HSA_STATUS_SUCCESS) {                                                       \
      const char *m_ = nullptr; hsa_status_string(r_, &m_);                               \
      std::printf("%s: %s\n", #e, m_ ? m_ : "?"); return 1; }                             \
  } while (0)

struct Found { hsa_agent_t gpu{}, cpu{}; bool have_gpu = false, have_cpu = false; int want = 0, seen = 0; };
static hsa_status_t onAgent(hsa_agent_t a, void *data) {
  Found *f = static_cast<Found *>(data);
  hsa_device_type_t t;
  hsa_agent_get_info(a, HSA_AGENT_INFO_DEVICE, &t);
  if (t == HSA_DEVICE_TYPE_GPU) { if (f->seen++ == f->want) { f->gpu = a; f->have_gpu = true; } }
  else if (t == HSA_DEVICE_TYPE_CPU && !f->have_cpu) { f->cpu = a; f->have_cpu = true; }
  return HSA_STATUS_SUCCESS;
}
struct Pools { hsa_amd_memory_pool_t kernarg{}, local{}; bool have_kernarg = false, have_local = false; };
static hsa_status_t onCpuPool(hsa_amd_memory_pool_t p, void *data) {
  Pools *ps = static_cast<Pools *>(data);
  hsa_amd_segment_t seg; uint32_t flags = 0;
  hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_SEGMENT, &seg);
  if (seg != HSA_AMD_SEGMENT_GLOBAL) return HSA_STATUS_SUCCESS;
  hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_GLOBAL_FLAGS, &flags);
  if ((flags & HSA_AMD_MEMORY_POOL_GLOBAL_FLAG_KERNARG_INIT) && !ps->have_kernarg) { ps->kernarg = p; ps->have_kernarg = true; }
  return HSA_STATUS_SUCCESS;
}
static hsa_status_t onGpuPool(hsa_amd_memory_pool_t p, void *data) {
  Pools *ps = static_cast<Pools *>(data);
  hsa_amd_segment_t seg; uint32_t flags = 0; bool alloc = false;
  hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_SEGMENT, &seg);
  if (seg != HSA_AMD_SEGMENT_GLOBAL) return HSA_STATUS_SUCCESS;
  hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_GLOBAL_FLAGS, &flags);
  hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_RUNTIME_ALLOC_ALLOWED, &alloc);
  if (alloc && (flags & HSA_AMD_MEMORY_POOL_GLOBAL_FLAG_COARSE_GRAINED) && !ps->have_local) { ps->local = p; ps->have_local = true; }
  return HSA_STATUS_SUCCESS;
}

struct Kernel { uint64_t object = 0; uint32_t kernarg = 0, group = 0, priv = 0; };
static int lookup(hsa_executable_t exe, hsa_agent_t agent, const char *name, Kernel *k) {
  hsa_executable_symbol_t sym;
  HSACHECK(hsa_executable_get_symbol_by_name(exe, name, &agent, &sym));
  HSACHECK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_OBJECT, &k->object));
  HSACHECK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_KERNARG_SEGMENT_SIZE, &k->kernarg));
  HSACHECK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_GROUP_SEGMENT_SIZE, &k->group));
  HSACHECK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_PRIVATE_SEGMENT_SIZE, &k->priv));
  return 0;
}

static int g_acquire[2] = {HSA_FENCE_SCOPE_SYSTEM, HSA_FENCE_SCOPE_SYSTEM};
static int g_release[2] = {HSA_FENCE_SCOPE_AGENT, HSA_FENCE_SCOPE_AGENT};
static void dispatch(hsa_queue_t *q, uint64_t index, const Kernel &k, uint32_t grid_threads, void *kernarg, int which = 0) {
  auto *base = static_cast<hsa_kernel_dispatch_packet_t *>(q->base_address);
  hsa_kernel_dispatch_packet_t *p = base + (index & (q->size - 1));
  p->workgroup_size_x = 256; p->workgroup_size_y = 1; p->workgroup_size_z = 1; p->reserved0 = 0;
  p->grid_size_x = grid_threads; p->grid_size_y = 1; p->grid_size_z = 1;
  p->private_segment_size = k.priv; p->group_segment_size = k.group;
  p->kernel_object = k.object; p->kernarg_address = kernarg; p->reserved2 = 0;
  p->completion_signal.handle = 0;
  const uint16_t header = (HSA_PACKET_TYPE_KERNEL_DISPATCH << HSA_PACKET_HEADER_TYPE) |
                          (1 << HSA_PACKET_HEADER_BARRIER) |
                          (g_acquire[which] << HSA_PACKET_HEADER_SCACQUIRE_FENCE_SCOPE) |
                          (g_release[which] << HSA_PACKET_HEADER_SCRELEASE_FENCE_SCOPE);
  const uint16_t setup = 1 << HSA_KERNEL_DISPATCH_PACKET_SETUP_DIMENSIONS;
  __atomic_store_n(reinterpret_cast<uint32_t *>(p), uint32_t(header) | (uint32_t(setup) << 16), __ATOMIC_RELEASE);
}

int main(int argc, char **argv) {
  if (argc < 2) { std::printf("usage: aql_probe <code object>\n"); return 1; }
  const int iters = argc > 2 ? std::atoi(argv[2]) : 2000;
  HIPCHECK(hipSetDevice(0));
  hipStream_t stream; HIPCHECK(hipStreamCreate(&stream));
  hipModule_t mod; HIPCHECK(hipModuleLoad(&mod, argv[1]));
  hipFunction_t fwriter, fpublish;
  HIPCHECK(hipModuleGetFunction(&fwriter, mod, "probeWriter"));
  HIPCHECK(hipModuleGetFunction(&fpublish, mod, "probePublish"));
  double *rows; HIPCHECK(hipMalloc(&rows, 256 * 23 * sizeof(double)));
  double *host_block; HIPCHECK(hipHostMalloc(&host_block, 4096, hipHostMallocMapped));
  std::memset(host_block, 0, 4096);
  double *host_values_dev; HIPCHECK(hipHostGetDevicePointer(reinterpret_cast<void **>(&host_values_dev), host_block, 0));
  auto *host_flag = reinterpret_cast<volatile unsigned long long *>(host_block + 8);
  unsigned long long *host_flag_dev = reinterpret_cast<unsigned long long *>(host_values_dev + 8);
  unsigned long long sequence = 0;
  int num_rows = 256;

  auto run = [&](const char *name, auto step) {
    std::vector<double> us;
    long wrong = 0;
    for (int i = 0; i < iters + 200; ++i) {
      const auto t0 = std::chrono::steady_clock::now();
      ++sequence;
      step(double(i));
      while (*host_flag != sequence) __builtin_ia32_pause();
      const auto t1 = std::chrono::steady_clock::now();
      if (i >= 200) us.push_back(std::chrono::duration<double, std::micro>(t1 - t0).count());
      // every row of this iteration must have reached the publishing kernel: sum = 5888 seed + 815488
      if (host_block[0] != 5888.0 * double(i) + 815488.0) ++wrong;
    }
    std::sort(us.begin(), us.end());
    double mean = 0; for (double v : us) mean += v; mean /= us.size();
    std::printf("%-44s median %6.2f us  mean %6.2f  p10 %6.2f  p90 %6.2f  (stale sums: %ld of %d)\n", name,
                us[us.size() / 2], mean, us[us.size() / 10], us[us.size() * 9 / 10], wrong, iters + 200);
    std::fflush(stdout);
  };

  // ---- through the HIP runtime --------------------------------------------------------------------
  run("hip: hipModuleLaunchKernel x 2", [&](double seed) {
    void *a1[] = {&rows, &seed};
    (void)hipModuleLaunchKernel(fwriter, 256, 1, 1, 256, 1, 1, 0, stream, a1, nullptr);
    void *a2[] = {&rows, &num_rows, &host_values_dev, &host_flag_dev, &sequence};
    (void)hipModuleLaunchKernel(fpublish, 1, 1, 1, 256, 1, 1, 0, stream, a2, nullptr);
  });
  HIPCHECK(hipStreamSynchronize(stream));

  // ---- own HSA queue ------------------------------------------------------------------------------
  HSACHECK(hsa_init());
  Found f; HSACHECK(hsa_iterate_agents(onAgent, &f));
  if (!f.have_gpu || !f.have_cpu) { std::printf("no agents\n"); return 1; }
  Pools pools;
  HSACHECK(hsa_amd_agent_iterate_memory_pools(f.cpu, onCpuPool, &pools));
  HSACHECK(hsa_amd_agent_iterate_memory_pools(f.gpu, onGpuPool, &pools));
  hsa_queue_t *q = nullptr;
  HSACHECK(hsa_queue_create(f.gpu, 1024, HSA_QUEUE_TYPE_SINGLE, nullptr, nullptr, UINT32_MAX, UINT32_MAX, &q));
  const int fd = open(argv[1], O_RDONLY);
  if (fd < 0) { std::printf("cannot open %s\n", argv[1]); return 1; }
  hsa_code_object_reader_t reader; HSACHECK(hsa_code_object_reader_create_from_file(fd, &reader));
  hsa_executable_t exe;
  HSACHECK(hsa_executable_create_alt(HSA_PROFILE_FULL, HSA_DEFAULT_FLOAT_ROUNDING_MODE_DEFAULT, nullptr, &exe));
  HSACHECK(hsa_executable_load_agent_code_object(exe, f.gpu, reader, nullptr, nullptr));
  HSACHECK(hsa_executable_freeze(exe, nullptr));
  Kernel kw, kp, kwt;
  if (lookup(exe, f.gpu, "probeWriter.kd", &kw) || lookup(exe, f.gpu, "probePublish.kd", &kp) ||
      lookup(exe, f.gpu, "probeWriterThrough.kd", &kwt)) return 1;
  std::printf("kernarg sizes %u / %u, group %u / %u, private %u / %u; kernarg pool %d, local pool %d\n",
              kw.kernarg, kp.kernarg, kw.group, kp.group, kw.priv, kp.priv, int(pools.have_kernarg), int(pools.have_local));

  struct WriterArgs { double *rows; double seed; };
  struct PublishArgs { const double *rows; int num_rows; int pad; double *host_values; unsigned long long *host_flag; unsigned long long sequence; };
  // fence scopes of the two packets: {acquire 1st, release 1st, acquire 2nd, release 2nd}; 0 none, 1 agent, 2 system
  const int configs[][4] = {{2, 1, 2, 1}, {1, 1, 1, 1}, {1, 1, 1, 0}, {2, 1, 1, 0}, {0, 1, 1, 0}, {0, 0, 0, 0},
                            {1, 0, 1, 0}, {1, 1, 0, 0}, {1, 0, 0, 0}, {1, 0, 1, 1},
                            {1, 0, 1, 0}, {0, 0, 1, 0}, {1, 0, 0, 0}};  // the last three: write-through writer
  for (int where = 0; where < 2 + 12; ++where) {
    const int cfg = where < 2 ? 0 : where - 1;
    const bool through = cfg >= 10;
    g_acquire[0] = configs[cfg][0]; g_release[0] = configs[cfg][1];
    g_acquire[1] = configs[cfg][2]; g_release[1] = configs[cfg][3];
    char *arena = nullptr;
    const size_t slot = 256, slots = 64;  // a ring of argument blocks: a block is not reused while its kernel may run
    if (where == 0) {
      if (!pools.have_kernarg) continue;
      HSACHECK(hsa_amd_memory_pool_allocate(pools.kernarg, slot * slots * 2, 0, reinterpret_cast<void **>(&arena)));
      HSACHECK(hsa_amd_agents_allow_access(1, &f.gpu, nullptr, arena));
    } else {
      if (!pools.have_local) continue;
      if (hsa_amd_memory_pool_allocate(pools.local, slot * slots * 2, 0, reinterpret_cast<void **>(&arena)) != HSA_STATUS_SUCCESS) {
        std::printf("device-memory kernargs: allocation failed\n"); continue; }
      if (hsa_amd_agents_allow_access(1, &f.cpu, nullptr, arena) != HSA_STATUS_SUCCESS) {
        std::printf("device-memory kernargs: the host cannot map it (no large BAR?)\n"); continue; }
    }
    uint64_t n = 0;
    char label[128];
    std::snprintf(label, sizeof label, "aql: kernargs in %s, fences %d%d/%d%d%s", where == 0 ? "host pool" : "device memory",
                  g_acquire[0], g_release[0], g_acquire[1], g_release[1], through ? " WT rows" : "");
    run(label,
        [&](double seed) {
          char *block = arena + (n % slots) * slot * 2;
          ++n;
          auto *wa = reinterpret_cast<WriterArgs *>(block);
          auto *pa = reinterpret_cast<PublishArgs *>(block + slot);
          wa->rows = rows; wa->seed = seed;
          pa->rows = rows; pa->num_rows = num_rows; pa->pad = 0; pa->host_values = host_values_dev;
          pa->host_flag = host_flag_dev; pa->sequence = sequence;
          const uint64_t idx = hsa_queue_add_write_index_relaxed(q, 2);
          dispatch(q, idx, through ? kwt : kw, 256 * 256, wa, 0);
          dispatch(q, idx + 1, kp, 256, pa, 1);
          hsa_signal_store_screlease(q->doorbell_signal, idx + 1);
        });
  }
  // ---- one kernel, rows straight to the host, the host adds them -----------------------------------
  {
    Kernel kh;
    if (lookup(exe, f.gpu, "probeWriterToHost.kd", &kh)) return 1;
    double *host_rows = nullptr;
    HIPCHECK(hipHostMalloc(&host_rows, 256 * 24 * sizeof(double), hipHostMallocMapped));
    std::memset(host_rows, 0, 256 * 24 * sizeof(double));
    double *host_rows_dev = nullptr;
    HIPCHECK(hipHostGetDevicePointer(reinterpret_cast<void **>(&host_rows_dev), host_rows, 0));
    char *arena = nullptr;
    const size_t slot = 256, slots = 256;
    if (pools.have_local &&
        hsa_amd_memory_pool_allocate(pools.local, slot * slots, 0, reinterpret_cast<void **>(&arena)) == HSA_STATUS_SUCCESS &&
        hsa_amd_agents_allow_access(1, &f.cpu, nullptr, arena) == HSA_STATUS_SUCCESS) {
      struct ToHostArgs { double *rows; double seed; unsigned long long sequence; };
      for (int scope = 1; scope >= 0; --scope) {
        g_acquire[0] = 1; g_release[0] = scope;
        std::vector<double> us;
        long wrong = 0;
        uint64_t n = 0;
        for (int i = 0; i < iters + 200; ++i) {
          const auto t0 = std::chrono::steady_clock::now();
          ++sequence;
          auto *a = reinterpret_cast<ToHostArgs *>(arena + (n++ % slots) * slot);
          a->rows = host_rows_dev; a->seed = double(i); a->sequence = sequence;
          _mm_sfence();
          const uint64_t idx = hsa_queue_add_write_index_relaxed(q, 1);
          dispatch(q, idx, kh, 256 * 256, a, 0);
          hsa_signal_store_screlease(q->doorbell_signal, idx);
          // the host's finalize: wait for every row's tag, add the rows in row order
          double sum = 0.0;
          for (int r = 0; r < 256; ++r) {
            volatile unsigned long long *tag = reinterpret_cast<volatile unsigned long long *>(host_rows + r * 24 + 23);
            while (*tag != sequence) __builtin_ia32_pause();
            const double *row = host_rows + r * 24;
            for (int k = 0; k < 23; ++k) sum += row[k];
          }
          const auto t1 = std::chrono::steady_clock::now();
          if (i >= 200) us.push_back(std::chrono::duration<double, std::micro>(t1 - t0).count());
          if (sum != 5888.0 * double(i) + 815488.0) ++wrong;
        }
        std::sort(us.begin(), us.end());
        std::printf("aql: ONE kernel, rows to host, host adds (release %d)  median %6.2f us  p10 %6.2f  p90 %6.2f  (wrong sums: %ld of %d)\n",
                    scope, us[us.size() / 2], us[us.size() / 10], us[us.size() * 9 / 10], wrong, iters + 200);
        std::fflush(stdout);
      }
    }
  }
  // ---- kernels queued back to back (the device-resident loop's situation): 64 pairs per wait -------
  {
    const int burst = 64, rounds = 40;
    auto timeBurst = [&](const char *name, auto enqueue) {
      std::vector<double> us;
      for (int r = 0; r < rounds; ++r) {
        const auto t0 = std::chrono::steady_clock::now();
        for (int b = 0; b < burst; ++b) { ++sequence; enqueue(double(b)); }
        while (*host_flag != sequence) __builtin_ia32_pause();
        us.push_back(std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / burst);
      }
      std::sort(us.begin(), us.end());
      std::printf("%-44s %6.2f us per pair, 64 pairs queued back to back (median of %d bursts)\n", name, us[us.size() / 2], rounds);
      std::fflush(stdout);
    };
    timeBurst("hip, back to back", [&](double seed) {
      void *a1[] = {&rows, &seed};
      (void)hipModuleLaunchKernel(fwriter, 256, 1, 1, 256, 1, 1, 0, stream, a1, nullptr);
      void *a2[] = {&rows, &num_rows, &host_values_dev, &host_flag_dev, &sequence};
      (void)hipModuleLaunchKernel(fpublish, 1, 1, 1, 256, 1, 1, 0, stream, a2, nullptr);
    });
    HIPCHECK(hipStreamSynchronize(stream));
    char *arena = nullptr;
    const size_t slot = 256, slots = 256;
    if (pools.have_local &&
        hsa_amd_memory_pool_allocate(pools.local, slot * slots * 2, 0, reinterpret_cast<void **>(&arena)) == HSA_STATUS_SUCCESS &&
        hsa_amd_agents_allow_access(1, &f.cpu, nullptr, arena) == HSA_STATUS_SUCCESS) {
      for (int scope = 2; scope >= 1; --scope) {
        g_acquire[0] = g_acquire[1] = scope; g_release[0] = g_release[1] = scope;
        uint64_t n = 0;
        char label[96];
        std::snprintf(label, sizeof label, "aql, back to back, fences %d%d/%d%d", scope, scope, scope, scope);
        timeBurst(label, [&](double seed) {
          char *block = arena + (n % slots) * slot * 2;
          ++n;
          auto *wa = reinterpret_cast<WriterArgs *>(block);
          auto *pa = reinterpret_cast<PublishArgs *>(block + slot);
          wa->rows = rows; wa->seed = seed;
          pa->rows = rows; pa->num_rows = num_rows; pa->pad = 0; pa->host_values = host_values_dev;
          pa->host_flag = host_flag_dev; pa->sequence = sequence;
          const uint64_t idx = hsa_queue_add_write_index_relaxed(q, 2);
          dispatch(q, idx, kw, 256 * 256, wa, 0);
          dispatch(q, idx + 1, kp, 256, pa, 1);
          hsa_signal_store_screlease(q->doorbell_signal, idx + 1);
        });
      }
    }
  }
  // the same hip loop once more (ordering effects)
  run("hip again", [&](double seed) {
    void *a1[] = {&rows, &seed};
    (void)hipModuleLaunchKernel(fwriter, 256, 1, 1, 256, 1, 1, 0, stream, a1, nullptr);
    void *a2[] = {&rows, &num_rows, &host_values_dev, &host_flag_dev, &sequence};
    (void)hipModuleLaunchKernel(fpublish, 1, 1, 1, 256, 1, 1, 0, stream, a2, nullptr);
  });
  HIPCHECK(hipStreamSynchronize(stream));
  hsa_queue_destroy(q);
  return 0;
}
