// stale_line_probe.hip — library-free reproducer for "a second-pass kernel read a few cache lines as an earlier
// launch left them" (tests/test_gpu_zz_multirank.py).  NO code of libbm_gar in here.
//
// Per iteration `it` (one stream, kernel boundaries only — no in-launch hand-off):
//   producer   mode k: kernel A rewrites ROWS x D words in place (non-temporal 16-byte loads of the old pattern, which it
//                      checks, non-temporal 16-byte stores of pattern(it, row, col)), grid-stride like a first pass;
//              mode h: the host writes the pattern into PAGEABLE memory and hipMemcpy's it over the rows (what
//                      `tensor.to(device)` does);
//   ranking    kernel R writes an 18-entry index table ord[t] (a function of `it`);
//   two tiny kernels (the reduce / rank launches of the real chain);
//   consumers  kernel B<V>, one workgroup per 1024 consecutive columns (the mapping of a second pass, different from
//              the producer's), fetches ord[] and then its 18 row pointers from the BY-VALUE table in the kernarg
//              segment at a run-time index (scalar loads), reads 16 bytes per row and lane and compares every word with
//              the pattern it must hold.  V: 0 = non-temporal loads, 1 = plain loads, 2 = `sc1` loads, 3 = an agent-scope
//              acquire fence first, then non-temporal loads, 4 = `sc0 sc1` loads.  The host permutes the table every
//              launch, so a stale kernarg line or a stale ord[] shows as "another row's pattern", a stale data line as
//              "an earlier iteration's pattern".  The order of the variants rotates with `it`.
// Mismatches are classified in the kernel and logged (iteration, variant, rank, row, column, bits, workgroup, XCC id).
//
//   hipcc --offload-arch=gfx950 -O2 -o stale_line_probe stale_line_probe.hip
//   for i in 1 2 3 4 5; do ./stale_line_probe k 3000 & done; wait          (five processes time-sharing one GPU)
//   ./stale_line_probe idle 60 8 &                                         (a process that only HOLDS a context and 8 GB)
#include <hip/hip_runtime.h>
#include <unistd.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CHECK(x)                                                                          \
  do {                                                                                    \
    hipError_t e_ = (x);                                                                  \
    if (e_ != hipSuccess) {                                                               \
      fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
      exit(2);                                                                            \
    }                                                                                     \
  } while (0)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
constexpr int ROWS = 20, RANKED = 18, LOGCAP = 2048, NVAR = 5;
struct Table { const uint32_t* p[64]; };  // 512 bytes by value

__host__ __device__ inline uint32_t pattern(uint32_t it, uint32_t row, uint32_t col) {
  uint32_t x = (it + 1) * 0x9E3779B1u ^ (row + 1) * 0x85EBCA77u ^ (col + 1) * 0xC2B2AE3Du;
  x ^= x >> 15; x *= 0x2C1B3C6Du; x ^= x >> 12; x *= 0x297A2D39u; x ^= x >> 15;
  return x;
}

struct Entry { uint32_t it, variant, t, row, col, got, want, cls, block, xcc, hwid, pad; };
struct Log {
  unsigned int count;
  unsigned int words[NVAR + 1];  // mismatching words per consumer variant; [NVAR] = the producer's own read-check
  Entry e[LOGCAP];
};

__device__ inline uint32_t classify(uint32_t got, uint32_t it, uint32_t row, uint32_t col) {
  for (uint32_t k = 1; k <= 6 && k <= it; ++k)
    if (got == pattern(it - k, row, col)) return k;                   // this row, k iterations ago
  for (uint32_t r = 0; r < ROWS; ++r)
    if (got == pattern(it, r, col)) return 100 + r;                   // another row, this iteration
  for (uint32_t r = 0; r < ROWS; ++r)
    for (uint32_t k = 1; k <= 3 && k <= it; ++k)
      if (got == pattern(it - k, r, col)) return 1000 + 100 * k + r;  // another row, k iterations ago
  return 0;
}

__device__ inline void report(Log* log, uint32_t it, uint32_t variant, uint32_t t, uint32_t row, uint32_t col,
                              uint32_t got, uint32_t want) {
  atomicAdd(&log->words[variant], 1u);
  const unsigned int slot = atomicAdd(&log->count, 1u);
  if (slot >= LOGCAP) return;
  Entry& e = log->e[slot];
  e.it = it; e.variant = variant; e.t = t; e.row = row; e.col = col; e.got = got; e.want = want;
  e.cls = classify(got, it, row, col);
  e.block = blockIdx.x;
  e.xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11));    // HW_REG_XCC_ID, 4 bits
  e.hwid = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));   // HW_REG_HW_ID
}

// ---- producer (mode k) ----
__global__ __launch_bounds__(256) void produce_kernel(Table rows, uint32_t it, uint32_t nvec, uint32_t d, Log* log) {
  for (uint32_t v = blockIdx.x * 256 + threadIdx.x; v < nvec; v += gridDim.x * 256) {
#pragma unroll 4
    for (uint32_t r = 0; r < ROWS; ++r) {
      u32x4* p = reinterpret_cast<u32x4*>(const_cast<uint32_t*>(rows.p[r])) + v;
      const u32x4 old = __builtin_nontemporal_load(p);
      u32x4 nw;
      for (int c = 0; c < 4; ++c) {
        const uint32_t col = v * 4 + c;
        if (it > 0 && old[c] != pattern(it - 1, r, col)) report(log, it, NVAR, 0, r, col, old[c], pattern(it - 1, r, col));
        nw[c] = pattern(it, r, col);
      }
      __builtin_nontemporal_store(nw, p);
    }
  }
  if (blockIdx.x == gridDim.x - 1 && threadIdx.x < d - nvec * 4)
    for (uint32_t r = 0; r < ROWS; ++r) {
      const uint32_t col = nvec * 4 + threadIdx.x;
      const_cast<uint32_t*>(rows.p[r])[col] = pattern(it, r, col);
    }
}

__global__ void rank_kernel(int32_t* ord, uint32_t it) {
  if (threadIdx.x < RANKED) ord[threadIdx.x] = (int32_t)((threadIdx.x + it) % ROWS);
}
__global__ void tiny_kernel(double* scratch, int n) {
  if ((int)threadIdx.x < n) scratch[threadIdx.x] = scratch[threadIdx.x] * 0.5 + 1.0;
}

template <int V>
__device__ inline u32x4 load16(const uint32_t* p) {
  const u32x4* q = reinterpret_cast<const u32x4*>(p);
  if constexpr (V == 0 || V == 3) {
    return __builtin_nontemporal_load(q);
  } else if constexpr (V == 1) {
    return *q;
  } else if constexpr (V == 2) {
    u32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(q) : "memory");
    return v;
  } else {
    u32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(q) : "memory");
    return v;
  }
}

// ---- consumer: table[ord[t]] with the table permuted by 3*it on the host: row of rank t = (t + 4*it) % ROWS ----
template <int V>
__global__ __launch_bounds__(256) void consume_kernel(Table rows, const int32_t* __restrict__ ord, uint32_t it,
                                                      uint32_t nvec, Log* log) {
  if constexpr (V == 3) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  typedef const uint32_t* __attribute__((address_space(4))) const* KargTable;
  const KargTable karg = (KargTable)__builtin_amdgcn_kernarg_segment_ptr();
  const int mine = (int)(threadIdx.x & 63) < RANKED
                       ? __hip_atomic_load(ord + (threadIdx.x & 63), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
  const uint32_t* ranked[RANKED];
#pragma unroll
  for (int t = 0; t < RANKED; ++t) ranked[t] = (const uint32_t*)karg[__builtin_amdgcn_readlane(mine, t)];
  const uint32_t nblk = (nvec + 255) / 256;
  for (uint32_t b = blockIdx.x; b < nblk; b += gridDim.x) {
    const uint32_t v = b * 256 + threadIdx.x;
    if (v >= nvec) continue;
    u32x4 x[RANKED];
#pragma unroll
    for (int t = 0; t < RANKED; ++t) x[t] = load16<V>(ranked[t] + v * 4);
#pragma unroll
    for (int t = 0; t < RANKED; ++t) {
      const uint32_t row = (t + 4 * it) % ROWS;
      for (int c = 0; c < 4; ++c) {
        const uint32_t col = v * 4 + c, want = pattern(it, row, col);
        if (x[t][c] != want) report(log, it, V, t, row, col, x[t][c], want);
      }
    }
  }
}

template <int V>
static void launch_consumer(const Table& tab, const int32_t* ord, uint32_t it, uint32_t nvec, Log* log, hipStream_t s) {
  const uint32_t nblk = (nvec + 255) / 256;
  hipLaunchKernelGGL(consume_kernel<V>, dim3(nblk < 16384 ? nblk : 16384), dim3(256), 0, s, tab, ord, it, nvec, log);
}

int main(int argc, char** argv) {
  const char* mode = argc > 1 ? argv[1] : "k";
  if (!strcmp(mode, "idle")) {  // hold a context and some memory, do nothing
    const int seconds = argc > 2 ? atoi(argv[2]) : 30;
    const size_t gb = argc > 3 ? atoi(argv[3]) : 4;
    std::vector<void*> held;
    for (size_t i = 0; i < gb; ++i) {
      void* p;
      CHECK(hipMalloc(&p, (size_t)1 << 30));
      CHECK(hipMemset(p, 0x5a, (size_t)1 << 30));
      held.push_back(p);
    }
    hipStream_t extra[3];
    for (auto& s : extra) CHECK(hipStreamCreate(&s));
    CHECK(hipDeviceSynchronize());
    sleep(seconds);
    printf("{\"pid\": %d, \"mode\": \"idle\", \"held_gb\": %zu}\n", (int)getpid(), gb);
    return 0;
  }
  const uint32_t iters = argc > 2 ? atoi(argv[2]) : 2000;
  const uint32_t d = argc > 3 ? atoi(argv[3]) : 200003;
  const int sync_every = argc > 4 ? atoi(argv[4]) : 7;
  const bool host_mode = mode[0] == 'h';
  const uint32_t nvec = d / 4;
  hipStream_t s = nullptr;  // the null stream, like torch's default
  uint32_t* row[ROWS];
  for (int r = 0; r < ROWS; ++r) CHECK(hipMalloc(&row[r], (size_t)d * 4 + 64));
  int32_t* ord;
  double* scratch;
  Log* log;
  CHECK(hipMalloc(&ord, 256));
  CHECK(hipMalloc(&scratch, 4096));
  CHECK(hipMalloc(&log, sizeof(Log)));
  CHECK(hipMemset(log, 0, sizeof(Log)));
  CHECK(hipMemset(scratch, 0, 4096));
  std::vector<uint32_t> host(host_mode ? (size_t)d : 0);  // pageable
  unsigned int seen = 0;
  for (uint32_t it = 0; it < iters; ++it) {
    if (host_mode) {
      for (int r = 0; r < ROWS; ++r) {
        for (uint32_t c = 0; c < d; ++c) host[c] = pattern(it, r, c);
        CHECK(hipMemcpy(row[r], host.data(), (size_t)d * 4, hipMemcpyHostToDevice));
      }
    } else {
      Table plain{};
      for (int r = 0; r < ROWS; ++r) plain.p[r] = row[r];
      hipLaunchKernelGGL(produce_kernel, dim3(1024), dim3(256), 0, s, plain, it, nvec, d, log);
    }
    hipLaunchKernelGGL(rank_kernel, dim3(1), dim3(64), 0, s, ord, it);
    hipLaunchKernelGGL(tiny_kernel, dim3(1), dim3(64), 0, s, scratch, 32);
    hipLaunchKernelGGL(tiny_kernel, dim3(1), dim3(64), 0, s, scratch, 32);
    Table tab{};
    for (int i = 0; i < ROWS; ++i) tab.p[i] = row[(i + 3 * it) % ROWS];
    for (int k = 0; k < NVAR; ++k) {
      switch ((k + it) % NVAR) {
        case 0: launch_consumer<0>(tab, ord, it, nvec, log, s); break;
        case 1: launch_consumer<1>(tab, ord, it, nvec, log, s); break;
        case 2: launch_consumer<2>(tab, ord, it, nvec, log, s); break;
        case 3: launch_consumer<3>(tab, ord, it, nvec, log, s); break;
        default: launch_consumer<4>(tab, ord, it, nvec, log, s); break;
      }
    }
    CHECK(hipGetLastError());
    if (sync_every > 0 && it % sync_every == 0) {  // a host round trip now and then, like `.floats()` / `.tolist()`
      CHECK(hipMemcpy(&seen, &log->count, sizeof(seen), hipMemcpyDeviceToHost));
      if (seen >= LOGCAP) break;
    }
  }
  CHECK(hipDeviceSynchronize());
  std::vector<char> raw(sizeof(Log));
  CHECK(hipMemcpy(raw.data(), log, sizeof(Log), hipMemcpyDeviceToHost));
  const Log* l = reinterpret_cast<const Log*>(raw.data());
  printf("{\"pid\": %d, \"mode\": \"%s\", \"iters\": %u, \"d\": %u, \"mismatch_words\": {\"nt\": %u, \"plain\": %u, \"sc1\": %u, "
         "\"acquire_nt\": %u, \"sc0sc1\": %u, \"producer_readback\": %u}, \"logged\": %u}\n",
         (int)getpid(), mode, iters, d, l->words[0], l->words[1], l->words[2], l->words[3], l->words[4], l->words[5],
         l->count);
  const unsigned int shown = l->count < 48 ? l->count : 48;
  for (unsigned int i = 0; i < shown; ++i) {
    const Entry& e = l->e[i];
    printf("  pid %d it %u variant %u rank %u row %u col %u (line %u, word %u of it) got %08x want %08x class %u wg %u xcc %u hwid %08x\n",
           (int)getpid(), e.it, e.variant, e.t, e.row, e.col,
           (unsigned)(((uintptr_t)row[e.row] + 4ull * e.col) >> 7 & 0xffffff), (unsigned)(((uintptr_t)row[e.row] / 4 + e.col) & 31),
           e.got, e.want, e.cls, e.block, e.xcc, e.hwid);
  }
  return 0;
}
