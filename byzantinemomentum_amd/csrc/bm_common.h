// bm_common.h — shared device/host helpers for libbm_gar.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <utility>
#include "../../include/bm_gar.h"

namespace bm {

// Row pointer table passed BY VALUE in the kernarg segment (512 B): no H2D copy,
// no host sync, indexable with compile-time indices (s_load from kernarg) or with a
// wave-uniform runtime index.  Replaces torch.stack (aggregators/median.py:39,
// trmean.py:79) — the n x d copy is never made.
struct RowTable {
  const float* p[BM_MAX_ROWS];
};

static inline int hip_code(hipError_t e) { return e == hipSuccess ? 0 : -(int)e; }

#define BM_LAUNCH_CHECK()                      \
  do {                                         \
    hipError_t e__ = hipGetLastError();        \
    if (e__ != hipSuccess) return hip_code(e__); \
  } while (0)

// ---------------------------------------------------------------------------
// Compile-time sorting network: Knuth's merge-exchange (TAOCP 5.2.2, Alg. M),
// valid for every N (not only powers of two).  The comparator list is a
// constexpr table so that every register index is a compile-time constant and
// the n values of a column stay in VGPRs.  Unused outputs are removed by the
// compiler's DCE, which turns the sorter into a selection network for the
// median for free.
// ---------------------------------------------------------------------------
template <int N>
struct MergeExchange {
  static constexpr int kMax = (N < 2) ? 1 : (N * 10);  // generous upper bound on #comparators
  struct Table {
    short a[kMax];
    short b[kMax];
    int count;
  };
  static constexpr Table make() {
    Table t{};
    t.count = 0;
    if (N < 2) return t;
    int tt = 0;
    while ((1 << tt) < N) ++tt;
    for (int p = 1 << (tt - 1); p > 0; p /= 2) {
      int q = 1 << (tt - 1), r = 0, d = p;
      while (true) {
        for (int i = 0; i < N - d; ++i)
          if ((i & p) == r) {
            t.a[t.count] = (short)i;
            t.b[t.count] = (short)(i + d);
            ++t.count;
          }
        if (q == p) break;
        d = q - p;
        q /= 2;
        r = p;
      }
    }
    return t;
  }
  static constexpr Table table = make();
};

// One comparator = v_min_f32 + v_max_f32.  Written as (non-volatile, hence DCE-able) inline asm:
// through fminf/fmaxf the compiler first canonicalises every loaded value (one extra v_max x,x per
// input, +10 % VALU on the median) because it must quiet signalling NaNs; columns that contain a
// NaN never use the network's result, so that is wasted work here.
__device__ __forceinline__ void cmp_exchange(float& a, float& b) {
  float lo, hi;
  asm("v_min_f32 %0, %1, %2" : "=v"(lo) : "v"(a), "v"(b));
  asm("v_max_f32 %0, %1, %2" : "=v"(hi) : "v"(a), "v"(b));
  a = lo;
  b = hi;
}

template <int N, size_t... I>
__device__ __forceinline__ void sort_network_impl(float (&x)[N], std::index_sequence<I...>) {
  (cmp_exchange(x[MergeExchange<N>::table.a[I]], x[MergeExchange<N>::table.b[I]]), ...);
}

// Ascending in-register sort of x[0..N).  Inputs must be NaN-free (v_min/v_max drop NaNs).
template <int N>
__device__ __forceinline__ void sort_network(float (&x)[N]) {
  if constexpr (N >= 2)
    sort_network_impl<N>(x, std::make_index_sequence<MergeExchange<N>::table.count>{});
}

// Correctly rounded x / m for a small positive integer m given rm = 1.0f/m (Markstein
// correction): three VALU ops instead of the ~10-op IEEE division sequence, same bits as
// torch's `.div_(m)` (aggregators/krum.py:80, bulyan.py:70) except for subnormal quotients and the sign of a zero
// quotient (-0 / m comes out as +0: equal as a value; tests/test_split_model.py checks the rest on a model).
__device__ __forceinline__ float div_small_int(float x, float m, float rm) {
  const float q = x * rm;
  const float r = __builtin_fmaf(-q, m, x);
  const float q1 = __builtin_fmaf(r, rm, q);
  // inf/nan inputs: r is nan, keep the plain product
  return (q1 == q1) ? q1 : q;
}

// 16/8/4-byte streaming loads of data that is read exactly once (non-temporal: do not
// displace useful lines from L2 / Infinity Cache).
template <int VEC>
struct VecLoad;
template <>
struct VecLoad<4> {
  using T = float __attribute__((ext_vector_type(4)));
};
template <>
struct VecLoad<2> {
  using T = float __attribute__((ext_vector_type(2)));
};
template <>
struct VecLoad<1> {
  using T = float;
};

template <int VEC>
__device__ __forceinline__ void load_stream(const float* p, float (&dst)[VEC]) {
  using T = typename VecLoad<VEC>::T;
  const T v = __builtin_nontemporal_load(reinterpret_cast<const T*>(p));
  if constexpr (VEC == 1) {
    dst[0] = v;
  } else {
#pragma unroll
    for (int c = 0; c < VEC; ++c) dst[c] = v[c];
  }
}

// Same with a wave-uniform base pointer and a 32-bit per-lane BYTE offset: the compiler emits
// `global_load_dwordx4 v, v_off, s[base:base+1]` (saddr form) — no 64-bit VALU address arithmetic
// and one offset register shared by all the rows of a column.
template <int VEC>
__device__ __forceinline__ void load_stream_off(const float* base, uint32_t byte_off, float (&dst)[VEC]) {
  load_stream<VEC>(reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + byte_off), dst);
}

template <int VEC>
__device__ __forceinline__ void store_stream_off(float* base, uint32_t byte_off, const float (&src)[VEC]);

template <int VEC>
__device__ __forceinline__ void store_stream(float* p, const float (&src)[VEC]) {
  using T = typename VecLoad<VEC>::T;
  T v;
  if constexpr (VEC == 1) {
    v = src[0];
  } else {
#pragma unroll
    for (int c = 0; c < VEC; ++c) v[c] = src[c];
  }
  __builtin_nontemporal_store(v, reinterpret_cast<T*>(p));
}

template <int VEC>
__device__ __forceinline__ void store_stream_off(float* base, uint32_t byte_off, const float (&src)[VEC]) {
  store_stream<VEC>(reinterpret_cast<float*>(reinterpret_cast<char*>(base) + byte_off), src);
}

// Stores of RESULT vectors (the aggregated gradient, averages, the Byzantine vector) with the default
// cache policy instead of non-temporal — the measured alternative, selected with BM_RESULT_NT=0.
// A/B on one GPU (profiles/r02_b_store_policy_ab.txt): kernel times are the same either way, but a whole
// C5 step takes 4.8-5.0 ms with cacheable result stores against 3.6-3.7 ms with non-temporal ones (dirty
// lines are written back at the kernel boundaries).  What a result store really costs is its place in the
// in-order vmcnt queue, see colwise_kernels.h.
template <int VEC>
__device__ __forceinline__ void store_result(float* p, const float (&src)[VEC]) {
  using T = typename VecLoad<VEC>::T;
  T v;
  if constexpr (VEC == 1) {
    v = src[0];
  } else {
#pragma unroll
    for (int c = 0; c < VEC; ++c) v[c] = src[c];
  }
  *reinterpret_cast<T*>(p) = v;
}
template <int VEC>
__device__ __forceinline__ void store_result_off(float* base, uint32_t byte_off, const float (&src)[VEC]) {
  store_result<VEC>(reinterpret_cast<float*>(reinterpret_cast<char*>(base) + byte_off), src);
}
// wave-uniform choice between the two policies (BM_RESULT_NT, A/B runs)
template <int VEC>
__device__ __forceinline__ void store_result_policy(float* p, const float (&src)[VEC], int nt) {
  if (nt)
    store_stream<VEC>(p, src);
  else
    store_result<VEC>(p, src);
}

// Grid cap of the plain column kernels (grid-stride above it).
constexpr int kColMaxBlocks = 256 * 64;

// Columns per launch such that every byte offset fits 32 bits (saddr addressing).
constexpr int64_t kMaxColsPerLaunch = (int64_t)1 << 29;

// Largest vector width (4, 2 or 1 floats) every pointer of the table and `extra` allow.
static inline int common_vec_width(const void* const* ptrs, int n, const void* extra) {
  uintptr_t bits = reinterpret_cast<uintptr_t>(extra);
  for (int i = 0; i < n; ++i) bits |= reinterpret_cast<uintptr_t>(ptrs[i]);
  if ((bits & 15u) == 0) return 4;
  if ((bits & 7u) == 0) return 2;
  return 1;
}

// Dynamic LDS beyond what a launch gets without asking: ONE rule for every kernel of the library.  A launch whose
// dynamic + static LDS exceeds 48 KB opts in through hipFuncAttributeMaxDynamicSharedMemorySize (gfx950 has 160 KB per
// workgroup; 48 KB is the most conservative default of the parts HIP runs on, so nothing here depends on a roomier
// one).  `static_bytes`: the __shared__ arrays the kernel declares besides its dynamic segment.
static inline int lds_opt_in(const void* kernel, size_t dynamic_bytes, size_t static_bytes) {
  if (dynamic_bytes + static_bytes <= 48u * 1024u) return 0;
  return hip_code(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dynamic_bytes));
}

// Grid size for a streaming kernel: enough workgroups to fill 256 CUs several times over,
// capped so that the grid-stride loop amortises launch/tail effects.
static inline int stream_grid(int64_t work_items, int block, int max_blocks) {
  int64_t g = (work_items + block - 1) / block;
  if (g < 1) g = 1;
  if (g > max_blocks) g = max_blocks;
  return (int)g;
}

// Compute units of the current device (the burst forms launch one workgroup per CU).
static inline int compute_units() {
  static int cached[16] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) dev = 0;
  if (cached[dev] == 0) {
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    cached[dev] = cus;
  }
  return cached[dev];
}

// "Am I the last workgroup of this launch to get here?" — the placement-independent hand-off of the CDNA4 guide
// (cdna_hip_programming.md, guideline 16): every wave drains its stores, the workgroup meets, ONE lane releases at
// agent scope (L2 write-back: the per-XCD L2s are not coherent with each other), takes a ticket from `counter`,
// and the workgroup holding the last ticket acquires at agent scope (its CU's L1 is invalidated) before it reads
// what the others wrote with plain loads.  `counter` must be zero when the launch starts: the kernel that precedes
// the launch on the stream zeroes it, every call.  Returns the same value on every lane of the workgroup.
__device__ __forceinline__ bool arrive_last(int* counter, int total, int* lds_flag) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const int ticket = atomicAdd(counter, 1);
    const int last = (ticket == total - 1) ? 1 : 0;
    if (last) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    *lds_flag = last;
  }
  __syncthreads();
  return *lds_flag != 0;
}

// One entry of a small index table that the PREVIOUS kernel on the stream wrote (the ranking, the selected rows), read
// at agent scope (`global_load_dword ... sc1`) by a vector lane instead of through the scalar cache.  Kernel boundaries
// make such tables visible by themselves; this form dates from round 5, when wrong coordinates out of Bulyan's second
// pass under GPU sharing were taken for a stale read of the ranking.  They were not (round 6: a library-free probe finds
// no stale word across kernel boundaries under sharing, and the failures followed the packed-fp32 instructions of that
// kernel — build.py).  The loads stay as they are: 72-256 bytes per launch, and one vector load + v_readlane
// per entry keeps the scalar registers free for the row pointers.
__device__ __forceinline__ int32_t load_index_coherent(const int32_t* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Deterministic block reduction (sum) of one double per thread; result valid on thread 0.
template <int BLOCK>
__device__ __forceinline__ double block_reduce_sum(double v, double* lds /* BLOCK/64 doubles */) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0) lds[wave] = v;
  __syncthreads();
  double r = 0.0;
  if (threadIdx.x == 0) {
#pragma unroll
    for (int w = 0; w < BLOCK / 64; ++w) r += lds[w];
  }
  __syncthreads();
  return r;
}

}  // namespace bm

namespace bm {
// Launch-shape knobs, read once from the environment (experiments only; defaults are the
// measured best on MI355X).
struct Tuning {
  int col_burst;       // BM_COL_BURST: iterations per CU from which median / trmean take their burst form (default 8; 0 = never, 1 = always: tests)
  int mean_burst;      // BM_MEAN_BURST: the same for the selected mean (default 8)
  int step_burst;      // BM_STEP_BURST: the same for bm_momentum_stats (default 8)
  int step_stream;     // BM_STEP_STREAM: 1 = the streaming form of bm_momentum_stats at every row count (tests)
  int pair_mode;       // BM_PAIR_MODE: 0 = centred bf16 Gram + accuracy gate (default), 1 = direct differences for every pair
  int pair_planes;     // BM_PAIR_PLANES (mode 0): 0 (default) by length, 2 or 3 forced
  int pair_dither;     // BM_PAIR_DITHER (mode 0, two planes): seed of the coordinate dither (default 0); -1 = round to nearest (A/B)
  double pair_tau;     // BM_PAIR_TAU: accuracy gate of mode 0 (see gram_to_sqdist_kernel); <= 0 disables
  int study_burst;     // BM_STUDY_BURST: iterations per CU from which bm_study_stats takes its burst form (default 8; 0 = never, 1 = always: tests)
  int step_stagger_us; // BM_STEP_STAGGER_US: start every other workgroup of an XCD this many microseconds late in the fused first pass of a Krum / Bulyan step (0 = off)
  int gram_steady;     // BM_GRAM_STEADY: 1 (default) = the condition-free steady-state loop of the Gram kernel, 0 = the generic loop only (A/B)
  int bulyan_short;    // BM_BULYAN_SHORT: 1 (default) = Bulyan pass 2 searches its window among the positions that straddle the median only (same bits), 0 = all positions (A/B)
  int pair_load_nt;    // BM_PAIR_LOAD_NT: 1 (default) = the Gram kernel's row loads carry the non-temporal hint, 0 = default cache policy (aligned rows; A/B)
  int rank_algo;       // BM_RANK_ALGO: how the rows' distances are put in order for the scores (rank_body.h): 0 (default) = counting up to 32 rows, a bitonic network per row beyond; 1 = bitonic, 2 = counting (A/B; same scores)
  int col_wide;        // BM_COL_WIDE: 1 (default) = median / trimmed mean at 29-52 rows take 16-byte columns, 0 = 8-byte ones (A/B)
  int brute_budget;    // BM_BRUTE_BUDGET: search-tree nodes per wave of the device Brute search before it gives up with status -2 (default 0 = 2^18; tests set a tiny one)
};
const Tuning& tuning();
Tuning& tuning_mutable();  // bm_tuning_set (A/B runs inside one process)
}  // namespace bm
