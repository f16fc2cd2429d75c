// pk_f32_probe.hip — library-free: are packed-fp32 VALU results right when several processes time-share one MI355X?
//
// What led here (profiles/r06_pass2_*): bulyan_pass2_kernel<25,5,4> returned a few wrong coordinates in ~1.5 % of its
// launches when 4-5 processes shared the GPU — always lanes 48..63 of a wave, always the same vector register, on inputs
// nobody had written for several launches — and NEVER when the same source was compiled with -fno-slp-vectorize, i.e.
// without the v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32 forms the SLP vectoriser builds from pairs of suffix sums
// (most of them with op_sel / op_sel_hi: one half of a register pair broadcast to both results).
//
// This probe has no library code.  A lane loads 18 x 16 bytes (non-temporal, like a second pass), then forms seven pairs
// of suffix sums per component twice: with v_pk_add_f32 (inline asm; FORM 0: `op_sel_hi:[1,0]`, the addend broadcast from
// the low half of its pair; FORM 1: no op_sel, the addend duplicated in both halves) and with scalar v_add_f32 in the
// same order.  fp32 addition is deterministic: any difference between the two is a wrong VALU result.
// Result on the MI355X pool (profiles/r06_fixcheck_summary.txt): zero differences in both forms, alone and with five
// processes sharing the GPU, 3 000 launches each — the packed additions by themselves are NOT reproducibly wrong; what
// fails is the kernel the compiler builds with them (497 packed instructions, 99 of them v_pk_mul_f32 / v_pk_fma_f32 with
// scalar register PAIRS as operands, interleaved with v_cmp / v_cndmask on VCC at 103 SGPRs), and only under
// time-sharing.  (A third form with hand-written scalar-pair operands differed from its scalar twin deterministically,
// on every lane and alone as well: an artefact of that inline asm, not the transient, and removed.)
// The library is built without packed fp32 (build.py).
//
//   hipcc --offload-arch=gfx950 -O2 -fno-slp-vectorize -o pk_f32_probe pk_f32_probe.hip
//   ./pk_f32_probe 3000                                   (alone)
//   for i in 1 2 3 4 5; do ./pk_f32_probe 3000 & done; wait   (five processes time-sharing the GPU)
#include <hip/hip_runtime.h>
#include <unistd.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                          \
  do {                                                                                    \
    hipError_t e_ = (x);                                                                  \
    if (e_ != hipSuccess) {                                                               \
      fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
      exit(2);                                                                            \
    }                                                                                     \
  } while (0)

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int R = 18, PAIRS = 7, LOGCAP = 1024;
struct Table { const float* p[R]; };
struct Entry { uint32_t it, form, v, lane, c, pair, half, got, want, xcc; };
struct Log { unsigned int count, wrong[3]; Entry e[LOGCAP]; };

// FORM 2 — what the narrowing of round 6 points at (profiles/r06_pass2_narrowing.txt: the failures follow the packed
// ADDITIONS of the suffix sums, `v_pk_add_f32 d, d, v[a:a+1] op_sel_hi:[1,0]`, whose source pair is (component 0,
// component 1) of a loaded 16-byte group while only component 0 is used — and every wrong coordinate was a component 1):
// the source pair is the loaded pair itself, its HIGH half is live data, and after the chains of component 0 the high
// halves are compared with copies taken before.  A difference means the instruction (or a wave switched out around
// it) damaged a register it only names.
__global__ __launch_bounds__(256) void pk_live_kernel(Table rows, uint32_t nvec, uint32_t it, Log* log, float* sink) {
  const uint32_t v = blockIdx.x * 256 + threadIdx.x;
  if (v >= nvec) return;
  f4 x[R];
#pragma unroll
  for (int t = 0; t < R; ++t) x[t] = __builtin_nontemporal_load(reinterpret_cast<const f4*>(rows.p[t]) + v);
  float kept[R];
#pragma unroll
  for (int t = 0; t < R; ++t) asm volatile("v_mov_b32 %0, %1" : "=v"(kept[t]) : "v"(x[t][1]));
  float keep = 0.0f;
#pragma unroll
  for (int q = 0; q < PAIRS; ++q) {
    const int i = 2 * q;
    f2 acc = {x[i][0], 0.0f};
#pragma unroll
    for (int t = i + 1; t < R; ++t) {
      f2 src = __builtin_shufflevector(x[t], x[t], 0, 1);  // (component 0, component 1): the registers of the load
      asm volatile("v_pk_add_f32 %0, %0, %1 op_sel_hi:[1,0]" : "+v"(acc), "+v"(src));
      x[t][0] = src[0];
      x[t][1] = src[1];
    }
    keep += acc.x + acc.y;
  }
#pragma unroll
  for (int t = 0; t < R; ++t) {
    float now = x[t][1];
    asm volatile("" : "+v"(now));
    if (__float_as_uint(now) != __float_as_uint(kept[t])) {
      atomicAdd(&log->wrong[2], 1u);
      const unsigned int slot = atomicAdd(&log->count, 1u);
      if (slot < LOGCAP)
        log->e[slot] = Entry{it, 2u, v, threadIdx.x & 63, 1u, (uint32_t)t, 1u, __float_as_uint(now), __float_as_uint(kept[t]),
                             (uint32_t)__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11))};
    }
  }
  if (keep == 12345.678f) sink[0] = keep;
}

template <int FORM>
__global__ __launch_bounds__(256) void pk_kernel(Table rows, uint32_t nvec, uint32_t it, Log* log, float* sink) {
  const uint32_t v = blockIdx.x * 256 + threadIdx.x;
  if (v >= nvec) return;
  f4 x[R];
#pragma unroll
  for (int t = 0; t < R; ++t) x[t] = __builtin_nontemporal_load(reinterpret_cast<const f4*>(rows.p[t]) + v);
  float keep = 0.0f;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
#pragma unroll
    for (int q = 0; q < PAIRS; ++q) {
      const int i = 2 * q;  // the pair (sum from rank i, sum from rank i + 1)
      f2 acc = {x[i][c], 0.0f};
      float lo = x[i][c], hi = 0.0f;
#pragma unroll
      for (int t = i + 1; t < R; ++t) {
        f2 add = {x[t][c], FORM == 0 ? -123.0f : x[t][c]};
        if (FORM == 0)
          asm volatile("v_pk_add_f32 %0, %0, %1 op_sel_hi:[1,0]" : "+v"(acc) : "v"(add));
        else
          asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(acc) : "v"(add));
        asm volatile("v_add_f32 %0, %0, %1" : "+v"(lo) : "v"(x[t][c]));
        asm volatile("v_add_f32 %0, %0, %1" : "+v"(hi) : "v"(x[t][c]));
      }
      const uint32_t got[2] = {__float_as_uint(acc.x), __float_as_uint(acc.y)};
      const uint32_t want[2] = {__float_as_uint(lo), __float_as_uint(hi)};
      for (int half = 0; half < 2; ++half)
        if (got[half] != want[half]) {
          atomicAdd(&log->wrong[FORM], 1u);
          const unsigned int slot = atomicAdd(&log->count, 1u);
          if (slot < LOGCAP)
            log->e[slot] = Entry{it, (uint32_t)FORM, v, threadIdx.x & 63, (uint32_t)c, (uint32_t)q, (uint32_t)half, got[half], want[half],
                                 (uint32_t)__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11))};
        }
      keep += acc.x + acc.y;
    }
  }
  if (keep == 12345.678f) sink[0] = keep;
}

int main(int argc, char** argv) {
  const uint32_t iters = argc > 1 ? atoi(argv[1]) : 2000;
  const uint32_t d = argc > 2 ? atoi(argv[2]) : 200003;
  const uint32_t nvec = d / 4;
  Table tab{};
  std::vector<float> host(d);
  for (int t = 0; t < R; ++t) {
    float* p;
    CHECK(hipMalloc(&p, (size_t)d * 4 + 64));
    uint32_t s = 12345u + 977u * t;
    for (uint32_t j = 0; j < d; ++j) { s = s * 1664525u + 1013904223u; host[j] = ((int)(s >> 8) % 20001 - 10000) * 1e-4f; }
    CHECK(hipMemcpy(p, host.data(), (size_t)d * 4, hipMemcpyHostToDevice));
    tab.p[t] = p;
  }
  Log* log;
  float* sink;
  CHECK(hipMalloc(&log, sizeof(Log)));
  CHECK(hipMemset(log, 0, sizeof(Log)));
  CHECK(hipMalloc(&sink, 64));
  const uint32_t grid = (nvec + 255) / 256;
  for (uint32_t it = 0; it < iters; ++it) {
    hipLaunchKernelGGL(pk_kernel<0>, dim3(grid), dim3(256), 0, 0, tab, nvec, it, log, sink);
    hipLaunchKernelGGL(pk_kernel<1>, dim3(grid), dim3(256), 0, 0, tab, nvec, it, log, sink);
    hipLaunchKernelGGL(pk_live_kernel, dim3(grid), dim3(256), 0, 0, tab, nvec, it, log, sink);
    if (it % 5 == 0) {  // a host round trip now and then, like the callers of the library
      unsigned int seen;
      CHECK(hipMemcpy(&seen, &log->count, 4, hipMemcpyDeviceToHost));
    }
  }
  CHECK(hipDeviceSynchronize());
  std::vector<char> raw(sizeof(Log));
  CHECK(hipMemcpy(raw.data(), log, sizeof(Log), hipMemcpyDeviceToHost));
  const Log* l = reinterpret_cast<const Log*>(raw.data());
  printf("{\"pid\": %d, \"launches_per_form\": %u, \"d\": %u, \"wrong_results\": {\"op_sel_hi_broadcast\": %u, \"plain_pairs\": %u, "
         "\"live_high_half_of_the_source_pair\": %u}}\n", (int)getpid(), iters, d, l->wrong[0], l->wrong[1], l->wrong[2]);
  for (unsigned int i = 0; i < (l->count < 64 ? l->count : 64); ++i) {
    const Entry& e = l->e[i];
    printf("  launch %u form %u group %u lane %u component %u pair %u half %u got %08x want %08x xcc %u\n", e.it, e.form, e.v, e.lane,
           e.c, e.pair, e.half, e.got, e.want, e.xcc);
  }
  return 0;
}
