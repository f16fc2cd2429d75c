/*
 * bm_gar.h — C ABI of libbm_gar.so, the MI355X (gfx950) implementation of the
 * Byzantine-robust gradient-aggregation hot path of LPD-EPFL/ByzantineMomentum.
 *
 * This is the drop-in boundary. Every entry point is what the reference's
 * `native` hook (aggregators/krum.py:22-26,82-96; bulyan.py:22-26,86-100;
 * median.py:22-26,41-49; brute.py:23-27,82-91) or its registry
 * (aggregators/__init__.py:71-86) would bind through an FFI for this path.
 * The reference-side binding (a ctypes stub) is shown in INTEGRATION.md.
 *
 * Conventions
 *   - `rows`   : HOST array of n DEVICE pointers, one per worker gradient
 *                (reference: `gradients: list[Tensor(d)]`, entries may alias,
 *                attacks/identical.py:86). Never written. n <= BM_MAX_ROWS.
 *   - `d`      : coordinates per gradient (fp32, contiguous).
 *   - `out`    : DEVICE pointer, caller-allocated, never aliases an input
 *                (aggregators/__init__.py:19).
 *   - `stream` : hipStream_t of the caller (torch.cuda.current_stream().cuda_stream);
 *                every launch goes there, no entry point synchronises the device.
 *   - return   : 0 on success; a negative value -hipError_t on a HIP failure;
 *                BM_EINVAL for an argument the kernels cannot serve.
 *   - The library owns no memory across calls; workspaces are caller-owned and
 *     sized by bm_workspace_bytes().
 */
#ifndef BM_GAR_H
#define BM_GAR_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BM_MAX_ROWS 64
#define BM_EINVAL   (-100000)
#define BM_ENOCOMM  (-100001) /* RCCL is not available in this process      */
#define BM_ECOMM    (-100002) /* an RCCL call failed                        */

/* Column-wise rules (coordinate-per-coordinate over the worker axis). */
enum bm_colwise_op {
  BM_OP_MEDIAN = 0, /* aggregators/median.py:31-39  lower median, rank (n-1)//2      */
  BM_OP_TRMEAN = 1, /* aggregators/trmean.py:24-33  mean of sorted ranks f..n-f-1    */
  BM_OP_PHOCAS = 2, /* aggregators/trmean.py:81-94  n-f closest to the trimmed mean  */
  BM_OP_MEAMED = 3  /* aggregators/trmean.py:96-109 n-f closest to the median        */
};

/* ABI version of this header; bumped on any signature change. */
int bm_abi_version(void);
/* TEST / MEASUREMENT ONLY — not part of the drop-in surface.  Sets one launch-shape knob by the name of its environment
 * variable (BM_COL_BURST, BM_PAIR_MODE, BM_BRUTE_BUDGET, ...: csrc/bm_common.h, struct Tuning) for the calls that
 * follow, so that an A/B of two settings can alternate inside one process; results never depend on a knob.  The knobs
 * are the library's only process-global mutable state ("no global mutable state besides a workspace
 * cache"): they are read from the environment once, and a production process never calls this.  No locking: the
 * caller must not run it concurrently with any other entry point.  BM_EINVAL for an unknown name. */
int bm_tuning_set(const char* name, int value);

/* Human-readable text for a code returned by any entry point. */
const char* bm_error_string(int code);

/* out[j] = op over {rows[i][j]}_i, 0 <= j < d.  f ignored for BM_OP_MEDIAN. */
int bm_colwise(int op, const float* const* rows, int n, int64_t d, int f,
               float* out, void* stream);

/* Workspace sizes (bytes) for the entry points below. kind: */
enum bm_ws_kind {
  BM_WS_PAIRWISE = 0, /* bm_pairwise_sqdist            */
  BM_WS_AKSEL    = 1, /* bm_aksel_pass1                */
  BM_WS_STATS    = 2, /* bm_stack_stats                */
  BM_WS_DOT      = 3, /* bm_multi_dot                  */
  BM_WS_STEP     = 4, /* bm_momentum_stats             */
  BM_WS_STUDY    = 5  /* bm_study_stats                */
};
int64_t bm_workspace_bytes(int kind, int n, int64_t d);

/* sq[i*n+j] = sum_k (rows[i][k]-rows[j][k])^2 as fp64, full symmetric n x n matrix,
 * zero diagonal.  Replaces the n(n-1)/2 x (sub, norm, .item()) loop of
 * aggregators/krum.py:41-48, bulyan.py:48-54, brute.py:43-45 (which take sqrt on top).
 * Deterministic: bitwise-equal rows give bitwise-equal distances.
 */
int bm_pairwise_sqdist(const float* const* rows, int n, int64_t d,
                       double* sq_nxn, void* ws, void* stream);

/* The same over ONE SHARD of a dimension-partitioned stack: d coordinates here, d_total >= d over
 * all shards.  The partial matrices of the shards add up to the matrix of the whole rows; the precision
 * plan of the pass (how each fp32 value is split for the matrix cores) follows d_total, so a job gives the
 * same plan whatever its world size.  bm_pairwise_sqdist(rows, n, d, ...) is this with d_total = d. */
int bm_pairwise_sqdist_shard(const float* const* rows, int n, int64_t d, int64_t d_total,
                             double* sq_nxn, void* ws, void* stream);

/* Score + stable rank on the device (one workgroup), from the squared distances.
 * mode BM_RANK_KRUM  : score_i = sum of the (n-f-1) smallest distances of row i
 *                      (aggregators/krum.py:50-62)
 * mode BM_RANK_BULYAN: score_i = sum of the m smallest distances of row i
 *                      (aggregators/bulyan.py:56-62)
 * Distances are sqrt(sq) in fp64, non-finite -> +inf (krum.py:46-47); sums are
 * fp64 in ascending order; order_out is the stable (lower index first) argsort of
 * the scores.  scores_out (n doubles) may be NULL. */
enum bm_rank_mode { BM_RANK_KRUM = 0, BM_RANK_BULYAN = 1 };
int bm_krum_rank(const double* sq_nxn, int n, int f, int m, int mode,
                 int32_t* order_out, double* scores_out, void* stream);

/* bm_pairwise_sqdist_shard + bm_krum_rank as one call for a single GPU (nothing to exchange between the two): the
 * third launch of the distance pass — the gated exact pass, which has nothing to do unless the accuracy gate listed
 * rows — ranks the rows from the final distances, so the ranking costs no launch of its own.  Same sq_nxn,
 * order_out, scores_out as the two calls.  ws as bm_pairwise_sqdist. */
int bm_pairwise_rank(const float* const* rows, int n, int64_t d, int64_t d_total, int f, int m, int mode,
                     double* sq_nxn, int32_t* order_out, double* scores_out, void* ws, void* stream);

/* out = (((0 + rows[idx[0]]) + rows[idx[1]]) + ...)/m, sequential fp32 like
 * `sum(...).div_(m)` at aggregators/krum.py:80, brute.py:80, aksel.py:64.
 * idx is a DEVICE array of m int32 (so no host sync between rank and mean).  A NEGATIVE entry means "there is no
 * selection" (bm_brute_select_device, status -2): out is then NaN everywhere. */
int bm_selected_mean(const float* const* rows, int n, const int32_t* idx, int m,
                     int64_t d, float* out, void* stream);

/* Bulyan pass 2 (aggregators/bulyan.py:64-84 with static scores): with
 * m_max = n-f-2, theta = n-2f-2, beta = theta-2f, for every coordinate
 *   sel[i] = mean(rows[order[i .. i+min(m, m_max-i)-1]]),  i < theta
 *   out    = mean of the beta values of sel closest to the lower median of sel.
 * order is the DEVICE output of bm_krum_rank(mode BULYAN). */
int bm_bulyan_pass2(const float* const* rows, int n, const int32_t* order, int f, int m,
                    int64_t d, float* out, void* stream);

/* Aksel pass 1 (aggregators/aksel.py:35-41): coordinate-wise lower median
 * (written to median_out if non-NULL) and sq[i] = sum_j (rows[i][j]-median[j])^2. */
int bm_aksel_pass1(const float* const* rows, int n, int64_t d, float* median_out,
                   double* sq_out, void* ws, void* stream);

/* tools/pytorch.py:97-125 in one pass: avg_out = sequential mean of the k rows;
 * out3 = { sum_j avg_j^2, sum_i sum_j (rows[i][j]-avg_j)^2, max_j |avg_j| }.
 * If scaled_out is non-NULL it receives avg + scale * (-avg) in the same pass, rounded like the
 * reference's `grad_att = grad_avg.neg(); grad_att.mul_(factor); byz = grad_avg.add_(grad_att)`:
 * the Byzantine vector of the "empire" attack with factor = scale (attacks/identical.py:63-86,
 * 129-134) comes for free with the honest-stack statistics the study block needs anyway
 * (attack.py:847).
 * attack_kind selects what scaled_out holds:
 *   BM_ATTACK_EMPIRE  avg + scale * (-avg)                         attacks/identical.py:129-134
 *   BM_ATTACK_LITTLE  avg + scale * sqrt(var_unbiased over rows)   attacks/identical.py:136-141
 *                     (pass a negative scale for the reference's `negative:True`) */
enum bm_attack_kind { BM_ATTACK_EMPIRE = 0, BM_ATTACK_LITTLE = 1,
                      /* OR-ed in: scaled_out = scale * direction only, without the average (what the factor
                       * search of attacks/identical.py:67-77 keeps as `grad_att`)                             */
                      BM_ATTACK_DIRECTION = 16 };
int bm_stack_stats(const float* const* rows, int k, int64_t d, float* avg_out,
                   float* scaled_out, float scale, int attack_kind, double* out3, void* ws,
                   void* stream);

/* The dot products of the study block (attack.py:851-868) in one pass:
 *   out[a*nc+b]   = <core[a], core[b]>   for the nc (<= 4) "core" vectors (symmetric, the
 *                   diagonal holds the squared norms), then
 *   out[nc*nc+e]  = <core[0], extra[e]>  for ne (<= 32) more vectors (the past sampled
 *                   averages of the curvature term, attack.py:863-865).  All fp64. */
int bm_multi_dot(const float* const* core, int nc, const float* const* extra, int ne,
                 int64_t d, double* out, void* ws, void* stream);

/* The study block of a step (attack.py:830,848-868) in ONE pass over the d-sized vectors:
 *   - statistics of the attack stack = f_real copies of byz (tools.compute_avg_dev_max, attack.py:848): the
 *     sequential mean a and the deviations are rebuilt per element from byz with the reference's operations;
 *     a is written to attack_avg_out only if that is non-NULL;
 *   - |defense|^2 and max|defense| (attack.py:851-852);
 *   - the Gram matrix of core = (sampled avg, honest avg, defense, attack avg) behind the six cosines
 *     (attack.py:854-859), <sampled avg, past_newest> (attack.py:861-862) and <sampled avg, curv>, curv being
 *     C = sum_i mu^i past_i: the curvature term mu * sum_i mu^i <s, past_i> of attack.py:863-866 as ONE dot;
 *   - |params - origin|^2 (attack.py:830) when both are non-NULL;
 *   - the update of C for the next step, in place, AFTER its dot product:
 *       curv_mode 0  no curvature term kept (curv, past_* ignored)
 *                 1  first step: C <- sampled avg (no dots with the past)
 *                 2  C <- sampled avg + mu * C
 *                 3  C <- sampled avg + mu * (C + oldest_weight * past_oldest), oldest_weight = -(mu^(P-1)):
 *                    the entry that leaves a full ring of P past averages is taken out first.
 * out (DEVICE, BM_STUDY_SLOTS doubles):
 *   [4a+b] <core_a, core_b> (row/column 3 zero when f_real = 0)   [16] <s, past_newest>   [17] <s, C before the update>
 *   [18] sum avg_a^2   [19] sum_i |a_i - avg_a|^2   [20] max|avg_a|   [21] max|defense|   [22] |params - origin|^2
 * ws: bm_workspace_bytes(BM_WS_STUDY).  f_real = 0: no attack (byz ignored). */
#define BM_STUDY_SLOTS 32
int bm_study_stats(const float* sampled_avg, const float* honest_avg, const float* defense, const float* byz,
                   int f_real, float* attack_avg_out, const float* past_newest, float* curv,
                   const float* past_oldest, int curv_mode, float mu, float oldest_weight, const float* params,
                   const float* origin, int64_t d, double* out, void* ws, void* stream);
/* bm_study_stats that also carries the momentum of the update (attack.py:836-838, `--momentum-at update`, the
 * reference's default placement): update_momentum <- fma(one_minus_damp, defense, momentum_mu * update_momentum), in
 * place, element by element with the bits of bm_multi_fma3(M, M, defense, mu, 1 - damp) — inside the pass that reads
 * the defense vector anyway (2 row units more instead of a 3-unit pass and a launch of its own).  update_momentum
 * NULL: exactly bm_study_stats.  The statistics are those of `defense`, not of the updated momentum. */
int bm_study_stats_update(const float* sampled_avg, const float* honest_avg, const float* defense, const float* byz,
                          int f_real, float* attack_avg_out, const float* past_newest, float* curv,
                          const float* past_oldest, int curv_mode, float mu, float oldest_weight, const float* params,
                          const float* origin, float* update_momentum, float momentum_mu, float one_minus_damp,
                          int64_t d, double* out, void* ws, void* stream);

/* sq_out[i] = |rows[i]|^2 (DEVICE, k doubles), every row read once: the `gradient.norm().item()` of
 * aggregators/cge.py:28-38 and of the clipping at attack.py:776-779,791-794 for all gradients in one call, without a
 * host round trip per gradient.  ws: bm_workspace_bytes(BM_WS_DOT). */
int bm_row_sqnorms(const float* const* rows, int k, int64_t d, double* sq_out, void* ws, void* stream);

/* order_out = stable argsort (ties to the lower index, NaN last) of n fp64 keys that live on
 * the device: the `d.sort(key=...)` of aggregators/aksel.py:48 without a host round trip. */
int bm_stable_argsort(const double* keys, int n, int32_t* order_out, void* stream);

/* y[i] = a*y[i] + b*x[i] for k vectors at once: worker momentum
 * `gmtm.mul_(mu).add_(grad, alpha=1-damp)` at attack.py:800-804. */
int bm_multi_axpby(float* const* y, const float* const* x, int k, int64_t d,
                   float a, float b, void* stream);

/* First pass of a simulation step in ONE kernel (attack.py:791-804,846-847 + attacks/identical.py:63-86):
 *   s_i      = clip_factors[i] * sampled[i]                       (clip_factors NULL: s_i = sampled[i])
 *   buffers[i] <- mu * buffers[i] + one_minus_damp * s_i, i < h   worker momentum, in place; these ARE
 *                                                                  the honest gradients the rule sees
 *   sampled_avg, honest_avg = sequential means of the ks sampled rows s_i / the h updated buffers
 *   byz_out  = honest_avg + scale * (-honest_avg)                 BM_ATTACK_EMPIRE
 *            = honest_avg + scale * sqrt(unbiased column variance) BM_ATTACK_LITTLE
 *   out6     = { sum avg_s^2, sum_i ||s_i-avg_s||^2, max|avg_s|,  sum avg_h^2, sum_i ||b_i-avg_h||^2, max|avg_h| }
 * i.e. tools.compute_avg_dev_max of both stacks (tools/pytorch.py:97-125) without re-reading them.
 * ks >= h; any of sampled_avg / honest_avg / byz_out may be NULL.  ws: bm_workspace_bytes(BM_WS_STEP). */
int bm_momentum_stats(const float* const* sampled, int ks, float* const* buffers, int h, int64_t d,
                      float mu, float one_minus_damp, const float* clip_factors, float* sampled_avg,
                      float* honest_avg, float* byz_out, float scale, int attack_kind, double* out6,
                      void* ws, void* stream);

/* The same first pass followed by a coordinate-wise rule over the h updated buffers and n_byz copies of byz_out, i.e.
 * defense_out = GAR(honests + [byz] * n_byz, f = rule_f) of attack.py:821 for rule_op = BM_OP_MEDIAN / TRMEAN / PHOCAS /
 * MEAMED (aggregators/median.py:39, trmean.py:33,81-109), with the results of bm_momentum_stats + bm_colwise (same
 * bits).  For the four rules over ks = h = 20 buffers and 1..6 Byzantine copies (n = 21..26: the
 * reference's n = 25, f = 5 among them), or 14 buffers and 11 copies (its n = 25, f = 11), the rule runs INSIDE the first pass, on the values it already holds in
 * registers: the rule's own pass over the n rows disappears (26 of the 97 row passes of such a step).  Any other
 * shape runs the two kernels one after the other.  byz_out and defense_out must be non-NULL, attack_kind without
 * BM_ATTACK_DIRECTION. */
int bm_momentum_stats_colwise(const float* const* sampled, int ks, float* const* buffers, int h, int64_t d,
                              float mu, float one_minus_damp, const float* clip_factors, float* sampled_avg,
                              float* honest_avg, float* byz_out, float scale, int attack_kind, int rule_op,
                              int rule_f, int n_byz, float* defense_out, double* out6, void* ws, void* stream);

/* The same first pass together with the squared distances of the n = h + n_byz rows (the h updated buffers and n_byz
 * copies of byz_out) that Krum / Bulyan / Brute rank next (attack.py:821 with a distance-based rule and worker momentum):
 * sq_nxn as bm_pairwise_sqdist_shard(rows, n, d, d_total, ...) would give it.  For ks = h = 20 with 1..6 Byzantine
 * copies (or 14 with 11) and gradients long enough for the burst form, the centred rows are contracted on the matrix cores INSIDE the first
 * pass, from the registers that hold them: the distance pass never re-reads the n rows (25 of the 115 row passes of
 * such a step).  The distances then differ from the stand-alone pass by its rounding (another centre, another
 * order: both within 1e-5 of fp64); everything else has the bits of bm_momentum_stats.  Any other shape runs the two
 * passes one after the other.  ws: bm_workspace_bytes(BM_WS_STEP); ws_pair: bm_workspace_bytes(BM_WS_PAIRWISE, n, d).
 * attack_kind without BM_ATTACK_DIRECTION. */
int bm_momentum_stats_sqdist(const float* const* sampled, int ks, float* const* buffers, int h, int64_t d,
                             int64_t d_total, float mu, float one_minus_damp, const float* clip_factors,
                             float* sampled_avg, float* honest_avg, float* byz_out, float scale, int attack_kind,
                             int n_byz, double* sq_nxn, double* out6, void* ws, void* ws_pair, void* stream);

/* The same two fusions when there are no momentum buffers: the honest rows are the k sampled rows themselves, as with
 * `--momentum-at update` (the reference's default placement, attack.py:809-810,837-839).  bm_stack_stats (average,
 * Byzantine vector, statistics; tools/pytorch.py:97-125, attacks/identical.py:63-86) together with
 *   bm_stack_stats_colwise: defense_out = rule(rows + [byz] * n_byz, f = rule_f), the rule of bm_colwise;
 *   bm_stack_stats_sqdist : sq_nxn = the squared distances of rows + [byz] * n_byz (n = k + n_byz).
 * For k = 20 with 1..6 Byzantine copies, or 14 with 11, 16-byte aligned rows and d a multiple of 4 (for the distances
 * also: gradients long enough for the burst form) it is ONE pass over the k rows; otherwise the stand-alone kernels
 * run one after the other.  out6[0..2] = out6[3..5] = { sum avg^2, sum_i |row_i - avg|^2, max|avg| }. */
int bm_stack_stats_colwise(const float* const* rows, int k, int64_t d, float* avg_out, float* byz_out, float scale,
                           int attack_kind, int rule_op, int rule_f, int n_byz, float* defense_out, double* out6,
                           void* ws, void* stream);
int bm_stack_stats_sqdist(const float* const* rows, int k, int64_t d, int64_t d_total, float* avg_out, float* byz_out,
                          float scale, int attack_kind, int n_byz, double* sq_nxn, double* out6, void* ws,
                          void* ws_pair, void* stream);

/* out[i] = b * q[i] + a * (p_scale[i] * p[i]) for k vectors (p_scale: DEVICE array of k floats or NULL;
 * entries of q may be one shared vector; out[i] may alias p[i]).  Every momentum placement of the loop:
 * worker (attack.py:800-804), server (:805-808), update (:838-839), Nesterov look-ahead (:762,767). */
int bm_multi_fma3(float* const* out, const float* const* p, const float* const* q, int k, int64_t d,
                  float a, float b, const float* p_scale, void* stream);
/* The same with b read from DEVICE memory (one double, rounded to fp32 like the host's conversion of the same number):
 * the Byzantine vector avg + factor * att (identical.py:82-84) from the factor bm_attack_line_search_device left on
 * the device, without a host round trip. */
int bm_multi_fma3_bdev(float* const* out, const float* const* p, const float* const* q, int k, int64_t d,
                       float a, const double* b_dev, const float* p_scale, void* stream);

/* Gradient clipping (attack.py:776-779,791-794) without a host round trip:
 * bm_clip_factors: factors[i] = clip/sqrt(row_sq[i]) if sqrt(row_sq[i]) > clip else 1 (device arrays);
 * bm_multi_scale : y[i] *= factors[i] in place; rows whose factor is 1 are not touched. */
int bm_clip_factors(const double* row_sq, int k, float clip, float* factors_out, void* stream);
int bm_multi_scale(float* const* y, int k, int64_t d, const float* factors, void* stream);

/* Brute subset search on the host (aggregators/brute.py:47-68): over all
 * C(n, n-f) subsets in lexicographic order, first subset of smallest diameter;
 * subsets touching a non-finite distance are skipped.  dist_nxn holds sqrt'ed
 * distances (host memory; entries [x*n + y] with x < y are read).  Writes n-f
 * ascending indices; returns 0, or BM_EINVAL if no finite subset exists.
 * The subsets are not enumerated (bisection over the distances, a search tree of
 * depth <= f per probe): n = 51, f = 12 — 1.6e11 subsets — takes 0.1 ms. */
int bm_brute_select(const double* dist_nxn, int n, int f, int32_t* sel_out);
/* The subset search of bm_brute_select ON THE DEVICE (one workgroup of 16 waves that probe 16 thresholds / 16 rows at
 * a time, csrc/brute.hip), from the SQUARED distances where bm_pairwise_sqdist left them: no copy out, no host search,
 * no copy in — distances -> search -> bm_selected_mean are three launches on one stream, and a HIP graph can record
 * them.  sel_out (DEVICE, BM_MAX_ROWS int32): the n-f rows ascending, then zeros — the index table bm_selected_mean
 * reads.  status (DEVICE, one int32): 0; -1 when every subset touches a non-finite distance (brute.py:56-57,68: the
 * reference then fails its assertion) — sel_out then holds n-f copies of the first row all of whose distances are
 * non-finite, so that the average that follows is non-finite where that row is; -2 when the search gave up on its
 * budget of 2^18 search-tree nodes per wave (the tree is exponential in f in the worst case; a crafted matrix must
 * not hold the stream for seconds) — sel_out then holds n-f times the index -1, which bm_selected_mean answers with
 * NaN in EVERY coordinate (never the average of some rows); bm_brute_select on the host has no such limit and is what
 * the Python host falls back to.
 * Same selections as bm_brute_select (tests/test_gpu_parity_r4.py). */
int bm_brute_select_device(const double* sq_nxn, int n, int f, int32_t* sel_out, int32_t* status, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Dim-sharded aggregation: one process per GPU, every rank holds all n rows
 * restricted to its d_local coordinates.  The reference has no counterpart (its aggregation is
 * single-device, attack.py:811-825); these are the entry points a multi-GPU binding would use.
 * RCCL is bound lazily with dlopen (no link-time dependency).  A NULL comm means "one rank".
 */
typedef struct bm_comm bm_comm;

int bm_comm_available(void);                  /* 1 if RCCL could be bound in this process        */
int bm_comm_unique_id(void* id128);           /* rank 0: 128 bytes to hand to every other rank   */
int bm_comm_init(bm_comm** out, int nranks, int rank, const void* id128);  /* collective         */
int bm_comm_destroy(bm_comm* comm);
int bm_comm_size(const bm_comm* comm);

/* In-place sum over the ranks of `count` doubles on the caller's stream (the n x n squared-distance
 * partials, the packed step statistics); no-op for one rank. */
int bm_allreduce_sum_f64(bm_comm* comm, double* buf, int64_t count, void* stream);
/* all[r*count .. (r+1)*count) = rank r's `mine`: the full output vector from the per-rank slices. */
int bm_allgather_f32(bm_comm* comm, const float* mine, float* all, int64_t count_per_rank, void* stream);

/* Multi-Krum / Bulyan of the local slice in ONE call: partial squared distances -> all-reduce ->
 * score + stable rank (identical on every rank) -> selected mean / Bulyan pass 2 of the slice
 * (aggregators/krum.py:31-80, bulyan.py:31-84).  order_out (DEVICE, BM_MAX_ROWS int32, may be NULL)
 * receives the ranking.  ws: bm_sharded_workspace_bytes(n, d_local).  d_total >= d_local is the length of the
 * WHOLE vectors (all shards): the plan of the distance pass follows it, and every rank must state the same number
 * (a short or empty trailing shard then plans exactly like its peers, whatever the world size). */
int64_t bm_sharded_workspace_bytes(int n, int64_t d_local);
int bm_sharded_krum(bm_comm* comm, const float* const* rows, int n, int64_t d_local, int64_t d_total, int f, int m,
                    float* out_local, int32_t* order_out, void* ws, void* stream);
int bm_sharded_bulyan(bm_comm* comm, const float* const* rows, int n, int64_t d_local, int64_t d_total, int f, int m,
                      float* out_local, int32_t* order_out, void* ws, void* stream);

/* The tail of bm_sharded_krum / bm_sharded_bulyan (all-reduce -> rank -> selected mean / pass 2) when the squared
 * distances of the local shard are already in the workspace: bm_momentum_stats_sqdist wrote them to
 * bm_sharded_sq_slot(ws), using bm_sharded_pair_workspace(ws) as its ws_pair.  rule: BM_RULE_KRUM or BM_RULE_BULYAN. */
double* bm_sharded_sq_slot(void* ws);
void* bm_sharded_pair_workspace(void* ws);
int bm_sharded_rule_from_sq(bm_comm* comm, int rule, const float* const* rows, int n, int64_t d_local, int f, int m,
                            float* out_local, int32_t* order_out, void* ws, void* stream);

/* ---------------------------------------------------------------------------------------------
 * One simulation step with worker-side momentum (attack.py:786-868) as ONE call on the caller's stream:
 * [clipping factors] -> bm_momentum_stats -> rule over buffers + [byz] * f_real -> bm_study_stats (attack /
 * defense statistics, study dots, l2 from the origin, curvature combination: one pass) -> one packed exchange.
 * attack_avg_out may be NULL (the attack average is only needed as a vector by callers that want it).
 * comm NULL = one rank; otherwise every collective of the dim-sharded step (row norms when clipping, the
 * n x n squared distances, the packed statistics) goes through it.  d = coordinates of this rank.
 *
 * stats_out (DEVICE, bm_step_stats_count() doubles, identical on every rank):
 *   [0] sum avg_s^2  [1] sum_i |s_i-avg_s|^2  [2] sum avg_h^2  [3] sum_i |b_i-avg_h|^2  [4] |defense|^2
 *   [5] sum avg_a^2  [6] sum_i |a_i-avg_a|^2  [7] |params-origin|^2
 *   [8 + 4a + b] <core_a, core_b>, core = (sampled avg, honest avg, defense, attack avg)
 *   [24] <sampled avg, past_newest>  [25] <sampled avg, curv>   [26..29] max|avg_s|, max|avg_h|, max|defense|, max|avg_a|
 * curv (in/out, may be NULL when nb_past = 0): C = sum_i mu^i past_i, updated for the NEXT step
 * (C <- s when past_count = 0, else C <- mu * (C + oldest_weight * past_oldest) ... see step.py);
 * past_oldest non-NULL only when the caller's ring of nb_past vectors is full (its last entry).
 */
enum bm_step_rule { BM_RULE_KRUM = 0, BM_RULE_BULYAN = 1, BM_RULE_MEDIAN = 2, BM_RULE_TRMEAN = 3,
                    BM_RULE_PHOCAS = 4, BM_RULE_MEAMED = 5,
                    BM_RULE_BRUTE = 6, BM_RULE_AVERAGE = 7 /* bm_attack_* only, not bm_step_worker */ };
typedef struct bm_step_params {
  int32_t n, f_decl, f_real;  /* workers, declared and real Byzantine ones; honest = n - f_real      */
  int32_t ks;                 /* sampled gradients (>= honest)                                        */
  int32_t rule;               /* bm_step_rule                                                          */
  int32_t m;                  /* Multi-Krum / Bulyan m, 0 = n - f_decl - 2                             */
  int32_t attack_kind;        /* bm_attack_kind                                                        */
  int32_t nb_past;            /* length P of the caller's ring of past sampled averages, 0 = none      */
  int32_t past_count;         /* entries in the ring before this step                                  */
  float attack_scale;         /* factor of the attack                                                  */
  float mu, one_minus_damp;   /* momentum, 1 - dampening                                               */
  float clip;                 /* gradient clipping threshold, <= 0 = off                               */
  float oldest_weight;        /* -(mu^(P-1)): weight that takes the leaving entry out of C              */
} bm_step_params;
int bm_step_stats_count(void);
int64_t bm_step_workspace_bytes(int n, int64_t d_local);
int bm_step_worker(bm_comm* comm, const bm_step_params* p, const float* const* sampled, float* const* buffers,
                   int64_t d, int64_t d_total /* >= d: all shards, as bm_sharded_krum */, float* defense_out,
                   float* sampled_avg_out, float* honest_avg_out, float* byz_out,
                   float* attack_avg_out, const float* past_newest, float* curv, const float* past_oldest,
                   const float* params, const float* origin, double* stats_out, void* ws, void* stream);

/* ONE candidate of the attacks' factor search against the trimmed mean, phocas or meamed, evaluate only
 * (attacks/identical.py:73-76 with aggregators/trmean.py:69-109 as the defense):
 *     out[0] = sum_j ( RULE(honests + [avg + t * dir] * copies)_j - avg_j )^2        (DEVICE, one double)
 * The candidate vector is never written (it is fma(t, dir, avg) in registers, the bits bm_multi_fma3 would store),
 * nor the rule's output: h + 2 rows read, nothing written, against h + 5 read and 2 written by candidate vector +
 * bm_colwise + objective.  Instances exist for n = h + copies in {11, 25, 51} (bm_colwise_eval_supported);
 * ws: bm_colwise_eval_workspace_bytes().  d == 0 is legal (an empty shard: out[0] = 0). */
int bm_colwise_eval_supported(int op, int n);
int64_t bm_colwise_eval_workspace_bytes(void);
int bm_colwise_eval(int op, const float* const* honests, int h, int copies, int64_t d, int f, const float* avg,
                    const float* dir, float t, double* out, void* ws, void* stream);
/* (Also op = BM_OP_MEDIAN with h = 2, copies = 1: the middle of (candidate, honests[0], honests[1]) — the median's own
 * factor search, where the two rows are order statistics of the honest rows formed once per search.)
 * The same with t read from DEVICE memory (one double, rounded to fp32 like the host's conversion of the same number):
 * the factor bm_search_device_next left there. */
int bm_colwise_eval_tdev(int op, const float* const* honests, int h, int copies, int64_t d, int f, const float* avg,
                         const float* dir, const double* t_dev, double* out, void* ws, void* stream);

/* ---------------------------------------------------------------------------------------------
 * The factor search of the "identical" attacks (attacks/identical.py:67-77, the reference's default
 * factor=-16) — HOST functions, no stream (but bm_attack_line_search_device).
 *
 * bm_search_*: the exploration of tools/misc.py:468-514 (best-effort arg-max over x >= 0 within a budget of
 * evaluations; reference defaults: start 0, delta 1, ratio 0.8) as a cursor the CALLER drives:
 *     bm_search_begin(&c, start, delta, ratio);
 *     repeat `evals` times: bm_search_propose(&c, &x); y = <evaluate at x>; bm_search_report(&c, y);
 *     answer: c.best_x
 * The evaluations are the caller's (a rule on the device, or bm_attack_objective on scalars); the library
 * never calls back.  bm_search is plain data owned by the caller.
 *
 * bm_attack_objective / bm_attack_line_search: for a rule whose output is the mean of a selected subset
 * (BM_RULE_KRUM with m, BM_RULE_BRUTE, BM_RULE_AVERAGE), the objective |GAR(honests + [avg + t*att]*k, f)
 * - avg|^2 of identical.py:72-76 evaluated from scalars only.  ext: HOST row-major (h+2) x (h+2) squared
 * distances (bm_pairwise_sqdist) among the h honest rows, their average and average + att.
 * sel_out (may be NULL): the rows the rule averages, indices >= h being Byzantine copies; count_out
 * their number.  bm_attack_line_search runs the whole search (negative: identical.py:70-71) and returns
 * the factor the attack then uses.  No d-sized vector is touched: 16 evaluations cost microseconds. */
typedef struct bm_search {
  double best_x, best_y;  /* incumbent: abscissa and value (valid once evaluations >= 1)           */
  double probe;           /* last abscissa proposed                                                 */
  double step, ratio;     /* current step; contraction factor, 0.5 < ratio < 1                      */
  int32_t phase;          /* 0 first evaluation, 1 growing, 2 shrinking                             */
  int32_t evaluations;    /* values reported so far                                                 */
  int32_t awaiting;       /* 1 between bm_search_propose and bm_search_report                       */
  int32_t reserved;
} bm_search;
int bm_search_begin(bm_search* c, double start, double delta, double ratio);
int bm_search_propose(bm_search* c, double* x_out);
int bm_search_report(bm_search* c, double y);
int bm_attack_objective(const double* ext, int h, int k, int f, int rule, int m, double t, double* y_out,
                        int32_t* sel_out, int32_t* count_out);
int bm_attack_line_search(const double* ext, int h, int k, int f, int rule, int m, int evals, int negative,
                          double* factor_out, double* trace_out);
/* bm_attack_line_search on the DEVICE, for BM_RULE_KRUM and BM_RULE_AVERAGE (BM_EINVAL for BM_RULE_BRUTE: its search
 * is the host form's): ext is the DEVICE matrix where bm_pairwise_sqdist left it, one workgroup evaluates the `evals`
 * candidates (csrc/search_device.hip) and writes out[0] = the factor, out[1 + 2e], out[2 + 2e] = abscissa and objective
 * of evaluation e (out: DEVICE, 1 + 2 * evals doubles).  Same candidates and the same bits as the host form; no copy,
 * no synchronisation — the factor is consumed where it is by bm_multi_fma3_bdev.  (MEASUREMENT ONLY: with
 * BM_SEARCH_TRACE=1 in the environment the kernel also writes 24 phase timestamps per evaluation behind the results and
 * `out` must hold 1 + 26 * evals doubles; scripts/search_kernel_probe.py.) */
int bm_attack_line_search_device(const double* ext, int h, int k, int f, int rule, int m, int evals, int negative,
                                 double* out, void* stream);
/* out[0] = |a - b|^2 (fp64, DEVICE): the objective `aggregated.sub_(grad_avg); aggregated.dot(aggregated)` of
 * identical.py:75-76 for a candidate whose rule was run on the vectors; one pass over the two vectors (the n x n
 * distance pass spends three launches on them).  ws: bm_colwise_eval_workspace_bytes().  Partial sums in a fixed order. */
int bm_sqdist2(const float* a, const float* b, int64_t d, double* out, void* ws, void* stream);
/* bm_attack_ranking on the DEVICE, the factor read from DEVICE memory (one double): order_out[0..63] (int32, DEVICE; the n
 * rows by rank, then zeros) for honests + [avg + t*att] * k, from the matrix where bm_pairwise_sqdist left it.  One
 * workgroup; no copy, no synchronisation.  k >= 1.  mode: BM_RANK_KRUM or BM_RANK_BULYAN (m as for bm_krum_rank). */
int bm_attack_ranking_device(const double* ext, int h, int k, int f, int mode, int m, const double* t_dev,
                             int32_t* order_out, void* stream);
/* out[0] = | pass2(honests + [avg + t * dir] * copies, order) - avg |^2 (fp64, DEVICE): bm_bulyan_pass2 on a stack whose
 * last `copies` rows are ONE candidate of the factor search (identical.py:67-77), evaluate only — the candidate is formed
 * in registers with the arithmetic of bm_multi_fma3, nothing is written.  `order` ranks the n = h + copies rows (indices
 * >= h name a copy), e.g. by bm_attack_ranking.  t from the host, or from DEVICE memory when t_dev != NULL (one double,
 * rounded to fp32).  Instances: (n, f) in {(11, 2), (25, 5), (51, 12)} with m = n - f - 2 (bm_bulyan_pass2_eval_supported),
 * d <= 2^29.  ws: bm_colwise_eval_workspace_bytes(). */
int bm_bulyan_pass2_eval_supported(int n, int f, int m);
int bm_bulyan_pass2_eval(const float* const* honests, int h, int copies, const int32_t* order, int f, int m, int64_t d,
                         const float* avg, const float* dir, float t, const double* t_dev, double* out, void* ws,
                         void* stream);
/* lo[j] = the value of rank il, hi[j] = the value of rank ih (0-based, ascending) among rows[0..h)[j]; a rank below 0 reads
 * -inf, a rank beyond h - 1 reads +inf; a NaN in the column makes both NaN.  One pass over the h rows (h <= 51:
 * bm_order_pair_supported).  With n = h + k, il = (n-1)/2 - k, ih = (n-1)/2 these are the lower medians (median.py:31-39)
 * of the rows with k copies of -inf, resp. +inf — the two vectors between which the median of `rows + [b] * k` moves
 * with b: the median's own factor search (identical.py:67-77) forms them once and evaluates every candidate as the
 * middle of three (bm_colwise_eval with BM_OP_MEDIAN).  Values of the rows, no arithmetic: the bits of the two median
 * calls it replaces. */
int bm_order_pair_supported(int h);
int bm_order_pair(const float* const* rows, int h, int64_t d, int il, int ih, float* lo, float* hi, void* stream);
/* The cursor of bm_search_* kept in DEVICE memory, for the searches whose candidates are evaluated by d-sized kernels
 * (median, trimmed mean, phocas, meamed, any rule): the host queues
 *     bm_search_device_next(state, NULL, negative, 0, start, delta, ratio, t, out)      first candidate
 *     repeat: <evaluation kernels that read t[0]: bm_multi_fma3_bdev, bm_colwise_eval_tdev, the rule, and leave the
 *              objective y[0] on the device>;  bm_search_device_next(state, y, negative, 0, ..., t, out)
 *     bm_search_device_next(state, y_last, negative, 1, ..., NULL, out)                  the factor
 * without waiting for any of them: `state` DEVICE memory of sizeof(bm_search), t DEVICE double (the candidate's signed
 * factor, identical.py:70-71), out DEVICE 1 + 2 * evals doubles laid out like bm_attack_line_search_device's.  Same
 * candidates as the host cursor (one definition, csrc/search_core.h).  start / delta / ratio are read on the first call. */
int bm_search_device_next(void* state, const double* y, int negative, int last, double start, double delta, double ratio,
                          double* t_out, double* out, void* stream);
/* The ranking bm_krum_rank(mode, m) would give for honests + [avg + t*att] * k, from the same scalars (order_out: n
 * int32, indices >= h being Byzantine copies): for the rules whose output needs the vectors but whose ranking does not
 * — Bulyan (aggregators/bulyan.py:48-62 rank, :64-84 second pass): a candidate of the factor search then costs
 * bm_bulyan_pass2 with this ranking instead of a distance pass over the n rows as well.  m <= 0: n - f - 2. */
int bm_attack_ranking(const double* ext, int h, int k, int f, int mode, int m, double t, int32_t* order_out);

#ifdef __cplusplus
}
#endif
#endif /* BM_GAR_H */
