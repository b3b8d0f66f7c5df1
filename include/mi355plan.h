/*
 * mi355plan.h -- C ABI of libmi355plan.so, the MI355X (gfx950) planning core.
 *
 * The reference (eleurent/rl-agents) is pure Python and has no FFI; this header is the boundary a
 * maintainer binds (ctypes / cffi ABI mode) to replace the bodies of the
 * reference functions cited on each entry point.  Citations are relative to /root/reference.
 *
 * Conventions
 *   - every function returns 0 on success or a negative MP_ERR_* code; it never throws.
 *     mp_last_error() returns a thread-local, human-readable message for the last failure.
 *   - mp_ctx owns one HIP device + one stream + growable device workspaces.  Not thread-safe per
 *     ctx; use one ctx per (process, GPU).
 *   - array arguments are C-contiguous.  `mem` selects where the caller's arrays live:
 *       MP_MEM_HOST   (0): host pointers; the call copies in, runs, copies out and synchronises.
 *       MP_MEM_DEVICE (1): device pointers on the ctx device (e.g. torch tensors' data_ptr());
 *                          the call only enqueues work on the ctx stream and returns.
 *     Output pointers may be NULL when the caller does not want that output.
 *   - all floating point is IEEE double evaluated in the reference's operation order with no
 *     fused multiply-add; integer indices are int32 on the device (tables are accepted as the
 *     int64 numpy arrays the reference holds and range-checked on upload).
 */
#ifndef MI355PLAN_H
#define MI355PLAN_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MP_ABI_VERSION 7

#define MP_OK 0
#define MP_ERR_HIP (-1)          /* a HIP runtime call failed (message has the hipError string)  */
#define MP_ERR_REWARD_RANGE (-2) /* OPD: reward outside [0,1] (deterministic.py:46-47 ValueError) */
#define MP_ERR_ALLOC (-3)
#define MP_ERR_ARG (-4)          /* invalid argument / unsupported configuration                  */
#define MP_ERR_MODE (-5)         /* model kind not valid for this call ("Unknown mode")           */

#define MP_MEM_HOST 0
#define MP_MEM_DEVICE 1
/* OR-ed into MP_MEM_HOST: `rng_state` alone is a DEVICE pointer (an mp_rng, below) while every other array is a host
 * array -- the 48-byte generator record per root then never crosses PCIe (it only has to when Python asks for it). */
#define MP_MEM_RNG_DEVICE 2

/* model kinds (value_iteration.py:51-61 `mode`) */
#define MP_MODE_DETERMINISTIC 0
#define MP_MODE_STOCHASTIC 1
#define MP_MODE_SPARSE 2
#define MP_MODE_CARTPOLE 3

typedef struct mp_ctx mp_ctx;
typedef struct mp_model mp_model;

const char *mp_last_error(void);
int mp_abi_version(void);

/* ---------------------------------------------------------------- context ------------------- */
/* `stream`: a hipStream_t to enqueue on (e.g. torch.cuda.current_stream().cuda_stream), or NULL
 * to let the ctx create and own a stream. */
int mp_ctx_create(int device, void *stream, mp_ctx **out);
int mp_ctx_destroy(mp_ctx *ctx);
int mp_ctx_set_stream(mp_ctx *ctx, void *stream);
/* Waits for the ctx's stream.  Returns MP_ERR_ARG -- once, then clears -- when a device-array call since the last
 * synchronisation had to clamp out-of-range roots (mp_uct_plan_models / mp_opd_plan_models with MP_MEM_DEVICE cannot validate
 * model_index / root_state on the host; host arrays are validated before the launch, as the reference's IndexError would be). */
int mp_ctx_synchronize(mp_ctx *ctx);
/* the sticky count behind that (read without clearing; exact only after the work was waited for) */
int mp_ctx_device_faults(mp_ctx *ctx, int32_t *count);
/* the hipStream_t the ctx enqueues on (its own or the caller's): lets the caller order events / collectives after its work */
int mp_ctx_get_stream(mp_ctx *ctx, void **stream);
/* device facts for reports: compute units, wavefront size, LDS bytes per workgroup, HBM bytes */
int mp_ctx_device_info(mp_ctx *ctx, int32_t *n_cu, int32_t *wave_size, int64_t *lds_bytes, int64_t *hbm_bytes,
                       char *name, int32_t name_cap);

/* ---------------------------------------------------------------- host-inclusive fast path --- */
/*
 * SURVEY.md 8(d) measures plan() with host arrays in and out (trainer/evaluation.py:168: agent.plan(observation) is a host
 * call).  Three things make that path fast; none changes a result:
 *   - ZERO-COPY: mp_host_alloc / mp_host_free hand out host memory that is pinned AND mapped into the device's address
 *     space (hipHostMalloc, portable | mapped).  When every array of a MP_MEM_HOST call lies in such memory, the kernels
 *     read the root states from and write the results to the caller's arrays directly over the bus: the call is one
 *     launch and one synchronisation, no copy is issued (every plan entry point; MP_NO_ZERO_COPY=1 disables).
 *     mp_uct_plan does so up to 65 536 roots (MP_ZERO_COPY_MAX); larger batches in such memory run as two chunks on two
 *     streams with asynchronous copies (the first chunk's results travel under the second chunk's kernel: faster than
 *     13 MB of kernel stores over the bus).  Any other host pointer is still accepted everywhere and goes through
 *     staging copies.
 *   - mp_uct_plan with mem = MP_MEM_HOST and ordinary (pageable) arrays splits batches of more than 65 536 roots into
 *     chunks and pipelines H2D(roots) -> kernel -> D2H(results) of different chunks over several HIP streams owned by
 *     the ctx (same kernels, same trees, same results: a root's plan depends on its own state and stream only).
 *     mp_last_kernel_ms then covers the whole pipelined region.  MP_PIPE_CHUNK=<roots> / MP_PIPE_STREAMS=<1..8>
 *     override; MP_PIPE_CHUNK=0 disables.
 *   - mp_rng: generator records resident on the device (see MP_MEM_RNG_DEVICE).
 */
int mp_host_alloc(mp_ctx *ctx, int64_t bytes, void **out);
int mp_host_free(mp_ctx *ctx, void *ptr);
/*
 * n numpy-PCG64 generator records uint64 [n,6] on the device -- the planners' np_random (tree_search/abstract.py:124-131)
 * for a batch of roots, continuing across plan() calls exactly as the host records would.
 *   mp_rng_set / mp_rng_get copy records [first, first+count) from / to a host array (synchronous).
 *   mp_rng_seed_sequence fills them as Generator(PCG64(SeedSequence(entropy + [first_key + i]))) for record first + i:
 *   `entropy` = n_words uint32 words (numpy's little-endian split of the entropy integers), the root's global index is
 *   appended as one more entropy integer -- rl_agents_amd's batch streams (AbstractPlanner.batch_rng_states).  Computed
 *   on the host in C (numpy's SeedSequence hashing restated; tests compare with numpy) and uploaded.
 *   mp_rng_device_ptr: the device address of record `first`, to pass as `rng_state` with MP_MEM_RNG_DEVICE (or with
 *   MP_MEM_DEVICE).
 * mp_seed_sequence_states is the host-only half: out uint64 [count,6].
 */
typedef struct mp_rng mp_rng;
int mp_rng_create(mp_ctx *ctx, int32_t n, mp_rng **out);
int mp_rng_free(mp_rng *rng);
int mp_rng_set(mp_rng *rng, int32_t first, int32_t count, const uint64_t *state6);
int mp_rng_get(mp_rng *rng, int32_t first, int32_t count, uint64_t *state6);
int mp_rng_seed_sequence(mp_rng *rng, int32_t first, int32_t count, const uint32_t *entropy, int32_t n_words,
                         int64_t first_key);
uint64_t *mp_rng_device_ptr(mp_rng *rng, int32_t first);
int mp_seed_sequence_states(const uint32_t *entropy, int32_t n_words, int64_t first_key, int32_t count, uint64_t *out);

/* ---------------------------------------------------------------- transition models ---------- */
/*
 * Deterministic finite-MDP tables: what the planners reach through env.step on a FiniteMDPEnv
 * (tree_search/abstract.py:158-161) and value iteration through mdp.transition / mdp.reward /
 * mdp.terminal (dynamic_programming/value_iteration.py:52-53,62-63).  Replaces
 * common/factory.py:119-134 safe_deepcopy_env: a cloned environment is an (int32 state,
 * int32 steps) pair on the device.
 *   transition int64 [M,S,A], reward double [M,S,A], terminal uint8 [S] or NULL.
 *   M > 1 = several models of one MDP (robust_value_iteration.py:21-27); tree search uses model 0.
 *   done_on_next: 0 -> terminated = terminal[s] (state acted FROM), 1 -> terminal[s'].
 *   max_steps: TimeLimit-style truncation for rollouts, 0 = none.
 * Always host pointers (model upload is outside every timed region).
 */
int mp_model_load_table(mp_ctx *ctx, int32_t M, int32_t S, int32_t A, const int64_t *transition,
                        const double *reward, const uint8_t *terminal, int32_t done_on_next, int32_t max_steps,
                        mp_model **out);
/*
 * A BATCH of N independent deterministic finite MDPs of one shape (S states, A actions each), one per episode of a
 * benchmark batch.  The reference evaluates one environment per process (trainer/evaluation.py:139-194,
 * scripts/experiments.py:102-106) and, for an environment that is not a FiniteMDPEnv (highway-v0), re-extracts its own
 * table with to_finite_mdp() at EVERY step (dynamic_programming/value_iteration.py:29-35, tree_search/abstract.py:59-62
 * through env_preprocessors): a batch of such episodes is N distinct tables, all of which change between two steps.
 *   transition int64 [N,S,A] (state indices LOCAL to each MDP, 0 <= t < S), reward double [N,S,A],
 *   terminal uint8 [N,S] or NULL; done_on_next / max_steps as mp_model_load_table.  N * S * A < 2^31.
 * The result is ONE table model over the disjoint union of the N state spaces -- GLOBAL state b * S + s, records packed
 * exactly as mp_model_load_table packs them -- so that every entry point that takes a deterministic table model works
 * on it unchanged and a root's plan depends on its own MDP only: mp_uct_plan*, mp_opd_plan, mp_ropd_plan,
 * mp_saopd_*, mp_policy_load* (per-state policies over the N * S global states), mp_env_step, mp_greedy_actions take
 * GLOBAL states; mp_uct_plan_models / mp_opd_plan_models below take (model_index, local state) pairs instead.
 * Tree exports report global states.  mp_vi_solve on it would test convergence over all N MDPs at once, which is not what
 * N agents do: use mp_vi_solve_batch (mp_vi_solve refuses a batch model).
 * mp_model_batch_info: *N = number of MDPs (1 for any other model), *S_each = states per MDP.
 */
int mp_model_load_table_batch(mp_ctx *ctx, int32_t N, int32_t S, int32_t A, const int64_t *transition, const double *reward,
                              const uint8_t *terminal, int32_t done_on_next, int32_t max_steps, mp_model **out);
int mp_model_batch_info(const mp_model *model, int32_t *N, int32_t *S_each);
/*
 * Replace the tables of MDPs [first, first + count) of a batch model (or, with first = 0 and count = 1, the whole of a
 * single-MDP table model from mp_model_load_table with M = 1): the per-step delta of a batch of highway episodes is each
 * episode's own table.  Arrays as mp_model_load_table_batch with N = count (host pointers; terminal must be given iff the
 * model was loaded with terminal flags).  Stream-ordered on the ctx stream through a pinned staging block owned by the
 * model: work enqueued earlier still reads the old tables, work enqueued later the new ones; nothing is synchronised
 * unless the previous update's copies are still in flight.  Only the touched records are re-packed.  Policies loaded for
 * the model become invalid (their fused records hold the old transitions).
 */
int mp_model_update_tables(mp_model *model, int32_t first, int32_t count, const int64_t *transition, const double *reward,
                           const uint8_t *terminal);
/*
 * Delta upload of a changed model (value_iteration.py:12-21,29-35 re-converts the env on every act):
 * replace n_rows rows of a deterministic table model (single, M = 1, or batch).
 *   rows int32 [n_rows]: GLOBAL state ids, distinct;  transition int64 [n_rows,A] (LOCAL next-state indices of the row's
 *   MDP), reward double [n_rows,A], terminal uint8 [n_rows] or NULL = the rows' terminal flags do not change.
 * With terminal == NULL only the records of the listed rows are re-packed; otherwise every record of the MDPs that own a
 * listed row is (a record carries terminal[next]: predecessors of a row whose flag changed must follow).  Stream-ordered
 * like mp_model_update_tables.
 */
int mp_model_update_rows(mp_model *model, int32_t n_rows, const int32_t *rows, const int64_t *transition, const double *reward,
                         const uint8_t *terminal);
/*
 * Environments that restrict the actions available in a state -- state.get_available_actions(), read by
 * DeterministicNode.expand (deterministic.py:32-35) -- as a table:
 *   available uint8 [S,A], non-zero = action a is listed in state s; every state needs at least one (else MP_ERR_ARG).
 * Host pointer.  Applies to the deterministic table model it is called on; mp_opd_plan then expands the available
 * actions only.  Policies loaded earlier for the model become invalid.  (MCTS reads availability through its POLICIES,
 * mcts.py:59-97: see mp_policy_load_listed.)
 */
int mp_model_set_available(mp_model *model, const uint8_t *available);
/*
 * Dense stochastic model (value_iteration.py:54-55, robust_value_iteration.py:55-56):
 *   transition double [M,S,A,S], reward double [M,S,A], terminal uint8 [S] or NULL.
 * mem = MP_MEM_DEVICE borrows the caller's device arrays (no copy; they must outlive the model).
 */
int mp_model_load_dense(mp_ctx *ctx, int32_t M, int32_t S, int32_t A, const double *transition,
                        const double *reward, const uint8_t *terminal, int32_t mem, mp_model **out);
/*
 * A block of source-state rows of a dense model, for value iteration sharded over GPUs: this
 * rank owns rows [row0, row0 + S_rows) of every model, i.e. transition double [M,S_rows,A,S_cols] (S_cols = |S|),
 * reward double [M,S_rows,A], terminal uint8 [S_rows] (flags of the owned rows) or NULL.  Use with mp_vi_backup.
 */
int mp_model_load_dense_rows(mp_ctx *ctx, int32_t M, int32_t S_rows, int32_t A, int32_t S_cols,
                             const double *transition, const double *reward, const uint8_t *terminal, int32_t mem,
                             mp_model **out);
/* Sparse model (value_iteration.py:56-59): transition double [S,A,B], next int64 [S,A,B]. */
int mp_model_load_sparse(mp_ctx *ctx, int32_t S, int32_t A, int32_t B, const double *transition,
                         const int64_t *next, const double *reward, const uint8_t *terminal, mp_model **out);
/*
 * Closed-form CartPole (gymnasium CartPole-v0/v1 dynamics, absent from this image; restated in
 * rl_agents_amd/envs/cartpole.py).  A clone is (x, x_dot, theta, theta_dot, steps).
 */
typedef struct {
    double gravity, masscart, masspole, length, force_mag, tau;
    double theta_threshold, x_threshold;
    int32_t max_steps;       /* TimeLimit: 200 for CartPole-v0 */
    int32_t euler;           /* 1 = explicit Euler (gymnasium default) */
} mp_cartpole_params;
int mp_model_load_cartpole(mp_ctx *ctx, const mp_cartpole_params *params, mp_model **out);
/*
 * CartPole's dynamics take math.sin / math.cos of the pole angle (gymnasium's cartpole.py, restated in
 * rl_agents_amd/envs/cartpole.py), i.e. the host C library's sin / cos -- which are not correctly rounded: "the" value is
 * what glibc's algorithm yields.  The kernels evaluate glibc's dbl-64 algorithm for |x| < 0.855469 themselves (a pole angle
 * stays below 0.21 rad), in the form that reproduces THIS host's libm:
 *   mp_libm_sincos_variant(): 1 = the FMA-contracted form (x86-64 CPUs with FMA + AVX2), 2 = every operation rounded
 *   (SSE2 / AVX variants), 0 = neither matched the host's sin / cos on the probe sample -- the device then uses its own
 *   math library and CartPole plans carry the tolerance of rounds 1-4 (>= 99.5 % of roots identical); host only, cached.
 *   mp_libm_sincos: the restated functions on the HOST, variant 1 / 2 (0 = libm itself): s[i], c[i] = sin, cos of x[i].
 *   mp_selftest_sincos: the same on the DEVICE (host arrays in / out) -- tests compare 10^7 angles with the host's libm;
 *   variant 3 / 4: the branch-free forms of variant 1 / 2 that the CartPole rollouts evaluate (|x| < 0.855 only).
 */
int mp_libm_sincos_variant(void);
int mp_libm_sincos(int32_t n, const double *x, int32_t variant, double *s, double *c);
int mp_selftest_sincos(mp_ctx *ctx, int32_t n, const double *x, int32_t variant, double *s, double *c);
int mp_model_free(mp_model *model);
int mp_model_info(const mp_model *model, int32_t *mode, int32_t *M, int32_t *S, int32_t *A, int32_t *B);

/* ---------------------------------------------------------------- value iteration ----------- */
/*
 * ValueIterationAgent.get_state_action_value (value_iteration.py:42-45) =
 * fixed_point_iteration (:65-73) of bellman_expectation(best_action_value(q)) (:47-63) from
 * Q0 = 0, with numpy.allclose(rtol, atol) early exit returning the PREVIOUS iterate; and, with
 * robust = 1, RobustValueIterationAgent.get_state_action_value (robust_value_iteration.py:39-58):
 * min over the M models, no terminal masking.
 *   Q_out double [S,A]; sweeps_out int32 [1] = Bellman sweeps actually executed.
 * Deterministic and sparse modes are bit-exact with the reference.  The dense mode has two forms (mp_vi_dense_mode):
 * in numpy's own order of roundings and additions -- bit-exact Q, V and sweep counts (the default) -- or on the f64
 * matrix cores, which accumulate in another order than numpy's pairwise sum (relative error ~1e-15 per sweep: Q within
 * 1e-12, greedy actions equal, sweep count within 1).
 * Deterministic models with |S| <= 16 384 are solved by ONE persistent launch whose workgroups hand V to each other and
 * therefore must all be resident at once.  On a GPU shared with other work that can fail (bounded spins, no hang):
 *   mem = MP_MEM_HOST  : the call notices and transparently solves again on the chained launches;
 *   mem = MP_MEM_DEVICE: asynchronous, nothing is read back: sweeps_out[0] = -1 and NaN in Q_out report the failure --
 *                        check sweeps_out, or set MP_VI_NO_PERSIST=1 to always use the chained launches.
 */
int mp_vi_solve(mp_ctx *ctx, mp_model *model, double gamma, int32_t iterations, double rtol, double atol,
                int32_t robust, double *Q_out, int32_t *sweeps_out, int32_t mem);
/*
 * N value-iteration agents at once: mp_vi_solve of every MDP of a batch model (mp_model_load_table_batch) in ONE launch,
 * one workgroup per MDP -- the fixed-point iteration of each MDP runs to ITS OWN allclose exit, exactly as N
 * ValueIterationAgent objects would (value_iteration.py:42-45,65-73 per agent; trainer/evaluation.py:139-194 runs one
 * agent per process).  Bit-exact with N mp_vi_solve calls on the N tables.
 *   Q_out double [N,S,A] (= [N*S, A] over global states: what mp_greedy_actions takes), sweeps_out int32 [N].
 * mem = MP_MEM_DEVICE only enqueues.  Rows of an MDP live in registers and its value vector in LDS when
 * S <= 4096 (any such S), else the value vector is double-buffered in LDS (S <= 10 200) or kept in global memory.
 */
int mp_vi_solve_batch(mp_ctx *ctx, mp_model *model, double gamma, int32_t iterations, double rtol, double atol,
                      double *Q_out, int32_t *sweeps_out, int32_t mem);
/* get_state_value (value_iteration.py:37-40): the V-form iteration.  V_out double [S]. */
int mp_vi_solve_v(mp_ctx *ctx, mp_model *model, double gamma, int32_t iterations, double rtol, double atol,
                  double *V_out, int32_t mem);
/* RobustValueIterationAgent.get_state_value (robust_value_iteration.py:32-37): V <- max_a min_m (R_m + gamma next_v_m(V)),
 * allclose on V, no terminal masking; deterministic and dense models. */
int mp_vi_solve_v_robust(mp_ctx *ctx, mp_model *model, double gamma, int32_t iterations, double rtol, double atol,
                         double *V_out, int32_t mem);
/*
 * One Bellman backup of a dense (or row-block) model: Q[rows,A] = min_m (R_m + gamma * mask(T_m . V)), the body of
 * bellman_expectation (value_iteration.py:54-55,62-63; robust_value_iteration.py:46-58).  V double [S_cols] in,
 * Q double [S_rows,A] out.  The row-sharded driver (rl_agents_amd/distributed.py) all_gathers V between backups.
 */
int mp_vi_backup(mp_ctx *ctx, mp_model *model, double gamma, int32_t robust, const double *V, double *Q, int32_t mem);
/*
 * How dense models (MP_MODE_STOCHASTIC) are contracted with V by every VI entry point of this ctx
 * (value_iteration.py:54-55: (T * v).sum(axis=-1) -- products rounded one by one, numpy's pairwise add.reduce):
 *   MP_VI_DENSE_MFMA   v_mfma_f64_16x16x4_f64 (fused, reordered accumulation): tolerance parity, see mp_vi_solve
 *   MP_VI_DENSE_EXACT  the reference's order restated (eight strided accumulators per block of <= 128 elements, the
 *                      halving recursion above it): bit-exact; both stream T once and are bound by HBM
 * Without a call the environment decides (MP_VI_DENSE=mfma|exact), else MP_VI_DENSE_EXACT: parity first, and at
 * 0.25 flop per byte the contraction is bound by HBM in both forms (measured: DESIGN.md 4.5).
 */
#define MP_VI_DENSE_MFMA 0
#define MP_VI_DENSE_EXACT 1
int mp_vi_dense_mode(mp_ctx *ctx, int32_t mode);
/* Host only (no GPU needed): the tables MP_VI_DENSE_EXACT sums a row of n elements by -- numpy's add.reduce: the
 * running sum, from the identity 0., of the pairwise sums (DOUBLE_pairwise_sum: blocks of at most 128, halves rounded
 * down to a multiple of 8) of the row's pieces of numpy.getbufsize() = 8192 elements.
 * leaves int32 [cap_leaves,2] {offset, length} left to right, leaf 0 = {0, 0} is the identity; nodes int32 [cap_nodes,2] {left, right} result
 * slots of the recursion's additions ordered by height (leaf l = slot l, addition k = slot n_leaves + k, the last one
 * is the row's sum); hoff int32 [cap_heights + 1]: additions of height h + 1 are [hoff[h], hoff[h+1]);
 * counts int32 [4] = {n_leaves, n_nodes, n_heights, most 8-element steps in a leaf}.  Arrays too small (or NULL) are
 * left alone: call once for the counts.  tests/test_host_logic.py replays the tables against numpy.add.reduce. */
int mp_vi_exact_plan(int32_t n, int32_t cap_leaves, int32_t *leaves, int32_t cap_nodes, int32_t *nodes, int32_t cap_heights,
                     int32_t *hoff, int32_t *counts);
/* Timing hook: run exactly `sweeps` Bellman sweeps (no early exit), Q left on the device. */
int mp_vi_sweeps(mp_ctx *ctx, mp_model *model, double gamma, int32_t sweeps, int32_t robust);

/* ---------------------------------------------------------------- UCT ----------------------- */
/*
 * MCTS.plan (tree_search/mcts.py:179-184) for n_roots independent roots: `episodes` iterations of
 * MCTS.run (:132-158: select by MCTSNode.selection_strategy :275-286 with random tie-break
 * abstract.py:296-311, expand :237-246, rollout MCTS.evaluate :160-177, backup
 * MCTSNode.update_branch :248-265), then AbstractPlanner.get_plan (abstract.py:143-156) with
 * MCTSNode.selection_rule (mcts.py:212-218).  Open loop (closed_loop = False).
 *   root_state  : table model int32 [n_roots]; cartpole double [n_roots,4]
 *   root_steps  : int32 [n_roots] env.steps at plan time (TimeLimit), or NULL = 0
 *   prior_p     : double [A]  prior over actions 0..A-1   (mcts.py:46-97 policies)
 *   rollout_p   : double [A]  rollout distribution; sampled as numpy Generator.choice(p=...)
 *   rng_state   : uint64 [n_roots,6] numpy PCG64 state {state_hi, state_lo, inc_hi, inc_lo,
 *                 has_uint32, uinteger}; advanced in place exactly as the reference's
 *                 planner.np_random would be (so plans are bit-identical at equal seeds)
 *   plans       : int32 [n_roots,max_plan_len], -1 padded;  plan_len int32 [n_roots]
 *   root_value  : double [n_roots];  root_child_count int64 [n_roots,A];
 *   root_child_value double [n_roots,A];  env_steps int64 [n_roots] (= len(planner.observations))
 */
int mp_uct_plan(mp_ctx *ctx, mp_model *model, int32_t n_roots, const void *root_state, const int32_t *root_steps,
                int32_t episodes, int32_t horizon, double gamma, double temperature, const double *prior_p,
                const double *rollout_p, uint64_t *rng_state, int32_t max_plan_len, int32_t *plans,
                int32_t *plan_len, double *root_value, int64_t *root_child_count, double *root_child_value,
                int64_t *env_steps, int32_t mem);
/*
 * mp_uct_plan on a batch model (mp_model_load_table_batch) with one MDP per root: root i plans on MDP model_index[i]
 * from its LOCAL state root_state[i] (model_index int32 [n_roots], values in [0, N); several roots may share an MDP).
 * Everything else as mp_uct_plan; equal to mp_uct_plan with root_state = model_index * S + root_state.
 */
int mp_uct_plan_models(mp_ctx *ctx, mp_model *model, int32_t n_roots, const int32_t *model_index, const int32_t *root_state,
                       const int32_t *root_steps, int32_t episodes, int32_t horizon, double gamma, double temperature,
                       const double *prior_p, const double *rollout_p, uint64_t *rng_state, int32_t max_plan_len, int32_t *plans,
                       int32_t *plan_len, double *root_value, int64_t *root_child_count, double *root_child_value,
                       int64_t *env_steps, int32_t mem);
/*
 * Per-state prior / rollout policies: MCTSWithPriorPolicyAgent (tree_search/mcts_with_prior.py:8-71) replaces the
 * planner's two policies by agent_policy_available (:47-62), the action distribution a prior agent gives for the state
 * the policy is asked about.  On a finite MDP that is a table:
 *   prior   double [S,A]  prior[s,a]   = probability handed to MCTSNode.expand when a node reached in state s is expanded
 *   rollout double [S,A]  rollout[s,:] = distribution MCTS.evaluate samples from in state s (Generator.choice(p=...))
 * Host pointers (policy upload is outside every timed region, like model upload).  Any |A| >= 2: mp_uct_plan_policy
 * (nodes' policy rows in registers) plans with 2..8 actions; beyond that mp_uct_plan_stochastic_policy does -- its loop
 * forms take any number of actions and deterministic tables as well as stochastic / sparse models.
 * A policy is tied to the model it was loaded for (its records are fused into the policy tables): use it with that
 * model only (checked) and free it before the model.
 * mp_uct_plan_policy = mp_uct_plan with (prior_p, rollout_p) looked up per state; everything else is identical,
 * including mp_uct_step_tree / mp_uct_tree_export on the trees it leaves.
 */
typedef struct mp_policy mp_policy;
int mp_policy_load(mp_ctx *ctx, mp_model *model, const double *prior, const double *rollout, mp_policy **out);
/*
 * Policies over restricted action sets: random_available_policy / preference_policy (mcts.py:59-97) and
 * agent_policy_available (mcts_with_prior.py:56-62) list, for a state, only the actions state.get_available_actions()
 * returns; MCTSNode.expand (mcts.py:237-246) creates one child per LISTED action and the exploration term scales with
 * len(children) = their number (mcts.py:286).
 *   listed uint8 [S,A]: non-zero = the prior policy lists action a in state s (an action may be listed with prior
 *   probability 0: it still gets a child); NULL = every action in every state (= mp_policy_load).  At least one action
 *   per state.  Unlisted actions must have rollout probability 0 unless the rollout policy ignores availability
 *   (random_policy, mcts.py:46-57) -- rollout[s,:] is simply the distribution to sample from.  |A| <= 8.
 */
int mp_policy_load_listed(mp_ctx *ctx, mp_model *model, const double *prior, const double *rollout,
                          const uint8_t *listed, mp_policy **out);
/* mp_policy_load_listed + the order in which the ROLLOUT policy lists the actions of each state: rollout_slot uint8 [S, A],
 * slot k of state s is column rollout_slot[s, k] (a permutation per state, zero-probability columns last; NULL = the
 * column order).  The columns -- the order of a node's children and of every tie-break -- follow the PRIOR policy's
 * listing; the rollout's inverse CDF (mcts.py:172, np_random.choice(actions, p=...)) runs over ITS listing.  The two
 * differ when one of them is the `random` policy, which lists np.arange(n) whatever the environment lists (mcts.py:46-57),
 * on an environment whose get_available_actions() is not ascending (highway-env: IDLE first). */
int mp_policy_load_ordered(mp_ctx *ctx, mp_model *model, const double *prior, const double *rollout,
                           const uint8_t *listed, const uint8_t *rollout_slot, mp_policy **out);
/* mp_policy_load_ordered with `rows` distribution rows: rows = S (one per state of the model), or rows = the states of ONE MDP
 * of a batch model (mp_model_load_table_batch) -- prior / rollout / listed / rollout_slot are then [rows, A] over LOCAL states
 * and serve every MDP of the batch: the planner's own policies (mcts.py:46-97) on a batch of environments that restrict their
 * actions identically (trainer/evaluation.py:139-194 run N at a time).  Deterministic table models with 2..8 actions and at
 * least 16 384 (s, a) pairs -- and every tiled call -- are fused by kernels on the device (ABI 7); results are identical. */
int mp_policy_load_rows(mp_ctx *ctx, mp_model *model, const double *prior, const double *rollout,
                        const uint8_t *listed, const uint8_t *rollout_slot, int32_t rows, mp_policy **out);
int mp_policy_free(mp_policy *policy);
int mp_uct_plan_policy(mp_ctx *ctx, mp_model *model, mp_policy *policy, int32_t n_roots, const void *root_state,
                       const int32_t *root_steps, int32_t episodes, int32_t horizon, double gamma, double temperature,
                       uint64_t *rng_state, int32_t max_plan_len, int32_t *plans, int32_t *plan_len, double *root_value,
                       int64_t *root_child_count, double *root_child_value, int64_t *env_steps, int32_t mem);
/*
 * AbstractPlanner.step_tree with step_strategy "subtree" (abstract.py:172-206): before the next mp_uct_plan, replace
 * the tree of every root left on this ctx by the last mp_uct_plan with the subtree of the root's child `actions[i]`
 * (a root that was never expanded starts a fresh tree, like step_by_reset).  The next mp_uct_plan must have the same
 * n_roots and model; it then continues on the kept statistics instead of starting from empty roots.
 * mp_uct_reset_tree drops the kept trees (step_by_reset, mcts.py:129-130).
 */
int mp_uct_step_tree(mp_ctx *ctx, int32_t n_roots, const int32_t *actions, int32_t mem);
int mp_uct_reset_tree(mp_ctx *ctx);
/* Node capacity per root of the trees currently on this ctx (array size for mp_uct_tree_export). */
int mp_uct_tree_capacity(mp_ctx *ctx, int32_t *cap);
/* Tree of root `root` after the last mp_uct_plan on this ctx, creation order (root = node 0, the
 * children of an expanded node are contiguous: first_child, n_children of them -- |A|, or the number of actions a
 * listed policy lists in the node's state).  Host arrays of capacity `cap` nodes. */
int mp_uct_tree_export(mp_ctx *ctx, int32_t root, int32_t cap, int32_t *n_nodes, int32_t *parent, int32_t *action,
                       int64_t *count, double *value, int32_t *first_child, int32_t *n_children);
/* Visit count of the node reached from root `root` by the action sequence `actions[0..n)` (host array) in the trees of the
 * last mp_uct_plan on this ctx -- what `AbstractPlanner.get_plan` (abstract.py:143-156) needs to know about the LAST
 * node of a closed-loop plan (an unvisited action node has no observation child yet) without exporting the tree.
 * *count = -1 when the path leaves the tree (a node on it is not expanded, or the action is not listed there). */
int mp_uct_path_count(mp_ctx *ctx, int32_t root, const int32_t *actions, int32_t n, int64_t *count);

/*
 * MCTS on STOCHASTIC finite MDPs (the `stochastic` [S,A,S] and `sparse` [S,A,B] modes of a finite-MDP env: models from
 * mp_model_load_dense / mp_model_load_sparse; deterministic table models are accepted too), open or closed loop.
 * The reference's planner steps deep copies of the env (mcts.py:183, tree_search/abstract.py:158-161) and the env
 * samples its next state with its OWN numpy generator, next = rng.choice(n, p=transition[s, a]); a copy carries a copy
 * of that generator (common/factory.py:119-134), so every episode of a plan starts from the same env generator state:
 *   env_rng_state uint64 [n_roots,6]: the env generator's record at plan time, read-only (NULL for deterministic models).
 * closed_loop != 0 (mcts.py:147, MCTSNode.get_child :267-273): an action node has one child per distinct observation
 * (= next state index) seen after it, created on first visit; `plans` then alternates action, observation key, action ...
 * as AbstractPlanner.get_plan walks such a tree (abstract.py:143-156); max_plan_len counts both.
 * prior_p / rollout_p: one distribution over the |A| actions for every state (restricted action sets and per-state
 * policies: mp_uct_plan_stochastic_policy below).  mp_model_set_episode_rules sets what a
 * table model gets at load time: done_on_next (terminated = terminal[s'] instead of terminal[s]) and the TimeLimit.
 * mp_uct_stoch_tree_export: the tree of `root` in creation order -- parent, key (action id or observed state), is_obs,
 * count, value -- a node's children in dict order are its children by ascending index; capacity from
 * mp_uct_stoch_tree_capacity.
 */
int mp_model_set_episode_rules(mp_model *model, int32_t done_on_next, int32_t max_steps);
int mp_uct_plan_stochastic(mp_ctx *ctx, mp_model *model, int32_t n_roots, const int32_t *root_state, const int32_t *root_steps,
                           int32_t episodes, int32_t horizon, double gamma, double temperature, const double *prior_p,
                           const double *rollout_p, int32_t closed_loop, uint64_t *rng_state, const uint64_t *env_rng_state,
                           int32_t max_plan_len, int32_t *plans, int32_t *plan_len, double *root_value,
                           int64_t *root_child_count, double *root_child_value, int64_t *env_steps, int32_t mem);
/* The same plan with PER-STATE policies (restricted action sets, mcts.py:59-97; prior agents, mcts_with_prior.py:47-62) from
 * mp_policy_load / _listed / _ordered on this model -- stochastic, sparse, or a deterministic table (whose per-state policies
 * over more than 8 actions plan here: the loop forms take any number of actions).  A node is expanded with the listed actions and
 * priors of the state the env clone is in at that moment (mcts.py:151-154,237-246) and keeps them -- in open loop later
 * episodes reach it in other states.  mp_uct_step_tree re-uses open-loop trees of either form. */
/*
 * AbstractPlanner.get_visits (abstract.py:163-167): the planner appends the observation of EVERY env step of a plan --
 * descent and rollouts -- to planner.observations (:158-161) and get_visits counts them.  Arms the NEXT
 * mp_uct_plan_stochastic / _policy call of this ctx (host arrays): visits int32 [n_roots, S] receives, per root, how often
 * the plan's env steps landed in each state (= observation of a finite MDP).  Rollouts leave no trace in the tree, so a
 * binding that wants get_visits replays the plan with its saved generator records through this pair
 * (rl_agents_amd/agents/tree_search/mcts.py: a private ctx, so the trees of the real plans stay untouched).
 */
int mp_uct_record_visits(mp_ctx *ctx, int32_t *visits);
int mp_uct_plan_stochastic_policy(mp_ctx *ctx, mp_model *model, mp_policy *policy, int32_t n_roots, const int32_t *root_state,
                                  const int32_t *root_steps, int32_t episodes, int32_t horizon, double gamma, double temperature,
                                  int32_t closed_loop, uint64_t *rng_state, const uint64_t *env_rng_state, int32_t max_plan_len,
                                  int32_t *plans, int32_t *plan_len, double *root_value, int64_t *root_child_count,
                                  double *root_child_value, int64_t *env_steps, int32_t mem);
/* stored child priors (0 for the root and observation nodes) of the tree LAST exported by mp_uct_stoch_tree_export, same order */
int mp_uct_stoch_tree_priors(mp_ctx *ctx, int32_t cap, double *prior);
int mp_uct_stoch_tree_capacity(mp_ctx *ctx, int32_t *cap);
int mp_uct_stoch_tree_export(mp_ctx *ctx, int32_t root, int32_t cap, int32_t *n_nodes, int32_t *parent, int32_t *key,
                             uint8_t *is_obs, int64_t *count, double *value);

/* ---------------------------------------------------------------- OPD ----------------------- */
/*
 * OptimisticDeterministicPlanner.plan (tree_search/deterministic.py:116-122) for n_roots
 * independent roots: budget // A times run() (:106-114: leaf = first maximal upper bound in
 * leaves order :110, DeterministicNode.expand :28-43 with update :45-65, backup_to_root :74-79),
 * then get_plan (abstract.py:143-156) with DeterministicNode.selection_rule (:21-26, random
 * tie-break on exactly equal lower bounds through rng_state).
 *   status int32 [n_roots]: MP_OK or MP_ERR_REWARD_RANGE per root (the reference raises).
 */
int mp_opd_plan(mp_ctx *ctx, mp_model *model, int32_t n_roots, const int32_t *root_state, int32_t budget,
                double gamma, double terminal_reward, uint64_t *rng_state, int32_t max_plan_len, int32_t *plans,
                int32_t *plan_len, double *root_lower, double *root_upper, int64_t *env_steps, int32_t *status,
                int32_t mem);
/* mp_opd_plan on a batch model with one MDP per root (see mp_uct_plan_models): model_index int32 [n_roots], root_state LOCAL. */
int mp_opd_plan_models(mp_ctx *ctx, mp_model *model, int32_t n_roots, const int32_t *model_index, const int32_t *root_state,
                       int32_t budget, double gamma, double terminal_reward, uint64_t *rng_state, int32_t max_plan_len,
                       int32_t *plans, int32_t *plan_len, double *root_lower, double *root_upper, int64_t *env_steps,
                       int32_t *status, int32_t mem);
/* Tree of root `root` after the last mp_opd_plan, creation order (the n_children children of an expanded node are
 * contiguous from first_child: |A|, or the available actions of its state); host arrays, capacity `cap`. */
int mp_opd_tree_export(mp_ctx *ctx, int32_t root, int32_t cap, int32_t *n_nodes, int32_t *parent, int32_t *action,
                       int32_t *state, int32_t *depth, double *reward, double *lower, double *upper, uint8_t *done,
                       int64_t *count, int32_t *first_child, int32_t *n_children);

/* ---------------------------------------------------------------- discrete robust OPD ------- */
/*
 * A joint environment of M models of one decision problem stepped together (agents/robust/robust.py:9-26 JointEnv): the
 * planner's clone is the M state indices, a step returns M rewards and M terminal flags.
 *   transition int64 [M,S,A], reward double [M,S,A], terminal uint8 [M,S] or NULL (each model has its own flags).
 * The model is also a table model (M models) for the other entry points, which use model 0 for tree search.
 */
int mp_model_load_joint(mp_ctx *ctx, int32_t M, int32_t S, int32_t A, const int64_t *transition, const double *reward,
                        const uint8_t *terminal, int32_t done_on_next, mp_model **out);
/*
 * Models that restrict their available actions: JointEnv.get_available_actions (robust.py:22-25) lists, for a joint
 * state, the UNION of what each model's env lists in its own state (a model without get_available_actions lists every
 * action); DeterministicNode.expand (deterministic.py:32-35) creates one child per listed action.
 *   available uint8 [M,S,A], host pointer; every (model, state) needs at least one action.  A tree node has
 *   n_children <= A children (mp_ropd_tree_export), env_steps counts one joint step per real child.
 */
int mp_model_set_available_joint(mp_model *model, const uint8_t *available);
/*
 * DiscreteRobustPlanner.plan (agents/robust/robust.py:28-40 over tree_search/deterministic.py:116-122) for n_roots
 * independent roots of a joint model: budget // A times { leaf = first maximal min_m U among the leaves (robust.py:37,
 * RobustNode.get_value_upper_bound :45-46); DeterministicNode.expand (deterministic.py:28-43) with the ndarray branch of
 * update (:45-65): per-model lower / upper bounds; backup_to_root (:74-79) on the minima over the models }, then get_plan
 * with selection_rule (:21-26) on min_m L.  Rewards outside [0,1] in ANY model: status MP_ERR_REWARD_RANGE.
 *   root_state int32 [n_roots,M] (usually the same state M times); root_lower / root_upper = min over the models of
 *   the root's bounds; everything else as mp_opd_plan.  Any number of models (the reference steps a list, robust.py:9-16).
 */
int mp_ropd_plan(mp_ctx *ctx, mp_model *model, int32_t n_roots, const int32_t *root_state, int32_t budget, double gamma,
                 double terminal_reward, uint64_t *rng_state, int32_t max_plan_len, int32_t *plans, int32_t *plan_len,
                 double *root_lower, double *root_upper, int64_t *env_steps, int32_t *status, int32_t mem);
/* Tree of root `root` after the last mp_ropd_plan, creation order (A children per expanded node); host arrays of capacity
 * `cap` nodes; state / reward / lower / upper / done are [cap,M]: a leaf's per-model values, an expanded node's
 * backed-up scalars repeated M times; n_children: children per node (contiguous from first_child; < A with restricted
 * action sets). */
int mp_ropd_tree_export(mp_ctx *ctx, int32_t root, int32_t cap, int32_t *n_nodes, int32_t *parent, int32_t *action,
                        int32_t *state, int32_t *depth, double *reward, double *lower, double *upper, uint8_t *done,
                        int64_t *count, int32_t *first_child, int32_t *n_children);

/* ---------------------------------------------------------------- state-aware OPD ----------- */
/*
 * StateAwarePlanner / StateAwareNode (tree_search/state_aware.py:10-137) over DeterministicNode.expand/update
 * (deterministic.py:28-65): optimistic planning that aggregates the tree nodes observed in the same state.
 * An mp_saopd holds, for n_planners independent planners of one table model, what a reference planner OBJECT holds
 * across plan() calls: state_values and state_nodes (dictionaries keyed by str(observation) = the state index, :81-83)
 * and every node ever created -- reset() (deterministic.py:102-104) only installs a new root and leaves list, so nodes
 * of earlier trees keep taking part in prune() (:28-40) and backup_to_root() (:42-63) through state_nodes.
 * mp_saopd_plan = step_by_reset + StateAwarePlanner.plan (:117-127) for every planner:
 *   root_state int32 [n]; budget // |A| iterations of run() (:93-107); accuracy / backup_aggregated_nodes /
 *   prune_suboptimal_leaves = the config of default_config (:85-91); gamma must not change between plans.
 *   rng_state uint64 [n,6]: get_plan's random tie-breaks (deterministic.py:21-26), drawn twice per plan as the
 *   reference does (deterministic.py:122 + state_aware.py:127).
 *   plans int32 [n,max_plan_len] (-1 padded), plan_len int32 [n], env_steps / updates int64 [n] (planner.step calls /
 *   Bellman backups of this plan), status int32 [n]: MP_OK, MP_ERR_REWARD_RANGE (ValueError, deterministic.py:46-47),
 *   MP_ERR_ARG (every leaf pruned: the reference's max() of an empty list, :95) or MP_ERR_ALLOC (backup queue full and
 *   no room to grow it: a full queue normally rolls the plan back and runs it again with a larger one).
 *   mem = MP_MEM_HOST: the call synchronises (it returns host arrays), reads the overflow flag and retries by itself.
 *   mem = MP_MEM_DEVICE: ASYNCHRONOUS, one launch and nothing read back; a planner whose queue fills up then reports
 *   MP_ERR_ALLOC in `status`, cannot be rolled back, and stays failed: every later call reports MP_ERR_ALLOC for it again
 *   (the other planners of the batch are unaffected).  The same holds for the planners left full when a host-mode call
 *   runs out of room to grow the queue.  A planner whose leaves were ALL pruned (the reference raises ValueError from
 *   max() there, state_aware.py:95) reports MP_ERR_ARG and, in the asynchronous mode -- where the caller goes on stepping
 *   the batch -- stays failed the same way.
 * The model must outlive the planners (they read its transition records).
 * mp_saopd_export: arena of one planner in creation order (node rows [root, n_nodes) are the current tree; `alive` =
 * "in planner.leaves"), arrays of capacity >= n_nodes (mp_saopd_info), and state_values double [S].
 * Restricted action sets (mp_model_set_available on the model; deterministic.py:32-35): an expansion still takes |A|
 * consecutive rows, the rows of unlisted actions are PHANTOMS -- lower = -inf, never alive, in no state's list, count 0 --
 * that a caller drops (rl_agents_amd/native.py StateAwarePlanners.export does, and renumbers); env_steps counts the
 * listed actions only.
 */
typedef struct mp_saopd mp_saopd;
int mp_saopd_create(mp_ctx *ctx, mp_model *model, int32_t n_planners, mp_saopd **out);
int mp_saopd_free(mp_saopd *planners);
int mp_saopd_plan(mp_ctx *ctx, mp_saopd *planners, const int32_t *root_state, int32_t budget, double gamma,
                  double terminal_reward, double accuracy, int32_t backup_aggregated_nodes,
                  int32_t prune_suboptimal_leaves, uint64_t *rng_state, int32_t max_plan_len, int32_t *plans,
                  int32_t *plan_len, int64_t *env_steps, int64_t *updates, int32_t *status, int32_t mem);
int mp_saopd_info(mp_saopd *planners, int32_t *n_planners, int32_t *n_nodes, int32_t *root, int32_t *n_states);
int mp_saopd_export(mp_saopd *planners, int32_t planner, int32_t cap, int32_t *parent, int32_t *action, int32_t *state,
                    int32_t *depth, double *reward, double *lower, uint8_t *done, int64_t *count, int32_t *first_child,
                    uint8_t *alive, double *state_values);

/* ---------------------------------------------------------------- batched evaluation -------- */
/*
 * The env side of Evaluation.step (trainer/evaluation.py:164-190) for n lock-step episodes of ONE deterministic table
 * model, on the device: for every episode i still alive, act = plans[i * plan_stride] (an empty plan, -1, steps label 0
 * and is LOGGED as -1: Evaluation.step raises "The agent did not plan any action" there, evaluation.py:168-170, and so
 * does the caller of this loop),
 *   reward = R[s, act];  done = terminal[s] (or terminal[s'] with done_on_next);  s <- T[s, act];  steps += 1;
 *   returns[i] += reward;  discounted[i] += reward * gpow[steps before];  actions_log[i * log_stride + steps before] = act;
 *   alive[i] = !(done || steps >= max_steps).
 * state / steps int32 [n], alive uint8 [n], returns / discounted double [n] (all in / out); gpow double [max_steps]
 * (gamma ** t as the host computes it); n_alive int32 [1] = episodes still alive after the step.  Episodes that are not
 * alive are left untouched.  The planners' root-state buffers ARE `state` / `steps`: no host round trip per step.
 * mp_greedy_actions: plans[i * plan_stride] = argmax_a Q[state[i], a] (first maximum: ValueIterationAgent.act,
 * value_iteration.py:35) -- the "plan" of a value-iteration agent for the same loop.
 */
int mp_env_step(mp_ctx *ctx, mp_model *model, int32_t n, int32_t *state, int32_t *steps, uint8_t *alive,
                const int32_t *plans, int32_t plan_stride, int32_t max_steps, const double *gpow, double *returns,
                double *discounted, int32_t *actions_log, int32_t log_stride, int32_t *n_alive, int32_t mem);
int mp_greedy_actions(mp_ctx *ctx, int32_t n, int32_t S, int32_t A, const double *Q, const int32_t *state, int32_t *plans,
                      int32_t plan_stride, int32_t mem);
/* mp_env_step for STOCHASTIC / sparse models: the next state is sampled from the episode's own env generator record
 * (env_rng uint64 [n, 6], numpy PCG64, advanced in place) exactly as FiniteMDPEnv.step does -- one Generator.random()
 * double, inverse CDF over the row.  The planner reads the same records (mp_uct_plan_stochastic's env_rng_state): every
 * plan's clones start from the env generator as it is at that step. */
int mp_env_step_stochastic(mp_ctx *ctx, mp_model *model, int32_t n, int32_t *state, int32_t *steps, uint8_t *alive,
                           const int32_t *plans, int32_t plan_stride, int32_t max_steps, const double *gpow, double *returns,
                           double *discounted, int32_t *actions_log, int32_t log_stride, int32_t *n_alive, uint64_t *env_rng,
                           int32_t mem);

/* ---------------------------------------------------------------- result exchange ------------ */
/*
 * The one exchange of the sharded planning path (per-root {plan, value, env_steps} to every rank; the
 * reference has no counterpart -- one process per experiment, scripts/experiments.py:102-106).  Both calls work on DEVICE
 * buffers only and only enqueue (on `stream`, a hipStream_t, or on the ctx stream when NULL): the caller runs ONE
 * all_gather_into_tensor (RCCL) on `packed` between them, no host hop.
 *   mp_pack_rows:   n_arrays (<= 8) per-root arrays src[k] with row widths width[k] bytes (multiples of 4) of this
 *                   rank's n_local roots -> packed uint8 [per][row_bytes], row_bytes = sum(width); rows n_local..per-1
 *                   (the padding that makes every rank's block the same size) are zero-filled.
 *   mp_unpack_rows: gathered uint8 [world][per][row_bytes] -> dst[k] [n_total][width[k]]; global row i comes from
 *                   rank r's block where [lo_r, hi_r) is the balanced contiguous split of n_total over world ranks
 *                   (the first n_total % world ranks hold one row more), i.e. rl_agents_amd.distributed.shard_bounds.
 * src / dst / width are HOST arrays of device pointers / ints.
 */
int mp_pack_rows(mp_ctx *ctx, void *stream, int32_t n_local, int32_t per, int32_t n_arrays, const void *const *src,
                 const int32_t *width, void *packed);
int mp_unpack_rows(mp_ctx *ctx, void *stream, int32_t n_total, int32_t world, int32_t per, int32_t n_arrays,
                   const void *packed, const int32_t *width, void *const *dst);

/*
 * The collective itself, for a consumer that has no torch.distributed (mp_comm_init / mp_gather_results): RCCL,
 * resolved at run time (dlopen of librccl.so.1 -- the copy already in the process if there is one, e.g. PyTorch's -- so the
 * library has no link-time dependency on it).  One communicator per ctx, one rank per GPU.
 *   mp_comm_unique_id: rank 0 creates the 128-byte id (ncclGetUniqueId); the caller hands it to the other ranks out of band
 *                      (a file, a socket, MPI: whatever launched the ranks).
 *   mp_comm_init:      ncclCommInitRank on the ctx's device; collective over all `world` ranks.
 *   mp_gather_results: ncclAllGather of this rank's packed block (per * row_bytes bytes, from mp_pack_rows) into
 *                      gathered [world][per][row_bytes] on `stream` (NULL: the ctx stream); device buffers, only enqueues.
 *                      mp_unpack_rows then spreads the rows into the per-root arrays.
 *   mp_comm_destroy:   ncclCommDestroy (also done by mp_ctx_destroy).
 * rl_agents_amd.distributed.ShardedDevicePlan uses torch.distributed's process group instead (the host side of this
 * package is PyTorch); tests/test_gpu_distributed.py checks that both deliver the same bytes.
 */
#define MP_COMM_UID_BYTES 128
int mp_comm_unique_id(void *uid);
int mp_comm_init(mp_ctx *ctx, int32_t rank, int32_t world, const void *uid);
int mp_gather_results(mp_ctx *ctx, void *stream, int32_t per, int32_t row_bytes, const void *packed, void *gathered);
int mp_comm_destroy(mp_ctx *ctx);

/* ---------------------------------------------------------------- helpers ------------------- */
/* OLOP.allocation (tree_search/olop.py:50-62) with OLOP.horizon (:42-44); host arithmetic. */
int mp_olop_allocation(int32_t budget, double gamma, int32_t *episodes, int32_t *horizon);
/* Timing of the last kernel batch enqueued by a plan / solve call, from HIP events recorded on
 * the ctx stream around the kernel launches only (no copies).  Synchronises the stream. */
int mp_last_kernel_ms(mp_ctx *ctx, double *ms, int32_t *n_launches);
/* Name of the kernel variant the last mp_uct_plan* / mp_vi_solve_batch call launched -- "uct_global", "uct_ldsr" (model resident
 * in LDS), "uct_lds", "uct_quad" (four lanes per root), "uct_lone" (one root per workgroup), "uct_lone_mw" (2 / 4 / 8 roots per workgroup, a
 * wavefront each, around one copy of the transitions), "uct_row_shared" / "uct_row_each" /
 * "uct_lone_each" (four roots per wavefront on DPP rows, trees in LDS: a shared model / one MDP per root; a wavefront per root),
 * "uct_policy", "uct_cartpole", "uct_global_spill"; "vi_batch_reg<own,block>", "vi_batch_cluster2|4|8" (K workgroups per MDP),
 * "vi_batch_wg_stream", "vi_batch_wg_lds", "vi_batch_wg_global": the host picks by model and batch size; reports and tests name
 * what ran.  Results do not depend on the variant. */
const char *mp_last_kernel_variant(mp_ctx *ctx);
/* Hardware self-test (no reference counterpart): LDS atomics of ONE wavefront instruction that hit the same address apply
 * in LANE ORDER on this device -- the state-aware OPD kernel's grouped backup relies on it (one ds_min_rtn_f64 of the group
 * leaders hands every leader the running minimum the reference's turn-by-turn loop would have read).  Runs `waves` wave
 * instructions with random lane masks / addresses / values; *violations = returned values that are not the lane-order
 * prefix minimum + wrong final cells (0 on a conforming device). */
int mp_selftest_lds_atomic_order(mp_ctx *ctx, int32_t waves, int64_t *violations);

#ifdef __cplusplus
}
#endif
#endif /* MI355PLAN_H */
