/*
 * planning_oracle.c -- CPU restatement of the reference planners.  TEST INFRASTRUCTURE ONLY.
 *
 * This file is the oracle the HIP path is checked against.  It may be used only by tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg; the product (rl_agents_amd/) never
 * imports, links or calls it.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks every function below bit-for-bit
 * against tests/golden/{vi,opd,uct,misc}.npz, which were produced by running the unmodified Python reference
 * (eleurent/rl-agents @ /root/reference) through tests/golden/gen/make_golden.py.
 *
 * Each function cites the reference lines (relative to /root/reference) it follows.  It is a
 * plain scalar restatement: linked-list-free arrays, the reference's O(#leaves) leaf scan, the
 * reference's evaluation order for every floating-point expression (Python float semantics =
 * IEEE double, `**` = libm pow, no fused multiply-add: build with -ffp-contract=off).
 *
 * Third-party arithmetic restated because it is not under /root/reference:
 *   - numpy.random.Generator(PCG64): pcg64 XSL-RR 128/64 step+output, next_double =
 *     (next64 >> 11) * 2^-53, next_uint32 buffering of the high half, and the bounded-integer
 *     draw used by Generator.integers/choice (Lemire multiply-shift with rejection on the
 *     buffered 32-bit stream).  Pinned against numpy 2.2.6 draws in tests/golden/misc.npz.
 *   - numpy add.reduce over a contiguous axis (pairwise summation, blocks of 128, 8 lanes, and -- rows longer than
 *     numpy.getbufsize() = 8192 elements -- the running sum over the 8192-element pieces the reduction's iterator cuts
 *     them into), needed for (T * v).sum(-1) in dense/sparse value iteration.  Pinned against numpy 2.2.6 ITSELF (the
 *     expression the reference evaluates) for rows of 8191 .. 50 000 elements in tests/test_oracle_vi_long_rows.py: the
 *     reference's goldens stop at 130 states.
 *   - numpy.allclose (rtol 1e-5, atol 1e-8) early exit.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_OK 0
#define ORC_ERR_REWARD_RANGE (-2) /* deterministic.py:46-47 ValueError */
#define ORC_ERR_ALLOC (-3)
#define ORC_ERR_ARG (-4)

/* ------------------------------------------------------------------ numpy PCG64 ---------- */
typedef struct {
    uint64_t s_hi, s_lo, inc_hi, inc_lo;
    uint64_t has_uint32, uinteger; /* numpy keeps one buffered 32-bit half */
} orc_pcg64;

typedef unsigned __int128 u128;

static inline uint64_t orc_pcg64_next64(orc_pcg64 *g)
{
    const u128 mult = ((u128)0x2360ED051FC65DA4ULL << 64) | 0x4385DF649FCCF645ULL;
    u128 s = ((u128)g->s_hi << 64) | g->s_lo;
    u128 inc = ((u128)g->inc_hi << 64) | g->inc_lo;
    s = s * mult + inc;
    g->s_hi = (uint64_t)(s >> 64);
    g->s_lo = (uint64_t)s;
    uint64_t x = g->s_hi ^ g->s_lo;
    unsigned rot = (unsigned)(g->s_hi >> 58);
    return (x >> rot) | (x << ((-rot) & 63));
}

static inline uint32_t orc_pcg64_next32(orc_pcg64 *g)
{
    if (g->has_uint32) {
        g->has_uint32 = 0;
        return (uint32_t)g->uinteger;
    }
    uint64_t n = orc_pcg64_next64(g);
    g->has_uint32 = 1;
    g->uinteger = (uint32_t)(n >> 32);
    return (uint32_t)(n & 0xffffffffu);
}

static inline double orc_pcg64_double(orc_pcg64 *g)
{
    return (double)(orc_pcg64_next64(g) >> 11) * (1.0 / 9007199254740992.0);
}

/* Generator.integers(0, k) / Generator.choice(arange(k)) for 1 <= k < 2^32:
 * k == 1 draws nothing; otherwise Lemire's method on the 32-bit stream. */
static inline uint32_t orc_pcg64_below(orc_pcg64 *g, uint32_t k)
{
    if (k <= 1) return 0;
    const uint32_t rng_excl = k;
    uint64_t m = (uint64_t)orc_pcg64_next32(g) * rng_excl;
    uint32_t leftover = (uint32_t)m;
    if (leftover < rng_excl) {
        const uint32_t threshold = (uint32_t)(-rng_excl) % rng_excl;
        while (leftover < threshold) {
            m = (uint64_t)orc_pcg64_next32(g) * rng_excl;
            leftover = (uint32_t)m;
        }
    }
    return (uint32_t)(m >> 32);
}

/* test hook: replay a mixed sequence of draws (ops[i] == 0 -> random(), k > 0 -> choice(arange(k))) */
int orc_pcg64_replay(uint64_t *state6, int n, const int32_t *ops, double *outs)
{
    orc_pcg64 g = {state6[0], state6[1], state6[2], state6[3], state6[4], state6[5]};
    for (int i = 0; i < n; ++i)
        outs[i] = ops[i] == 0 ? orc_pcg64_double(&g) : (double)orc_pcg64_below(&g, (uint32_t)ops[i]);
    state6[0] = g.s_hi; state6[1] = g.s_lo; state6[2] = g.inc_hi; state6[3] = g.inc_lo;
    state6[4] = g.has_uint32; state6[5] = g.uinteger;
    return ORC_OK;
}

/* Generator.choice(actions, 1, p=p): idx = searchsorted(cdf, random(), side='right') with
 * cdf = cumsum(p) / cumsum(p)[-1]; the cdf is computed by the caller with numpy itself. */
static inline int orc_cdf_pick(const double *cdf, int n, double u)
{
    int idx = 0;
    while (idx < n && cdf[idx] <= u) ++idx;
    return idx;
}

int orc_pchoice_replay(uint64_t *state6, int n_p, const double *cdf, int n, int32_t *outs)
{
    orc_pcg64 g = {state6[0], state6[1], state6[2], state6[3], state6[4], state6[5]};
    for (int i = 0; i < n; ++i) outs[i] = orc_cdf_pick(cdf, n_p, orc_pcg64_double(&g));
    state6[0] = g.s_hi; state6[1] = g.s_lo; state6[2] = g.inc_hi; state6[3] = g.inc_lo;
    state6[4] = g.has_uint32; state6[5] = g.uinteger;
    return ORC_OK;
}

/* ------------------------------------------------------------------ OLOP.allocation ------ */
/* olop.py:42-44 */
static int orc_olop_horizon(int episodes, double gamma)
{
    int h = (int)ceil(log((double)episodes) / (2.0 * log(1.0 / gamma)));
    return h > 1 ? h : 1;
}

/* olop.py:50-62; returns ORC_ERR_ARG where the reference raises ValueError */
int orc_olop_allocation(int budget, double gamma, int32_t *episodes_out, int32_t *horizon_out)
{
    for (int episodes = 1; episodes < budget; ++episodes) {
        if ((long long)episodes * orc_olop_horizon(episodes, gamma) > budget) {
            int e = episodes - 1 > 1 ? episodes - 1 : 1;
            *episodes_out = e;
            *horizon_out = orc_olop_horizon(e, gamma);
            return ORC_OK;
        }
    }
    return ORC_ERR_ARG;
}

/* ------------------------------------------------------------------ value iteration ------ */
/* numpy pairwise summation of a contiguous double vector (add.reduce inner loop). */
static double orc_pairwise_sum(const double *a, long n)
{
    if (n < 8) {
        double res = 0.0;
        for (long i = 0; i < n; ++i) res += a[i];
        return res;
    } else if (n <= 128) {
        double r[8];
        long i;
        for (int j = 0; j < 8; ++j) r[j] = a[j];
        for (i = 8; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; ++j) r[j] += a[i + j];
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; ++i) res += a[i];
        return res;
    } else {
        long n2 = n / 2;
        n2 -= n2 % 8;
        return orc_pairwise_sum(a, n2) + orc_pairwise_sum(a + n2, n - n2);
    }
}

/*
 * numpy add.reduce of a contiguous double vector as the ufunc machinery runs it: the reduction's iterator hands the inner
 * loop at most `bufsize` = 8192 elements at a time (numpy.getbufsize(), the default the reference runs with) even when
 * nothing needs buffering, so a row longer than 8192 is NOT one pairwise sum: it is the running sum, from the identity
 * 0., of the pairwise sums of its 8192-element pieces.  (Found in round 4 while restating the order on the device:
 * tests/test_oracle_vi_long_rows.py pins this function on numpy itself for rows of 8193 .. 50 000 elements; up to 8192
 * elements -- every golden of the reference -- it is 0. + orc_pairwise_sum, as before.)
 */
#define ORC_NPY_BUFSIZE 8192
static double orc_add_reduce(const double *a, long n)
{
    double res = 0.0;
    for (long off = 0; off < n; off += ORC_NPY_BUFSIZE)
        res += orc_pairwise_sum(a + off, n - off < ORC_NPY_BUFSIZE ? n - off : ORC_NPY_BUFSIZE);
    return res;
}

/* numpy.isclose for one pair (a = old value, b = new value) */
static inline int orc_isclose(double a, double b, double rtol, double atol)
{
    if (isfinite(a) && isfinite(b)) return fabs(a - b) <= atol + rtol * fabs(b);
    return a == b;
}

/*
 * One Bellman backup for M models (M = 1, robust = 0: value_iteration.py:51-63;
 * M >= 1, robust = 1: robust_value_iteration.py:39-58).
 *   mode 0 deterministic: T int64 [M,S,A]
 *   mode 1 stochastic   : P double [M,S,A,S]
 *   mode 2 sparse       : P double [S,A,B], NXT int64 [S,A,B]   (plain VI only)
 * v[S] in, q[S,A] out.  `scratch` holds max(S, B) doubles.
 */
static void orc_bellman(int mode, int M, int S, int A, int B, const int64_t *T, const double *P,
                        const int64_t *NXT, const double *R, const uint8_t *term, int robust,
                        double gamma, const double *v, double *q, double *scratch)
{
    for (int s = 0; s < S; ++s) {
        for (int a = 0; a < A; ++a) {
            double best = 0.0;
            for (int m = 0; m < M; ++m) {
                const long sa = ((long)m * S + s) * A + a;
                double next_v;
                if (mode == 0) {
                    next_v = v[T[sa]];
                } else if (mode == 1) {
                    const double *row = P + sa * (long)S;
                    for (int k = 0; k < S; ++k) scratch[k] = row[k] * v[k];
                    next_v = orc_add_reduce(scratch, S);
                } else {
                    for (int b = 0; b < B; ++b) scratch[b] = P[sa * B + b] * v[NXT[sa * B + b]];
                    next_v = orc_add_reduce(scratch, B);
                }
                if (!robust && term && term[s]) next_v = 0.0; /* value_iteration.py:62 */
                const double qm = R[sa] + gamma * next_v;
                if (m == 0 || qm < best) best = qm;            /* robust_value_iteration.py:46-48 */
            }
            q[(long)s * A + a] = best;
        }
    }
}

/*
 * The stochastic branch of orc_bellman for a BLOCK of source-state rows (value_iteration.py:54-55,62-63;
 * robust_value_iteration.py:46-58): the backup of row s reads only P[m,s,:,:], R[m,s,:], term[s] and the whole v, so
 * a checker can walk a model that does not fit host memory block by block (tests/test_gpu_bench_sizes.py) and the
 * row-sharded solver's ranks can be checked one at a time.  P double [M,rows,A,S_cols], R double [M,rows,A],
 * term uint8 [rows] or NULL, v double [S_cols] in, q double [rows,A] out.
 */
int orc_dense_backup_rows(int M, int rows, int A, int S_cols, const double *P, const double *R, const uint8_t *term,
                          int robust, double gamma, const double *v, double *q)
{
    double *scratch = malloc(((long)S_cols + 8) * sizeof(double));
    if (!scratch) return ORC_ERR_ALLOC;
    for (int s = 0; s < rows; ++s) {
        for (int a = 0; a < A; ++a) {
            double best = 0.0;
            for (int m = 0; m < M; ++m) {
                const long sa = ((long)m * rows + s) * A + a;
                const double *row = P + sa * (long)S_cols;
                for (int k = 0; k < S_cols; ++k) scratch[k] = row[k] * v[k];
                double next_v = orc_add_reduce(scratch, S_cols);
                if (!robust && term && term[s]) next_v = 0.0;
                const double qm = R[sa] + gamma * next_v;
                if (m == 0 || qm < best) best = qm;
            }
            q[(long)s * A + a] = best;
        }
    }
    free(scratch);
    return ORC_OK;
}

/* value_iteration.py:42-45,65-73 (fixed_point_iteration on Q, allclose early exit returns the
 * PREVIOUS iterate) and robust_value_iteration.py:39-44. */
int orc_vi_solve(int mode, int M, int S, int A, int B, const int64_t *T, const double *P,
                 const int64_t *NXT, const double *R, const uint8_t *term, int robust, double gamma,
                 int iterations, double rtol, double atol, double *q_out, int32_t *sweeps_out)
{
    const long n = (long)S * A;
    double *q = calloc(n, sizeof(double)), *qn = malloc(n * sizeof(double));
    double *v = malloc(S * sizeof(double)), *scratch = malloc(((S > B ? S : B) + 8) * sizeof(double));
    if (!q || !qn || !v || !scratch) return ORC_ERR_ALLOC;
    int sweeps = 0;
    for (int it = 0; it < iterations; ++it) {
        for (int s = 0; s < S; ++s) { /* best_action_value: q.max(axis=-1) */
            double m = q[(long)s * A];
            for (int a = 1; a < A; ++a) if (q[(long)s * A + a] > m) m = q[(long)s * A + a];
            v[s] = m;
        }
        orc_bellman(mode, M, S, A, B, T, P, NXT, R, term, robust, gamma, v, qn, scratch);
        ++sweeps;
        int close = 1;
        for (long i = 0; i < n && close; ++i) close = orc_isclose(q[i], qn[i], rtol, atol);
        if (close) break;
        double *t = q; q = qn; qn = t;
    }
    memcpy(q_out, q, n * sizeof(double));
    if (sweeps_out) *sweeps_out = sweeps;
    free(q); free(qn); free(v); free(scratch);
    return ORC_OK;
}

/* value_iteration.py:37-40 get_state_value (V-form iteration); robust_value_iteration.py:32-37 with robust = 1 */
int orc_vi_solve_v(int mode, int M, int S, int A, int B, const int64_t *T, const double *P, const int64_t *NXT,
                   const double *R, const uint8_t *term, int robust, double gamma, int iterations, double rtol,
                   double atol, double *v_out)
{
    const long n = (long)S * A;
    double *q = malloc(n * sizeof(double)), *v = calloc(S, sizeof(double)), *vn = malloc(S * sizeof(double));
    double *scratch = malloc(((S > B ? S : B) + 8) * sizeof(double));
    if (!q || !v || !vn || !scratch) return ORC_ERR_ALLOC;
    for (int it = 0; it < iterations; ++it) {
        /* robust: best_action_value(worst_case(bellman_expectation(v))), robust_value_iteration.py:32-37 */
        orc_bellman(mode, M, S, A, B, T, P, NXT, R, term, robust, gamma, v, q, scratch);
        for (int s = 0; s < S; ++s) {
            double m = q[(long)s * A];
            for (int a = 1; a < A; ++a) if (q[(long)s * A + a] > m) m = q[(long)s * A + a];
            vn[s] = m;
        }
        int close = 1;
        for (int s = 0; s < S && close; ++s) close = orc_isclose(v[s], vn[s], rtol, atol);
        if (close) break;
        double *t = v; v = vn; vn = t;
    }
    memcpy(v_out, v, S * sizeof(double));
    free(q); free(v); free(vn); free(scratch);
    return ORC_OK;
}

/* ------------------------------------------------------------------ table environment ---- */
/* Deterministic finite-MDP env clone: rl_agents_amd/envs/finite_mdp.py restates the absent
 * `finite_mdp` package; a clone (common/factory.py:119-134 safe_deepcopy_env) is {state, steps}. */
typedef struct {
    int S, A;
    const int64_t *T;    /* [S,A] */
    const double *R;     /* [S,A] */
    const uint8_t *term; /* [S]   */
    int done_on_next;    /* 0: terminated = term[s] (default), 1: term[s'] */
    int max_steps;       /* 0 = no truncation */
    /* closed-form CartPole clone (rl_agents_amd/envs/cartpole.py restates gymnasium's CartPole, absent here):
     * cp = {gravity, masscart, masspole, length, force_mag, tau, theta_threshold, x_threshold}, NULL = table env */
    const double *cp;
} orc_env;

/* rl_agents_amd/envs/cartpole.py step(), operation for operation (Python floats, libm sin/cos) */
static inline void orc_cartpole_step(const orc_env *e, double *x4, int32_t *steps, int a, double *reward,
                                     int *terminated, int *truncated)
{
    const double gravity = e->cp[0], masscart = e->cp[1], masspole = e->cp[2], length = e->cp[3];
    const double force_mag = e->cp[4], tau = e->cp[5], theta_thr = e->cp[6], x_thr = e->cp[7];
    const double total_mass = masspole + masscart, polemass_length = masspole * length;
    double x = x4[0], x_dot = x4[1], theta = x4[2], theta_dot = x4[3];
    const double force = a == 1 ? force_mag : -force_mag;
    const double costheta = cos(theta), sintheta = sin(theta);
    const double temp = (force + polemass_length * (theta_dot * theta_dot) * sintheta) / total_mass;
    const double thetaacc = (gravity * sintheta - costheta * temp) /
                            (length * (4.0 / 3.0 - masspole * (costheta * costheta) / total_mass));
    const double xacc = temp - polemass_length * thetaacc * costheta / total_mass;
    x = x + tau * x_dot;
    x_dot = x_dot + tau * xacc;
    theta = theta + tau * theta_dot;
    theta_dot = theta_dot + tau * thetaacc;
    x4[0] = x; x4[1] = x_dot; x4[2] = theta; x4[3] = theta_dot;
    *terminated = x < -x_thr || x > x_thr || theta < -theta_thr || theta > theta_thr;
    *reward = 1.0; /* a clone is never stepped again after its first termination inside a planner */
    *steps += 1;
    *truncated = e->max_steps > 0 && *steps >= e->max_steps;
}

static inline void orc_env_step(const orc_env *e, int32_t *s, int32_t *steps, int a, double *reward,
                                int *terminated, int *truncated)
{
    const long sa = (long)(*s) * e->A + a;
    const int32_t sn = (int32_t)e->T[sa];
    *reward = e->R[sa];
    *terminated = e->done_on_next ? e->term[sn] : e->term[*s];
    *s = sn;
    *steps += 1;
    *truncated = e->max_steps > 0 && *steps >= e->max_steps;
}

/* ------------------------------------------------------------------ OPD ------------------ */
/*
 * deterministic.py:9-122 for one root.  Node arrays are in creation order (root = 0, the children of an expanded
 * leaf are contiguous), capacity 1 + (budget/A)*A.
 * avail: uint8 [S*A] flags of the actions state.get_available_actions() lists in each state (deterministic.py:32-35;
 * NULL = range(action_space.n)): an expansion creates one child per available action, in increasing action order.
 * Outputs may be NULL.  Returns ORC_ERR_REWARD_RANGE where the reference raises ValueError.
 */
int orc_opd_plan(int S, int A, const int64_t *T, const double *R, const uint8_t *term, int done_on_next,
                 int32_t s0, int budget, double gamma, double terminal_reward, uint64_t *rng6,
                 int max_plan_len, int32_t *plan, int32_t *plan_len, double *root_lower, double *root_upper,
                 int64_t *env_steps,
                 /* optional tree export, capacity n_nodes = 1 + (budget/A)*A */
                 int32_t *t_parent, int32_t *t_action, int32_t *t_state, int32_t *t_depth, double *t_reward,
                 double *t_lower, double *t_upper, uint8_t *t_done, int64_t *t_count, int32_t *t_first_child,
                 const uint8_t *avail, int32_t *t_n_children, int32_t *n_nodes_out)
{
    orc_env env = {S, A, T, R, term, done_on_next, 0, NULL};
    const int K = budget / A; /* deterministic.py:118 */
    const int cap = 1 + K * A;
    int32_t *parent = malloc(cap * sizeof(int32_t)), *action = malloc(cap * sizeof(int32_t));
    int32_t *state = malloc(cap * sizeof(int32_t)), *depth = malloc(cap * sizeof(int32_t));
    int32_t *first_child = malloc(cap * sizeof(int32_t)), *leaves = malloc(cap * sizeof(int32_t));
    int32_t *n_children = malloc(cap * sizeof(int32_t));
    double *lower = malloc(cap * sizeof(double)), *upper = malloc(cap * sizeof(double));
    double *reward = malloc(cap * sizeof(double));
    uint8_t *done = malloc(cap);
    int64_t *count = malloc(cap * sizeof(int64_t));
    if (!parent || !action || !state || !depth || !first_child || !leaves || !n_children || !lower || !upper || !reward ||
        !done || !count)
        return ORC_ERR_ALLOC;
    int rc = ORC_OK;
    int64_t steps_taken = 0;
    /* deterministic.py:10-19: root */
    parent[0] = -1; action[0] = -1; state[0] = s0; depth[0] = 0; first_child[0] = -1; n_children[0] = 0;
    lower[0] = 0; upper[0] = 0; reward[0] = 0; done[0] = 0; count[0] = 1;
    int n_nodes = 1, n_leaves = 1;
    leaves[0] = 0;
    for (int k = 0; k < K && rc == ORC_OK; ++k) {
        /* deterministic.py:110: max(self.leaves, key=U) -> first maximal element in list order */
        int li = 0;
        for (int i = 1; i < n_leaves; ++i)
            if (upper[leaves[i]] > upper[leaves[li]]) li = i;
        const int leaf = leaves[li];
        /* deterministic.py:29: leaves.remove(self) keeps the order of the others */
        memmove(leaves + li, leaves + li + 1, (n_leaves - li - 1) * sizeof(int32_t));
        --n_leaves;
        first_child[leaf] = n_nodes;
        for (int a = 0; a < A; ++a) { /* deterministic.py:32-43 */
            if (avail && !avail[(long)state[leaf] * A + a]) continue;
            const int c = n_nodes++;
            ++n_children[leaf];
            parent[c] = leaf; action[c] = a; depth[c] = depth[leaf] + 1; first_child[c] = -1; n_children[c] = 0;
            int32_t s = state[leaf], st = 0;
            double r; int terminated, truncated;
            orc_env_step(&env, &s, &st, a, &r, &terminated, &truncated);
            ++steps_taken;
            state[c] = s;
            leaves[n_leaves++] = c;
            /* deterministic.py:45-65 update() */
            if (!(0 <= r) || !(r <= 1)) { rc = ORC_ERR_REWARD_RANGE; break; }
            const int d = depth[c];
            reward[c] = r; done[c] = (uint8_t)terminated;
            lower[c] = lower[leaf] + pow(gamma, d - 1) * r;
            upper[c] = lower[c] + pow(gamma, d) / (1 - gamma);
            if (terminated) {
                const double nv = lower[c] + terminal_reward * pow(gamma, d) / (1 - gamma);
                lower[c] = nv; upper[c] = nv;
            }
            count[c] = 1;
            for (int n = c; n >= 0; n = parent[n]) count[n] += 1;
        }
        if (rc != ORC_OK) break;
        /* deterministic.py:74-79 backup_to_root */
        for (int n = leaf; n >= 0; n = parent[n]) {
            if (n_children[n] == 0) break; /* `if self.children:` -- unreachable while every state lists an action */
            double ml = lower[first_child[n]], mu = upper[first_child[n]];
            for (int j = 1; j < n_children[n]; ++j) {
                if (lower[first_child[n] + j] > ml) ml = lower[first_child[n] + j];
                if (upper[first_child[n] + j] > mu) mu = upper[first_child[n] + j];
            }
            lower[n] = ml; upper[n] = mu;
        }
    }
    if (rc == ORC_OK) {
        /* abstract.py:143-156 get_plan with deterministic.py:21-26 selection_rule */
        orc_pcg64 g = {rng6[0], rng6[1], rng6[2], rng6[3], rng6[4], rng6[5]};
        int n = 0, len = 0;
        while (n_children[n] > 0) {
            const int fc = first_child[n], kc = n_children[n];
            double m = lower[fc];
            for (int j = 1; j < kc; ++j) if (lower[fc + j] > m) m = lower[fc + j];
            /* Node.random_argmax (abstract.py:304-311): choice over ALL maximal children (any number of them: |A| > 64) */
            int nt = 0, j = 0;
            for (int q = 0; q < kc; ++q) nt += lower[fc + q] == m;
            int pick = (int)orc_pcg64_below(&g, (uint32_t)nt);
            for (int q = 0; q < kc; ++q) if (lower[fc + q] == m && pick-- == 0) { j = q; break; }
            if (plan && len < max_plan_len) plan[len] = action[fc + j];
            ++len;
            n = fc + j;
        }
        if (plan) for (int i = len; i < max_plan_len; ++i) plan[i] = -1;
        if (plan_len) *plan_len = len;
        rng6[0] = g.s_hi; rng6[1] = g.s_lo; rng6[2] = g.inc_hi; rng6[3] = g.inc_lo;
        rng6[4] = g.has_uint32; rng6[5] = g.uinteger;
        if (root_lower) *root_lower = lower[0];
        if (root_upper) *root_upper = upper[0];
    }
    if (env_steps) *env_steps = steps_taken;
    for (int i = 0; i < n_nodes; ++i) {
        if (t_parent) t_parent[i] = parent[i];
        if (t_action) t_action[i] = action[i];
        if (t_state) t_state[i] = state[i];
        if (t_depth) t_depth[i] = depth[i];
        if (t_reward) t_reward[i] = reward[i];
        if (t_lower) t_lower[i] = lower[i];
        if (t_upper) t_upper[i] = upper[i];
        if (t_done) t_done[i] = done[i];
        if (t_count) t_count[i] = count[i];
        if (t_first_child) t_first_child[i] = n_children[i] > 0 ? first_child[i] : -1;
        if (t_n_children) t_n_children[i] = n_children[i];
    }
    if (n_nodes_out) *n_nodes_out = n_nodes;
    free(parent); free(action); free(state); free(depth); free(first_child); free(leaves); free(n_children);
    free(lower); free(upper); free(reward); free(done); free(count);
    return rc;
}

/* ------------------------------------------------------------------ discrete robust OPD ---- */
/*
 * agents/robust/robust.py:28-50 (DiscreteRobustPlanner / RobustNode) over deterministic.py:28-79, for one root.
 * The planner's `state` is a joint environment of M models stepped together (robust.py:9-16 JointEnv): a joint state
 * is M state indices, a step returns M rewards and M terminal flags (ndarrays), so DeterministicNode.update takes its
 * ndarray branch (deterministic.py:54-59) and a LEAF's value_lower / value_upper are vectors over the models.
 * RobustNode reads them through np.min (robust.py:42-49): the leaf to expand is the first maximal min_m U (robust.py:37),
 * backup_to_root (deterministic.py:74-79) replaces an expanded node's bounds by the SCALARS max_c min_m L_c / max_c
 * min_m U_c, and the plan follows max min_m L with random ties (deterministic.py:21-26).
 *   T int64 [M,S,A], R double [M,S,A], term uint8 [M,S] (NULL = none), s0 int32 [M].
 * Tree export (capacity 1 + (budget/A)*A): t_lower / t_upper [cap*M] hold a node's vector (an expanded node's scalar
 * repeated M times), t_state / t_reward / t_done [cap*M] the joint observation, rewards and flags.
 */
int orc_ropd_plan(int M, int S, int A, const int64_t *T, const double *R, const uint8_t *term, int done_on_next,
                  const int32_t *s0, int budget, double gamma, double terminal_reward, uint64_t *rng6,
                  int max_plan_len, int32_t *plan, int32_t *plan_len, double *root_lower, double *root_upper,
                  int64_t *env_steps,
                  int32_t *t_parent, int32_t *t_action, int32_t *t_state, int32_t *t_depth, double *t_reward,
                  double *t_lower, double *t_upper, uint8_t *t_done, int64_t *t_count, int32_t *t_first_child,
                  int32_t *n_nodes_out,
                  /* restricted action sets: avail [M,S,A] flags of each model's own get_available_actions(); the joint
                   * environment lists the UNION over the models' current states, ascending (robust.py:22-25); NULL = all.
                   * t_n_children: children per node (they are contiguous from first_child, keyed by t_action). */
                  const uint8_t *avail, int32_t *t_n_children)
{
    const int K = budget / A; /* deterministic.py:118: budget // state.action_space.n */
    const int cap = 1 + K * A;
    int32_t *parent = malloc(cap * sizeof(int32_t)), *action = malloc(cap * sizeof(int32_t));
    int32_t *state = malloc((size_t)cap * M * sizeof(int32_t)), *depth = malloc(cap * sizeof(int32_t));
    int32_t *first_child = malloc(cap * sizeof(int32_t)), *leaves = malloc(cap * sizeof(int32_t));
    int32_t *n_children = calloc(cap, sizeof(int32_t));
    double *lower = malloc((size_t)cap * M * sizeof(double)), *upper = malloc((size_t)cap * M * sizeof(double));
    double *reward = calloc((size_t)cap * M, sizeof(double));
    uint8_t *done = calloc((size_t)cap * M, 1);
    int64_t *count = malloc(cap * sizeof(int64_t));
    if (!parent || !action || !state || !depth || !first_child || !leaves || !lower || !upper || !reward || !done || !count)
        return ORC_ERR_ALLOC;
#define VMIN(v, n) ({ double m_ = (v)[(long)(n) * M]; for (int q_ = 1; q_ < M; ++q_) if ((v)[(long)(n) * M + q_] < m_) m_ = (v)[(long)(n) * M + q_]; m_; })
    int rc = ORC_OK;
    int64_t steps_taken = 0;
    parent[0] = -1; action[0] = -1; depth[0] = 0; first_child[0] = -1; count[0] = 1;
    for (int m = 0; m < M; ++m) { state[m] = s0[m]; lower[m] = 0; upper[m] = 0; }
    int n_nodes = 1, n_leaves = 1;
    leaves[0] = 0;
    for (int k = 0; k < K && rc == ORC_OK; ++k) {
        /* robust.py:37: max(self.leaves, key=np.min(value_upper)) -> first maximal element in list order */
        int li = 0;
        double best = VMIN(upper, leaves[0]);
        for (int i = 1; i < n_leaves; ++i) {
            const double u = VMIN(upper, leaves[i]);
            if (u > best) { best = u; li = i; }
        }
        const int leaf = leaves[li];
        memmove(leaves + li, leaves + li + 1, (n_leaves - li - 1) * sizeof(int32_t));
        --n_leaves;
        first_child[leaf] = n_nodes;
        for (int a = 0; a < A && rc == ORC_OK; ++a) { /* deterministic.py:32-43 over state.get_available_actions() */
            if (avail) {
                int listed = 0;
                for (int m = 0; m < M; ++m) listed |= avail[((long)m * S + state[(long)leaf * M + m]) * A + a] != 0;
                if (!listed) continue;
            }
            const int c = n_nodes++;
            n_children[leaf] += 1;
            parent[c] = leaf; action[c] = a; depth[c] = depth[leaf] + 1; first_child[c] = -1;
            const int d = depth[c];
            ++steps_taken; /* one planner.step (of the joint environment) per child */
            leaves[n_leaves++] = c;
            for (int m = 0; m < M; ++m) { /* JointEnv.step: every model steps its own state */
                const int32_t sm = state[(long)leaf * M + m];
                const long sa = ((long)m * S + sm) * A + a;
                const int32_t sn = (int32_t)T[sa];
                const double r = R[sa];
                const int terminated = term ? (done_on_next ? term[(long)m * S + sn] : term[(long)m * S + sm]) : 0;
                state[(long)c * M + m] = sn;
                reward[(long)c * M + m] = r; done[(long)c * M + m] = (uint8_t)terminated;
                if (!(0 <= r) || !(r <= 1)) rc = ORC_ERR_REWARD_RANGE; /* np.all(0 <= reward), np.all(reward <= 1) */
                /* deterministic.py:51-59 with ndarray reward / done */
                double lo = lower[(long)leaf * M + m] + pow(gamma, d - 1) * r;
                double up = lo + pow(gamma, d) / (1 - gamma);
                if (terminated) {
                    const double nv = lo + terminal_reward * pow(gamma, d) / (1 - gamma);
                    lo = nv; up = nv;
                }
                lower[(long)c * M + m] = lo; upper[(long)c * M + m] = up;
            }
            if (rc != ORC_OK) break;
            count[c] = 1;
            for (int n = c; n >= 0; n = parent[n]) count[n] += 1;
        }
        if (rc != ORC_OK) break;
        /* deterministic.py:74-79 backup_to_root with RobustNode.get_value_*_bound = np.min (robust.py:42-46) */
        for (int n = leaf; n >= 0; n = parent[n]) {
            double ml = VMIN(lower, first_child[n]), mu = VMIN(upper, first_child[n]);
            for (int a = 1; a < n_children[n]; ++a) {
                const double l = VMIN(lower, first_child[n] + a), u = VMIN(upper, first_child[n] + a);
                if (l > ml) ml = l;
                if (u > mu) mu = u;
            }
            for (int m = 0; m < M; ++m) { lower[(long)n * M + m] = ml; upper[(long)n * M + m] = mu; }
        }
    }
    if (rc == ORC_OK) {
        orc_pcg64 g = {rng6[0], rng6[1], rng6[2], rng6[3], rng6[4], rng6[5]};
        int n = 0, len = 0;
        while (first_child[n] >= 0) {
            const int fc = first_child[n];
            double m = VMIN(lower, fc);
            for (int a = 1; a < n_children[n]; ++a) { const double l = VMIN(lower, fc + a); if (l > m) m = l; }
            int nt = 0, a = 0; /* choice over all maximal children, however many */
            for (int q = 0; q < n_children[n]; ++q) nt += VMIN(lower, fc + q) == m;
            int pick = (int)orc_pcg64_below(&g, (uint32_t)nt);
            for (int q = 0; q < n_children[n]; ++q) if (VMIN(lower, fc + q) == m && pick-- == 0) { a = q; break; }
            if (plan && len < max_plan_len) plan[len] = action[fc + a];
            ++len;
            n = fc + a;
        }
        if (plan) for (int i = len; i < max_plan_len; ++i) plan[i] = -1;
        if (plan_len) *plan_len = len;
        rng6[0] = g.s_hi; rng6[1] = g.s_lo; rng6[2] = g.inc_hi; rng6[3] = g.inc_lo;
        rng6[4] = g.has_uint32; rng6[5] = g.uinteger;
        if (root_lower) *root_lower = VMIN(lower, 0);
        if (root_upper) *root_upper = VMIN(upper, 0);
    }
#undef VMIN
    if (env_steps) *env_steps = steps_taken;
    for (int i = 0; i < n_nodes; ++i) {
        if (t_parent) t_parent[i] = parent[i];
        if (t_action) t_action[i] = action[i];
        if (t_depth) t_depth[i] = depth[i];
        if (t_count) t_count[i] = count[i];
        if (t_first_child) t_first_child[i] = first_child[i];
        if (t_n_children) t_n_children[i] = n_children[i];
        for (int m = 0; m < M; ++m) {
            if (t_state) t_state[(long)i * M + m] = state[(long)i * M + m];
            if (t_reward) t_reward[(long)i * M + m] = reward[(long)i * M + m];
            if (t_lower) t_lower[(long)i * M + m] = lower[(long)i * M + m];
            if (t_upper) t_upper[(long)i * M + m] = upper[(long)i * M + m];
            if (t_done) t_done[(long)i * M + m] = done[(long)i * M + m];
        }
    }
    if (n_nodes_out) *n_nodes_out = n_nodes;
    free(parent); free(action); free(state); free(depth); free(first_child); free(leaves); free(n_children);
    free(lower); free(upper); free(reward); free(done); free(count);
    return rc;
}

int orc_ropd_plan_batch(int M, int S, int A, const int64_t *T, const double *R, const uint8_t *term, int done_on_next,
                        int n_roots, const int32_t *s0 /* [n_roots,M] */, int budget, double gamma, double terminal_reward,
                        uint64_t *rng6, int max_plan_len, int32_t *plans, int32_t *plan_len, double *root_lower,
                        double *root_upper, int64_t *env_steps, int32_t *status, int n_threads, const uint8_t *avail)
{
#pragma omp parallel for schedule(dynamic, 4) num_threads(n_threads > 0 ? n_threads : 1)
    for (int i = 0; i < n_roots; ++i) {
        int rc = orc_ropd_plan(M, S, A, T, R, term, done_on_next, s0 + (long)i * M, budget, gamma, terminal_reward,
                               rng6 + (long)i * 6, max_plan_len, plans ? plans + (long)i * max_plan_len : NULL,
                               plan_len ? plan_len + i : NULL, root_lower ? root_lower + i : NULL,
                               root_upper ? root_upper + i : NULL, env_steps ? env_steps + i : NULL, NULL, NULL, NULL, NULL,
                               NULL, NULL, NULL, NULL, NULL, NULL, NULL, avail, NULL);
        if (status) status[i] = rc;
    }
    return ORC_OK;
}

/* ------------------------------------------------------------------ UCT ------------------ */
/*
 * mcts.py:100-184 (MCTS planner) + mcts.py:203-286 (MCTSNode) for one root.
 * Policies (mcts.py:46-97, mcts_with_prior.py:47-62), selected by state_policy:
 *   0  prior[a] / rollout_cdf[a]: one distribution over actions 0..A-1 for every state (a table env without
 *      get_available_actions); the cdf is numpy's cumsum(p)/cumsum(p)[-1].
 *   1  prior / rollout_cdf are [S][A] tables indexed by the state the policy is asked about
 *      (MCTSWithPriorPolicyAgent.agent_policy: both policies come from a prior agent's distribution in that state).
 *   2  LISTED policies, what a policy function literally returns for a state that may restrict its actions
 *      (state.get_available_actions(), mcts.py:59-73,88-97, mcts_with_prior.py:56-62): for state s the prior policy
 *      returns the pol_n[s] actions pol_act[s][0..n) with probabilities prior[s][0..n), the rollout policy the
 *      pol_n[S + s] actions pol_act[S*A + s*A + 0..n) with cdf rollout_cdf[s][0..n).  MCTSNode.expand creates one child
 *      per listed action (mcts.py:237-246), so nodes have a variable number of children.
 * A child's prior is stored when its parent is expanded (mcts.py:237-246).
 * closed_loop != 0 (mcts.py:147, MCTSNode.get_child :267-273): after every selection step the node reached is the
 *   child of the action node keyed by str(observation) -- created on first visit with prior 0 -- where the observation
 *   of a finite-MDP env is the index of the state reached.  The env is deterministic, so an action node can only ever
 *   observe one state (checked: ORC_ERR_ARG otherwise).
 * Node arrays are in creation order; the children made by one expansion are contiguous (first_child, n_children).
 * Capacity: 1 + episodes*A nodes (one expansion per episode at most, mcts.py:151-154), twice that with closed_loop.
 */
int orc_uct_plan(int S, int A, const int64_t *T, const double *R, const uint8_t *term, int done_on_next,
                 int max_steps, int32_t s0, int32_t steps0, int episodes, int horizon, double gamma,
                 double temperature, const double *prior, const double *rollout_cdf, uint64_t *rng6,
                 int max_plan_len, int32_t *plan, int32_t *plan_len, int64_t *env_steps,
                 /* optional tree export, capacity as above */
                 int32_t *t_parent, int32_t *t_action, int64_t *t_count, double *t_value,
                 int32_t *t_first_child, int32_t *n_nodes_out,
                 /* CartPole roots: cp = 8 parameters, x0 = root state (4 doubles); NULL, NULL = table env */
                 const double *cp, const double *x0,
                 /* step_strategy "subtree" (abstract.py:195-206): the tree kept from the previous plan, creation
                  * order with contiguous children (n_init = 0: fresh root).  Tree exports then need capacity
                  * n_init + episodes*A. */
                 int n_init, const int64_t *init_count, const double *init_value, const int32_t *init_first_child,
                 /* per-state policies; t_prior / init_prior: the stored child priors (export / kept tree), or NULL */
                 int state_policy, double *t_prior, const double *init_prior,
                 /* listed policies (state_policy == 2), closed loop, and the matching export / kept-tree arrays */
                 const int32_t *pol_n, const int32_t *pol_act, int closed_loop, int32_t *t_n_children,
                 uint8_t *t_is_obs, const int32_t *init_n_children, const int32_t *init_action)
{
    orc_env env = {S, A, T, R, term, done_on_next, max_steps, cp};
    if (state_policy && cp) return ORC_ERR_ARG;
    if (state_policy == 2 && (!pol_n || !pol_act)) return ORC_ERR_ARG;
    if (closed_loop && (cp || n_init > 0)) return ORC_ERR_ARG; /* the reference's subtree step breaks on observation keys */
    const int cap = ((n_init > 0 ? n_init : 1) + episodes * A) * (closed_loop ? 2 : 1);
    int32_t *parent = malloc(cap * sizeof(int32_t)), *action = malloc(cap * sizeof(int32_t));
    int32_t *first_child = malloc(cap * sizeof(int32_t)), *n_children = malloc(cap * sizeof(int32_t));
    uint8_t *is_obs = malloc(cap);
    int64_t *count = malloc(cap * sizeof(int64_t));
    double *value = malloc(cap * sizeof(double)), *gpow = malloc((horizon + 1) * sizeof(double));
    double *nprior = malloc(cap * sizeof(double));
    double *score = malloc((A > 0 ? A : 1) * sizeof(double));
    int *ties = malloc((A > 0 ? A : 1) * sizeof(int));
    if (!parent || !action || !first_child || !n_children || !is_obs || !count || !value || !gpow || !nprior || !score || !ties)
        return ORC_ERR_ALLOC;
    for (int h = 0; h <= horizon; ++h) gpow[h] = pow(gamma, h); /* Python: gamma ** h */
    orc_pcg64 g = {rng6[0], rng6[1], rng6[2], rng6[3], rng6[4], rng6[5]};
    /* mcts.py:129-130 reset(): fresh root (value 0, count 0, prior 1) */
    parent[0] = -1; action[0] = -1; first_child[0] = -1; n_children[0] = 0; is_obs[0] = 0; count[0] = 0; value[0] = 0; nprior[0] = 1;
    int n_nodes = 1;
    int rc = ORC_OK;
    if (n_init > 0) { /* continue on the re-rooted tree (the new root keeps the prior it had as a child) */
        if (init_prior) nprior[0] = init_prior[0];
        for (int i = 0; i < n_init; ++i) {
            count[i] = init_count[i]; value[i] = init_value[i]; first_child[i] = init_first_child[i]; is_obs[i] = 0;
            n_children[i] = first_child[i] < 0 ? 0 : (init_n_children ? init_n_children[i] : A);
            for (int j = 0; j < n_children[i]; ++j) {
                const int c = first_child[i] + j;
                parent[c] = i; action[c] = init_action ? init_action[c] : j;
                nprior[c] = init_prior ? init_prior[c] : prior[action[c]];
            }
        }
        n_nodes = n_init;
    }
    int64_t steps_taken = 0;
    for (int ep = 0; ep < episodes && rc == ORC_OK; ++ep) { /* mcts.py:179-184 */
        int32_t s = s0, st = steps0;     /* safe_deepcopy_env(state) */
        double x4[4] = {0, 0, 0, 0};
        if (cp) memcpy(x4, x0, sizeof(x4));
        int node = 0, depth = 0, terminal = 0, truncated = 0;
        double total_reward = 0;
        /* mcts.py:143-149 selection */
        while (depth < horizon && n_children[node] > 0 && !terminal) {
            const int fc = first_child[node], k = n_children[node];
            /* mcts.py:275-286: value + temperature * len(parent.children) * prior / (count + 1) */
            double m = 0;
            for (int j = 0; j < k; ++j) {
                score[j] = value[fc + j] + temperature * k * nprior[fc + j] / (double)(count[fc + j] + 1);
                if (j == 0 || score[j] > m) m = score[j];
            }
            int nt = 0;
            for (int j = 0; j < k; ++j) if (score[j] == m) ties[nt++] = j; /* abstract.py:296-311 */
            const int j = ties[orc_pcg64_below(&g, (uint32_t)nt)];
            const int a = action[fc + j];
            double r;
            if (cp) orc_cartpole_step(&env, x4, &st, a, &r, &terminal, &truncated);
            else orc_env_step(&env, &s, &st, a, &r, &terminal, &truncated);
            ++steps_taken;
            total_reward += gpow[depth] * r;
            node = fc + j;
            if (closed_loop) { /* MCTSNode.get_child(action, observation), mcts.py:267-273 */
                if (n_children[node] == 0) {
                    const int c = n_nodes++;
                    parent[c] = node; action[c] = s; is_obs[c] = 1; first_child[c] = -1; n_children[c] = 0;
                    count[c] = 0; value[c] = 0; nprior[c] = 0;
                    first_child[node] = c; n_children[node] = 1;
                } else if (action[first_child[node]] != s) {
                    rc = ORC_ERR_ARG;
                    break;
                }
                node = first_child[node];
            }
            ++depth;
        }
        if (rc != ORC_OK) break;
        /* mcts.py:151-154 expansion: one child per action the prior policy lists for this state */
        if (n_children[node] == 0 && depth < horizon && (!terminal || node == 0)) {
            const int k = state_policy == 2 ? pol_n[s] : A;
            first_child[node] = n_nodes; n_children[node] = k;
            for (int j = 0; j < k; ++j) {
                const int c = n_nodes++;
                const int a = state_policy == 2 ? pol_act[(long)s * A + j] : j;
                parent[c] = node; action[c] = a; first_child[c] = -1; n_children[c] = 0; is_obs[c] = 0; count[c] = 0; value[c] = 0;
                nprior[c] = state_policy ? prior[(long)s * A + j] : prior[j]; /* prior_policy(state, observation) */
            }
        }
        /* mcts.py:156-157,160-177 rollout */
        if (!terminal) {
            for (int h = depth; h < horizon; ++h) {
                const double u = orc_pcg64_double(&g);
                int a;
                if (state_policy == 2)
                    a = pol_act[(long)S * A + (long)s * A + orc_cdf_pick(rollout_cdf + (long)s * A, pol_n[S + s], u)];
                else
                    a = orc_cdf_pick(state_policy ? rollout_cdf + (long)s * A : rollout_cdf, A, u);
                double r; int term_h, trunc_h;
                if (cp) orc_cartpole_step(&env, x4, &st, a, &r, &term_h, &trunc_h);
                else orc_env_step(&env, &s, &st, a, &r, &term_h, &trunc_h);
                ++steps_taken;
                total_reward += gpow[h] * r;
                if (term_h || trunc_h) break;
            }
        }
        /* mcts.py:248-265 update_branch */
        for (int n = node; n >= 0; n = parent[n]) {
            count[n] += 1;
            value[n] += 1.0 / (double)count[n] * (total_reward - value[n]);
        }
    }
    /* abstract.py:143-156 get_plan with mcts.py:212-218 selection_rule; under an action node of a closed-loop tree the
     * "action" is the observation key of its single child */
    int n = 0, len = 0;
    while (rc == ORC_OK && n_children[n] > 0) {
        const int fc = first_child[n], k = n_children[n];
        int64_t mc = count[fc];
        for (int j = 1; j < k; ++j) if (count[fc + j] > mc) mc = count[fc + j];
        int best = -1;
        for (int j = 0; j < k; ++j)
            if (count[fc + j] == mc && (best < 0 || value[fc + j] > value[fc + best])) best = j;
        if (plan && len < max_plan_len) plan[len] = action[fc + best];
        ++len;
        n = fc + best;
    }
    if (plan) for (int i = len; i < max_plan_len; ++i) plan[i] = -1;
    if (plan_len) *plan_len = len;
    if (env_steps) *env_steps = steps_taken;
    rng6[0] = g.s_hi; rng6[1] = g.s_lo; rng6[2] = g.inc_hi; rng6[3] = g.inc_lo;
    rng6[4] = g.has_uint32; rng6[5] = g.uinteger;
    for (int i = 0; i < n_nodes; ++i) {
        if (t_parent) t_parent[i] = parent[i];
        if (t_action) t_action[i] = action[i];
        if (t_count) t_count[i] = count[i];
        if (t_value) t_value[i] = value[i];
        if (t_first_child) t_first_child[i] = first_child[i];
        if (t_prior) t_prior[i] = nprior[i];
        if (t_n_children) t_n_children[i] = n_children[i];
        if (t_is_obs) t_is_obs[i] = is_obs[i];
    }
    if (n_nodes_out) *n_nodes_out = n_nodes;
    free(parent); free(action); free(first_child); free(n_children); free(is_obs); free(count); free(value); free(gpow);
    free(nprior); free(score); free(ties);
    return rc;
}

/*
 * MCTS on a STOCHASTIC finite MDP (mcts.py:132-184 over a FiniteMDPEnv in `stochastic` / `sparse` mode, restated in
 * rl_agents_amd/envs/finite_mdp.py: next state = rng.choice(n, p=row) with the ENV's generator), open or closed loop.
 * What differs from orc_uct_plan:
 *   - every episode steps a deep copy of the env (mcts.py:183 safe_deepcopy_env), and the copy includes the env's own
 *     numpy generator: each episode starts from the SAME env generator state env_rng6 (never advanced for the caller);
 *   - closed loop (mcts.py:147, get_child :267-273): an action node has one child per DISTINCT observation seen after it
 *     (key = the next state index, prior 0), created on first visit, in first-visit order.
 * mode: 0 deterministic T [S,A]; 1 dense P [S,A,S]; 2 sparse P [S,A,B] + NXT [S,A,B].  Policies: one distribution over
 * actions 0..A-1 for every state (prior [A], rollout_cdf [A]).
 * Tree arrays in creation order, capacity `cap` >= 1 + episodes * (A + horizon): parent, key (action or observation),
 * is_obs, count, value, prior.  A node's children in dict order = its children by ascending id.
 */
int orc_uct_plan_stoch(int mode, int S, int A, int B, const int64_t *T, const double *P, const int64_t *NXT,
                       const double *R, const uint8_t *term, int done_on_next, int max_steps, int32_t s0, int32_t steps0,
                       int episodes, int horizon, double gamma, double temperature, const double *prior,
                       const double *rollout_cdf, int closed_loop, uint64_t *rng6, const uint64_t *env_rng6,
                       int max_plan_len, int32_t *plan, int32_t *plan_len, int64_t *env_steps, double *root_value,
                       int cap, int32_t *t_parent, int32_t *t_key, uint8_t *t_is_obs, int64_t *t_count, double *t_value,
                       double *t_prior, int32_t *n_nodes_out,
                       /* per-state policies as in orc_uct_plan: 0 = prior [A] / rollout cdf [A]; 1 = tables [S, A]; 2 = listed
                        * (pol_n [2 S]: listed actions per state of the prior, then of the rollout policy; pol_act [2 S A]: their
                        * ids; prior / rollout_cdf [S, A] by list position).  A node is expanded with the list of the state the
                        * env is in AT THAT MOMENT (mcts.py:151-154,237-246) and keeps it. */
                       int state_policy, const int32_t *pol_n, const int32_t *pol_act)
{
    if (cap < 1 + episodes * (A + horizon)) return ORC_ERR_ARG;
    if (state_policy == 2 && (!pol_n || !pol_act)) return ORC_ERR_ARG;
    const int W = mode == 1 ? S : B;
    int32_t *parent = malloc(cap * sizeof(int32_t)), *key = malloc(cap * sizeof(int32_t));
    int32_t *first = malloc(cap * sizeof(int32_t)), *last = malloc(cap * sizeof(int32_t)), *next = malloc(cap * sizeof(int32_t));
    int32_t *n_ch = malloc(cap * sizeof(int32_t));
    uint8_t *is_obs = malloc(cap);
    int64_t *count = malloc(cap * sizeof(int64_t));
    double *value = malloc(cap * sizeof(double)), *nprior = malloc(cap * sizeof(double));
    double *gpow = malloc((horizon + 1) * sizeof(double)), *cdf = malloc(((W > 0 ? W : 1) + 1) * sizeof(double));
    double *score = malloc((A > 0 ? A : 1) * sizeof(double));
    int *ties = malloc((A > 0 ? A : 1) * sizeof(int)), *kids = malloc((A > 0 ? A : 1) * sizeof(int));
    if (!parent || !key || !first || !last || !next || !n_ch || !is_obs || !count || !value || !nprior || !gpow || !cdf || !score ||
        !ties || !kids)
        return ORC_ERR_ALLOC;
    for (int h = 0; h <= horizon; ++h) gpow[h] = pow(gamma, h);
    orc_pcg64 g = {rng6[0], rng6[1], rng6[2], rng6[3], rng6[4], rng6[5]};
    int n_nodes = 0;
#define NEW_NODE(par_, key_, obs_, prior_) ({ const int c_ = n_nodes++; parent[c_] = (par_); key[c_] = (key_); is_obs[c_] = (obs_); \
        count[c_] = 0; value[c_] = 0; nprior[c_] = (prior_); first[c_] = last[c_] = next[c_] = -1; n_ch[c_] = 0; \
        if ((par_) >= 0) { if (last[par_] < 0) first[par_] = c_; else next[last[par_]] = c_; last[par_] = c_; n_ch[par_] += 1; } c_; })
    NEW_NODE(-1, -1, 0, 1.0); /* mcts.py:129-130 reset() */
    int64_t steps_taken = 0;
    for (int ep = 0; ep < episodes; ++ep) {
        int32_t s = s0, st = steps0; /* safe_deepcopy_env(state): state, step counter AND the env's generator */
        orc_pcg64 eg = {env_rng6[0], env_rng6[1], env_rng6[2], env_rng6[3], env_rng6[4], env_rng6[5]};
        int node = 0, depth = 0, terminal = 0;
        double total_reward = 0;
#define ENV_STEP(a_, r_out, term_out, trunc_out) do { \
            const long sa_ = (long)s * A + (a_); int32_t sn_; \
            if (mode == 0) sn_ = (int32_t)T[sa_]; \
            else { /* Generator.choice(n, p=row): cdf = cumsum(p); cdf /= cdf[-1]; searchsorted(cdf, random(), 'right') */ \
                double acc_ = 0; for (int j_ = 0; j_ < W; ++j_) { acc_ += P[sa_ * W + j_]; cdf[j_] = acc_; } \
                for (int j_ = 0; j_ < W; ++j_) cdf[j_] /= acc_; \
                const int idx_ = orc_cdf_pick(cdf, W, orc_pcg64_double(&eg)); \
                sn_ = mode == 1 ? idx_ : (int32_t)NXT[sa_ * W + idx_]; } \
            (r_out) = R[sa_]; (term_out) = term ? (done_on_next ? term[sn_] : term[s]) : 0; \
            s = sn_; st += 1; (trunc_out) = max_steps > 0 && st >= max_steps; ++steps_taken; } while (0)
        while (depth < horizon && n_ch[node] > 0 && !terminal) { /* mcts.py:143-149; node: the root or an observation node
                                                                   * (closed loop) / an action node (open loop) */
            const int k = n_ch[node];
            int j = 0;
            double m = 0;
            for (int c = first[node]; c >= 0; c = next[c], ++j) {
                kids[j] = c;
                score[j] = value[c] + temperature * k * nprior[c] / (double)(count[c] + 1); /* mcts.py:275-286 */
                if (j == 0 || score[j] > m) m = score[j];
            }
            int nt = 0;
            for (j = 0; j < k; ++j) if (score[j] == m) ties[nt++] = j;
            const int child = kids[ties[orc_pcg64_below(&g, (uint32_t)nt)]];
            double r; int trunc;
            ENV_STEP(key[child], r, terminal, trunc);
            (void)trunc;
            total_reward += gpow[depth] * r;
            node = child;
            if (closed_loop) { /* get_child(action, observation): the child keyed by str(observation), made on first visit */
                int o = -1;
                for (int c = first[node]; c >= 0; c = next[c]) if (key[c] == s) { o = c; break; }
                if (o < 0) o = NEW_NODE(node, s, 1, 0.0);
                node = o;
            }
            ++depth;
        }
        if (n_ch[node] == 0 && depth < horizon && (!terminal || node == 0)) { /* mcts.py:151-154 */
            const int k = state_policy == 2 ? pol_n[s] : A;
            for (int j = 0; j < k; ++j)
                NEW_NODE(node, state_policy == 2 ? pol_act[(long)s * A + j] : j, 0, state_policy ? prior[(long)s * A + j] : prior[j]);
        }
        if (!terminal)
            for (int h = depth; h < horizon; ++h) { /* mcts.py:160-177 */
                const double u = orc_pcg64_double(&g);
                int a;
                if (state_policy == 2)
                    a = pol_act[(long)S * A + (long)s * A + orc_cdf_pick(rollout_cdf + (long)s * A, pol_n[S + s], u)];
                else
                    a = orc_cdf_pick(state_policy ? rollout_cdf + (long)s * A : rollout_cdf, A, u);
                double r; int term_h, trunc_h;
                ENV_STEP(a, r, term_h, trunc_h);
                total_reward += gpow[h] * r;
                if (term_h || trunc_h) break;
            }
        for (int n = node; n >= 0; n = parent[n]) { /* mcts.py:248-265 */
            count[n] += 1;
            value[n] += 1.0 / (double)count[n] * (total_reward - value[n]);
        }
        if (n_nodes + A + horizon > cap && ep + 1 < episodes) { /* cannot happen with the documented capacity */ }
    }
#undef ENV_STEP
    /* abstract.py:143-156 get_plan, mcts.py:212-218 selection_rule at every level (action nodes AND observation nodes) */
    int n = 0, len = 0;
    while (n_ch[n] > 0) {
        int64_t mc = -1;
        for (int c = first[n]; c >= 0; c = next[c]) if (count[c] > mc) mc = count[c];
        int best = -1;
        for (int c = first[n]; c >= 0; c = next[c])
            if (count[c] == mc && (best < 0 || value[c] > value[best])) best = c;
        if (plan && len < max_plan_len) plan[len] = key[best];
        ++len;
        n = best;
    }
    if (plan) for (int i = len; i < max_plan_len; ++i) plan[i] = -1;
    if (plan_len) *plan_len = len;
    if (env_steps) *env_steps = steps_taken;
    if (root_value) *root_value = value[0];
    rng6[0] = g.s_hi; rng6[1] = g.s_lo; rng6[2] = g.inc_hi; rng6[3] = g.inc_lo; rng6[4] = g.has_uint32; rng6[5] = g.uinteger;
    for (int i = 0; i < n_nodes; ++i) {
        if (t_parent) t_parent[i] = parent[i];
        if (t_key) t_key[i] = key[i];
        if (t_is_obs) t_is_obs[i] = is_obs[i];
        if (t_count) t_count[i] = count[i];
        if (t_value) t_value[i] = value[i];
        if (t_prior) t_prior[i] = nprior[i];
    }
    if (n_nodes_out) *n_nodes_out = n_nodes;
#undef NEW_NODE
    free(parent); free(key); free(first); free(last); free(next); free(n_ch); free(is_obs); free(count); free(value);
    free(nprior); free(gpow); free(cdf); free(score); free(ties); free(kids);
    return ORC_OK;
}

int orc_uct_plan_stoch_batch(int mode, int S, int A, int B, const int64_t *T, const double *P, const int64_t *NXT,
                             const double *R, const uint8_t *term, int done_on_next, int max_steps, int n_roots,
                             const int32_t *s0, const int32_t *steps0, int episodes, int horizon, double gamma,
                             double temperature, const double *prior, const double *rollout_cdf, int closed_loop,
                             uint64_t *rng6, const uint64_t *env_rng6, int max_plan_len, int32_t *plans, int32_t *plan_len,
                             int64_t *env_steps, double *root_value, int n_threads, int state_policy, const int32_t *pol_n,
                             const int32_t *pol_act)
{
    const int cap = 1 + episodes * (A + horizon);
    int bad = 0;
#pragma omp parallel for schedule(dynamic, 4) num_threads(n_threads > 0 ? n_threads : 1)
    for (int i = 0; i < n_roots; ++i) {
        int rc = orc_uct_plan_stoch(mode, S, A, B, T, P, NXT, R, term, done_on_next, max_steps, s0[i], steps0 ? steps0[i] : 0,
                                    episodes, horizon, gamma, temperature, prior, rollout_cdf, closed_loop, rng6 + (long)i * 6,
                                    env_rng6 + (long)i * 6, max_plan_len, plans ? plans + (long)i * max_plan_len : NULL,
                                    plan_len ? plan_len + i : NULL, env_steps ? env_steps + i : NULL,
                                    root_value ? root_value + i : NULL, cap, NULL, NULL, NULL, NULL, NULL, NULL, NULL, state_policy,
                                    pol_n, pol_act);
        if (rc != ORC_OK) bad = rc;
    }
    return bad;
}

/*
 * AbstractPlanner.step_by_subtree (abstract.py:195-206): the tree is replaced by the subtree of the root's child
 * `action`; an unexpanded root or child-less choice starts a new tree (n_out = 0 -> caller plans from a fresh root).
 * Nodes are re-numbered breadth-first so that children stay contiguous.  Arrays of capacity n_in.
 * n_children / node_action: per-node child counts and action keys of trees with restricted action sets (NULL: every
 * expanded node has the A children 0..A-1).
 */
int orc_uct_reroot(int A, int n_in, const int64_t *count, const double *value, const int32_t *first_child, int action,
                   int64_t *o_count, double *o_value, int32_t *o_first_child, int32_t *n_out,
                   const double *prior, double *o_prior /* stored child priors, or NULL */,
                   const int32_t *n_children, const int32_t *node_action, int32_t *o_n_children, int32_t *o_action)
{
    *n_out = 0;
    if (n_in < 1 || first_child[0] < 0) return ORC_OK;
    int chosen = -1; /* `if action in self.root.children` */
    const int k0 = n_children ? n_children[0] : A;
    for (int j = 0; j < k0; ++j)
        if ((node_action ? node_action[first_child[0] + j] : j) == action) chosen = first_child[0] + j;
    if (chosen < 0) return ORC_OK;
    int32_t *src = malloc((size_t)n_in * 2 * sizeof(int32_t));
    if (!src) return ORC_ERR_ALLOC;
    int32_t *src_act = src + n_in;
    int head = 0, tail = 1;
    src[0] = chosen; src_act[0] = -1;
    while (head < tail) {
        const int o = src[head];
        o_count[head] = count[o];
        o_value[head] = value[o];
        if (prior && o_prior) o_prior[head] = prior[o];
        if (o_action) o_action[head] = src_act[head];
        const int k = first_child[o] < 0 ? 0 : (n_children ? n_children[o] : A);
        if (o_n_children) o_n_children[head] = k;
        if (k > 0) {
            o_first_child[head] = tail;
            for (int j = 0; j < k; ++j) {
                src[tail + j] = first_child[o] + j;
                src_act[tail + j] = node_action ? node_action[first_child[o] + j] : j;
            }
            tail += k;
        } else {
            o_first_child[head] = -1;
        }
        ++head;
    }
    *n_out = tail;
    free(src);
    return ORC_OK;
}

/*
 * Batch drivers for the cpu_baseline leg of bench.py and for many-root parity checks:
 * independent roots, one RNG state each, OpenMP over roots (the reference's own fan-out is one
 * process per experiment, scripts/experiments.py:105).  Per-root outputs only (no trees).
 */
int orc_uct_plan_batch(int S, int A, const int64_t *T, const double *R, const uint8_t *term, int done_on_next,
                       int max_steps, int n_roots, const int32_t *s0, const int32_t *steps0, int episodes,
                       int horizon, double gamma, double temperature, const double *prior,
                       const double *rollout_cdf, uint64_t *rng6 /* [n_roots,6] */, int max_plan_len,
                       int32_t *plans /* [n_roots,max_plan_len] */, int32_t *plan_len, double *root_value,
                       int64_t *root_child_count /* [n_roots,A] */, double *root_child_value /* [n_roots,A] */,
                       int64_t *env_steps /* [n_roots] */, int n_threads,
                       const double *cp, const double *x0 /* [n_roots,4] or NULL */, int state_policy,
                       const int32_t *pol_n, const int32_t *pol_act)
{
    int rc_all = ORC_OK;
    const int cap = 1 + episodes * A;
#pragma omp parallel for schedule(dynamic, 16) num_threads(n_threads > 0 ? n_threads : 1)
    for (int i = 0; i < n_roots; ++i) {
        int64_t *cnt = malloc(cap * sizeof(int64_t));
        double *val = malloc(cap * sizeof(double));
        int32_t *act = malloc(cap * sizeof(int32_t)), *nch = malloc(cap * sizeof(int32_t));
        int32_t nn = 0;
        int rc = orc_uct_plan(S, A, T, R, term, done_on_next, max_steps, s0 ? s0[i] : 0, steps0 ? steps0[i] : 0, episodes,
                              horizon, gamma, temperature, prior, rollout_cdf, rng6 + (long)i * 6, max_plan_len,
                              plans ? plans + (long)i * max_plan_len : NULL, plan_len ? plan_len + i : NULL,
                              env_steps ? env_steps + i : NULL, NULL, act, cnt, val, NULL, &nn, cp,
                              x0 ? x0 + (long)i * 4 : NULL, 0, NULL, NULL, NULL, state_policy, NULL, NULL,
                              pol_n, pol_act, 0, nch, NULL, NULL, NULL);
        if (rc != ORC_OK) {
#pragma omp critical
            rc_all = rc;
        }
        if (root_value) root_value[i] = val[0];
        for (int a = 0; a < A; ++a) {
            if (root_child_count) root_child_count[(long)i * A + a] = 0;
            if (root_child_value) root_child_value[(long)i * A + a] = 0;
        }
        for (int j = 0; nn > 1 && j < nch[0]; ++j) { /* the root's children are nodes 1 .. n_children(root) */
            if (root_child_count) root_child_count[(long)i * A + act[1 + j]] = cnt[1 + j];
            if (root_child_value) root_child_value[(long)i * A + act[1 + j]] = val[1 + j];
        }
        free(cnt); free(val); free(act); free(nch);
    }
    return rc_all;
}

int orc_opd_plan_batch(int S, int A, const int64_t *T, const double *R, const uint8_t *term, int done_on_next,
                       int n_roots, const int32_t *s0, int budget, double gamma, double terminal_reward,
                       uint64_t *rng6, int max_plan_len, int32_t *plans, int32_t *plan_len, double *root_lower,
                       double *root_upper, int64_t *env_steps, int32_t *status /* [n_roots] */, int n_threads,
                       const uint8_t *avail)
{
#pragma omp parallel for schedule(dynamic, 4) num_threads(n_threads > 0 ? n_threads : 1)
    for (int i = 0; i < n_roots; ++i) {
        int rc = orc_opd_plan(S, A, T, R, term, done_on_next, s0[i], budget, gamma, terminal_reward,
                              rng6 + (long)i * 6, max_plan_len, plans ? plans + (long)i * max_plan_len : NULL,
                              plan_len ? plan_len + i : NULL, root_lower ? root_lower + i : NULL,
                              root_upper ? root_upper + i : NULL, env_steps ? env_steps + i : NULL, NULL, NULL,
                              NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL, avail, NULL, NULL);
        if (status) status[i] = rc;
    }
    return ORC_OK;
}

/* ------------------------------------------------------------------ state-aware OPD -------- */
/*
 * tree_search/state_aware.py:10-137 (StateAwareNode / StateAwarePlanner) for one root, over deterministic.py's
 * expand/update.  Observations of a finite-MDP env are state indices, so the planner's two dictionaries keyed by
 * str(observation) are arrays over states:
 *   sv[s]            state_values (defaultdict of 1/(1-gamma), state_aware.py:83)
 *   head/tail[s] + next_same[node]   state_nodes[s], the list of nodes observed in s, in append order (:20-22)
 * The planner object -- and with it both dictionaries and every node ever created -- outlives plan() calls:
 * reset() (deterministic.py:102-104) only installs a new root and a new leaves list, and plan() (:117-123) only
 * re-initialises the root state's entries.  So the node arrays are an ARENA over all plans of one planner
 * (fresh != 0 starts a new planner); nodes of earlier trees still take part in prune() and backup_to_root()
 * through state_nodes, exactly as in the reference.  alive[n] = "n in planner.leaves".
 * Returns ORC_ERR_ARG where the reference raises (max() of an empty leaves list, arena too small).
 */
int orc_saopd_plan(int S, int A, const int64_t *T, const double *R, const uint8_t *term, int done_on_next,
                   int32_t s0, int budget, double gamma, double terminal_reward, double accuracy,
                   int backup_aggregated_nodes, int prune_suboptimal_leaves, uint64_t *rng6, int max_plan_len,
                   int32_t *plan, int32_t *plan_len, int64_t *env_steps, int64_t *updates_total,
                   /* planner state: arena of capacity cap nodes, n_nodes in/out, root out */
                   int fresh, int cap, int32_t *n_nodes_io, int32_t *root_out, int32_t *parent, int32_t *action,
                   int32_t *state, int32_t *depth, double *reward, double *lower, uint8_t *done, int64_t *count,
                   int32_t *first_child, uint8_t *alive, int32_t *next_same, double *sv, int32_t *head, int32_t *tail,
                   /* restricted action sets (deterministic.py:32-35): avail [S,A] flags = state.get_available_actions(),
                    * NULL = all; n_children [cap] (in / out with the arena): children are contiguous from first_child,
                    * keyed by `action` */
                   const uint8_t *avail, int32_t *n_children)
{
    orc_env env = {S, A, T, R, term, done_on_next, 0, NULL};
    const int K = budget / A; /* deterministic.py:118 */
    const double vmax = 1 / (1 - gamma);
    int n_nodes = fresh ? 0 : *n_nodes_io;
    if (n_nodes + 1 + K * A > cap) return ORC_ERR_ARG;
    if (fresh)
        for (int s = 0; s < S; ++s) { sv[s] = vmax; head[s] = -1; tail[s] = -1; }
    double *gpow = malloc((K + 3) * sizeof(double));
    int qcap = 1024, qh = 0, qt = 0;
    int32_t *queue = malloc(qcap * sizeof(int32_t));
    if (!gpow || !queue) return ORC_ERR_ALLOC;
    for (int d = 0; d < K + 3; ++d) gpow[d] = pow(gamma, d);
#define SA_U(n) (lower[n] + gpow[depth[n]] * sv[state[n]]) /* state_aware.py:65-67 */
    /* reset(): the previous leaves list is dropped, a new root is created */
    for (int i = 0; i < n_nodes; ++i) alive[i] = 0;
    const int root = n_nodes++;
    parent[root] = -1; action[root] = -1; state[root] = s0; depth[root] = 0; reward[root] = 0; lower[root] = 0;
    done[root] = 0; count[root] = 1; first_child[root] = -1; alive[root] = 1; next_same[root] = -1; n_children[root] = 0;
    /* plan(), state_aware.py:117-121 */
    head[s0] = tail[s0] = root;
    sv[s0] = vmax;
    int rc = ORC_OK;
    int64_t steps_taken = 0, updates = 0;
    for (int k = 0; k < K && rc == ORC_OK; ++k) {
        /* run(), :96-107: first maximal U in leaves order (= creation order of the surviving leaves) */
        int leaf = -1;
        double bu = 0;
        for (int i = root; i < n_nodes; ++i)
            if (alive[i]) {
                const double u = SA_U(i);
                if (leaf < 0 || u > bu) { leaf = i; bu = u; }
            }
        if (leaf < 0) { rc = ORC_ERR_ARG; break; } /* max() arg is an empty sequence */
        /* deterministic.py:28-43 expand */
        alive[leaf] = 0;
        first_child[leaf] = n_nodes;
        n_children[leaf] = 0;
        for (int a = 0; a < A; ++a) {
            if (avail && !avail[(long)state[leaf] * A + a]) continue;
            const int c = n_nodes++;
            n_children[leaf] += 1;
            n_children[c] = 0;
            parent[c] = leaf; action[c] = a; depth[c] = depth[leaf] + 1; first_child[c] = -1;
            int32_t s = state[leaf], st = 0;
            double r; int terminated, truncated;
            orc_env_step(&env, &s, &st, a, &r, &terminated, &truncated);
            ++steps_taken;
            state[c] = s;
            alive[c] = 1;
            /* deterministic.py:45-65 update(), called by state_aware.py:15-16 */
            if (!(0 <= r) || !(r <= 1)) { rc = ORC_ERR_REWARD_RANGE; break; }
            const int d = depth[c];
            reward[c] = r; done[c] = (uint8_t)terminated;
            lower[c] = lower[leaf] + gpow[d - 1] * r;
            if (terminated) lower[c] = lower[c] + terminal_reward * gpow[d] / (1 - gamma);
            count[c] = 1;
            for (int n = c; n >= 0; n = parent[n]) count[n] += 1;
            /* state_aware.py:19-22 */
            next_same[c] = -1;
            if (head[s] < 0) head[s] = c; else next_same[tail[s]] = c;
            tail[s] = c;
            /* :24-26 terminal states are worth 0 */
            if (terminated && sv[s] - 0 > 0) sv[s] = 0;
        }
        if (rc != ORC_OK) break;
        /* state_aware.py:42-63 backup_to_root */
        qh = qt = 0;
        queue[qt++] = leaf;
        while (qh < qt) {
            const int node = queue[qh++];
            double delta = 0;
            if (first_child[node] >= 0) {
                int bc = first_child[node];
                double bcu = SA_U(bc);
                for (int a = 1; a < n_children[node]; ++a) {
                    const double u = SA_U(first_child[node] + a);
                    if (u > bcu) { bc = first_child[node] + a; bcu = u; }
                }
                const double backup = reward[bc] + gamma * sv[state[bc]];
                delta = sv[state[node]] - backup; /* update_value, :109-119 */
                if (delta > 0) sv[state[node]] = backup;
                ++updates;
            }
            for (int nb = head[state[node]]; nb >= 0; nb = next_same[nb])
                if (parent[nb] >= 0 && (nb == node || backup_aggregated_nodes) &&
                    delta > accuracy * (1 - gamma) * gpow[depth[nb] - 1]) {
                    if (qt == qcap) {
                        if (qh > 0) { memmove(queue, queue + qh, (qt - qh) * sizeof(int32_t)); qt -= qh; qh = 0; }
                        if (qt == qcap) {
                            qcap *= 2;
                            queue = realloc(queue, qcap * sizeof(int32_t));
                            if (!queue) { free(gpow); return ORC_ERR_ALLOC; }
                        }
                    }
                    queue[qt++] = parent[nb];
                }
        }
        /* run(), :106-107 + prune(), :28-40: leaves in reverse order */
        if (prune_suboptimal_leaves)
            for (int i = n_nodes - 1; i >= root; --i) {
                if (!alive[i]) continue;
                const double vub = SA_U(i);
                for (int nd = head[state[i]]; nd >= 0; nd = next_same[nd])
                    if (nd != i && SA_U(nd) >= vub && depth[nd] >= depth[i] && (first_child[nd] >= 0 || alive[nd])) {
                        alive[i] = 0;
                        break;
                    }
            }
    }
#undef SA_U
    if (rc == ORC_OK) {
        /* abstract.py:143-156 get_plan with deterministic.py:21-26 selection_rule (value_lower is the path's
         * discounted reward: state_aware.py never backs lower bounds up) */
        orc_pcg64 g = {rng6[0], rng6[1], rng6[2], rng6[3], rng6[4], rng6[5]};
        int len = 0;
        /* get_plan() runs TWICE per plan(): OptimisticDeterministicPlanner.plan returns one (deterministic.py:122)
         * that StateAwarePlanner.plan drops before calling it again (state_aware.py:122-127); both descents draw
         * from the generator on ties, the second one is returned */
        for (int pass = 0; pass < 2; ++pass) {
            int n = root;
            len = 0;
            while (first_child[n] >= 0) {
                const int fc = first_child[n];
                double m = lower[fc];
                for (int a = 1; a < n_children[n]; ++a) if (lower[fc + a] > m) m = lower[fc + a];
                int nt = 0, a = 0; /* choice over all maximal children, however many */
                for (int q = 0; q < n_children[n]; ++q) nt += lower[fc + q] == m;
                int pick = (int)orc_pcg64_below(&g, (uint32_t)nt);
                for (int q = 0; q < n_children[n]; ++q) if (lower[fc + q] == m && pick-- == 0) { a = q; break; }
                if (plan && len < max_plan_len) plan[len] = action[fc + a];
                ++len;
                n = fc + a;
            }
        }
        if (plan) for (int i = len; i < max_plan_len; ++i) plan[i] = -1;
        if (plan_len) *plan_len = len;
        rng6[0] = g.s_hi; rng6[1] = g.s_lo; rng6[2] = g.inc_hi; rng6[3] = g.inc_lo;
        rng6[4] = g.has_uint32; rng6[5] = g.uinteger;
    }
    if (env_steps) *env_steps = steps_taken;
    if (updates_total) *updates_total = updates;
    *n_nodes_io = n_nodes;
    if (root_out) *root_out = root;
    free(gpow); free(queue);
    return rc;
}

/* First plan() of n fresh StateAwarePlanner objects, OpenMP over planners (cpu_baseline leg of bench.py). */
int orc_saopd_plan_batch(int S, int A, const int64_t *T, const double *R, const uint8_t *term, int done_on_next,
                         int n, const int32_t *s0, int budget, double gamma, double terminal_reward, double accuracy,
                         int backup_aggregated_nodes, int prune_suboptimal_leaves, uint64_t *rng6 /* [n,6] */,
                         int max_plan_len, int32_t *plans, int32_t *plan_len, int64_t *env_steps, int64_t *updates,
                         int32_t *status, int n_threads, const uint8_t *avail)
{
    const int cap = 1 + (budget / A) * A;
#pragma omp parallel for schedule(dynamic, 4) num_threads(n_threads > 0 ? n_threads : 1)
    for (int i = 0; i < n; ++i) {
        int32_t *i32 = malloc((size_t)cap * 7 * sizeof(int32_t) + (size_t)S * 2 * sizeof(int32_t));
        double *f64 = malloc((size_t)cap * 2 * sizeof(double) + (size_t)S * sizeof(double));
        int64_t *cnt = malloc((size_t)cap * sizeof(int64_t));
        uint8_t *u8 = malloc((size_t)cap * 2);
        int32_t nn = 0, root = 0;
        int rc = ORC_ERR_ALLOC;
        if (i32 && f64 && cnt && u8)
            rc = orc_saopd_plan(S, A, T, R, term, done_on_next, s0[i], budget, gamma, terminal_reward, accuracy,
                                backup_aggregated_nodes, prune_suboptimal_leaves, rng6 + (long)i * 6, max_plan_len,
                                plans ? plans + (long)i * max_plan_len : NULL, plan_len ? plan_len + i : NULL,
                                env_steps ? env_steps + i : NULL, updates ? updates + i : NULL, 1, cap, &nn, &root,
                                i32, i32 + cap, i32 + 2 * cap, i32 + 3 * cap, f64, f64 + cap, u8, cnt, i32 + 4 * cap,
                                u8 + cap, i32 + 5 * cap, f64 + 2 * cap, i32 + 7 * cap, i32 + 7 * cap + S, avail, i32 + 6 * cap);
        if (status) status[i] = rc;
        free(i32); free(f64); free(cnt); free(u8);
    }
    return ORC_OK;
}
