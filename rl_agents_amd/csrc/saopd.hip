// saopd.hip -- state-aware optimistic planning (tree_search/state_aware.py) for many independent planners.
//
// The reference's StateAwarePlanner keeps, next to the OPD tree, two dictionaries keyed by str(observation):
// state_values (an upper bound of V per observed state) and state_nodes (every node observed in that state).
// One iteration (run(), :93-107) = argmax-U leaf over ALL leaves (U depends on the moving state values, so no leaf
// order can be cached), expand, a queue-driven Bellman backup over aggregated nodes (backup_to_root, :42-63) and a
// prune pass over all leaves in reverse order (:28-40).  Both dictionaries and every node ever created outlive
// plan() calls (reset() only installs a new root and leaves list), so the planner state is an arena.
//
// Mapping.  The backup queue and the prune pass are order-dependent chains (first-in-first-out processing against
// moving state values; pruned leaves stop dominating later ones), so a planner is sequential and the parallel axis is
// planners.  Two kernels, same results:
//   saopd_wave_kernel (default)  one planner per WAVEFRONT, planner-major arrays: the serial phases run as uniform code
//                                (a dependent access is one cache line per wave), scans / children / prune candidates
//                                use the 64 lanes.  Lowest latency, and 64x more waves to hide it.
//   saopd_kernel (MP_SAOPD_MODEL=lane)  one planner per LANE, node-major arrays [node][planner]: all planners of a batch
//                                run the same iterations on the same node ids, so the loops over "all nodes" are
//                                coalesced 64-wide; per-state dictionaries and list walks are gathers.
// Layouts below are given for the lane mapping ([row][planner]); the wave mapping transposes them.
//
//   node   SaNode[cap][n] 16 B = {f64 lower, i32 next_same (state_nodes list link), u32 meta}
//          meta = depth | HAS_CHILDREN | ALIVE ("in planner.leaves"): a list walk costs one dwordx4 per element
//          state / parent / first_child i32 [cap][n], reward f64 [cap][n], done u8 [cap][n]
//   state  sv f64 [S][n] (state_values, default 1/(1-gamma)), head / tail i32 [S][n] (state_nodes)
//   stamp  i32 [S][n]: the last iteration in which a state's value or node list changed.  prune() of a leaf depends
//          only on its state's value and list (U of same-state nodes, depths, has-children / in-leaves flags, and
//          those flags only ever change in the direction that removes dominators), so a leaf that survived the
//          previous pass survives this one unless its state changed: the pass walks the lists of the leaves in
//          changed states only -- same result, a small fraction of the reference's O(leaves x list) work
//   queue  ring buffer per planner of 16-byte descriptors {state, node, delta}: the reference's list.pop(0) queue kept
//          lazily (one entry per pending state update, expanded into "the parents of that state's nodes" when it
//          reaches the front); overflow is reported per planner, never dropped silently
//   tables gamma**d, terminal_reward*gamma**d/(1-gamma), accuracy*(1-gamma)*gamma**(d-1) from the host (libm pow)
//
// Round 2 of the wave kernel (same results, the order-dependent semantics kept by sequential halves over registers):
//   prune   the rows of the changed states are COLLECTED by one coalesced scan of state[] (dead rows marked in bit 1 of
//           done[], the old arena represented by a per-plan list of its rows with children), held one per lane in four
//           register sets and tested against every candidate leaf with one ballot -- no list walk
//   backup  every state's list also exists as chunks of 15 ids (lstate / lpool); a group of |A| lanes evaluates one
//           neighbour's parent, 64 / |A| neighbours of several pending descriptors per pass; applied in queue and list
//           order with register patching / re-evaluation when a state value moves (first plan of fresh planners)
//   meta    bits 24..29 = the action that led to the node (its sibling group = the children of its parent)
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <new>
#include <vector>

#include "common.hpp"
#include "pcg64.hpp"
#include "wave.hpp"

namespace mp {

struct alignas(16) SaNode {
    double lower;
    int32_t next_same;
    uint32_t meta;
};
static_assert(sizeof(SaNode) == 16, "SaNode must be one dwordx4");
constexpr uint32_t SA_ALIVE = 0x80000000u, SA_CHILDREN = 0x40000000u, SA_DEPTH = 0x00ffffffu;
// bits 24..29 of meta: the action that led to the node (wave kernel: its sibling group starts at id - action, so a
// backup finds the children of a node's parent without reading the parent's first_child)
// A node record as ONE 16-byte load: the compiler otherwise splits the struct load into a dwordx2 and one or two dword
// loads when the fields are used apart, and at 8 waves per SIMD the wave kernel is bound by the number of vector-memory
// instructions it issues.
__device__ __forceinline__ SaNode load_node(const SaNode *p)
{
    // (round 4: no barrier behind the load any more.  An `asm volatile` that made all four words live kept the load one
    // dwordx4, but it also made the compiler WAIT for it where it stood: the leaf scan requested its four rows one after the
    // other with a full round trip each.  Plain: 5.92 against 6.08 ms for light planners, 5.4 / 6.35 against 5.6 / 6.6 ms for
    // the following plans.)
    const uint4 r = *reinterpret_cast<const uint4 *>(p);
    SaNode n;
    n.lower = __hiloint2double((int)r.y, (int)r.x); n.next_same = (int32_t)r.z; n.meta = r.w;
    return n;
}
constexpr int SA_ACT_SHIFT = 24;
constexpr uint32_t SA_ACT = 0x3f000000u;

} // namespace mp

// the planners of a batch: everything a reference StateAwarePlanner object holds across plan() calls
struct mp_saopd {
    mp_ctx *ctx = nullptr;
    mp_model *model = nullptr;
    uint64_t model_serial = 0;
    int n = 0, S = 0, A = 0;
    int cap = 0;        // node rows allocated
    int n_nodes = 0;    // node rows in use (same for every planner of the batch)
    int root = -1;      // row of the current roots
    int qcap = 0;
    double gamma = 0.0; // fixed at the first plan (sv defaults depend on it)
    mp::SaNode *node = nullptr;
    int32_t *state = nullptr, *parent = nullptr, *first_child = nullptr;
    double *reward = nullptr;
    uint8_t *done = nullptr;
    int2 *oldlive = nullptr; // wave kernel: {row, state} of the rows of earlier plans that have children (rebuilt by every plan)
    // wave kernel: every state's node list again as chunks of 15 ids + link (16 ints), rebuilt from the linked lists by every
    // plan and kept up to date by its appends: the backup reads a popped state's list 15 neighbours per load
    int4 *lstate = nullptr;     // [n][S] {count, head chunk, tail chunk, -}
    int32_t *lpool = nullptr;   // [n][pool_ints]
    long pool_ints = 0;
    double *sv = nullptr;
    int32_t *head = nullptr, *tail = nullptr, *queue = nullptr, *stamp = nullptr;
    int iters = 0;      // iterations run so far (stamps are unique across plans)
    // copies of the per-state dictionaries and the generator states taken before a plan: a plan whose backup queue
    // overflows is rolled back and run again with a larger queue
    double *snap_sv = nullptr;
    int32_t *snap_head = nullptr, *snap_tail = nullptr, *snap_stamp = nullptr, *overflow = nullptr;
    uint64_t *snap_rng = nullptr;
    std::vector<mp_ctx::Block> blocks; // every device block of this batch, with its size: they go back to the ctx's block cache
    int32_t *cost = nullptr, *order = nullptr; // wave kernel: Bellman backups of the last plan per planner / this plan's dispatch order
    bool cost_valid = false;
    int wave = 0;       // 1: one planner per wavefront, planner-major arrays ([planner][node], [planner][state],
                        //    [planner][slot]); 0: one planner per lane, node-major arrays
    // element (row i, planner r) of a node array = i * node_si + r * node_sr, likewise states and queue slots
    long node_si() const { return wave ? 1 : n; }
    long node_sr() const { return wave ? cap : 1; }
    long state_si() const { return wave ? 1 : n; }
    long state_sr() const { return wave ? S : 1; }
};

namespace mp {

struct SaArgs {
    int n, S, A, K, root, n_prev, prev_root, qcap, done_on_next, max_plan_len;
    int backup_aggregated, prune, fresh, iter_base;
    int32_t *overflow; // [0]: set when a planner's backup queue is full; [1 + r]: planner r FAILED for good (sticky)
    int sticky;        // 1: a full queue marks the planner failed (asynchronous device mode: no roll-back, no retry)
    int cap;  // wave kernel: node rows allocated per planner
    int scap; // wave kernel: scratch entries per lane in the prune pass (qcap / 64 unless MP_SAOPD_LANE_SCRATCH)
    int prune_rows; // wave kernel: rows the prune pass takes through its register sets (256; MP_SAOPD_PRUNE_ROWS: test knob)
    int par_backup; // wave kernel: 1 = grouped parallel backup over chunked state lists (host: the first plan of fresh planners)
    int lds_rows, lds_qcap; // LDS-resident wave kernel: node rows held in LDS (>= rows after this plan), queue ints in LDS
    int tab_lds;            // wave kernel with the dictionaries in LDS: depth-table entries held in LDS
    int tab_plain;          // the other wave kernels: depth-table entries held in LDS (all K + 3 unless the budget is huge)
    int csr_old;            // wave kernel, dictionaries in LDS: the earlier plans' rows bucketed by state: 0 never, 1 from the fourth plan, 2 always
    const int32_t *order;   // wave kernel: workgroup b plans planner order[b] (longest expected plan first), or nullptr
    int32_t *cost;          // wave kernel: Bellman backups this plan ran, per planner (the next plan's dispatch order)
    double gamma, vmax;
    const Rec *rec;
    const double *tab; // gpow[K+3] | trg[K+3] | acc[K+3]
    const int32_t *root_state;
    uint64_t *rng;
    SaNode *node;
    int32_t *state, *parent, *first_child;
    double *reward;
    uint8_t *done;
    int2 *oldlive;    // [planner][cap] {row, state}: scratch of the wave kernel's prune scan
    int4 *lstate;     // chunked per-state lists (wave kernel): {count, head chunk, tail chunk, -} per state
    int32_t *lpool;
    long pool_ints;
    double *sv;
    int32_t *head, *tail, *queue, *stamp;
    int32_t *plans, *plan_len, *status;
    int64_t *env_steps, *updates;
};

__global__ __launch_bounds__(64) void saopd_init_kernel(int n, int S, double vmax, double *sv, int32_t *head, int32_t *tail,
                                                        int32_t *stamp)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)n * S) return;
    sv[i] = vmax; head[i] = -1; tail[i] = -1; stamp[i] = -1;
}

__global__ __launch_bounds__(64) void saopd_kernel(SaArgs p)
{
    extern __shared__ __attribute__((aligned(16))) double lds_d[];
    const int ntab = 3 * (p.K + 3);
    for (int i = threadIdx.x; i < ntab; i += blockDim.x) lds_d[i] = p.tab[i];
    __syncthreads();
    const double *gpow = lds_d, *trg = lds_d + (p.K + 3), *acc = lds_d + 2 * (p.K + 3);
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= p.n) return;
    if (p.overflow[1 + r]) { // a planner whose backup queue overflowed in an asynchronous call: its dictionaries are broken
        if (p.plans)
            for (int i = 0; i < p.max_plan_len; ++i) p.plans[(long)r * p.max_plan_len + i] = -1;
        if (p.plan_len) p.plan_len[r] = 0;
        if (p.status) p.status[r] = MP_ERR_ALLOC;
        if (p.env_steps) p.env_steps[r] = 0;
        if (p.updates) p.updates[r] = 0;
        return;
    }
    const long n = p.n;
    const int A = p.A;
    // [row][planner] addressing
    auto ND = [&](int i) -> SaNode & { return p.node[(long)i * n + r]; };
    auto ST = [&](int i) -> int32_t & { return p.state[(long)i * n + r]; };
    auto PA = [&](int i) -> int32_t & { return p.parent[(long)i * n + r]; };
    auto FC = [&](int i) -> int32_t & { return p.first_child[(long)i * n + r]; };
    auto RW = [&](int i) -> double & { return p.reward[(long)i * n + r]; };
    auto SV = [&](int s) -> double & { return p.sv[(long)s * n + r]; };
    auto HD = [&](int s) -> int32_t & { return p.head[(long)s * n + r]; };
    auto TL = [&](int s) -> int32_t & { return p.tail[(long)s * n + r]; };
    auto SM = [&](int s) -> int32_t & { return p.stamp[(long)s * n + r]; };
    // ring-buffer slot q of this planner (slot-major: a planner-major layout puts the 64 lanes of a wave 4*qcap bytes
    // apart, i.e. in one cache set and one L2 channel)
    const int qmask = p.qcap - 1;
    auto QU = [&](unsigned q) -> int32_t & { return p.queue[(long)(q & (unsigned)qmask) * n + r]; };
    const unsigned dcap = (unsigned)p.qcap >> 2; // backup-queue descriptors: 4 ints each
    auto QD = [&](unsigned q, int f) -> int32_t & { return p.queue[(long)(((q & (dcap - 1)) << 2) + f) * n + r]; };
    const uint32_t done_bit = p.done_on_next ? 2u : 1u;

    // reset() (deterministic.py:102-104): the previous leaves list is dropped, a new root is installed
    for (int i = p.prev_root; i < p.n_prev; ++i) {
        const uint32_t m = ND(i).meta;
        if (m & SA_ALIVE) ND(i).meta = m & ~SA_ALIVE;
    }
    const int root = p.root;
    const int32_t s0 = p.root_state[r];
    {
        SaNode nd;
        nd.lower = 0.0; nd.next_same = -1; nd.meta = SA_ALIVE; // depth 0
        ND(root) = nd;
        ST(root) = s0; PA(root) = -1; FC(root) = -1; RW(root) = 0.0;
        p.done[(long)root * n + r] = 0;
    }
    // plan() (state_aware.py:117-121): the root state's entries start over
    HD(s0) = root; TL(s0) = root;
    SV(s0) = p.vmax;
    int n_nodes = root + 1;
    int status = MP_OK;
    long steps_taken = 0, updates = 0;
    // U of a node (state_aware.py:65-67): value_lower + gamma**depth * state_values[observation]
    auto U_of = [&](const SaNode &nd, int i) { return nd.lower + gpow[nd.meta & SA_DEPTH] * SV(ST(i)); };

    for (int k = 0; k < p.K && status == MP_OK; ++k) {
        // ---- run() :95: max(leaves, key=U), first maximum in leaves order = ascending node id among the alive
        const int cur = p.iter_base + k; // stamp of this iteration
        int leaf = -1;
        double bu = 0.0;
        constexpr int UN = 8; // rows in flight: the loads of a chunk are independent, only the compare is ordered
        for (int i0 = root; i0 < n_nodes; i0 += UN) {
            SaNode nd[UN];
            int32_t st[UN];
            double sv[UN];
#pragma unroll
            for (int j = 0; j < UN; ++j) {
                const int i = min(i0 + j, n_nodes - 1);
                nd[j] = ND(i);
                st[j] = ST(i);
            }
#pragma unroll
            for (int j = 0; j < UN; ++j) sv[j] = (nd[j].meta & SA_ALIVE) ? SV(st[j]) : 0.0;
#pragma unroll
            for (int j = 0; j < UN; ++j)
                if (i0 + j < n_nodes && (nd[j].meta & SA_ALIVE)) {
                    const double u = nd[j].lower + gpow[nd[j].meta & SA_DEPTH] * sv[j];
                    if (leaf < 0 || u > bu) { leaf = i0 + j; bu = u; }
                }
        }
        if (leaf < 0) { // the reference raises: max() of an empty sequence.  Asynchronous mode: the caller goes on stepping the
            status = MP_ERR_ARG; // batch, so the planner stays failed (its tree has no leaf to plan from)
            if (p.sticky) p.overflow[1 + r] = MP_ERR_ARG;
            break;
        }
        // ---- expand (deterministic.py:28-43) + update (:45-65, state_aware.py:15-26)
        const SaNode lf = ND(leaf);
        const int dl = (int)(lf.meta & SA_DEPTH);
        ND(leaf).meta = (lf.meta & ~SA_ALIVE) | SA_CHILDREN;
        FC(leaf) = n_nodes;
        const int32_t sl = ST(leaf);
        for (int a = 0; a < A; ++a) {
            const int c = n_nodes + a;
            const Rec rc = p.rec[(long)sl * A + a];
            if (!(rc.flags & 4u)) {
                // deterministic.py:32-35: an action state.get_available_actions() does not list gets no child.  Its slot
                // stays in the id space (ids advance by |A| per expansion) as a PHANTOM row: lower = -inf (never the best
                // child of a Bellman backup), not alive (never a leaf), in no state's list (never a neighbour, never a
                // dominator), dead for the prune scan; the export drops it.
                SaNode ph;
                ph.lower = -INFINITY; ph.next_same = -1; ph.meta = (uint32_t)(dl + 1);
                ND(c) = ph;
                ST(c) = rc.next; PA(c) = leaf; FC(c) = -1; RW(c) = 0.0;
                p.done[(long)c * n + r] = 2;
                continue;
            }
            const bool terminated = (rc.flags & done_bit) != 0;
            ++steps_taken;
            if (!(0.0 <= rc.reward) || !(rc.reward <= 1.0)) { status = MP_ERR_REWARD_RANGE; break; }
            const int d = dl + 1;
            double lower = lf.lower + gpow[d - 1] * rc.reward;
            if (terminated) lower = lower + trg[d];
            const int32_t s = rc.next;
            SaNode nd;
            nd.lower = lower; nd.next_same = -1; nd.meta = SA_ALIVE | (uint32_t)d;
            ND(c) = nd;
            ST(c) = s; PA(c) = leaf; FC(c) = -1; RW(c) = rc.reward;
            p.done[(long)c * n + r] = terminated ? 1 : 0;
            // state_nodes[str(observation)].append(self)
            const int32_t t = TL(s);
            if (t < 0) HD(s) = c; else ND(t).next_same = c;
            TL(s) = c;
            SM(s) = cur;
            // terminal states are worth 0 (update_value(observation, 0))
            if (terminated && SV(s) - 0.0 > 0.0) SV(s) = 0.0;
        }
        if (status != MP_OK) break;
        n_nodes += A;
        // ---- backup_to_root (state_aware.py:42-63): first-in-first-out over a queue that holds duplicates.
        // Planners of a wave have queues and lists of different lengths, so the nested loops (pop / walk the list
        // of the popped node's state) are flattened into ONE loop in which every lane does one step of its own
        // state machine per trip: a trip count of max-over-lanes(total steps) instead of a sum of per-pop maxima.
        // The reference's queue is kept LAZILY: an update of state s by node x appends "the parents of the nodes of
        // list(s)" -- one 16-byte descriptor {s, x, delta} -- and the list is walked when the descriptor reaches the
        // front.  Lists, parents and thresholds do not change during a backup, so the order of the pops is the
        // reference's, while the ring holds one entry per pending update instead of one per list element (tiny state
        // spaces put most of the tree in one list: materialised, 120 nodes over 3 states overflowed 4096 entries).
        {
            unsigned qh = 0, qt = 0;
            QD(qt, 0) = -1; QD(qt, 1) = leaf; // {-1, node}: a single node (the expanded leaf itself)
            ++qt;
            int src = -1, nb = -1;
            double src_delta = 0.0;
            bool active = true;
            while (active) {
                int target = -1; // the node this trip pops (:48)
                if (nb < 0) {    // front descriptor
                    if (qh == qt) { active = false; continue; }
                    const int32_t ds = QD(qh, 0);
                    src = QD(qh, 1);
                    if (ds < 0) {
                        target = src;
                    } else {
                        src_delta = __hiloint2double(QD(qh, 3), QD(qh, 2));
                        nb = HD(ds);
                    }
                    ++qh;
                } else {         // one neighbour (:58-63)
                    const SaNode nd = ND(nb);
                    const int par = PA(nb);
                    if (par >= 0 && (nb == src || p.backup_aggregated) && src_delta > acc[nd.meta & SA_DEPTH]) target = par;
                    nb = nd.next_same;
                }
                if (target >= 0) { // Bellman backup of the popped node (:49-56)
                    const int32_t sn = ST(target);
                    const int fc = FC(target);
                    if (fc >= 0) {
                        int bc = fc;
                        double bcu = U_of(ND(fc), fc);
                        for (int a = 1; a < A; ++a) {
                            const double u = U_of(ND(fc + a), fc + a);
                            if (u > bcu) { bc = fc + a; bcu = u; }
                        }
                        const double backup = RW(bc) + p.gamma * SV(ST(bc));
                        const double old = SV(sn);
                        const double delta = old - backup; // update_value (:109-119)
                        ++updates;
                        if (delta > 0.0) { // (thresholds are >= 0: nothing is appended otherwise)
                            SV(sn) = backup; SM(sn) = cur;
                            if (qt - qh >= dcap) {
                                status = MP_ERR_ALLOC; *p.overflow = 1; active = false;
                                if (p.sticky) p.overflow[1 + r] = MP_ERR_ALLOC;
                                continue;
                            }
                            QD(qt, 0) = sn; QD(qt, 1) = target;
                            QD(qt, 2) = __double2loint(delta); QD(qt, 3) = __double2hiint(delta);
                            ++qt;
                        }
                    }
                }
            }
        }
        if (status != MP_OK) break;
        // ---- prune (run() :106-107, prune() :28-40): leaves in reverse order; a pruned leaf stops dominating.
        // Pass 1 (uniform over rows, coalesced): the alive leaves whose state changed this iteration, in reverse
        // order, into the (now idle) queue buffer.  Pass 2: the list walks, flattened like the backup loop.
        if (p.prune) {
            int nc = 0;
            for (int i0 = n_nodes - 1; i0 >= root; i0 -= UN) {
                SaNode mes[UN];
                int32_t sts[UN], stamps[UN];
#pragma unroll
                for (int j = 0; j < UN; ++j) {
                    const int i = max(i0 - j, root);
                    mes[j] = ND(i);
                    sts[j] = ST(i);
                }
#pragma unroll
                for (int j = 0; j < UN; ++j) stamps[j] = (mes[j].meta & SA_ALIVE) ? SM(sts[j]) : -1;
#pragma unroll
                for (int j = 0; j < UN; ++j)
                    if (i0 - j >= root && (mes[j].meta & SA_ALIVE) && stamps[j] == cur) QU((unsigned)nc++) = i0 - j;
            }
            int ci = 0, i = -1, nd_i = -1, dm = 0;
            uint32_t my_meta = 0;
            double svs = 0.0, vub = 0.0;
            while (ci < nc || nd_i >= 0) {
                if (nd_i < 0) { // next candidate leaf
                    i = QU((unsigned)ci++);
                    const SaNode me = ND(i);
                    const int32_t s = ST(i);
                    svs = SV(s); // every node of the list is in state s
                    my_meta = me.meta;
                    dm = (int)(me.meta & SA_DEPTH);
                    vub = me.lower + gpow[dm] * svs;
                    nd_i = HD(s);
                } else { // one node of state_nodes[str(observation)]
                    const SaNode nd = ND(nd_i);
                    const int dn = (int)(nd.meta & SA_DEPTH);
                    if (nd_i != i && nd.lower + gpow[dn] * svs >= vub && dn >= dm && (nd.meta & (SA_CHILDREN | SA_ALIVE))) {
                        ND(i).meta = my_meta & ~SA_ALIVE;
                        nd_i = -1;
                    } else {
                        nd_i = nd.next_same;
                    }
                }
            }
        }
    }
    // ---- get_plan (abstract.py:143-156) with DeterministicNode.selection_rule (deterministic.py:21-26), TWICE:
    // OptimisticDeterministicPlanner.plan computes one (deterministic.py:122) that StateAwarePlanner.plan drops
    // before computing the one it returns (state_aware.py:122-127); both draw from the generator on ties
    int len = 0;
    if (status == MP_OK) {
        Pcg64 g;
        g.load(p.rng + (long)r * 6);
        for (int pass = 0; pass < 2; ++pass) {
            int node = root;
            len = 0;
            int fc = FC(node);
            while (fc >= 0) {
                double m = ND(fc).lower;
                for (int a = 1; a < A; ++a) {
                    const double l = ND(fc + a).lower;
                    m = l > m ? l : m;
                }
                int nt = 0;
                for (int a = 0; a < A; ++a) nt += ND(fc + a).lower == m ? 1 : 0;
                int pick = nt > 1 ? (int)g.below((uint32_t)nt) : 0;
                int act = 0;
                for (int a = 0; a < A; ++a)
                    if (ND(fc + a).lower == m) {
                        if (pick == 0) { act = a; break; }
                        --pick;
                    }
                if (pass == 1 && p.plans && len < p.max_plan_len) p.plans[(long)r * p.max_plan_len + len] = act;
                ++len;
                node = fc + act;
                fc = FC(node);
            }
        }
        g.store(p.rng + (long)r * 6);
    }
    if (p.plans)
        for (int i = len; i < p.max_plan_len; ++i) p.plans[(long)r * p.max_plan_len + i] = -1;
    if (p.plan_len) p.plan_len[r] = len;
    if (p.status) p.status[r] = status;
    if (p.env_steps) p.env_steps[r] = steps_taken;
    if (p.updates) p.updates[r] = updates;
}


// ---- one planner per WAVEFRONT (planner-major arrays).  The order-dependent parts (backup queue, list walks, list
// appends, plan descent) run as uniform code -- every lane computes the same thing, lane 0 stores -- so a dependent
// access is one cache line for the whole wave instead of 64 scattered ones; the data-parallel parts use the lanes:
// leaf argmax over 64 rows per trip (DPP reduction), the |A| children of an expansion and of a Bellman backup one
// per lane, the candidate leaves of a prune pass 64 rows per trip (ballot).  Same results as saopd_kernel; a plan's
// latency is that of ONE planner's chain (~12x shorter than a lane's, whose wave waits for its slowest lane and
// pays 64 scattered lines per access), and there are 64x more waves to hide it.
// LDSR: the planner's whole working set -- every node row of its arena, the per-state dictionaries, the backup queue --
// is staged into LDS for the plan and written back at the end (39 KB at the reference's GridWorld configuration: four
// planners per CU).  The kernel is a chain of dependent accesses (queue front -> list element -> parent -> children ->
// state values), ~15 000 per plan: from LDS each costs ~100 cycles instead of a ~600-2 000-cycle L2 / HBM round trip.
// The host selects it while the arena fits (first plans of a planner; later plans fall back to the global-memory form).
#ifndef MP_SAOPD_DICT_WAVES
#define MP_SAOPD_DICT_WAVES 8 // the same for the variant with the dictionaries in LDS (5 KB of LDS per planner)
#endif
#ifndef MP_SAOPD_MIN_WAVES
#define MP_SAOPD_MIN_WAVES 8 // waves per SIMD the register allocation must admit (106 SGPRs would cap the kernel at 6)
#endif
// Ordering of the wave kernel's uniform phases: lane 0 stores, all lanes load later.  A workgroup is one wavefront and
// the texture path takes a wavefront's vector-memory operations in order (a store that hits in the L1 updates the line),
// so a later load sees an earlier store without waiting for its acknowledgement; only the compiler must keep the order.
// MP_SAOPD_SYNC restores the waits (A/B).
#ifdef MP_SAOPD_SYNC
#define SA_ORDER() __syncthreads()
#else
#define SA_ORDER() __builtin_amdgcn_wave_barrier()
#endif
// LDSD (round 4): only the per-state DICTIONARIES live in LDS for the plan -- state values, list heads / tails, stamps and the
// chunked lists' records, 36 B per state (3.6 KB at S = 100, which leaves five waves per SIMD) -- staged in at the start and
// written back at the end.  Every dependent chain of the plan ends in one of them (a child's or a parent's state value, a
// popped state's list, a scanned row's stamp): those hops become LDS reads, off the texture path the kernel is bound by.
// SA_SYNC: a wave's stores followed by its own (other lanes') loads.  From global memory the texture path's order is enough
// (see SA_ORDER); the LDS-resident form keeps the barrier it was validated with.
#define SA_SYNC() do { if constexpr (LDSR) __syncthreads(); else SA_ORDER(); } while (0)
template <bool LDSR, bool LDSD = false>
__global__ __launch_bounds__(64, LDSR ? 1 : (LDSD ? MP_SAOPD_DICT_WAVES : MP_SAOPD_MIN_WAVES)) void saopd_wave_kernel(SaArgs p)
{
    extern __shared__ __attribute__((aligned(16))) double lds_d[];
    // the depth tables in LDS; LDSD keeps only their first tab_lds entries there (deeper nodes read the global copy), so that
    // tables + dictionaries stay within the 5 KB per planner that eight waves per SIMD leave
    const int TLD = LDSD ? p.tab_lds : p.tab_plain;
    const int ntab = 3 * TLD;
    const int lane = threadIdx.x;
    for (int i = lane; i < ntab; i += 64) {
        const int t = i / TLD;
        lds_d[i] = p.tab[t * (p.K + 3) + (i - t * TLD)];
    }
    struct DepthTab {
        const double *l, *g;
        int nl;
        // The LDS read is unconditional and the global one a branch no lane of a shallow tree takes.  (Written as
        // `d < nl ? l[d] : g[d]` the compiler selects between the two POINTERS and issues one flat_load through the texture path
        // for every table access -- the path this kernel is short of; explicit address spaces keep it from forming the select.)
        __device__ __forceinline__ double operator[](int d) const
        {
            typedef const __attribute__((address_space(3))) double *lds_cdp;
            typedef const __attribute__((address_space(1))) double *glb_cdp;
            double v = ((lds_cdp)l)[d < nl ? d : nl - 1];
            if (d >= nl) v = ((glb_cdp)g)[d];
            return v;
        }
    };
    const DepthTab gpow{lds_d, p.tab, TLD}, trg{lds_d + TLD, p.tab + (p.K + 3), TLD}, acc{lds_d + 2 * TLD, p.tab + 2 * (p.K + 3), TLD};
    constexpr int DCAP = LDSD ? 4 : 128; // (512 B of LDS after the tables, reserved by the host's size computation; unused since the
                              // prune pass finds the rows of changed states by their stamps)
    // Workgroups start in index order and a planner is one sequential chain whose length varies 7x with its root state
    // (1 700 .. 13 000 Bellman backups on the reference's grid): a long planner that starts in the last round IS the
    // kernel's tail.  The host passes the planners sorted by their expected cost, longest first.
    const int r = p.order ? p.order[blockIdx.x] : (int)blockIdx.x;
    const int A = p.A;
    if (p.overflow[1 + r]) { // failed for good in an earlier asynchronous call (see saopd_kernel)
        if (lane == 0) {
            if (p.plans)
                for (int i = 0; i < p.max_plan_len; ++i) p.plans[(long)r * p.max_plan_len + i] = -1;
            if (p.plan_len) p.plan_len[r] = 0;
            if (p.status) p.status[r] = p.overflow[1 + r]; // the code it failed with (MP_ERR_ALLOC: queue full; MP_ERR_ARG: every leaf pruned)
            if (p.env_steps) p.env_steps[r] = 0;
            if (p.updates) p.updates[r] = 0;
        }
        return;
    }
    const long nb = (long)r * p.cap, sb = (long)r * p.S, qb = (long)r * p.qcap;
    // LDS carve (LDSR): [tables | dirty 128 i32 | node rows 16 B | reward f64 | sv f64 | state, parent, first_child i32 rows |
    // head, tail, stamp i32 [S] | queue i32]
    const int LR = p.lds_rows;
    SaNode *l_node = reinterpret_cast<SaNode *>(lds_d + ((ntab + DCAP / 2 + 1) & ~1)); // 16-byte aligned
    double *l_reward = reinterpret_cast<double *>(l_node + LR);
    double *l_sv = l_reward + LR;
    int32_t *l_state = reinterpret_cast<int32_t *>(l_sv + p.S);
    int32_t *l_parent = l_state + LR, *l_fc = l_parent + LR;
    int32_t *l_head = l_fc + LR, *l_tail = l_head + p.S, *l_stamp = l_tail + p.S;
    int32_t *l_queue = l_stamp + p.S;
    // LDS carve (LDSD): [tables | 4 i32 | sv f64 [S] | list records int4 [S] | head, tail, stamp, mark i32 [S]]
    double *d_sv = lds_d + ((ntab + DCAP / 2 + 1) & ~1);
    int4 *d_ls = reinterpret_cast<int4 *>(d_sv + p.S);
    int32_t *d_head = reinterpret_cast<int32_t *>(d_ls + p.S), *d_tail = d_head + p.S, *d_stamp = d_tail + p.S;
    uint32_t *d_mark = reinterpret_cast<uint32_t *>(d_stamp + p.S); // backup: the first group of the pass that may write a state
    SaNode *const node_b = LDSR ? l_node : p.node + nb;
    int32_t *const state_b = LDSR ? l_state : p.state + nb;
    int32_t *const parent_b = LDSR ? l_parent : p.parent + nb;
    int32_t *const fc_b = LDSR ? l_fc : p.first_child + nb;
    double *const reward_b = LDSR ? l_reward : p.reward + nb;
    double *const sv_b = LDSR ? l_sv : (LDSD ? d_sv : p.sv + sb);
    int32_t *const head_b = LDSR ? l_head : (LDSD ? d_head : p.head + sb);
    int32_t *const tail_b = LDSR ? l_tail : (LDSD ? d_tail : p.tail + sb);
    int32_t *const stamp_b = LDSR ? l_stamp : (LDSD ? d_stamp : p.stamp + sb);
    int32_t *const queue_b = LDSR ? l_queue : p.queue + qb;
    const int qcap = LDSR ? p.lds_qcap : p.qcap;
    if (LDSR) { // stage in: the rows of earlier plans and the dictionaries (a fresh planner starts from the defaults)
        for (int i = lane; i < p.n_prev; i += 64) {
            l_node[i] = p.node[nb + i];
            l_state[i] = p.state[nb + i]; l_parent[i] = p.parent[nb + i]; l_fc[i] = p.first_child[nb + i];
            l_reward[i] = p.reward[nb + i];
        }
        for (int s = lane; s < p.S; s += 64) {
            l_sv[s] = p.fresh ? p.vmax : p.sv[sb + s];
            l_head[s] = p.fresh ? -1 : p.head[sb + s];
            l_tail[s] = p.fresh ? -1 : p.tail[sb + s];
            l_stamp[s] = p.fresh ? -1 : p.stamp[sb + s];
        }
    }
    if (LDSD)
        for (int s = lane; s < p.S; s += 64) {
            d_sv[s] = p.fresh ? p.vmax : p.sv[sb + s];
            d_head[s] = p.fresh ? -1 : p.head[sb + s];
            d_tail[s] = p.fresh ? -1 : p.tail[sb + s];
            d_stamp[s] = p.fresh ? -1 : p.stamp[sb + s];
            d_mark[s] = 0xffffffffu;
        }
    __syncthreads();
    auto ND = [&](int i) -> SaNode & { return node_b[i]; };
    auto ST = [&](int i) -> int32_t & { return state_b[i]; };
    auto PA = [&](int i) -> int32_t & { return parent_b[i]; };
    auto FC = [&](int i) -> int32_t & { return fc_b[i]; };
    auto RW = [&](int i) -> double & { return reward_b[i]; };
    auto SV = [&](int s) -> double & { return sv_b[s]; };
    auto HD = [&](int s) -> int32_t & { return head_b[s]; };
    auto TL = [&](int s) -> int32_t & { return tail_b[s]; };
    auto SM = [&](int s) -> int32_t & { return stamp_b[s]; };
    const unsigned dcap = (unsigned)qcap >> 2; // backup-queue descriptors: 4 ints each
    // one 16-byte descriptor {state, node, delta lo, delta hi} per slot: one vector-memory instruction per push / pop (at 8
    // waves per SIMD the backup is bound by the number of vector-memory instructions it issues, ~8 cycles each per CU)
    auto QD4 = [&](unsigned q) -> int4 & { return reinterpret_cast<int4 *>(queue_b)[q & (dcap - 1)]; };
    const uint32_t done_bit = p.done_on_next ? 2u : 1u;
    const double ninf = -INFINITY;
    const bool l0 = lane == 0;

    // reset(): drop the previous leaves list (64 rows per trip), install the new root
    for (int i = p.prev_root + lane; i < p.n_prev; i += 64) {
        const uint32_t m = ND(i).meta;
        // (bit 1 of done[]: the row is DEAD -- neither an alive leaf nor a node with children: it can neither be pruned nor
        // dominate, and the prune scan skips it)
        if (m & SA_ALIVE) { ND(i).meta = m & ~SA_ALIVE; p.done[nb + i] = (uint8_t)(p.done[nb + i] | 2); }
    }
    const int root = p.root;
    const int32_t s0 = p.root_state[r];
    if (l0) {
        SaNode nd;
        nd.lower = 0.0; nd.next_same = -1; nd.meta = SA_ALIVE;
        ND(root) = nd;
        ST(root) = s0; PA(root) = -1; FC(root) = -1; RW(root) = 0.0;
        p.done[nb + root] = 0;
        HD(s0) = root; TL(s0) = root; // plan(): the root state's entries start over
        SV(s0) = p.vmax;
    }
    __syncthreads();
    // the rows of EARLIER plans that can still matter to a prune test: those with children (their leaves were dropped by
    // reset()).  Listed once per plan, in id order, so that the per-iteration scan below reads a quarter of the old arena
    // and nothing of its dead rows -- the scan's cost would otherwise grow with every plan of an episode.
    int2 *old_b = p.oldlive + nb;
    int n_old = 0;
    // LDSD, later plans (no chunked lists in use): the same rows BUCKETED BY STATE -- a counting sort through LDS counters
    // (lane-ordered atomics keep the ids ascending inside a state), {count, offset} per state in the idle list records -- so that
    // an iteration touches the old rows of its few changed states only, not all of them: the prune pass of the sixteenth
    // plan of an episode costs what the second one's does.
    auto wave_incl_scan = [&](int v) {
        for (int d = 1; d < 64; d <<= 1) {
            const int o = __shfl_up(v, d);
            if (lane >= d) v += o;
        }
        return v;
    };
    // (building the buckets costs two passes over the old rows: it pays from the fourth plan of an episode on -- measured 5.7 against
    // 5.25 ms for a second plan, 4.8 against 5.2 ms for an eighth; MP_SAOPD_CSR=1 buckets from the second plan on, =0 never)
    const bool csr = LDSD && p.prune && !(p.par_backup != 0 && A <= 32) && p.S <= 128 && root > 0 &&
                     (p.csr_old == 2 || (p.csr_old == 1 && root >= 3 * (1 + p.K * A)));
    if (csr) {
        for (int s = lane; s < p.S; s += 64) d_mark[s] = 0u;
        __syncthreads();
        for (int i0 = 0; i0 < root; i0 += 64) {
            const int i = i0 + lane;
            // (the state is range-checked: a planner whose earlier plan broke off -- every leaf pruned -- has rows nobody wrote)
            bool live = i < root && (ND(i).meta & SA_CHILDREN) != 0;
            const int st = live ? ST(i) : 0;
            live = live && (unsigned)st < (unsigned)p.S;
            if (live && i >= HD(st)) atomicAdd(&d_mark[st], 1u); // (rows in front of the list head were dropped by a root restart)
        }
        __syncthreads();
        const int c0 = lane < p.S ? (int)d_mark[lane] : 0, c1 = lane + 64 < p.S ? (int)d_mark[lane + 64] : 0;
        const int i0s = wave_incl_scan(c0);
        const int tot0 = __builtin_amdgcn_readlane(i0s, 63);
        const int i1s = wave_incl_scan(c1) + tot0;
        if (lane < p.S) { d_ls[lane] = make_int4(c0, i0s - c0, 0, 0); d_mark[lane] = 0u; }
        if (lane + 64 < p.S) { d_ls[lane + 64] = make_int4(c1, i1s - c1, 0, 0); d_mark[lane + 64] = 0u; }
        __syncthreads();
        for (int i0 = 0; i0 < root; i0 += 64) {
            const int i = i0 + lane;
            bool live = i < root && (ND(i).meta & SA_CHILDREN) != 0;
            const int st = live ? ST(i) : 0;
            live = live && (unsigned)st < (unsigned)p.S;
            if (live && i >= HD(st)) old_b[d_ls[st].y + (int)atomicAdd(&d_mark[st], 1u)] = make_int2(i, st);
        }
        __syncthreads();
        for (int s = lane; s < p.S; s += 64) d_mark[s] = 0xffffffffu;
        __syncthreads();
    } else if (p.prune) {
        const unsigned long long lt0 = (1ULL << lane) - 1ULL;
        for (int i0 = 0; i0 < root; i0 += 64) {
            const int i = i0 + lane;
            const bool live = i < root && (ND(i).meta & SA_CHILDREN) != 0;
            const unsigned long long bm = ballot64(live);
            if (live) old_b[n_old + __popcll(bm & lt0)] = make_int2(i, ST(i)); // (a row with children is never dead: no flag byte)
            n_old += __popcll(bm);
        }
        __syncthreads();
    }
    // Every state's node list a second time as CHUNKS of 15 ids + a link (16 ints = one 64-byte line): the backup below
    // reads a popped state's list 15 neighbours per load and evaluates their parents' backups in 15 lanes at once,
    // instead of chasing next_same one element per dependent round trip.  The linked lists stay the authority (the
    // other passes, the lane kernel and the roll-back use them): the chunks are rebuilt from them here, by every plan,
    // and kept up to date by this plan's appends.
    constexpr int CH = 15;
    const bool par_backup = p.par_backup != 0 && A <= 32; // a group of |A| lanes per list element, at least two groups per pass
    int4 *const ls_b = LDSD ? d_ls : p.lstate + sb; // one 16-byte record per state: one load / one store where three arrays took three
    int32_t *pool_b = p.lpool + (long)r * p.pool_ints;
    auto PL = [&](int chunk, int f) -> int32_t & { return pool_b[(chunk << 4) + f]; };
    int pool_top = 0;
    if (par_backup) {
        // one walk per state, the states spread over the lanes; chunks are handed out by a counter in LDS (the 512 bytes
        // after the tables), so a state's chunks need not be adjacent -- they are linked
        int *ctr = reinterpret_cast<int *>(lds_d + ntab);
        if (l0) *ctr = 0;
        __syncthreads();
        for (int s = lane; s < p.S; s += 64) {
            int c = 0, chunk = -1, kq = CH;
            int first = -1;
            for (int i = HD(s); i >= 0; i = ND(i).next_same) {
                if (kq == CH) {
                    const int nc = atomicAdd(ctr, 1);
                    if (chunk >= 0) PL(chunk, CH) = nc; else first = nc;
                    chunk = nc; kq = 0;
                }
                PL(chunk, kq) = i;
                ++kq; ++c;
            }
            if (chunk >= 0) PL(chunk, CH) = -1;
            ls_b[s] = make_int4(c, first, chunk, 0);
        }
        __syncthreads();
        pool_top = *ctr;
        __syncthreads();
    }
    int n_nodes = root + 1;
    int status = MP_OK;
    long steps_taken = 0, updates = 0;
#ifdef MP_PROFILE
    long long t_ph[5] = {0, 0, 0, 0, 0}, t_mark = clock64();
    long pf_nd = 0; int pf_fallback = 0, pf_ndmax = 0; long pf_reval = 0, pf_pass = 0, pf_pops = 0;
#define SA_PROF(i) do { const long long t_now = clock64(); t_ph[i] += t_now - t_mark; t_mark = t_now; } while (0)
#else
#define SA_PROF(i)
#endif

    for (int k = 0; k < p.K && status == MP_OK; ++k) {
        const int cur = p.iter_base + k;
        SA_PROF(4);
        // ---- max(leaves, key=U): 64 rows per trip, then (max U, lowest id) across the lanes
        double bu = ninf, b_lower = 0.0;
        uint32_t b_meta = 0;
        int32_t b_state = 0; // the lane's best leaf's record: the expansion takes the winner's from its lane, not from memory
        int leaf = 0x7fffffff;
        constexpr int WU = 4; // 4 x 64 rows per trip: the three levels of loads (node, state, state value) each in flight together
        for (int i0 = root + lane; i0 < n_nodes; i0 += 64 * WU) {
            SaNode nd[WU];
            int32_t st[WU];
            double sv[WU];
#pragma unroll
            for (int j = 0; j < WU; ++j) {
                const int i = min(i0 + 64 * j, n_nodes - 1);
                nd[j] = load_node(&ND(i));
                st[j] = ST(i);
            }
#pragma unroll
            for (int j = 0; j < WU; ++j) sv[j] = (nd[j].meta & SA_ALIVE) ? SV(st[j]) : 0.0;
#pragma unroll
            for (int j = 0; j < WU; ++j) {
                const int i = i0 + 64 * j;
                if (i < n_nodes && (nd[j].meta & SA_ALIVE)) {
                    const double u = nd[j].lower + gpow[nd[j].meta & SA_DEPTH] * sv[j];
                    if (u > bu || leaf == 0x7fffffff) { // ascending ids within a lane: first maximum
                        bu = u; leaf = i;
                        b_lower = nd[j].lower; b_meta = nd[j].meta; b_state = st[j];
                    }
                }
            }
        }
        wave_argmax(bu, leaf);
        if (leaf == 0x7fffffff) { // max() of an empty leaves list (see saopd_kernel: sticky in asynchronous mode)
            status = MP_ERR_ARG;
            if (p.sticky && lane == 0) p.overflow[1 + r] = MP_ERR_ARG;
            break;
        }
        SA_PROF(0);
        // ---- expand + update: one child per lane, then the list appends in action order
        const int owner = (leaf - root) & 63; // rows are dealt to the lanes round-robin from the root
        SaNode lf;
        lf.lower = bcast_lane(b_lower, owner); lf.next_same = -1; lf.meta = (uint32_t)__builtin_amdgcn_readlane((int)b_meta, owner);
        const int dl = (int)(lf.meta & SA_DEPTH);
        const int32_t sl = __builtin_amdgcn_readlane(b_state, owner);
        const int g = n_nodes;
        bool bad = false, term_c = false, real_c = false;
        int32_t s_c = 0;
        double c0_lower = ninf, c0_rew = 0.0; // the lane's child as stored (its parent's Bellman backup below needs no read-back)
        if (lane < A) {
            const Rec rc = p.rec[(long)sl * A + lane];
            // deterministic.py:32-35: the slot of an action state.get_available_actions() does not list is a PHANTOM row
            // (see saopd_kernel): lower = -inf, not alive, in no state's list, dead for the prune scan
            real_c = (rc.flags & 4u) != 0;
            term_c = real_c && (rc.flags & done_bit) != 0;
            bad = real_c && (!(0.0 <= rc.reward) || !(rc.reward <= 1.0));
            const int d = dl + 1;
            double lower = lf.lower + gpow[d - 1] * rc.reward;
            if (term_c) lower = lower + trg[d];
            if (!real_c) lower = -INFINITY;
            s_c = rc.next;
            const int c = g + lane;
            SaNode nd;
            nd.lower = lower; nd.next_same = -1; nd.meta = (real_c ? SA_ALIVE : 0u) | ((uint32_t)lane << SA_ACT_SHIFT) | (uint32_t)d;
            ND(c) = nd;
            c0_lower = lower; c0_rew = real_c ? rc.reward : 0.0;
            ST(c) = s_c; PA(c) = leaf; FC(c) = -1; RW(c) = c0_rew;
            p.done[nb + c] = real_c ? (term_c ? 1 : 0) : 2;
        }
        const unsigned long long real_mask = ballot64(real_c);
        if (l0) {
            ND(leaf).meta = (lf.meta & ~SA_ALIVE) | SA_CHILDREN;
            FC(leaf) = g;
        }
        steps_taken += __popcll(real_mask); // planner.step calls: one per listed action
        if (any64(bad)) { status = MP_ERR_REWARD_RANGE; break; }
        SA_SYNC();
        // state_nodes[str(observation)].append(child), update_value(observation, 0), child by child in action order.  What an
        // append reads -- its state's list tail, value (and chunk record) -- is fetched for ALL children at once, lane a
        // for child a (one round trip instead of |A|); the appends then run in order over registers, and a later sibling
        // in the SAME state gets the earlier one's result patched into its registers, as if it had read it.
        {
            int32_t my_t = -1;
            double my_sv = 0.0;
            int4 my_ls = make_int4(0, -1, -1, 0);
            if (lane < A) {
                my_t = TL(s_c);
                my_sv = SV(s_c);
                if (par_backup) my_ls = ls_b[s_c];
            }
            for (int a = 0; a < A; ++a) {
                if (!((real_mask >> a) & 1ULL)) continue; // a phantom slot joins no list and moves no state value (uniform)
                const int32_t s = __builtin_amdgcn_readlane(s_c, a);
                const bool term = __builtin_amdgcn_readlane((int)term_c, a) != 0;
                const int c = g + a;
                const int32_t t = __builtin_amdgcn_readlane(my_t, a);
                const double svs = bcast_lane(my_sv, a);
                const double sv_new = (term && svs - 0.0 > 0.0) ? 0.0 : svs;
                if (l0) {
                    if (t < 0) HD(s) = c; else ND(t).next_same = c;
                    TL(s) = c;
                    SM(s) = cur; // the state's list changed in this iteration: its leaves are prune candidates
                    if (term && svs - 0.0 > 0.0) SV(s) = 0.0;
                }
                int4 ls_new = make_int4(0, -1, -1, 0);
                if (par_backup) { // the same append on the chunked list
                    const int cnt = __builtin_amdgcn_readlane(my_ls.x, a), hd = __builtin_amdgcn_readlane(my_ls.y, a),
                              tl = __builtin_amdgcn_readlane(my_ls.z, a);
                    const bool fresh_chunk = cnt % CH == 0;
                    const int chunk = fresh_chunk ? pool_top : tl;
                    ls_new = make_int4(cnt + 1, cnt == 0 ? chunk : hd, chunk, 0);
                    if (l0) {
                        if (fresh_chunk) {
                            if (cnt != 0) PL(tl, CH) = chunk;
                            PL(chunk, CH) = -1;
                        }
                        PL(chunk, cnt % CH) = c;
                        ls_b[s] = ls_new;
                    }
                    pool_top += fresh_chunk ? 1 : 0;
                }
                if (lane > a && lane < A && s_c == s) { my_t = c; my_sv = sv_new; my_ls = ls_new; }
            }
            SA_ORDER();
        }
        n_nodes += A;
        SA_PROF(1);
        // ---- backup_to_root: uniform first-in-first-out loop, the |A| children of a popped node one per lane
        // (lazy queue of {state, node, delta} descriptors: see saopd_kernel)
        {
            unsigned qh = 0, qt = 0;
            {   // The first pop of backup_to_root is the expanded leaf itself: its children are still in this wave's registers
                // (lower bound, depth dl + 1, state, reward as stored above), so its Bellman backup reads the state values only
                // -- no trip through the queue, no read-back of the rows just written.
                double u = ninf, bk = 0.0;
                int a_id = 0x7fffffff;
                if (lane < A) {
                    const double svc = SV(s_c);
                    u = c0_lower + gpow[dl + 1] * svc;
                    bk = c0_rew + p.gamma * svc;
                    a_id = lane;
                }
                const double old = SV(sl);
                if (A <= 16) row0_argmax(u, a_id); else wave_argmax(u, a_id); // first maximal U in action order
                const double backup = __shfl(bk, a_id);
                const double delta = old - backup;
                ++updates;
                if (delta > 0.0) { // (an empty queue cannot be full)
                    if (l0) {
                        SV(sl) = backup; SM(sl) = cur;
                        QD4(qt) = make_int4(sl, leaf, __double2loint(delta), __double2hiint(delta));
                    }
                    ++qt;
                    SA_ORDER();
                }
            }
            if (!par_backup) {
                int src = -1, nbr = -1;
                double src_delta = 0.0;
                // The walk of a popped state's list is software-pipelined: the record and the parent of the NEXT list element
                // are requested as soon as the current element's link is known, so they arrive while the current element's
                // backup (two more round trips) is in flight.  Nothing the backup writes (state values, stamps, queue) is
                // part of a node record, so the early read sees what a late one would.
                SaNode nd_nbr;
                nd_nbr.lower = 0.0; nd_nbr.next_same = -1; nd_nbr.meta = 0;
                int par_nbr = -1;
                while ((nbr >= 0 || qh != qt) && status == MP_OK) {
                    int node = -1, group = -1; // group: first child of `node` when it is known without reading FC(node)
                    if (nbr < 0) { // front descriptor
                        const int4 dq = QD4(qh);
                        const int32_t ds = dq.x;
                        src = dq.y;
                        if (ds < 0) {
                            node = src;
                        } else {
                            src_delta = __hiloint2double(dq.w, dq.z);
                            nbr = HD(ds);
                            if (nbr >= 0) { nd_nbr = load_node(&ND(nbr)); par_nbr = PA(nbr); }
                        }
                        ++qh;
                    } else {       // one neighbour
                        const SaNode nd = nd_nbr;
                        const int par = par_nbr;
                        if (par >= 0 && (nbr == src || p.backup_aggregated) && src_delta > acc[nd.meta & SA_DEPTH]) {
                            node = par;
                            group = nbr - (int)((nd.meta & SA_ACT) >> SA_ACT_SHIFT); // the siblings of nbr = the children of par
                        }
                        nbr = nd.next_same;
                        if (nbr >= 0) { nd_nbr = load_node(&ND(nbr)); par_nbr = PA(nbr); }
                    }
                    if (node < 0) continue;
                    const int32_t sn = ST(node);
                    const int fc = group >= 0 ? group : FC(node);
                    if (fc >= 0) {
                        double u = ninf, bk = 0.0;
                        int a_id = 0x7fffffff;
                        if (lane < A) {
                            const int c = fc + lane;
                            const SaNode nd = load_node(&ND(c));
                            const double svc = SV(ST(c));
                            u = nd.lower + gpow[nd.meta & SA_DEPTH] * svc;
                            bk = RW(c) + p.gamma * svc;
                            a_id = lane;
                        }
                        const double old = SV(sn); // (requested with the children's values, not after the argmax)
                        if (A <= 16) row0_argmax(u, a_id); else wave_argmax(u, a_id); // first maximal U in action order
                        const double backup = __shfl(bk, a_id);
                        const double delta = old - backup;
                        ++updates;
                        if (delta > 0.0) {
                            if (qt - qh >= dcap) {
                                status = MP_ERR_ALLOC;
                                if (l0) { *p.overflow = 1; if (p.sticky) p.overflow[1 + r] = MP_ERR_ALLOC; }
                                break;
                            }
                            if (l0) {
                                SV(sn) = backup; SM(sn) = cur;
                                QD4(qt) = make_int4(sn, node, __double2loint(delta), __double2hiint(delta));
                            }
                            ++qt;
                            SA_ORDER();
                        }
                    }
                }
        
            } else {
                // ---- the backups of a popped state's neighbours, up to 64 / |A| at a time.  PARALLEL: a GROUP of |A| lanes takes
                // one list element -- all of them its record, its parent and the parent state's value (one line each per
                // group), lane a of the group child a of the parent (sibling group of the element: record, state, that
                // state's value, reward) -- and the group computes the parent's Bellman backup from the state values as
                // they are NOW: eight vector-memory instructions serve 12-16 neighbours where the element-by-element
                // loop issued eight per neighbour.  SEQUENTIAL, in list order: compare with the parent state's value,
                // count, write and push as the reference does.  A write changes ONE state value; it reaches a later group
                // either as its `old` (same parent state: patched in the register) or through one of its children's states
                // (each lane keeps its child's state: the group is re-evaluated when its turn comes).  Nothing else the
                // evaluation reads changes during a backup (node records, lists), so every group's numbers are those of
                // the reference's turn-by-turn loop.
                const int my_g = lane / A, my_a = lane - my_g * A, npp = 64 / A; // group, action, groups per pass
                const int g_lead = my_g * A;
                int v_node = -1, v_sn = -1, v_sc = -1;
                double v_old = 0.0, v_backup = 0.0;
                bool v_cond = false;
                // the lane's child as fetched: a re-evaluation (a state value moved) needs no memory access beyond sv[]
                double c_lower = 0.0, c_rew = 0.0;
                int c_depth = 0;
                // the group's Bellman backup from the state values as they are NOW (sv[] only)
                auto score = [&](bool mine) {
                    if (!mine) return;
                    v_old = SV(v_sn);
                    const double svc = SV(v_sc);
                    const double u = c_lower + gpow[c_depth] * svc;
                    const double bk = c_rew + p.gamma * svc;
                    double bu = ninf, bkb = 0.0; // first maximal U in action order, over the lanes of the group
                    for (int q = 0; q < A; ++q) {
                        const double uq = __shfl(u, g_lead + q), bq = __shfl(bk, g_lead + q);
                        if (q == 0 || uq > bu) { bu = uq; bkb = bq; }
                    }
                    v_backup = bkb;
                };
                // node_given >= 0: the descriptor names the node itself (the expanded leaf); otherwise the node is the parent
                // of list element `nbr`.  Every lane of a taking group runs this with the same nbr.
                auto eval = [&](bool mine, int nbr, int node_given, int src_, double src_delta_) {
                    if (!mine) return;
                    int node_ = node_given, fc = -1;
                    v_cond = false;
                    if (node_given >= 0) {
                        fc = FC(node_given);
                    } else {
                        const uint32_t nmeta = ND(nbr).meta; // (depth and action: four of the record's sixteen bytes)
                        const int par = PA(nbr);
                        if (par >= 0 && (nbr == src_ || p.backup_aggregated) && src_delta_ > acc[nmeta & SA_DEPTH]) {
                            node_ = par;
                            fc = nbr - (int)((nmeta & SA_ACT) >> SA_ACT_SHIFT); // the siblings of nbr = the children of par
                        }
                    }
                    if (node_ < 0 || fc < 0) return;
                    v_cond = true;
                    v_node = node_;
                    v_sn = ST(node_);
                    const int c = fc + my_a;
                    const SaNode cd = load_node(&ND(c));
                    v_sc = ST(c);
                    c_lower = cd.lower; c_depth = (int)(cd.meta & SA_DEPTH); c_rew = RW(c);
                    score(true);
                };
                // the sequential half over the group leaders in `todo` (ascending = list order)
                auto apply = [&](unsigned long long todo, int src_, double src_delta_, int my_nbr, int my_given) {
                    unsigned long long dirty = 0ULL;
                    const unsigned long long gmask = (A == 64 ? ~0ULL : ((1ULL << A) - 1ULL));
                    while (todo && status == MP_OK) {
                        const int j = __ffsll((long long)todo) - 1; // the group's first lane
                        todo &= todo - 1;
                        if (dirty & (gmask << j)) {
                            score(lane >= j && lane < j + A); // a child's state value moved
#ifdef MP_PROFILE
                            ++pf_reval;
#endif
                        }
                        const double backup = bcast_lane(v_backup, j), old = bcast_lane(v_old, j);
                        const int sn = __builtin_amdgcn_readlane(v_sn, j), node = __builtin_amdgcn_readlane(v_node, j);
                        const double delta = old - backup;
                        ++updates;
                        if (delta > 0.0) {
                            if (qt - qh >= dcap) {
                                status = MP_ERR_ALLOC;
                                if (l0) { *p.overflow = 1; if (p.sticky) p.overflow[1 + r] = MP_ERR_ALLOC; }
                                break;
                            }
                            if (l0) {
                                SV(sn) = backup; SM(sn) = cur;
                                QD4(qt) = make_int4(sn, node, __double2loint(delta), __double2hiint(delta));
                            }
                            ++qt;
                            SA_ORDER();
                            bool hit = false;
                            if (v_cond && lane >= j + A) {
                                if (v_sn == sn) v_old = backup;
                                hit = v_sc == sn;
                            }
                            dirty |= ballot64(hit);
                        }
                    }
                };
                // The same sequential half for ALL groups of a pass at once (LDSD: the dictionaries are in LDS).  What orders
                // the groups is (i) a later group whose parent is in a state an earlier group wrote -- its `old` is the
                // running minimum of that state's value -- and (ii) a later group with a CHILD in such a state (its backup is
                // stale).  (i): LDS atomics of one wave instruction apply in lane order (tools/lds_atomic_order.hip, and
                // tests/test_gpu_batch.py checks it on the box), so ONE ds_min_rtn_f64 of the group leaders onto sv[] hands
                // every leader the value the turn-by-turn loop would have read, and leaves the minimum.  (ii): the groups that
                // may write (old > backup with the unpatched old: a superset of the writers) mark their parent's state with
                // their group index; every lane looks its child's state up; the first group that finds an earlier mark ends the
                // batch: the groups before it are applied together (queue slots by rank among the writers), the marked ones
                // are re-evaluated, and the rest goes through the next round.
                auto apply_vec = [&](bool valid, int src_, double src_delta_, int my_nbr, int my_given) {
                    bool in_r = valid; // lanes of the groups not yet applied
                    const unsigned long long gmask = (A == 64 ? ~0ULL : ((1ULL << A) - 1ULL));
                    const unsigned long long ltm = (1ULL << lane) - 1ULL;
                    while (any64(in_r) && status == MP_OK) {
                        const bool leader = in_r && my_a == 0;
                        const bool cand = leader && v_old > v_backup;
                        if (cand) atomicMin(&d_mark[v_sn], (uint32_t)my_g);
                        SA_ORDER();
                        const uint32_t m = in_r ? d_mark[v_sc] : 0xffffffffu;
                        const unsigned long long stale = ballot64(m < (uint32_t)my_g);
                        const int c_g = stale ? (__ffsll((long long)stale) - 1) / A : 64; // the first group with a stale child value
                        const bool now = in_r && my_g < c_g;
                        double ret = 0.0;
                        bool w = false;
                        if (now && cand) {
                            ret = __hip_atomic_fetch_min(&sv_b[v_sn], v_backup, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            w = v_backup < ret;
                        }
                        const unsigned long long wm = ballot64(w);
                        const int n_w = __popcll(wm);
                        updates += __popcll(ballot64(now && leader));
                        if (n_w) {
                            if (qt - qh + (unsigned)n_w > dcap) {
                                status = MP_ERR_ALLOC;
                                if (l0) { *p.overflow = 1; if (p.sticky) p.overflow[1 + r] = MP_ERR_ALLOC; }
                            } else if (w) {
                                const double delta = ret - v_backup;
                                SM(v_sn) = cur;
                                QD4(qt + (unsigned)__popcll(wm & ltm)) = make_int4(v_sn, v_node, __double2loint(delta), __double2hiint(delta));
                            }
                            qt += (unsigned)n_w;
                        }
                        if (cand) d_mark[v_sn] = 0xffffffffu;
                        SA_ORDER();
                        in_r = in_r && my_g >= c_g;
                        if (!stale) break;
                        // the remaining groups: re-evaluate those with a child in a state a group applied above may have
                        // written; the others re-read their parent state's value (it may have moved)
                        const unsigned long long nm = ballot64(in_r && m < (uint32_t)c_g);
                        const bool redo = in_r && ((nm >> g_lead) & gmask) != 0ULL;
#ifdef MP_PROFILE
                        pf_reval += __popcll(ballot64(redo && my_a == 0));
#endif
                        score(redo);
                        if (in_r && !redo) v_old = SV(v_sn);
                    }
                };
                const int nq_max = npp < 16 ? npp : 16;
                while (qh != qt && status == MP_OK) {
                    // ---- several pending descriptors per pass.  Early in a plan a state's list holds a handful of nodes, so
                    // one descriptor fills three or four of the 64 / |A| groups; the descriptors behind it are already in
                    // the queue (first in, first out: whatever this pass pushes comes after them), so the longest run of
                    // leading descriptors whose lists are single chunks and fit the groups together is evaluated in ONE
                    // pass and applied in queue order, list order within a descriptor -- the order of the
                    // element-by-element loop, with the same register patching when a state value moves.
                    const int nq = (int)(qt - qh) < nq_max ? (int)(qt - qh) : nq_max;
                    int4 dq_l = make_int4(-1, -1, 0, 0);
                    int cnt_l = 0, head_l = -1;
                    if (lane < nq) {
                        dq_l = QD4(qh + lane);
                        cnt_l = 1; // (ds < 0: the expanded leaf itself is one item)
                        if (dq_l.x >= 0) { const int4 ls = ls_b[dq_l.x]; cnt_l = ls.x; head_l = ls.y; }
                    }
                    int incl = cnt_l; // inclusive prefix sum of the item counts over the descriptor lanes
                    for (int d = 1; d < 16; d <<= 1) {
                        const int o = __shfl_up(incl, d);
                        if (lane >= d) incl += o;
                    }
                    const unsigned long long okm = ballot64(lane < nq && cnt_l <= CH && incl <= npp);
                    const int nb = okm == ~0ULL ? 64 : __ffsll((long long)~okm) - 1; // leading descriptors of the batch
                    if (nb > 0) {
                        const int total = __builtin_amdgcn_readlane(incl, nb - 1);
                        int own = 0; // the descriptor group my_g's item belongs to
                        for (int i = 0; i < nb; ++i) own += __shfl(incl, i) <= my_g ? 1 : 0;
                        own = own < nb ? own : nb - 1;
                        const int incl_prev = __shfl(incl, (own - 1) & 63); // (unconditional: a cross-lane read under a divergent
                        const int before = own ? incl_prev : 0;            //  branch would read lanes that sit the branch out)
                        const int ds_o = __shfl(dq_l.x, own), src_o = __shfl(dq_l.y, own), head_o = __shfl(head_l, own);
                        const double delta_o = __hiloint2double(__shfl(dq_l.w, own), __shfl(dq_l.z, own));
                        const bool mine = my_g < total && my_g < npp;
                        const int my_nbr = (mine && ds_o >= 0) ? PL(head_o, my_g - before) : -1;
                        const int my_given = (mine && ds_o < 0) ? src_o : -1;
                        qh += nb;
                        v_cond = false;
#ifdef MP_PROFILE
                        ++pf_pass;
#endif
                        eval(mine, my_nbr, my_given, ds_o < 0 ? -1 : src_o, delta_o);
                        if constexpr (LDSD) apply_vec(mine && v_cond, ds_o < 0 ? -1 : src_o, delta_o, my_nbr, my_given);
                        else apply(ballot64(mine && v_cond && my_a == 0), ds_o < 0 ? -1 : src_o, delta_o, my_nbr, my_given);
                        continue;
                    }
                    // the first pending descriptor alone is longer than a pass: chunk by chunk
                    const int4 dq = QD4(qh);
                    ++qh;
                    const int src_ = dq.y;
                    const double src_delta_ = __hiloint2double(dq.w, dq.z);
                    const int4 ls = ls_b[dq.x];
                    int remaining = ls.x;
                    int w = (remaining > 0 && lane <= CH) ? PL(ls.y, lane) : -1; // 15 ids + the link in one 64-byte read
                    while (remaining > 0 && status == MP_OK) {
                        const int here = remaining < CH ? remaining : CH;
                        const int nxt = __builtin_amdgcn_readlane(w, CH);
                        const int w_now = w;
                        remaining -= here;
                        if (remaining > 0) w = lane <= CH ? PL(nxt, lane) : -1; // the next chunk arrives under this one's work
                        for (int t0 = 0; t0 < here && status == MP_OK; t0 += npp) {
                            const int t = t0 + my_g;
                            const int nb_t = __shfl(w_now, t & 63);
                            const int my_nbr = (my_g < npp && t < here) ? nb_t : -1;
                            v_cond = false;
#ifdef MP_PROFILE
                            ++pf_pass;
#endif
                            eval(my_nbr >= 0, my_nbr, -1, src_, src_delta_);
                            if constexpr (LDSD) apply_vec(my_nbr >= 0 && v_cond, src_, src_delta_, my_nbr, -1);
                            else apply(ballot64(my_nbr >= 0 && v_cond && my_a == 0), src_, src_delta_, my_nbr, -1);
                        }
                    }
                }
            }
        }
        if (status != MP_OK) break;
        SA_PROF(2);
        // ---- prune: candidate leaves (alive, state changed this iteration) 64 rows per trip in reverse order, each
        // candidate's list walked as uniform code
        // Leaves of different states never interact in the pass, and within one state the pass order is descending id:
        // one changed state per lane -- pass A stacks the alive leaves of its list (ascending ids) in the lane's slice
        // of the idle queue buffer, pass B pops them (descending) and walks the list for a dominator.
        bool serial_prune = p.prune != 0;
        // Prune as ONE scan of the arena instead of list walks.  A leaf's test reads its state's value and node list; only
        // leaves of states whose value or list changed in this iteration can newly fail it (stamp == cur).  Walking those
        // lists is a chain of dependent reads (17-24 links on average, 84 at most on the reference's grid, 1.6-1.9 alive
        // leaves per changed state each walking it again: 65-75 % of a later plan's time).  The rows of the changed
        // states are instead COLLECTED by a coalesced scan of state[] over the whole arena (independent reads: 24-40 per
        // lane, all in flight) into the idle queue buffer, in id order; at most 256 of them (4.3 changed states x ~20
        // rows per iteration on the grid: 80-100 on average, 240 at most) are then held one per lane in four register sets, and every candidate leaf --
        // descending id, as the reference's reversed leaves list -- is tested against all of them at once with one
        // ballot per set; alive flags are updated in the registers (a pruned leaf stops dominating later candidates)
        // and written back at the end.  More rows than that, or the test knob, take the serial pass below.
        if (p.prune && p.scap >= 2) {
            int32_t *recs = queue_b; // {row, state} pairs of the rows whose state is dirty
            constexpr int NS = 4; // register sets of 64 rows
            const int pcap = (qcap - 64 * NS) >> 1; // pairs that fit in front of the per-state selection area
            int n_d = 0;
            const unsigned long long lt = (1ULL << lane) - 1ULL;
            if (csr) { // the old rows of the CHANGED states: straight from their buckets into the pairs, state by state
                const int4 b0 = lane < p.S ? d_ls[lane] : make_int4(0, 0, 0, 0), b1 = lane + 64 < p.S ? d_ls[lane + 64] : make_int4(0, 0, 0, 0);
                const bool h0 = lane < p.S && b0.x > 0 && SM(lane) == cur, h1 = lane + 64 < p.S && b1.x > 0 && SM(lane + 64) == cur;
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    unsigned long long todo = ballot64(half ? h1 : h0);
                    while (todo) {
                        const int l = __ffsll((long long)todo) - 1;
                        todo &= todo - 1;
                        const int cnt = __builtin_amdgcn_readlane(half ? b1.x : b0.x, l), off = __builtin_amdgcn_readlane(half ? b1.y : b0.y, l);
                        const int st = l + 64 * half;
                        for (int k = lane; k < cnt; k += 64)
                            if (n_d + k < pcap) { recs[2 * (n_d + k)] = old_b[off + k].x; recs[2 * (n_d + k) + 1] = st; }
                        n_d += cnt;
                    }
                }
            }
            // scan positions: the listed old rows, then the rows of this plan (root .. n_nodes - 1)
            const int n_pos = n_old + (n_nodes - root);
            // The register sets: rst[q] = the row's state (-1: none), ids[q] = its row, rows in ascending id over (set, lane).
            // key = the row's depth while it can dominate (alive or with children), -1 otherwise.
            auto process_sets = [&](const int (&ids)[NS], int (&rst)[NS]) {
                int key[NS];
                uint32_t rmeta[NS];
                double rval[NS];
                uint32_t changed = 0, nonempty = 0;
#pragma unroll
                for (int q = 0; q < NS; ++q) {
                    rmeta[q] = 0; rval[q] = ninf; key[q] = -1;
                    if (!any64(rst[q] >= 0)) continue; // (uniform: an empty set costs nothing)
                    nonempty |= 1u << q;
                    if (rst[q] >= 0) {
                        const SaNode nd = load_node(&ND(ids[q]));
                        rmeta[q] = nd.meta;
                        rval[q] = nd.lower + gpow[nd.meta & SA_DEPTH] * SV(rst[q]);
                        key[q] = (nd.meta & (SA_CHILDREN | SA_ALIVE)) ? (int)(nd.meta & SA_DEPTH) : -1;
                    }
                }
                // candidates: alive rows, descending id = the last set from its last lane down, then the sets before it.
                // One candidate against one set = three compares (state, value, depth-or-dead).
#pragma unroll
                for (int q = NS - 1; q >= 0; --q) {
                    if (!(nonempty & (1u << q))) continue;
                    unsigned long long todo = ballot64(rst[q] >= 0 && (rmeta[q] & SA_ALIVE));
                    while (todo) {
                        const int l = 63 - __clzll((long long)todo);
                        todo &= ~(1ULL << l);
                        const int cs = __builtin_amdgcn_readlane(rst[q], l);
                        const int cd = (int)(__builtin_amdgcn_readlane((int)rmeta[q], l) & SA_DEPTH);
                        const double cv = bcast_lane(rval[q], l);
                        unsigned long long dom = 0ULL;
#pragma unroll
                        for (int t = 0; t < NS; ++t) {
                            if (!(nonempty & (1u << t))) continue;
                            unsigned long long m = ballot64(rst[t] == cs && rval[t] >= cv && key[t] >= cd);
                            if (t == q) m &= ~(1ULL << l);
                            dom |= m;
                        }
                        if (dom && lane == l) { rmeta[q] &= ~SA_ALIVE; key[q] = -1; changed |= 1u << q; }
                    }
                }
#pragma unroll
                for (int q = 0; q < NS; ++q)
                    if (changed & (1u << q)) { // a pruned leaf: not alive, no children -- dead from now on
                        const int i = ids[q];
                        ND(i).meta = rmeta[q];
                        p.done[nb + i] = (uint8_t)(p.done[nb + i] | 2);
                    }
            };
            // (Tried: hits left where a single-trip scan found them, no compaction through memory -- slower, 6.05 -> 6.45 ms for
            // light planners: uncompacted, the ~30 rows occupy all four register sets and every candidate pays four compares.)
            for (int i0 = 0; i0 < n_pos; i0 += 64 * WU) {
                int32_t rows[WU], sts[WU], stamps[WU], heads[WU];
                uint8_t dead[WU];
#pragma unroll
                for (int j = 0; j < WU; ++j) {
                    const int q = min(i0 + 64 * j + lane, n_pos - 1);
                    rows[j] = root + (q - n_old); sts[j] = 0; dead[j] = 0;
                    if (q < n_old) { const int2 o = old_b[q]; rows[j] = o.x; sts[j] = o.y; }
                }
#pragma unroll
                for (int j = 0; j < WU; ++j)
                    if (min(i0 + 64 * j + lane, n_pos - 1) >= n_old) {
                        sts[j] = ST(rows[j]);
                        dead[j] = p.done[nb + rows[j]];
                    }
#pragma unroll
                for (int j = 0; j < WU; ++j) { stamps[j] = SM(sts[j]); heads[j] = HD(sts[j]); }
#pragma unroll
                for (int j = 0; j < WU; ++j) {
                    const int i = rows[j];
                    // a state's list = its rows from the list head on (lists are in id order; plan() starts the root
                    // state's list over, which drops that state's older rows)
                    const bool hit = i0 + 64 * j + lane < n_pos && stamps[j] == cur && i >= heads[j] && !(dead[j] & 2);
                    const unsigned long long bm = ballot64(hit);
                    if (hit) {
                        const int pos = n_d + __popcll(bm & lt);
                        if (pos < pcap) { recs[2 * pos] = i; recs[2 * pos + 1] = sts[j]; }
                    }
                    n_d += __popcll(bm);
                }
            }
#ifdef MP_PROFILE
            pf_nd += n_d; pf_fallback += n_d > 64 * NS ? 1 : 0; pf_ndmax = n_d > pf_ndmax ? n_d : pf_ndmax;
#endif
            if (n_d <= pcap) {
                SA_SYNC(); // the pairs are read back by other lanes
                // One group of rows (record j: row id_at(j), state st_at(j), j < cnt <= 64 NS, ascending id) through the NS
                // register sets: record lane + 64 q; a record is identified by (set, lane).
                auto process = [&](int cnt, auto id_at, auto st_at) {
                    int ids[NS], rst[NS];
#pragma unroll
                    for (int q = 0; q < NS; ++q) {
                        const int j = lane + 64 * q;
                        ids[q] = 0; rst[q] = -1;
                        if (j < cnt) { ids[q] = id_at(j); rst[q] = st_at(j); }
                    }
                    process_sets(ids, rst);
                };
                if (n_d <= p.prune_rows) { // the usual case: every changed state at once
                    process(n_d, [&](int j) { return recs[2 * j]; }, [&](int j) { return recs[2 * j + 1]; });
                    serial_prune = false;
                } else {
                    // long episodes (lists grow with every plan): one changed state per round through the same register
                    // sets -- its rows are picked out of the pairs in order and marked done; only rows that do not fit the
                    // scratch buffer take the serial pass
                    int32_t *sel = recs + 2 * n_d;
                    const int selcap = ((qcap - 2 * n_d) >> 2) & ~1; // room for {row, meta, value} of one state's rows
                    // One state with more listed rows than the register sets hold (an agent that stays in a small region
                    // for many plans): its rows' {meta, value} are computed once into the scratch area and every
                    // candidate streams them back in coalesced chunks -- still no list walk.
                    auto process_long = [&](int cnt, int st_u) {
                        int32_t *mt = sel + selcap;
                        double *vl = reinterpret_cast<double *>(mt + selcap);
                        const double svs = SV(st_u);
                        for (int j = lane; j < cnt; j += 64) {
                            const SaNode nd = load_node(&ND(sel[j]));
                            mt[j] = (int32_t)nd.meta;
                            vl[j] = nd.lower + gpow[nd.meta & SA_DEPTH] * svs;
                        }
                        __syncthreads();
                        for (int c0 = (cnt - 1) & ~63; c0 >= 0; c0 -= 64) {
                            const int jc = c0 + lane;
                            const uint32_t mc = jc < cnt ? (uint32_t)mt[jc] : 0u;
                            const double vc = jc < cnt ? vl[jc] : ninf;
                            unsigned long long todo = ballot64(jc < cnt && (mc & SA_ALIVE));
                            while (todo) {
                                const int l = 63 - __clzll((long long)todo);
                                todo &= ~(1ULL << l);
                                const int cd = (int)(__builtin_amdgcn_readlane((int)mc, l) & SA_DEPTH);
                                const double cv = bcast_lane(vc, l);
                                bool dom = false;
                                for (int t0 = 0; t0 < cnt; t0 += 64) {
                                    const int jt = t0 + lane;
                                    if (jt < cnt && jt != c0 + l) {
                                        const uint32_t m2 = (uint32_t)mt[jt];
                                        dom |= vl[jt] >= cv && (int)(m2 & SA_DEPTH) >= cd && (m2 & (SA_CHILDREN | SA_ALIVE)) != 0;
                                    }
                                }
                                if (any64(dom)) {
                                    if (lane == l) {
                                        const uint32_t nm = (uint32_t)mt[jc] & ~SA_ALIVE;
                                        mt[jc] = (int32_t)nm;
                                        const int i = sel[jc];
                                        ND(i).meta = nm;
                                        p.done[nb + i] = (uint8_t)(p.done[nb + i] | 2);
                                    }
                                    __syncthreads(); // later candidates read the flag through memory
                                }
                            }
                        }
                    };
                    bool too_long = false;
                    int start = 0;
                    for (;;) {
                        int pj = -1, ps = -1;
                        for (int j0 = start; j0 < n_d; j0 += 64) {
                            const int j = j0 + lane;
                            const int stj = j < n_d ? recs[2 * j + 1] : -1;
                            const unsigned long long bm = ballot64(stj >= 0);
                            if (bm) {
                                const int l = __ffsll((long long)bm) - 1;
                                pj = j0 + l; ps = __builtin_amdgcn_readlane(stj, l);
                                break;
                            }
                            start = j0 + 64;
                        }
                        if (pj < 0) break;
                        int m = 0;
                        for (int j0 = pj & ~63; j0 < n_d; j0 += 64) {
                            const int j = j0 + lane;
                            const int stj = j < n_d ? recs[2 * j + 1] : -1;
                            const bool hit = stj == ps;
                            const unsigned long long bm = ballot64(hit);
                            if (hit) {
                                const int pos = m + __popcll(bm & lt);
                                if (pos < selcap) sel[pos] = recs[2 * j];
                                recs[2 * j + 1] = -1 - stj;
                            }
                            m += __popcll(bm);
                        }
                        if (m > selcap) { too_long = true; break; }
                        __syncthreads();
                        if (m <= p.prune_rows) process(m, [&](int j) { return sel[j]; }, [&](int) { return ps; });
                        else process_long(m, ps);
                        __syncthreads();
                    }
                    serial_prune = too_long;
                }
            }
            SA_SYNC();
        }
        if (serial_prune)
            for (int ib = n_nodes - 1; ib >= root; ib -= 64 * WU) {
              // candidate flags of 4 x 64 rows with the loads of each level in flight together
              bool cands[WU];
              {
                uint32_t ms[WU];
                int32_t sts[WU], stamps[WU];
#pragma unroll
                for (int j = 0; j < WU; ++j) {
                    const int i = max(ib - 64 * j - lane, root);
                    ms[j] = ND(i).meta;
                    sts[j] = ST(i);
                }
#pragma unroll
                for (int j = 0; j < WU; ++j) stamps[j] = (ms[j] & SA_ALIVE) ? SM(sts[j]) : -1;
#pragma unroll
                for (int j = 0; j < WU; ++j) cands[j] = ib - 64 * j - lane >= root && (ms[j] & SA_ALIVE) && stamps[j] == cur;
              }
#pragma unroll
              for (int jc = 0; jc < WU; ++jc) {
                const int i0 = ib - 64 * jc;
                if (i0 < root) break;
                unsigned long long todo = ballot64(cands[jc]);
                while (todo) {
                    const int j = __ffsll((long long)todo) - 1; // lane 0 holds the highest row of the chunk
                    todo &= todo - 1;
                    const int i = i0 - j;
                    const SaNode me = ND(i);
                    const int32_t s = ST(i);
                    const double svs = SV(s);
                    const int dm = (int)(me.meta & SA_DEPTH);
                    const double vub = me.lower + gpow[dm] * svs;
                    for (int nd_i = HD(s); nd_i >= 0;) {
                        const SaNode nd = ND(nd_i);
                        const int dn = (int)(nd.meta & SA_DEPTH);
                        if (nd_i != i && nd.lower + gpow[dn] * svs >= vub && dn >= dm && (nd.meta & (SA_CHILDREN | SA_ALIVE))) {
                            if (l0) { ND(i).meta = me.meta & ~SA_ALIVE; p.done[nb + i] = (uint8_t)(p.done[nb + i] | 2); }
                            break;
                        }
                        nd_i = nd.next_same;
                    }
                    __syncthreads();
                }
              }
            }
        SA_PROF(3);
    }
#ifdef MP_PROFILE
    if (r == 0 && l0)
        printf("saopd prof planner0: leaf-scan %lld  expand+append %lld  backup %lld  prune %lld  other %lld (clock64 ticks), nodes %d..%d\n",
               t_ph[0], t_ph[1], t_ph[2], t_ph[3], t_ph[4], root, n_nodes);
    if (r == 0 && l0) printf("saopd prof planner0: backup passes %ld, re-evaluated groups %ld, updates %ld\n", pf_pass, pf_reval, updates);
    if (r == 0 && l0) printf("saopd prof planner0: rows of dirty states %ld over %d iterations (max %d), serial fallbacks %d\n", pf_nd, p.K, pf_ndmax, pf_fallback);
#endif
    if (status != MP_OK) {
        // A plan that broke off (every leaf pruned, a reward out of range, a full queue) leaves the rest of its row range
        // unwritten, and the host counts the whole range as used: later plans of the batch scan it.  Make those rows what a
        // phantom is -- no children, not alive, dead for the prune scan, state 0.
        for (int i = n_nodes + lane; i < root + 1 + p.K * A; i += 64) {
            SaNode nd;
            nd.lower = ninf; nd.next_same = -1; nd.meta = 0;
            ND(i) = nd;
            ST(i) = 0; PA(i) = -1; FC(i) = -1; RW(i) = 0.0;
            p.done[nb + i] = 2;
        }
    }
    // ---- get_plan, twice (see saopd_kernel), uniform
    int len = 0;
    if (status == MP_OK) {
        Pcg64 gen;
        gen.load(p.rng + (long)r * 6);
        for (int pass = 0; pass < 2; ++pass) {
            int node = root;
            len = 0;
            int fc = FC(node);
            while (fc >= 0) {
                const double l = lane < A ? ND(fc + lane).lower : ninf;
                const double m = A <= 16 ? row0_max(l) : wave_max(l);
                const unsigned long long ties = ballot64(lane < A && l == m);
                const int nt = __popcll(ties);
                int pick = nt > 1 ? (int)gen.below((uint32_t)nt) : 0;
                unsigned long long t = ties;
                while (pick-- > 0) t &= t - 1;
                const int act = __ffsll((long long)t) - 1;
                if (pass == 1 && l0 && p.plans && len < p.max_plan_len) p.plans[(long)r * p.max_plan_len + len] = act;
                ++len;
                node = fc + act;
                fc = FC(node);
            }
        }
        if (l0) gen.store(p.rng + (long)r * 6);
    }
    if (l0) {
        if (p.plans)
            for (int i = len; i < p.max_plan_len; ++i) p.plans[(long)r * p.max_plan_len + i] = -1;
        if (p.plan_len) p.plan_len[r] = len;
        if (p.status) p.status[r] = status;
        if (p.env_steps) p.env_steps[r] = steps_taken;
        if (p.updates) p.updates[r] = updates;
        if (p.cost) p.cost[r] = (int32_t)(updates < 0x7fffffffL ? updates : 0x7fffffffL) + p.K; // (>= 1 once planned)
    }
    if (LDSR) { // write back: every node record (flags and list links of older rows change too), this plan's rows, the dictionaries
        __syncthreads();
        const int n_wb = status == MP_OK ? n_nodes : root + 1 + p.K * A; // (a broken-off plan: with the rows made harmless above)
        for (int i = lane; i < n_wb; i += 64) p.node[nb + i] = l_node[i];
        for (int i = root + lane; i < n_wb; i += 64) {
            p.state[nb + i] = l_state[i]; p.parent[nb + i] = l_parent[i]; p.first_child[nb + i] = l_fc[i];
            p.reward[nb + i] = l_reward[i];
        }
        for (int i = p.prev_root + lane; i < root; i += 64) p.first_child[nb + i] = l_fc[i]; // (unchanged; kept simple)
        for (int s = lane; s < p.S; s += 64) {
            p.sv[sb + s] = l_sv[s]; p.head[sb + s] = l_head[s]; p.tail[sb + s] = l_tail[s]; p.stamp[sb + s] = l_stamp[s];
        }
    }
    if (LDSD) {
        __syncthreads();
        for (int s = lane; s < p.S; s += 64) {
            p.sv[sb + s] = d_sv[s]; p.head[sb + s] = d_head[s]; p.tail[sb + s] = d_tail[s]; p.stamp[sb + s] = d_stamp[s];
        }
    }
}

// ---- dispatch order of the wave kernel: planners by expected cost, longest first (a counting sort over 1 024 cost
// buckets by ONE workgroup; the order inside a bucket is whatever the atomics make it -- the plans do not depend on it).
// key[r] = by_state ? model_cost[root_state[r]] : cost[r]
__global__ __launch_bounds__(1024) void saopd_order_kernel(int n, int S, const int32_t *cost, const int32_t *by_state, const int32_t *root_state,
                                                           int32_t *order)
{
    __shared__ int hist[1024];
    __shared__ int kmax;
    const int t = threadIdx.x;
    hist[t] = 0;
    if (t == 0) kmax = 0;
    __syncthreads();
    // (root states of device arrays cannot be checked on the host: a state outside [0, S) -- which the plan kernel reports in
    // `status` -- sorts as "nothing known" instead of reading beyond the [S] table)
    auto key = [&](int r) {
        if (!by_state) return cost[r];
        const int rs = root_state[r];
        return (unsigned)rs < (unsigned)S ? by_state[rs] : 0;
    };
    int m = 0;
    for (int r = t; r < n; r += 1024) m = max(m, key(r));
    atomicMax(&kmax, m);
    __syncthreads();
    const long km = kmax;
    if (km <= 0) { // nothing known: index order
        for (int r = t; r < n; r += 1024) order[r] = r;
        return;
    }
    auto bucket = [&](int r) { const long k = key(r); return 1023 - (int)((k < 0 ? 0 : k) * 1023 / km); }; // bucket 0 = the longest
    for (int r = t; r < n; r += 1024) atomicAdd(&hist[bucket(r)], 1);
    __syncthreads();
    if (t == 0) {
        int acc = 0;
        for (int b = 0; b < 1024; ++b) { const int c = hist[b]; hist[b] = acc; acc += c; }
    }
    __syncthreads();
    for (int r = t; r < n; r += 1024) order[atomicAdd(&hist[bucket(r)], 1)] = r;
}

// what the first plan of a fresh planner cost, remembered by root state with the model
__global__ __launch_bounds__(256) void saopd_learn_kernel(int n, int S, const int32_t *root_state, const int32_t *cost, const int32_t *status,
                                                          int32_t *model_cost)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n || (status && status[r] != MP_OK)) return;
    const int rs = root_state[r];
    if ((unsigned)rs < (unsigned)S) model_cost[rs] = cost[r]; // (never a store beyond the [S] table, whatever the status says)
}

// grow a node array from old_cap to new_cap rows per planner, keeping the used_rows rows in use
// undo the list appends of a rolled-back plan: after the tails are restored, every tail is the end of its list again
// host mode, no room left to grow the queue: the planners that still report MP_ERR_ALLOC stay failed
__global__ __launch_bounds__(64) void saopd_mark_failed_kernel(int n, const int32_t *status, int32_t *overflow)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < n && status[r] == MP_ERR_ALLOC) overflow[1 + r] = MP_ERR_ALLOC;
}

__global__ __launch_bounds__(64) void saopd_fix_tails_kernel(int n, int S, long node_si, long node_sr, long state_si, long state_sr,
                                                             const int32_t *tail, SaNode *node)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)n * S) return;
    const long r = i / S, st = i % S;
    const int32_t t = tail[st * state_si + r * state_sr];
    if (t >= 0) node[(long)t * node_si + r * node_sr].next_same = -1;
}

// device blocks of a planner batch: through the ctx's block cache (common.hpp), remembered with their sizes
template <typename T>
static hipError_t sa_alloc(mp_saopd *pl, T **out, size_t bytes)
{
    void *p = nullptr;
    size_t got = bytes;
    const hipError_t e = ctx_block_alloc(pl->ctx, &p, bytes, &got);
    if (e != hipSuccess) return e;
    pl->blocks.push_back({p, got});
    *out = static_cast<T *>(p);
    return hipSuccess;
}
static void sa_release(mp_saopd *pl, void *p)
{
    if (!p) return;
    for (size_t i = 0; i < pl->blocks.size(); ++i)
        if (pl->blocks[i].p == p) {
            ctx_block_release(pl->ctx, p, pl->blocks[i].bytes);
            pl->blocks[i] = pl->blocks.back();
            pl->blocks.pop_back();
            return;
        }
    (void)hipFree(p);
}

template <typename T>
static int grow_rows(mp_saopd *pl, T **buf, size_t used_rows, size_t old_cap, size_t new_cap, size_t n, bool planner_major, hipStream_t st)
{
    T *nw = nullptr;
    MP_HIP(sa_alloc(pl, &nw, new_cap * n * sizeof(T)));
    if (*buf) {
        if (planner_major) { // [planner][row]: every planner's prefix moves to its new pitch
            if (used_rows)
                MP_HIP(hipMemcpy2DAsync(nw, new_cap * sizeof(T), *buf, old_cap * sizeof(T), used_rows * sizeof(T), n,
                                        hipMemcpyDeviceToDevice, st));
        } else {             // [row][planner]: one contiguous prefix
            MP_HIP(hipMemcpyAsync(nw, *buf, used_rows * n * sizeof(T), hipMemcpyDeviceToDevice, st));
        }
        MP_HIP(hipStreamSynchronize(st));
        sa_release(pl, *buf);
    }
    *buf = nw;
    return MP_OK;
}

} // namespace mp

using namespace mp;

extern "C" {

int mp_saopd_create(mp_ctx *ctx, mp_model *model, int32_t n_planners, mp_saopd **out)
{
    if (!ctx || !model || !out) return fail(MP_ERR_ARG, "mp_saopd_create: NULL argument");
    if (model->mode != MP_MODE_DETERMINISTIC || !model->rec)
        return fail(MP_ERR_MODE, "mp_saopd_create: state-aware planning needs a deterministic table model");
    if (n_planners < 1) return fail(MP_ERR_ARG, "mp_saopd_create: n_planners = %d", n_planners);
    MP_HIP(hipSetDevice(ctx->device));
    mp_saopd *pl = new (std::nothrow) mp_saopd;
    if (!pl) return fail(MP_ERR_ALLOC, "mp_saopd_create: out of memory");
    pl->ctx = ctx; pl->model = model; pl->model_serial = model->serial; pl->n = n_planners; pl->S = model->S; pl->A = model->A;
    // one planner per wavefront unless asked otherwise (MP_SAOPD_MODEL=lane: one per lane, the first implementation)
    const char *force = getenv("MP_SAOPD_MODEL");
    pl->wave = !(force && force[0] == 'l');
    if (pl->wave && model->A > 64) pl->wave = 0;
    const size_t sn = (size_t)pl->S * pl->n;
    if (sa_alloc(pl, &pl->sv, sn * 8) != hipSuccess || sa_alloc(pl, &pl->head, sn * 4) != hipSuccess ||
        sa_alloc(pl, &pl->tail, sn * 4) != hipSuccess || sa_alloc(pl, &pl->stamp, sn * 4) != hipSuccess ||
        sa_alloc(pl, &pl->snap_sv, sn * 8) != hipSuccess || sa_alloc(pl, &pl->snap_head, sn * 4) != hipSuccess ||
        sa_alloc(pl, &pl->snap_tail, sn * 4) != hipSuccess || sa_alloc(pl, &pl->snap_stamp, sn * 4) != hipSuccess ||
        sa_alloc(pl, &pl->snap_rng, (size_t)pl->n * 6 * 8) != hipSuccess || sa_alloc(pl, &pl->overflow, 4 * (size_t)(1 + pl->n)) != hipSuccess ||
        sa_alloc(pl, &pl->lstate, sn * 16) != hipSuccess || sa_alloc(pl, &pl->cost, 4 * (size_t)pl->n) != hipSuccess ||
        sa_alloc(pl, &pl->order, 4 * (size_t)pl->n) != hipSuccess) {
        mp_saopd_free(pl);
        return fail(MP_ERR_ALLOC, "mp_saopd_create: device allocation failed (%zu states x planners)", sn);
    }
    if (hipMemsetAsync(pl->overflow, 0, 4 * (size_t)(1 + pl->n), ctx->stream) != hipSuccess) { // (stream order: the block may be a recycled one)
        mp_saopd_free(pl);
        return fail(MP_ERR_HIP, "mp_saopd_create: hipMemset failed");
    }
    *out = pl;
    return MP_OK;
}

int mp_saopd_free(mp_saopd *pl)
{
    if (!pl) return MP_OK;
    // (a batch whose ctx is gone cannot recycle: the ctx destroys its cache; planners are freed before their ctx)
    for (const auto &b : pl->blocks) ctx_block_release(pl->ctx, b.p, b.bytes);
    pl->blocks.clear();
    delete pl;
    return MP_OK;
}

int mp_saopd_plan(mp_ctx *ctx, mp_saopd *pl, const int32_t *root_state, int32_t budget, double gamma,
                  double terminal_reward, double accuracy, int32_t backup_aggregated_nodes,
                  int32_t prune_suboptimal_leaves, uint64_t *rng_state, int32_t max_plan_len, int32_t *plans,
                  int32_t *plan_len, int64_t *env_steps, int64_t *updates, int32_t *status, int32_t mem)
{
    if (!ctx || !pl || !root_state || !rng_state) return fail(MP_ERR_ARG, "mp_saopd_plan: NULL argument");
    if (!mem_valid(mem)) return fail(MP_ERR_ARG, "mp_saopd_plan: unknown mem flags %d", mem);
    const int rmem = mem_rng(mem); // MP_MEM_RNG_DEVICE: the generator records are device-resident (mp_rng) also with host arrays
    mem = mem_arrays(mem);
    if (pl->ctx != ctx) return fail(MP_ERR_ARG, "mp_saopd_plan: planners belong to another context");
    if (budget < 0 || max_plan_len < 0) return fail(MP_ERR_ARG, "mp_saopd_plan: bad sizes");
    if (!(gamma >= 0.0 && gamma < 1.0)) return fail(MP_ERR_ARG, "mp_saopd_plan: gamma must be in [0, 1)");
    const bool fresh = pl->n_nodes == 0;
    if (!fresh && gamma != pl->gamma)
        return fail(MP_ERR_ARG, "mp_saopd_plan: gamma changed (%g -> %g) on a planner that holds state values", pl->gamma, gamma);
    MP_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const int A = pl->A, n = pl->n;
    const int K = budget / A; // deterministic.py:118
    if (K + 2 > (int)SA_DEPTH) return fail(MP_ERR_ARG, "mp_saopd_plan: budget too large");
    // backup queue: the reference's list holds duplicates and reaches ~10x the node count of a plan (4 600 entries at
    // 325 nodes on the 10x10 grid); 32 entries per node of this plan, at least 4096, a power of two.  MP_SAOPD_QUEUE
    // sets the entries per planner; a full queue is reported per planner (MP_ERR_ALLOC), never dropped silently.
    // The queue is empty between plans, so it can be replaced when a larger budget comes along.
    {
        long want = 32L * (1 + K * A);
        int q = 4096;
        if (const char *e = getenv("MP_SAOPD_QUEUE")) { want = atol(e); q = 2; }
        while (q < want && q < (1 << 28)) q <<= 1;
        if (q > pl->qcap) {
            if (pl->queue) { MP_HIP(hipStreamSynchronize(ctx->stream)); sa_release(pl, pl->queue); pl->queue = nullptr; }
            if (sa_alloc(pl, &pl->queue, (size_t)pl->n * q * 4) != hipSuccess)
                return fail(MP_ERR_ALLOC, "mp_saopd_plan: %zu B for the backup queues", (size_t)pl->n * q * 4);
            pl->qcap = q;
        }
        if (1 + K * A > pl->qcap) // the prune pass lists its candidate leaves in the idle queue
            return fail(MP_ERR_ARG, "mp_saopd_plan: queue of %d entries is smaller than the %d nodes of a plan", pl->qcap, 1 + K * A);
    }
    const int need = pl->n_nodes + 1 + K * A;
    if (need > pl->cap) {
        const int new_cap = need + (need - pl->cap < 4096 ? need / 2 : 0); // some slack for the following plans
        const size_t o = (size_t)pl->n_nodes, oc = (size_t)pl->cap;
        const bool pm = pl->wave != 0;
        MP_TRY(grow_rows(pl, &pl->node, o, oc, new_cap, n, pm, st));
        MP_TRY(grow_rows(pl, &pl->state, o, oc, new_cap, n, pm, st));
        MP_TRY(grow_rows(pl, &pl->parent, o, oc, new_cap, n, pm, st));
        MP_TRY(grow_rows(pl, &pl->first_child, o, oc, new_cap, n, pm, st));
        MP_TRY(grow_rows(pl, &pl->reward, o, oc, new_cap, n, pm, st));
        MP_TRY(grow_rows(pl, &pl->done, o, oc, new_cap, n, pm, st));
        MP_TRY(grow_rows(pl, &pl->oldlive, o, oc, new_cap, n, pm, st));
        pl->cap = new_cap;
    }
    {   // chunk pool: at most one partly filled chunk per non-empty state, contents rebuilt by every plan (nothing to keep)
        const long states = pl->S < pl->cap ? pl->S : pl->cap;
        const long want = 16L * (states + pl->cap / 15 + 2);
        if (want > pl->pool_ints) {
            if (pl->lpool) { MP_HIP(hipStreamSynchronize(st)); sa_release(pl, pl->lpool); pl->lpool = nullptr; }
            if (sa_alloc(pl, &pl->lpool, (size_t)n * want * 4) != hipSuccess)
                return fail(MP_ERR_ALLOC, "mp_saopd_plan: %zu B for the state-list chunks", (size_t)n * want * 4);
            pl->pool_ints = want;
        }
    }
    // tables with the reference's own operations
    std::vector<double> tab((size_t)3 * (K + 3));
    double *gpow = tab.data(), *trg = gpow + (K + 3), *acc = trg + (K + 3);
    for (int d = 0; d < K + 3; ++d) {
        gpow[d] = pow(gamma, (double)d);                                          // gamma ** depth
        trg[d] = terminal_reward * gpow[d] / (1 - gamma);                          // deterministic.py:60
        acc[d] = d >= 1 ? accuracy * (1 - gamma) * pow(gamma, (double)(d - 1)) : 0.0; // state_aware.py:62
    }
    double *d_tab = nullptr;
    MP_TRY(upload_tables(ctx, 3, tab, &d_tab));

    SaArgs a;
    a.n = n; a.S = pl->S; a.A = A; a.K = K; a.root = pl->n_nodes; a.n_prev = pl->n_nodes;
    a.prev_root = pl->root < 0 ? 0 : pl->root; a.qcap = pl->qcap;
    a.done_on_next = pl->model->done_on_next; a.max_plan_len = max_plan_len;
    a.backup_aggregated = backup_aggregated_nodes ? 1 : 0; a.prune = prune_suboptimal_leaves ? 1 : 0; a.fresh = fresh;
    a.gamma = gamma; a.vmax = 1 / (1 - gamma);
    a.rec = pl->model->rec; a.tab = d_tab;
    a.node = pl->node; a.state = pl->state; a.parent = pl->parent; a.first_child = pl->first_child;
    a.reward = pl->reward; a.done = pl->done; a.oldlive = pl->oldlive; a.lstate = pl->lstate;
    a.lpool = pl->lpool; a.pool_ints = pl->pool_ints; a.sv = pl->sv; a.head = pl->head; a.tail = pl->tail; a.queue = pl->queue;
    a.stamp = pl->stamp; a.iter_base = pl->iters; a.cap = pl->cap;
    auto lane_scratch = [&]() {
        int v = pl->qcap >> 6;
        if (const char *e = getenv("MP_SAOPD_LANE_SCRATCH")) { // test knob: a tiny slice forces the serial prune pass
            const int w = atoi(e);
            if (w >= 0 && w < v) v = w;
        }
        return v;
    };
    a.scap = lane_scratch();
    // The grouped parallel backup pays where the backups are: the first plan of fresh planners runs ~4 200 of them, the
    // following plans ~200 -- there the upkeep of the chunked lists (rebuild at the start, two more accesses per append)
    // costs more than it saves (16 384 planners: 7.8 / 9.0 ms against 6.9 / 7.9).  MP_SAOPD_PAR_BACKUP=0|1 forces it.
    a.par_backup = fresh ? 1 : 0;
    if (const char *e = getenv("MP_SAOPD_PAR_BACKUP")) a.par_backup = e[0] == '1';
    a.csr_old = 1;
    if (const char *e = getenv("MP_SAOPD_CSR")) a.csr_old = e[0] == '0' ? 0 : 2;
    a.prune_rows = 256;
    if (const char *e = getenv("MP_SAOPD_PRUNE_ROWS")) { // test knob: 0 = every changed state through the streamed form
        const int v = atoi(e);
        a.prune_rows = v < 0 ? 0 : (v > 256 ? 256 : v);
    }
    int32_t *d_rs = nullptr;
    MP_TRY(stage_in(ctx, WS_IO0, root_state, (size_t)n, mem, &d_rs));
    a.root_state = d_rs;
    MP_TRY(stage_in(ctx, WS_IO2, (const uint64_t *)rng_state, (size_t)n * 6, rmem, &a.rng));
    MP_TRY(stage_out_alloc(ctx, WS_IO3, plans, (size_t)n * max_plan_len, mem, &a.plans));
    MP_TRY(stage_out_alloc(ctx, WS_IO4, plan_len, (size_t)n, mem, &a.plan_len));
    MP_TRY(stage_out_alloc(ctx, WS_IO5, status, (size_t)n, mem, &a.status));
    MP_TRY(stage_out_alloc(ctx, WS_IO6, env_steps, (size_t)n, mem, &a.env_steps));
    MP_TRY(stage_out_alloc(ctx, WS_IO7, updates, (size_t)n, mem, &a.updates));

    // tables (+ 512 B the wave kernel uses as a counter).  The wave kernels keep at most 2 560 depth entries (60 KB) in LDS and
    // read deeper nodes' entries from the global copy, so the budget is not bounded by LDS (round 4); the lane kernel holds them all
    a.tab_plain = pl->wave ? std::min(K + 3, 2560) : K + 3;
    const size_t lds = (size_t)3 * a.tab_plain * sizeof(double) + 128 * sizeof(int32_t);
    if (lds > 64 * 1024)
        return fail(MP_ERR_ARG, "mp_saopd_plan: budget %d needs %zu B of LDS tables (> 64 KiB) in the one-planner-per-lane kernel", budget, lds);
    // LDS-resident variant of the wave kernel: the arena after this plan, the dictionaries and a (smaller) queue in LDS.
    // A queue overflow there rolls the plan back like any other and the retry runs on the global-memory form.
    a.lds_rows = need;
    a.lds_qcap = 4096;
    while (a.lds_qcap < 2 * (1 + K * A) && a.lds_qcap < (1 << 20)) a.lds_qcap <<= 1; // the prune pass lists a plan's leaves in it
    if (getenv("MP_SAOPD_QUEUE") && pl->qcap < a.lds_qcap) a.lds_qcap = pl->qcap; // (test knob: a deliberately small queue)
    const size_t lds_res = ((lds + 15) & ~(size_t)15) + 16 + (size_t)need * (16 + 8 + 12) + (size_t)pl->S * (8 + 12) + (size_t)a.lds_qcap * 4;
    // Measured (GridWorld 10x10, budget 500, first plan / following plans): 1 planner 4.8 / 2.4 ms against 6.1 / 3.1 ms
    // from global memory, 64 planners 11.6 / 5.6 ms against 14.4 / 7.0 ms -- but 4 096 planners 32 / 40 ms against
    // 22 / 16 ms: the chain is bound by the wave's own instruction latency (~300 ns per dependent step even from LDS), so
    // what a big batch needs is resident waves, which 39 KB of LDS per planner takes away.  Hence: latency mode only,
    // at most one planner per CU.
    bool use_lds = pl->wave && lds_res <= kLdsBytes - 1024 && n <= ctx->prop.multiProcessorCount;
    if (const char *e = getenv("MP_SAOPD_LDS")) use_lds = pl->wave && lds_res <= kLdsBytes - 1024 && e[0] == '1';
    // mem = MP_MEM_DEVICE: ASYNCHRONOUS -- one launch, nothing read back.  A planner whose backup queue fills up cannot be
    // rolled back and retried then: it reports MP_ERR_ALLOC and stays failed (every later call reports it again), the
    // other planners of the batch are unaffected.  With host arrays the call synchronises anyway, reads the overflow
    // word and retries with a larger queue.
    // dictionaries in LDS (LDSD): while they leave five planners per SIMD
    // (5 KB per planner = 32 per CU = eight waves per SIMD; the depth tables take what the dictionaries leave, at least 16 entries)
    const long dict_room = 5 * 1024 - 32 - (long)pl->S * (8 + 16 + 16);
    a.tab_lds = (int)std::min<long>(K + 3, dict_room / 24);
    const size_t lds_dict = (size_t)a.tab_lds * 24 + 32 + (size_t)pl->S * (8 + 16 + 16);
    bool use_dict = pl->wave && a.tab_lds >= 16;
    if (const char *e = getenv("MP_SAOPD_DICT")) use_dict = use_dict && e[0] == '1';
    // with the dictionaries in LDS and the grouped backup the global-memory arena is as fast for ONE planner as the all-in-LDS
    // latency mode (first plan, light / heavy root: 1.4 / 3.3 ms against 1.4 / 4.6 ms): MP_SAOPD_LDS=1 still forces that one
    if (use_dict && !getenv("MP_SAOPD_LDS")) use_lds = false;
    const bool async = mem == MP_MEM_DEVICE;
    a.sticky = async ? 1 : 0;
    if (async) use_lds = false; // (its small queue relies on the retry)
    const int scap_global = a.scap;
    if (use_lds) a.scap = a.lds_qcap >> 6 < a.scap ? a.lds_qcap >> 6 : a.scap;
    // what a plan changes outside its own node rows: the per-state dictionaries, the list links of older tail nodes
    // and the generator states.  They are copied first, so that a plan that fills its backup queue (values decaying
    // to floating-point underflow make the reference run tens of thousands of backups) can be rolled back and run
    // again with a larger queue.  Costs one 4-byte read-back (a stream synchronisation) per call.
    const size_t sn = (size_t)pl->S * n;
    a.overflow = pl->overflow;
    // (an asynchronous call never rolls back -- a full queue is sticky there -- so it takes no snapshot: ADVICE r2)
    if (!fresh && !async) {
        MP_HIP(hipMemcpyAsync(pl->snap_sv, pl->sv, sn * 8, hipMemcpyDeviceToDevice, st));
        MP_HIP(hipMemcpyAsync(pl->snap_head, pl->head, sn * 4, hipMemcpyDeviceToDevice, st));
        MP_HIP(hipMemcpyAsync(pl->snap_tail, pl->tail, sn * 4, hipMemcpyDeviceToDevice, st));
        MP_HIP(hipMemcpyAsync(pl->snap_stamp, pl->stamp, sn * 4, hipMemcpyDeviceToDevice, st));
    }
    if (!async) MP_HIP(hipMemcpyAsync(pl->snap_rng, a.rng, (size_t)n * 48, hipMemcpyDeviceToDevice, st));
    size_t max_queue_bytes = (size_t)8 << 30; // per batch; MP_SAOPD_QUEUE_LIMIT_MB overrides
    if (const char *e = getenv("MP_SAOPD_QUEUE_LIMIT_MB")) max_queue_bytes = (size_t)atol(e) << 20;
    int launches = 0;
    // Dispatch order (wave kernel, batches beyond one residency round = 32 planners per CU): by the cost of the planner's
    // previous plan, or -- fresh planners -- by what first plans from the same root state cost on this model so far.
    // Only the ORDER in which workgroups start depends on it; MP_SAOPD_ORDER=0 keeps the index order.
    a.order = nullptr; a.cost = pl->wave ? pl->cost : nullptr;
    bool ordered = pl->wave && n > 32 * ctx->prop.multiProcessorCount;
    if (const char *e = getenv("MP_SAOPD_ORDER")) ordered = pl->wave && e[0] == '1';
    const bool have_cost = fresh ? pl->model->sa_cost != nullptr : pl->cost_valid;
    if (pl->wave && fresh && !pl->model->sa_cost && hipMalloc(&pl->model->sa_cost, 4 * (size_t)pl->S) == hipSuccess)
        MP_HIP(hipMemsetAsync(pl->model->sa_cost, 0, 4 * (size_t)pl->S, st));
    MP_TRY(kernels_begin(ctx));
    if (ordered && have_cost) { // (inside the timed region: the sort is part of what a plan costs)
        hipLaunchKernelGGL(saopd_order_kernel, dim3(1), dim3(1024), 0, st, n, pl->model->S, pl->cost, fresh ? pl->model->sa_cost : nullptr, d_rs, pl->order);
        a.order = pl->order;
        ++launches;
    }
    for (;;) {
        MP_HIP(hipMemsetAsync(pl->overflow, 0, 4, st));
        if (fresh) {
            const long tot = (long)n * pl->S;
            hipLaunchKernelGGL(saopd_init_kernel, dim3((unsigned)((tot + 63) / 64)), dim3(64), 0, st, n, pl->S, a.vmax, pl->sv,
                               pl->head, pl->tail, pl->stamp);
            ++launches;
        }
        if (pl->wave && use_lds) {
            if (lds_res > 64 * 1024)
                MP_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(saopd_wave_kernel<true>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_res));
            hipLaunchKernelGGL(saopd_wave_kernel<true>, dim3((unsigned)n), dim3(64), lds_res, st, a);
        } else if (pl->wave && use_dict) hipLaunchKernelGGL((saopd_wave_kernel<false, true>), dim3((unsigned)n), dim3(64), lds_dict, st, a);
        else if (pl->wave) hipLaunchKernelGGL(saopd_wave_kernel<false>, dim3((unsigned)n), dim3(64), lds, st, a);
        else hipLaunchKernelGGL(saopd_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), lds, st, a);
        ++launches;
        if (async) break;
        int32_t ovf = 0;
        MP_HIP(hipMemcpyAsync(&ovf, pl->overflow, 4, hipMemcpyDeviceToHost, st));
        MP_HIP(hipStreamSynchronize(st));
        MP_HIP(hipGetLastError());
        const size_t bigger = (size_t)n * pl->qcap * 4 * 4;
        if (!ovf) break;
        if (use_lds) {
            // the small LDS queue filled up: roll back and run the plan on the global-memory form (full-size queue)
            use_lds = false;
            a.scap = scap_global;
        } else {
            if (bigger > max_queue_bytes) { // out of room: the full planners keep MP_ERR_ALLOC, now and in later calls
                if (a.status) {
                    hipLaunchKernelGGL(saopd_mark_failed_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, st, n, a.status, pl->overflow);
                    ++launches;
                }
                break;
            }
            // roll back and run again with a queue four times as large
            sa_release(pl, pl->queue);
            pl->queue = nullptr;
            if (sa_alloc(pl, &pl->queue, bigger) != hipSuccess) return fail(MP_ERR_ALLOC, "mp_saopd_plan: %zu B for the backup queues", bigger);
            pl->qcap *= 4;
            a.queue = pl->queue; a.qcap = pl->qcap;
            a.scap = lane_scratch();
        }
        if (!fresh) {
            MP_HIP(hipMemcpyAsync(pl->sv, pl->snap_sv, sn * 8, hipMemcpyDeviceToDevice, st));
            MP_HIP(hipMemcpyAsync(pl->head, pl->snap_head, sn * 4, hipMemcpyDeviceToDevice, st));
            MP_HIP(hipMemcpyAsync(pl->tail, pl->snap_tail, sn * 4, hipMemcpyDeviceToDevice, st));
            MP_HIP(hipMemcpyAsync(pl->stamp, pl->snap_stamp, sn * 4, hipMemcpyDeviceToDevice, st));
            hipLaunchKernelGGL(saopd_fix_tails_kernel, dim3((unsigned)((sn + 63) / 64)), dim3(64), 0, st, n, pl->S, pl->node_si(),
                               pl->node_sr(), pl->state_si(), pl->state_sr(), pl->tail, pl->node);
            ++launches;
        }
        MP_HIP(hipMemcpyAsync(a.rng, pl->snap_rng, (size_t)n * 48, hipMemcpyDeviceToDevice, st));
    }
    if (pl->wave) {
        pl->cost_valid = true;
        if (fresh && pl->model->sa_cost) { // remember the first plans' costs by root state
            hipLaunchKernelGGL(saopd_learn_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, n, pl->model->S, d_rs, pl->cost, a.status,
                               pl->model->sa_cost);
            ++launches;
        }
    }
    MP_TRY(kernels_end(ctx, launches));
    MP_HIP(hipGetLastError());
    pl->gamma = gamma;
    pl->iters += K;
    pl->root = pl->n_nodes;
    pl->n_nodes = need;

    MP_TRY(stage_out_copy(ctx, rng_state, a.rng, (size_t)n * 6, rmem));
    MP_TRY(stage_out_copy(ctx, plans, a.plans, (size_t)n * max_plan_len, mem));
    MP_TRY(stage_out_copy(ctx, plan_len, a.plan_len, (size_t)n, mem));
    MP_TRY(stage_out_copy(ctx, status, a.status, (size_t)n, mem));
    MP_TRY(stage_out_copy(ctx, env_steps, a.env_steps, (size_t)n, mem));
    MP_TRY(stage_out_copy(ctx, updates, a.updates, (size_t)n, mem));
    if (mem == MP_MEM_HOST) MP_HIP(hipStreamSynchronize(st));
    return MP_OK;
}

int mp_saopd_info(mp_saopd *pl, int32_t *n_planners, int32_t *n_nodes, int32_t *root, int32_t *n_states)
{
    if (!pl) return fail(MP_ERR_ARG, "mp_saopd_info: planner is NULL");
    if (n_planners) *n_planners = pl->n;
    if (n_nodes) *n_nodes = pl->n_nodes;
    if (root) *root = pl->root;
    if (n_states) *n_states = pl->S;
    return MP_OK;
}

int mp_saopd_export(mp_saopd *pl, int32_t planner, int32_t cap, int32_t *parent, int32_t *action, int32_t *state,
                    int32_t *depth, double *reward, double *lower, uint8_t *done, int64_t *count, int32_t *first_child,
                    uint8_t *alive, double *state_values)
{
    if (!pl) return fail(MP_ERR_ARG, "mp_saopd_export: planner is NULL");
    if (planner < 0 || planner >= pl->n) return fail(MP_ERR_ARG, "mp_saopd_export: planner %d out of range", planner);
    if (cap < pl->n_nodes) return fail(MP_ERR_ARG, "mp_saopd_export: capacity %d < %d nodes", cap, pl->n_nodes);
    MP_HIP(hipSetDevice(pl->ctx->device));
    MP_HIP(hipStreamSynchronize(pl->ctx->stream));
    const size_t nn = (size_t)pl->n_nodes, n = (size_t)pl->n;
    // the entries of one planner: element (i, planner) of an array with row stride si and planner stride sr
    auto column_of = [&](void *dst, const void *src, size_t elem, size_t rows, long si, long sr) -> int {
        MP_HIP(hipMemcpy2D(dst, elem, (const char *)src + (size_t)planner * sr * elem, (size_t)si * elem, elem, rows,
                           hipMemcpyDeviceToHost));
        return MP_OK;
    };
    auto column = [&](void *dst, const void *src, size_t elem, size_t rows) -> int {
        return column_of(dst, src, elem, rows, pl->node_si(), pl->node_sr());
    };
    (void)n;
    std::vector<SaNode> hn(nn);
    std::vector<int32_t> hpar(nn), hfc(nn);
    if (nn) {
        MP_TRY(column(hn.data(), pl->node, sizeof(SaNode), nn));
        MP_TRY(column(hpar.data(), pl->parent, 4, nn));
        MP_TRY(column(hfc.data(), pl->first_child, 4, nn));
        if (state) MP_TRY(column(state, pl->state, 4, nn));
        if (reward) MP_TRY(column(reward, pl->reward, 8, nn));
        if (done) MP_TRY(column(done, pl->done, 1, nn));
    }
    if (state_values) MP_TRY(column_of(state_values, pl->sv, 8, (size_t)pl->S, pl->state_si(), pl->state_sr()));
    for (size_t i = 0; i < nn; ++i) {
        if (parent) parent[i] = hpar[i];
        if (first_child) first_child[i] = hfc[i];
        if (lower) lower[i] = hn[i].lower;
        if (depth) depth[i] = (int32_t)(hn[i].meta & SA_DEPTH);
        if (alive) alive[i] = (hn[i].meta & SA_ALIVE) ? 1 : 0;
        if (done) done[i] &= 1; // (bit 1 is the wave kernel's dead-row mark)
        if (action) action[i] = -1;
    }
    if (action)
        for (size_t i = 0; i < nn; ++i)
            if (hfc[i] >= 0)
                for (int a = 0; a < pl->A; ++a) action[hfc[i] + a] = a;
    if (count) {
        // DeterministicNode starts at count 1 and update() adds 1 along the whole root->child sequence
        // (deterministic.py:17,64-65): count = 1 + subtree size, the root of a tree (no update of its own) one less
        // (phantom rows -- slots of actions the env does not list, lower = -inf -- are not nodes: they count for nothing)
        for (size_t i = 0; i < nn; ++i) count[i] = hn[i].lower == -INFINITY ? 0 : 1;
        for (size_t i = nn; i-- > 0;)
            if (hpar[i] >= 0) count[hpar[i]] += count[i];
        for (size_t i = 0; i < nn; ++i) count[i] = hpar[i] >= 0 ? count[i] + 1 : count[i];
    }
    return MP_OK;
}

} // extern "C"
