// Search / rules kernels: one wavefront (one 64-thread workgroup) per game or tree.
#pragma once
#include "az_device.h"

// ---------------------------------------------------------------------------
// RNG draws (wave-uniform; lane 0 advances the cursors)
// ---------------------------------------------------------------------------

__device__ __forceinline__ unsigned long long rng_u64(const Dev &D, int g)
{
    unsigned long long c = D.rng_ctr[g];
    WSYNC();
    if (lane_id() == 0) D.rng_ctr[g] = c + 1;
    return mix64(D.seed + 0x9E3779B97F4A7C15ULL * (unsigned long long)(g + 1) + mix64(c));
}

__device__ __forceinline__ double rng_u01(const Dev &D, int g)
{
    return (double)(rng_u64(D, g) >> 11) * (1.0 / 9007199254740992.0);
}

__device__ int rng_tid(const Dev &D, int g)
{
    if (D.rng_replay) {
        long long i = D.rt_i[g];
        WSYNC();
        if (i >= D.rt_n[g]) { if (lane_id() == 0) D.err[g] = -5; return 0; }
        int v = D.rt[g][i];
        if (lane_id() == 0) D.rt_i[g] = i + 1;
        return v;
    }
    return (int)(rng_u64(D, g) % (unsigned long long)D.transform_types);
}

__device__ double rng_uniform(const Dev &D, int g)
{
    if (D.rng_replay) {
        long long i = D.ru_i[g];
        WSYNC();
        if (i >= D.ru_n[g]) { if (lane_id() == 0) D.err[g] = -5; return 0.0; }
        double v = D.ru[g][i];
        if (lane_id() == 0) D.ru_i[g] = i + 1;
        return v;
    }
    return rng_u01(D, g);
}

// gamma(alpha, 1), the three algorithms of CPython's random.gammavariate (math_helper.py:13-17 calls it with
// alpha = d_alpha, beta = 1): Ahrens-Dieter GS for alpha < 1, the exponential for alpha == 1, Cheng's rejection
// method for alpha > 1.  Per-lane independent draw from a hashed stream; the rejection loops are bounded (a draw is
// accepted with probability > 0.5 per trial in every branch).
__device__ double gamma_variate(double alpha, unsigned long long key)
{
    auto u01 = [&](int i) { return (double)(mix64(key + (unsigned long long)i) >> 11) * (1.0 / 9007199254740992.0); };
    double x = 0.0;
    if (alpha > 1.0) {
        const double ainv = sqrt(2.0 * alpha - 1.0), bbb = alpha - 1.3862943611198906, ccc = alpha + ainv;
        for (int it = 0; it < 64; it++) {
            const double u1 = u01(2 * it);
            if (!(1e-7 < u1 && u1 < 0.9999999)) continue;
            const double u2 = 1.0 - u01(2 * it + 1);
            const double v = log(u1 / (1.0 - u1)) / ainv;
            x = alpha * exp(v);
            const double z = u1 * u1 * u2, r = bbb + ccc * v - x;
            if (r + 2.504077396776274 - 4.5 * z >= 0.0 || r >= log(z)) break;
        }
        return x;
    }
    if (alpha == 1.0) return -log(1.0 - u01(0));
    for (int it = 0; it < 64; it++) {
        double u = u01(2 * it);
        double u1 = u01(2 * it + 1);
        double bb = (2.718281828459045 + alpha) / 2.718281828459045;
        double p = bb * u;
        x = p <= 1.0 ? pow(p, 1.0 / alpha) : -log((bb - p) / alpha);
        if (p > 1.0) { if (u1 <= pow(x, alpha - 1.0)) break; }
        else if (u1 <= exp(-x)) break;
    }
    return x;
}

// P <- (1-eps)*P + eps*eta over the root's children in insertion order
// (search/mcts.py:285-291; eta is NOT normalised, math_helper.py:13-17)
template <typename R>
__device__ void apply_noise(const Dev &D, int g, int t)
{
    R *const eP = static_cast<R *>(D.eP);
    const int lane = lane_id();
    int rb = D.rootBlock[t];
    if (rb < 0) return;
    size_t bb = (size_t)t * D.BC;
    int start = D.bStart[bb + rb], n = D.bCount[bb + rb];
    const Span sp = span_of(D, t, start);
    R keep = (R)(1.0 - D.d_epsilon);
    long long base = 0;
    unsigned long long key = 0;
    if (D.rng_replay) {
        base = D.re_i[g];
        WSYNC();
        if (base + n > D.re_n[g]) { if (lane == 0) D.err[g] = -5; return; }
        if (lane == 0) D.re_i[g] = base + n;
    } else {
        key = rng_u64(D, g);
    }
    for (int c = lane; c < n; c += AZ_WAVE) {
        double eta = D.rng_replay ? D.re[g][base + c] : gamma_variate(D.d_alpha, key + 1000ULL * (unsigned long long)(c + 1));
        R p = eP[sp.at(start + c)];
        R a = keep * p;
        eP[sp.at(start + c)] = a + (R)(D.d_epsilon * eta);
    }
    WSYNC();
}

// ---------------------------------------------------------------------------
// select: descend one tree per slot to a leaf, replaying the rules on the way
// (search/mcts.py:173-217, :261-280)
// ---------------------------------------------------------------------------

// Finished games' trajectories packed back to back: block q copies the rows of slot slots[q] to row starts[q] of the
// packed arrays (moves int16 per ply, visits int32[O] per ply).
__global__ __launch_bounds__(256) void k_traj_gather(Dev D, const int32_t *__restrict__ slots, const int32_t *__restrict__ starts, int n,
                                                    int16_t *__restrict__ moves, int32_t *__restrict__ visits)
{
    const int q = blockIdx.x;
    if (q >= n) return;
    const int g = slots[q], O = D.geo.O;
    const int np = min(D.ply[g], D.traj_plies);
    const size_t r0 = (size_t)starts[q];
    for (int i = threadIdx.x; i < np; i += blockDim.x) moves[r0 + i] = D.tr_moves[(size_t)g * D.traj_plies + i];
    const int32_t *src = D.tr_visits + (size_t)g * D.traj_plies * O;
    int32_t *dst = visits + r0 * O;
    for (size_t i = threadIdx.x; i < (size_t)np * O; i += blockDim.x) dst[i] = src[i];
}

#ifdef AZ_STAMP_SELECT
// development build only: shader-clock stamps of one wavefront's phases in k_select (block 0, lane 0)
__device__ unsigned long long g_sel_stamps[16];
#define SEL_STAMP(i) do { if (blockIdx.x == 0 && lane == 0) g_sel_stamps[i] += __builtin_amdgcn_s_memtime() - t_prev; t_prev = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define SEL_STAMP(i) do {} while (0)
#endif

template <typename R>
__device__ __forceinline__ void select_one(const Dev &D, const int sub, WLds &L)
{
    R *const eW = static_cast<R *>(D.eW);
    const R *const eP = static_cast<const R *>(D.eP);
    const int g = blockIdx.x;
    const int li = sub * D.G + g;  // row of this sub-round's leaf
    const int lane = lane_id();
#ifdef AZ_STAMP_SELECT
    unsigned long long t_prev = __builtin_amdgcn_s_memtime();
#endif
    const Geo G = D.geo;
    int phase = D.phase[g];
    // a slot stopped by an error keeps its pending leaf untouched: once the cause is removed (a full tree arena that the
    // host has enlarged) the finish step completes that playout
    if (D.err[g] != 0) return;
    if (lane == 0) D.leaf_valid[li] = 0;
    if (phase != PH_NEED_START && phase != PH_IN_SEARCH) return;
    // sub-rounds after the first only add playouts to a search that is under way and has budget left
    if (sub > 0 && (phase != PH_IN_SEARCH || D.issued[g] >= D.max_playout)) return;
    const int side = D.side[g];
    const int t = 2 * g + side;
    const size_t bb = (size_t)t * D.BC;

    PS s;
    load_scalars(D, g, s);
    load_ring(D, g, L);
    const SK sk = slot_sk(D, g);   // superko record: the playout appends behind the game's entries
    if (G.game == G_GO) { s.is_end = 0; s.passes_b = 0; s.passes_w = 0; }  // GameState.copy(), env/go.py:242

    int root_expand = 0;
    int rootN = D.rootN[t];
    WSYNC();
    SEL_STAMP(0);  // state load
    if (phase == PH_NEED_START) {
        rootN += 1;  // self._root.visit(), mcts.py:261
        if (D.rootBlock[t] < 0) {
            root_expand = 1;
        } else {
            if (D.dflag[g]) apply_noise<R>(D, g, t);
            phase = PH_IN_SEARCH;
            if (lane == 0) { D.phase[g] = PH_IN_SEARCH; D.pdone[g] = 0; D.issued[g] = 0; }
            if (D.max_playout == 0) {
                if (lane == 0) { D.phase[g] = PH_SEARCH_DONE; D.rootN[t] = rootN; }
                return;
            }
        }
    }
    // the playout's transform id (one draw per playout, mcts.py:208) is fetched here so that the dependent loads of the
    // replay stream overlap the descent; it is used at the leaf
    const int tid = D.transform_on ? rng_tid(D, g) : 0;
    int node = NODE_ROOT, node_block = -1, depth = 0;
    unsigned long long levels = 0, scanned = 0;
    if (!root_expand) {
        rootN += 1;  // node.visit() of the root inside _playout
        int blk = D.rootBlock[t];
        int parentN = rootN;
        SEL_STAMP(1);  // search start / noise
        while (blk >= 0) {
            int start = D.bStart[bb + blk], n = D.bCount[bb + blk];
            const Span sp = span_of(D, t, start);
            // Q+U in the evaluator's precision, left to right as mcts.py:88-90; first maximum wins.  A node has at most
            // AZ_NJ * 64 children: all their statistics are requested before any is used (one memory latency per level,
            // not one per 64 children); a slot past the end re-reads the last child and is ignored.
            R sq = (R)sqrt((double)parentN);
            const R cp = (R)D.c_punt;
            int cN[AZ_NJ];
            R cW[AZ_NJ], cP[AZ_NJ];
#pragma unroll
            for (int j = 0; j < AZ_NJ; j++) {
                const size_t ci = sp.at(start + min(lane + j * AZ_WAVE, n - 1));
                cN[j] = D.eN[ci]; cW[j] = eW[ci]; cP[j] = eP[ci];
            }
            R bv = -INFINITY;
            int bi = 0x7fffffff, bN = cN[0];
#pragma unroll
            for (int j = 0; j < AZ_NJ; j++) {
                const int c = lane + j * AZ_WAVE;
                const int N = cN[j];
                R u = cp * cP[j];
                u = u * sq;
                u = u / ((R)1.0 + (R)N);
                R q = N == 0 ? (R)0.0 : cW[j] / (R)N;
                R v = q + u;
                if (c < n && v > bv) { bv = v; bi = c; bN = N; }
            }
            // the action and the child block of this lane's own best candidate are fetched while the lanes compare
            const size_t mine = sp.at(start + (bi == 0x7fffffff ? min(lane, n - 1) : bi));
            if (bi == 0x7fffffff) bN = cN[0];
            const int myAct = D.eAct[mine], myChild = D.eChild[mine];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                R ov = __shfl_xor(bv, o, 64);
                int oi = __shfl_xor(bi, o, 64);
                if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
            }
            levels++; scanned += (unsigned long long)n;
            if (bi == 0x7fffffff) bi = 0;  // all-NaN scores: Python's max keeps the first item
            int e = start + bi;
            if (depth >= D.MAXD) { if (lane == 0) D.err[g] = -6; return; }
            if (lane == 0) {
                D.path_node[(size_t)li * D.MAXD + depth] = node;
                D.path_player[(size_t)li * D.MAXD + depth] = (int8_t)s.cur;
            }
            depth++;
            const int wl = bi & (AZ_WAVE - 1);  // child c sits in lane c % 64
            int a = __shfl(myAct, wl, 64);
            SEL_STAMP(2);  // child scan + arg-max
            dev_do_move(G, L, s, a, &sk);
            SEL_STAMP(3);  // rule move
            int N = __shfl(bN, wl, 64) + 1;  // child.visit()
            WSYNC();
            if (lane == 0) {
                D.eN[sp.at(e)] = N;
                if (D.K > 1) eW[sp.at(e)] -= (R)D.vloss;  // virtual loss, taken back by k_finish
            }
            node = e; node_block = blk;
            parentN = N;
            blk = __shfl(myChild, wl, 64);
            SEL_STAMP(4);  // visit store + child pointer
        }
    }
    if (lane == 0) {
        D.rootN[t] = rootN;
        if (!root_expand) D.issued[g] += 1;
    }

    // leaf: transform id, legality on the un-transformed state, evaluator input
    SEL_STAMP(5);
    int nlegal = dev_legal_mask(G, L, s, &sk);
    SEL_STAMP(6);  // legal mask
    uint8_t *gm = D.leaf_mask + (size_t)li * AZ_OP;
    for (int i = lane; i < AZ_OP / 4; i += AZ_WAVE)
        reinterpret_cast<int *>(gm)[i] = reinterpret_cast<int *>(L.mask)[i];
    int terminal = (nlegal == 0) || s.is_end;
    int win = 0;
    if (terminal && !root_expand) win = dev_winner(G, L, s);
    SEL_STAMP(7);  // mask store, winner
    dev_features(G, L, s, tid, D.planes + (size_t)li * 17 * G.NC);
    SEL_STAMP(8);  // feature planes
    if (lane == 0) {
        D.leaf_valid[li] = 1;
        D.leaf_node[li] = node;
        D.leaf_block[li] = node_block;
        D.leaf_player[li] = s.cur;
        D.leaf_end[li] = terminal;
        D.leaf_winner[li] = win;
        D.leaf_tid[li] = tid;
        D.leaf_root[li] = root_expand;
        D.path_len[li] = depth;
        atomicAdd(&D.ctr[(size_t)(blockIdx.x & (AZ_CTR_BUCKETS - 1)) * C_N + C_LEVELS], levels);
        atomicAdd(&D.ctr[(size_t)(blockIdx.x & (AZ_CTR_BUCKETS - 1)) * C_N + C_CHILDREN], scanned);
        atomicAdd(&D.ctr[(size_t)(blockIdx.x & (AZ_CTR_BUCKETS - 1)) * C_N + C_RULEMOVES], (unsigned long long)depth);
        atomicAdd(&D.ctr[(size_t)AZ_CTR_BUCKETS * C_N + (blockIdx.x & (AZ_CTR_BUCKETS - 1))], 1ULL);
        atomicAdd(&D.ctr[(size_t)(blockIdx.x & (AZ_CTR_BUCKETS - 1)) * C_N + C_EVALS], 1ULL);
    }
}

// k_select: one sub-round.  k_select_all: all leaves_per_round sub-rounds one after the other in one launch; they are
// sequential by construction (each sees the virtual losses of the ones before), and one wavefront per game runs them
// in program order, so fusing them only saves the launches.
template <typename R>
__global__ __launch_bounds__(64) void k_select(Dev D, int sub)
{
    __shared__ WLds L;
    select_one<R>(D, sub, L);
}

// (a kernel of its own: inside the loop the compiler keeps so much lane-invariant state in registers that the single
// sub-round form, which serves thousands of games at once, would drop to one wave per SIMD)
template <typename R>
__global__ __launch_bounds__(64) void k_select_all(Dev D)
{
    __shared__ WLds L;
    for (int j = 0; j < D.K; j++) {
        select_one<R>(D, j, L);
        WSYNC();
    }
}

// ---------------------------------------------------------------------------
// finish: expand or terminal value, then backup (mcts.py:217-232, 196-202, 283)
// ---------------------------------------------------------------------------

// returns 1 when the leaf has been dealt with, 0 when it is left pending (nothing to do, or an error stopped the slot)
template <typename R>
__device__ __forceinline__ int finish_one(const Dev &D, const int sub)
{
    R *const eW = static_cast<R *>(D.eW);
    R *const eP = static_cast<R *>(D.eP);
    R *const rootW = static_cast<R *>(D.rootW);
    const int g = blockIdx.x;
    const int li = sub * D.G + g;
    const int lane = lane_id();
    const Geo G = D.geo;
    if (!D.leaf_valid[li] || D.err[g] != 0) return 0;
    const int side = D.side[g];
    const int t = 2 * g + side;
    const size_t bb = (size_t)t * D.BC;
    const int root_expand = D.leaf_root[li];
    const int node = D.leaf_node[li];
    const size_t en = node == NODE_ROOT ? 0 : edge_at(D, t, node);   // the leaf's own edge
    const int player = D.leaf_player[li];
    const int tid = D.leaf_tid[li];
    R v = static_cast<const R *>(D.values)[li];
    const uint8_t *mask = D.leaf_mask + (size_t)li * AZ_OP;
    const R *pri = static_cast<const R *>(D.priors) + (size_t)li * G.O;
    const R vl = D.K > 1 ? (R)D.vloss : (R)0.0;
    WSYNC();
    // with several leaves in flight an earlier sub-round of this round may already have expanded the node
    const int has_children = D.K > 1 && (node == NODE_ROOT ? D.rootBlock[t] >= 0 : D.eChild[en] >= 0);
    const int expandable = root_expand ? 1 : !D.leaf_end[li];
    if (expandable && has_children) {
        if (lane == 0) {
            R add = (R)(-player) * v;
            if (node == NODE_ROOT) rootW[t] += add; else eW[en] += add + vl;
        }
    } else if (expandable) {
        int bi = D.nBlocks[t], start = D.nEdges[t];
        // count legal children first so that a full arena is detected before writing
        int n = 0;
        for (int i0 = 0; i0 < G.O; i0 += AZ_WAVE) {
            int i = i0 + lane;
            int legal = 0;
            if (i < G.O) {
                int a = (i < G.NC && tid) ? transform_src(i, G.H, G.W, tid) : i;
                legal = mask[a];
            }
            n += __popcll(__ballot(legal));
        }
        // a full arena is an error, or (tree_full_policy 1, never at a root) the node simply stays a leaf for this
        // playout: its value is backed up as usual and the event is counted
        // (the tree's own limits first; then the pages the new children need, which the shared pool may not have)
        int full = (bi >= D.BC || start + n > D.EC);
        if (!full && n > 0 && !tree_reserve(D, t, start + n)) full = 1;
        if (full && (D.full_policy != 1 || root_expand)) { if (lane == 0) D.err[g] = -4; return 0; }
        const Span sp = span_of(D, t, start);
        int off = 0;
        for (int i0 = 0; i0 < G.O && !full; i0 += AZ_WAVE) {
            int i = i0 + lane;
            int legal = 0, a = 0;
            if (i < G.O) {
                a = (i < G.NC && tid) ? transform_src(i, G.H, G.W, tid) : i;
                legal = mask[a];
            }
            unsigned long long bal = __ballot(legal);
            int pos = off + __popcll(bal & ((1ULL << lane) - 1ULL));
            if (legal) {
                size_t e = sp.at(start + pos);
                D.eN[e] = 0; eW[e] = (R)0.0; eP[e] = pri[i]; D.eChild[e] = -1; D.eAct[e] = (int16_t)a;
            }
            off += __popcll(bal);
        }
        if (lane == 0) {
            if (full) atomicAdd(&D.ctr[(size_t)(blockIdx.x & (AZ_CTR_BUCKETS - 1)) * C_N + C_SKIPPED], 1ULL);
            if (n > 0 && !full) {
                D.bStart[bb + bi] = start; D.bCount[bb + bi] = n;
                D.bOwner[bb + bi] = node; D.bParent[bb + bi] = D.leaf_block[li];
                D.nBlocks[t] = bi + 1; D.nEdges[t] = start + n;
                int32_t *pk = D.peak + 2 * (blockIdx.x & (AZ_CTR_BUCKETS - 1));
                if (start + n > pk[0]) atomicMax(&pk[0], start + n);
                if (bi + 1 > pk[1]) atomicMax(&pk[1], bi + 1);
                if (node == NODE_ROOT) D.rootBlock[t] = bi; else D.eChild[en] = bi;
                atomicAdd(&D.ctr[(size_t)(blockIdx.x & (AZ_CTR_BUCKETS - 1)) * C_N + C_EDGES], (unsigned long long)n);
            }
            // node.update(value): raw value at the root, -player*value below it
            R add = root_expand ? v : (R)(-player) * v;
            if (node == NODE_ROOT) rootW[t] += add; else eW[en] += add + vl;
        }
    } else {
        v = (R)D.leaf_winner[li];  // value = state.get_winner()
        if (lane == 0) {
            R add = (R)(-player) * v;
            if (node == NODE_ROOT) rootW[t] += add; else eW[en] += add + vl;
        }
    }
    WSYNC();
    if (root_expand) {
        if (D.dflag[g]) apply_noise<R>(D, g, t);
        if (lane == 0) {
            D.pdone[g] = 0; D.issued[g] = 0;
            D.phase[g] = D.max_playout == 0 ? PH_SEARCH_DONE : PH_IN_SEARCH;
        }
        return 1;
    }
    // ancestors: node.update(-current_player * simres_value)
    int plen = D.path_len[li];
    for (int d = lane; d < plen; d += AZ_WAVE) {
        int nd = D.path_node[(size_t)li * D.MAXD + d];
        R add = (R)(-(int)D.path_player[(size_t)li * D.MAXD + d]) * v;
        if (nd == NODE_ROOT) rootW[t] += add; else eW[edge_at(D, t, nd)] += add + vl;
    }
    if (lane == 0) {
        int pd = D.pdone[g] + 1;
        D.pdone[g] = pd;
        if (pd >= D.max_playout) D.phase[g] = PH_SEARCH_DONE;
    }
    return 1;
}

// sub as in k_select: one sub-round, or (sub < 0) all of them in order within one launch
template <typename R>
__global__ __launch_bounds__(64) void k_finish(Dev D, int sub)
{
    const int first = sub >= 0 ? sub : 0, last = sub >= 0 ? sub + 1 : D.K;
    for (int j = first; j < last; j++) {
        const int done = finish_one<R>(D, j);
        WSYNC();
        // a leaf is consumed exactly once: running the finish step again (after a tree arena was enlarged) only
        // completes what an error left pending
        if (done && lane_id() == 0) D.leaf_valid[j * D.G + blockIdx.x] = 0;
        WSYNC();
    }
}

// ---------------------------------------------------------------------------
// weighted_random_choice over N_a / sum N (search/mcts.py:234-243,
// math_helper.py:20-23, CPython random.choices): returns the action
// ---------------------------------------------------------------------------

__device__ int dev_choose(const Dev &D, WLds &L, int t, double u, int *n_out)
{
    const int lane = lane_id();
    size_t bb = (size_t)t * D.BC;
    int rb = D.rootBlock[t];
    if (rb < 0) { *n_out = 0; return NOMOVE; }
    int start = D.bStart[bb + rb], n = D.bCount[bb + rb];
    const Span sp = span_of(D, t, start);
    *n_out = n;
    for (int c = lane; c < n; c += AZ_WAVE) L.libs[c] = D.eN[sp.at(start + c)];
    WSYNC();
    if (lane == 0) {
        long long total = 0;
        for (int c = 0; c < n; c++) total += L.libs[c];
        double acc = 0.0;
        for (int c = 0; c < n; c++) {
            double w = (double)L.libs[c] / (double)total;
            acc = c == 0 ? w : acc + w;
            L.cum[c] = acc;
        }
        double x = u * (L.cum[n - 1] + 0.0);
        int lo = 0, hi = n - 1;
        while (lo < hi) {
            int mid = (lo + hi) / 2;
            if (x < L.cum[mid]) hi = mid; else lo = mid + 1;
        }
        L.tmp[0] = lo;
    }
    WSYNC();
    int idx = L.tmp[0];
    WSYNC();
    return D.eAct[sp.at(start + idx)];
}

// one ply of Game.start for every slot whose search is complete
// (game/gameplay.py:52-69): pick the move, log it, play it, queue the re-roots
__global__ __launch_bounds__(64) void k_move(Dev D)
{
    __shared__ WLds L;
    const int g = blockIdx.x;
    const int lane = lane_id();
    const Geo G = D.geo;
    if (D.phase[g] != PH_SEARCH_DONE || !D.automode[g] || D.err[g] != 0) return;
    const int side = D.side[g];
    const int t = 2 * g + side;
    const size_t bb = (size_t)t * D.BC;
    double u = rng_uniform(D, g);
    int n = 0;
    int mv = dev_choose(D, L, t, u, &n);
    if (mv == NOMOVE) { if (lane == 0) D.err[g] = -7; return; }
    int ply = D.ply[g];
    if (D.record && ply < D.traj_plies) {
        int32_t *row = D.tr_visits + ((size_t)g * D.traj_plies + ply) * G.O;
        for (int i = lane; i < G.O; i += AZ_WAVE) row[i] = 0;
        WSYNC();
        int start = D.bStart[bb + D.rootBlock[t]];
        const Span sp = span_of(D, t, start);
        for (int c = lane; c < n; c += AZ_WAVE) row[D.eAct[sp.at(start + c)]] = D.eN[sp.at(start + c)];
        if (lane == 0) D.tr_moves[(size_t)g * D.traj_plies + ply] = (int16_t)mv;
    }
    PS s;
    load_scalars(D, g, s);
    load_ring(D, g, L);
    const SK sk = slot_sk(D, g);
    dev_do_move(G, L, s, mv, &sk);
    store_ring(D, g, L);
    store_scalars(D, g, s);
    int over = s.is_end || s.turns > D.max_turn;
    int win = 0;
    if (over) win = dev_winner(G, L, s);
    if (lane == 0) {
        D.pending[2 * g] = mv;
        D.pending[2 * g + 1] = mv;
        D.ply[g] = ply + 1;
        D.side[g] = side ^ 1;
        atomicAdd(&D.ctr[(size_t)(blockIdx.x & (AZ_CTR_BUCKETS - 1)) * C_N + C_MOVES], 1ULL);
        if (over) {
            D.tr_winner[g] = win;
            D.phase[g] = PH_GAME_DONE;
            atomicAdd(&D.ctr[(size_t)(blockIdx.x & (AZ_CTR_BUCKETS - 1)) * C_N + C_GAMES], 1ULL);
        } else {
            D.phase[g] = PH_NEED_START;
            D.dflag[g] = s.turns <= D.dirichlet_before;
        }
    }
}

// ---------------------------------------------------------------------------
// update_with_move (search/mcts.py:336-346): keep the chosen child's subtree,
// compacting the tree's arena in place (blocks keep their creation order, so
// every block only ever moves towards lower addresses)
// ---------------------------------------------------------------------------

template <typename R>
__global__ __launch_bounds__(64) void k_reroot(Dev D)
{
    R *const eW = static_cast<R *>(D.eW);
    R *const eP = static_cast<R *>(D.eP);
    R *const rootW = static_cast<R *>(D.rootW);
    R *const rootP = static_cast<R *>(D.rootP);
    const int t = blockIdx.x;
    const int lane = lane_id();
    const int mv = D.pending[t];
    if (mv == NOMOVE) return;
    WSYNC();
    if (lane == 0) D.pending[t] = NOMOVE;
    const size_t bb = (size_t)t * D.BC;
    const int rb = D.rootBlock[t];
    int found = -1;
    if (rb >= 0) {
        int start = D.bStart[bb + rb], n = D.bCount[bb + rb];
        const Span sp = span_of(D, t, start);
        for (int c0 = 0; c0 < n; c0 += AZ_WAVE) {
            int c = c0 + lane;
            int hit = (c < n) && (D.eAct[sp.at(start + min(c, n - 1))] == mv);
            unsigned long long bal = __ballot(hit);
            if (bal) { found = start + c0 + __ffsll((long long)bal) - 1; break; }
        }
    }
    if (found < 0) {  // MCTreeNode(None, 1.0)
        if (lane == 0) {
            D.rootN[t] = 0; rootW[t] = (R)0.0; rootP[t] = (R)1.0;
            D.rootBlock[t] = -1; D.nBlocks[t] = 0; D.nEdges[t] = 0;
        }
        tree_release(D, t, 0);
        return;
    }
    const size_t ef = edge_at(D, t, found);
    const int r = D.eChild[ef];
    const int nN = D.eN[ef];
    const R nW = eW[ef], nP = eP[ef];
    const int nb = D.nBlocks[t];
    WSYNC();
    if (lane == 0) { D.rootN[t] = nN; rootW[t] = nW; rootP[t] = nP; }
    if (r < 0) {
        if (lane == 0) { D.rootBlock[t] = -1; D.nBlocks[t] = 0; D.nEdges[t] = 0; }
        tree_release(D, t, 0);
        return;
    }
    // 1. liveness: a block lives iff it is r or its parent block lives (parent index < own index)
    for (int b = lane; b <= r; b += AZ_WAVE) D.bMap[bb + b] = (b == r);
    WSYNC();
    for (int c0 = r + 1; c0 < nb; c0 += AZ_WAVE) {
        int b = c0 + lane;
        int valid = b < nb;
        int par = valid ? D.bParent[bb + b] : -1;
        int live = valid && par >= 0 && par < c0 && D.bMap[bb + par] != 0;
        unsigned long long m = __ballot(live);
        for (;;) {
            int nl = valid && !live && par >= c0 && ((m >> (par - c0)) & 1ULL);
            live |= nl;
            unsigned long long m2 = __ballot(live);
            if (m2 == m) break;
            m = m2;
        }
        if (valid) D.bMap[bb + b] = live;
        WSYNC();
    }
    // 2. new block index (bMap) and address shift (bShift) of every live block
    int nbNew = 0, neNew = 0;
    for (int c0 = r; c0 < nb; c0 += AZ_WAVE) {
        int b = c0 + lane;
        int live = (b < nb) && D.bMap[bb + b] != 0;
        int cnt = live ? D.bCount[bb + b] : 0;
        int st = live ? D.bStart[bb + b] : 0;
        unsigned long long m = __ballot(live);
        int idx = nbNew + __popcll(m & ((1ULL << lane) - 1ULL));
        int incl = cnt;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            int up = __shfl_up(incl, o, 64);
            if (lane >= o) incl += up;
        }
        int excl = incl - cnt;
        int tot = __shfl(incl, 63, 64);
        WSYNC();
        if (b < nb) {
            D.bMap[bb + b] = live ? idx : -1;
            D.bShift[bb + b] = live ? st - (neNew + excl) : 0;
        }
        nbNew += __popcll(m);
        neNew += tot;
    }
    WSYNC();
    // 3. slide edges and rewrite the block table, lowest block first
    for (int b = r; b < nb; b++) {
        int ni = D.bMap[bb + b];
        if (ni < 0) continue;
        int os = D.bStart[bb + b], cnt = D.bCount[bb + b], sh = D.bShift[bb + b];
        int owner = D.bOwner[bb + b], par = D.bParent[bb + b];
        int ns = os - sh;
        int nowner = -1, npar = -1;
        if (b != r) { nowner = owner - D.bShift[bb + par]; npar = D.bMap[bb + par]; }
        const Span so = span_of(D, t, os), sn = span_of(D, t, ns);
        WSYNC();
        for (int c0 = 0; c0 < cnt; c0 += AZ_WAVE) {
            int c = c0 + lane;
            int N = 0, ch = -1; R Wv = 0, P = 0; int16_t a = 0;
            if (c < cnt) {
                size_t e = so.at(os + c);
                N = D.eN[e]; Wv = eW[e]; P = eP[e]; ch = D.eChild[e]; a = D.eAct[e];
                if (ch >= 0) ch = D.bMap[bb + ch];
            }
            WSYNC();
            if (c < cnt) {
                size_t e = sn.at(ns + c);
                D.eN[e] = N; eW[e] = Wv; eP[e] = P; D.eChild[e] = ch; D.eAct[e] = a;
            }
            WSYNC();
        }
        if (lane == 0) {
            D.bStart[bb + ni] = ns; D.bCount[bb + ni] = cnt;
            D.bOwner[bb + ni] = nowner; D.bParent[bb + ni] = npar;
        }
        WSYNC();
    }
    if (lane == 0) { D.rootBlock[t] = 0; D.nBlocks[t] = nbNew; D.nEdges[t] = neNew; }
    tree_release(D, t, (neNew + (1 << D.PS) - 1) >> D.PS);   // the pages behind the kept subtree go back to the pool
}

// ---------------------------------------------------------------------------
// batched rules / features entry kernels (env modules, state_converter)
// ---------------------------------------------------------------------------

__device__ __forceinline__ void init_state(const Geo &G, WLds &L, PS &s)
{
    int *p = reinterpret_cast<int *>(&L.ring[0][0]);
    for (int q = lane_id(); q < 8 * AZ_CP / 4; q += AZ_WAVE) p[q] = 0;
    WSYNC();
    s.head = 0; s.cur = 1; s.ko = -1; s.turns = 0; s.nhist = 0; s.last1 = NOMOVE; s.last2 = NOMOVE;
    s.is_end = 0; s.winner = 0; s.winner_set = 0; s.passes_b = 0; s.passes_w = 0; s.pris_b = 0; s.pris_w = 0;
    s.stones = 0; s.nprev = 0; s.h0 = 0; s.h1 = 0;
    if (G.game == G_REVERSI) {  // env/reversi.py:20-23
        int h = G.H / 2;
        if (lane_id() == 0) {
            L.ring[0][(h - 1) * G.W + (h - 1)] = -1; L.ring[0][h * G.W + h] = -1;
            L.ring[0][h * G.W + (h - 1)] = 1; L.ring[0][(h - 1) * G.W + h] = 1;
        }
        s.stones = 4;
        WSYNC();
    }
}

__global__ __launch_bounds__(64) void k_reset(Dev D, const int32_t *slots, int n, int automode)
{
    __shared__ WLds L;
    int q = blockIdx.x;
    if (q >= n) return;
    int g = slots ? slots[q] : q;
    PS s;
    init_state(D.geo, L, s);
    store_ring(D, g, L);
    store_scalars(D, g, s);
    tree_release(D, 2 * g, 0);
    tree_release(D, 2 * g + 1, 0);
    if (lane_id() == 0) {
        D.phase[g] = automode ? PH_NEED_START : PH_IDLE;
        D.side[g] = 0; D.dflag[g] = automode ? (0 <= D.dirichlet_before) : 0; D.pdone[g] = 0; D.issued[g] = 0;
        D.automode[g] = automode; D.ply[g] = 0; D.err[g] = 0; D.leaf_valid[g] = 0; D.tr_winner[g] = 0;
        for (int k = 0; k < 2; k++) {
            int t = 2 * g + k;
            D.rootN[t] = 0; D.rootBlock[t] = -1;
            if (D.f64) { static_cast<double *>(D.rootW)[t] = 0.0; static_cast<double *>(D.rootP)[t] = 1.0; }
            else { static_cast<float *>(D.rootW)[t] = 0.0f; static_cast<float *>(D.rootP)[t] = 1.0f; }
            D.nBlocks[t] = 0; D.nEdges[t] = 0; D.pending[t] = NOMOVE;
        }
    }
}

__global__ __launch_bounds__(64) void k_rules_mask(Dev D, const int32_t *slots, int n, uint8_t *out)
{
    __shared__ WLds L;
    int q = blockIdx.x;
    if (q >= n) return;
    int g = slots[q];
    PS s;
    load_scalars(D, g, s);
    load_ring(D, g, L);
    const SK sk = slot_sk(D, g);
    dev_legal_mask(D.geo, L, s, &sk);
    for (int i = lane_id(); i < D.geo.O; i += AZ_WAVE) out[(size_t)q * D.geo.O + i] = L.mask[i];
}

__global__ __launch_bounds__(64) void k_rules_domove(Dev D, const int32_t *slots, int n, const int32_t *actions,
                                                     const int32_t *colors, int32_t *status, int32_t *ended)
{
    __shared__ WLds L;
    int q = blockIdx.x;
    if (q >= n) return;
    int g = slots[q], a = actions[q];
    PS s;
    load_scalars(D, g, s);
    load_ring(D, g, L);
    int color = colors ? colors[q] : 0;
    if (color != 0) s.cur = color;  // legality is judged for the stated colour (env/go.py:511-514)
    const SK sk = slot_sk(D, g);
    dev_legal_mask(D.geo, L, s, &sk);
    int ok = a >= 0 && a < D.geo.O && L.mask[a];
    WSYNC();
    if (!ok) {
        if (lane_id() == 0) { status[q] = -3; ended[q] = s.is_end; }
        return;
    }
    if (D.geo.superko && s.nhist >= D.sk_cap) {   // the superko record of this slot is full: the move is refused, not forgotten
        if (lane_id() == 0) { status[q] = -7; ended[q] = s.is_end; }
        return;
    }
    dev_do_move(D.geo, L, s, a, &sk);
    store_ring(D, g, L);
    store_scalars(D, g, s);
    if (lane_id() == 0) { status[q] = 0; ended[q] = s.is_end; }
}

__global__ __launch_bounds__(64) void k_rules_winner(Dev D, const int32_t *slots, int n, int32_t *out)
{
    __shared__ WLds L;
    int q = blockIdx.x;
    if (q >= n) return;
    int g = slots[q];
    PS s;
    load_scalars(D, g, s);
    load_ring(D, g, L);
    int w = dev_winner(D.geo, L, s);
    if (lane_id() == 0) out[q] = w;
}

__global__ __launch_bounds__(64) void k_features(Dev D, const int32_t *slots, int n, const int32_t *tids, int8_t *out)
{
    __shared__ WLds L;
    int q = blockIdx.x;
    if (q >= n) return;
    int g = slots[q];
    PS s;
    load_scalars(D, g, s);
    load_ring(D, g, L);
    dev_features(D.geo, L, s, tids ? tids[q] : 0, out + (size_t)q * 17 * D.geo.NC);
}

// GameState.transform(id): board and the 7 history boards, in place
__global__ __launch_bounds__(64) void k_transform(Dev D, int g, int tid)
{
    __shared__ WLds L;
    __shared__ int8_t T[8][AZ_CP];
    PS s;
    load_scalars(D, g, s);
    load_ring(D, g, L);
    for (int i = lane_id(); i < D.geo.NC; i += AZ_WAVE) {
        int src = transform_src(i, D.geo.H, D.geo.W, tid);
        for (int r = 0; r < 8; r++) T[r][i] = L.ring[r][src];
    }
    WSYNC();
    for (int i = lane_id(); i < D.geo.NC; i += AZ_WAVE)
        for (int r = 0; r < 8; r++) L.ring[r][i] = T[r][i];
    store_ring(D, g, L);
}

// copy one slot's state between engines' arrays (same device)
__global__ __launch_bounds__(64) void k_state_copy(Dev Dd, int gd, Dev Ds, int gs, int ref_semantics)
{
    __shared__ WLds L;
    PS s;
    load_scalars(Ds, gs, s);
    load_ring(Ds, gs, L);
    if (ref_semantics && Ds.geo.game == G_GO) { s.is_end = 0; s.passes_b = 0; s.passes_w = 0; }
    if (Dd.geo.superko) {   // copy() hands the move list and previous_hashes on (go.py:247-253); the host checked the sizes
        const SK a = slot_sk(Ds, gs), b = slot_sk(Dd, gd);
        for (int i = lane_id(); i < min(s.nhist, b.cap); i += AZ_WAVE) b.moves[i] = a.moves[i];
        for (int i = lane_id(); i < 2 * min(s.nprev, b.cap); i += AZ_WAVE) b.hash[i] = a.hash[i];
    }
    store_ring(Dd, gd, L);
    store_scalars(Dd, gd, s);
}

__global__ void k_begin(Dev D, const int32_t *slots, int n, const int32_t *sides, const int32_t *dflags)
{
    int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n) return;
    int g = slots[q];
    D.phase[g] = PH_NEED_START; D.side[g] = sides[q]; D.dflag[g] = dflags[q]; D.pdone[g] = 0; D.issued[g] = 0; D.automode[g] = 0;
}

__global__ void k_set_pending(Dev D, const int32_t *slots, int n, const int32_t *moves, int side_sel)
{
    int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n) return;
    int g = slots[q];
    if (side_sel < 0 || side_sel == 0) D.pending[2 * g] = moves[q];
    if (side_sel < 0 || side_sel == 1) D.pending[2 * g + 1] = moves[q];
}

__global__ __launch_bounds__(64) void k_choose(Dev D, int g, int side, double u, int32_t *out)
{
    __shared__ WLds L;
    int n = 0;
    int mv = dev_choose(D, L, 2 * g + side, u, &n);
    if (lane_id() == 0) out[0] = mv;
}

// after the tree arenas were enlarged: slots stopped by a full arena may go on
__global__ void k_clear_tree_full(Dev D)
{
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g < D.G && D.err[g] == -4) D.err[g] = 0;
}

// A leaf row needs the evaluator unless it is a terminal state below the root: that one takes its value from the rules
// (mcts.py:229-231) and never looks at priors or value.  (The reference still calls its evaluator there and throws the
// answer away.)
__device__ __forceinline__ bool leaf_needs_eval(const Dev &D, int row)
{
    return D.leaf_valid[row] && (D.leaf_root[row] || !D.leaf_end[row]);
}

// Rows that need the evaluator, packed to the front of `rows` in row order; their number goes to count[0] and is added
// to total[0].  One workgroup, n_rows = leaves_per_round * n_slots.
__global__ __launch_bounds__(1024) void k_compact(Dev D, int n_rows, int32_t *__restrict__ rows, int32_t *__restrict__ count,
                                                  unsigned long long *__restrict__ total)
{
    __shared__ int wtot[16];
    __shared__ int base;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) base = 0;
    __syncthreads();
    for (int r0 = 0; r0 < n_rows; r0 += 1024) {
        const int r = r0 + tid;
        const bool live = r < n_rows && leaf_needs_eval(D, r);
        const unsigned long long m = __ballot(live);
        if (lane == 0) wtot[wave] = __popcll(m);
        __syncthreads();
        if (live) {
            int off = base;
            for (int w = 0; w < wave; w++) off += wtot[w];
            rows[off + __popcll(m & ((1ull << lane) - 1ull))] = r;
        }
        __syncthreads();
        if (tid == 0) {
            int t = 0;
            for (int w = 0; w < 16; w++) t += wtot[w];
            base += t;
        }
        __syncthreads();
    }
    if (tid == 0) { count[0] = base; total[0] += (unsigned long long)base; }
}

// ---------------------------------------------------------------------------
// Two-network matches (train/parallel/evaluator.py:81-84: Game(chal, best) or Game(best, chal)).
// Splits the round's valid leaves by the network that owns the searching tree: side s of slot g is
// evaluated by network net_of_first[g] ^ s.  One workgroup; each list keeps slot order.
// rows[net * G + i] = i-th slot of that network, counts[net] = length.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_partition(Dev D, const int32_t *__restrict__ net_of_first, int32_t *__restrict__ rows,
                                                    int32_t *__restrict__ counts, unsigned long long *__restrict__ totals)
{
    __shared__ int wtot[2][16];
    __shared__ int base[2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid < 2) base[tid] = 0;
    __syncthreads();
    for (int g0 = 0; g0 < D.G; g0 += 1024) {
        const int g = g0 + tid;
        int net = -1;
        if (g < D.G && leaf_needs_eval(D, g)) net = (net_of_first[g] ^ D.side[g]) & 1;
        const unsigned long long m0 = __ballot(net == 0), m1 = __ballot(net == 1);
        if (lane == 0) { wtot[0][wave] = __popcll(m0); wtot[1][wave] = __popcll(m1); }
        __syncthreads();
        if (net >= 0) {
            int off = base[net];
            for (int w = 0; w < wave; w++) off += wtot[net][w];
            off += __popcll((net ? m1 : m0) & ((1ull << lane) - 1ull));
            rows[(size_t)net * D.G + off] = g;
        }
        __syncthreads();
        if (tid < 2) {
            int t = 0;
            for (int w = 0; w < 16; w++) t += wtot[tid][w];
            base[tid] += t;
        }
        __syncthreads();
    }
    if (tid < 2) { counts[tid] = base[tid]; totals[tid] += (unsigned long long)base[tid]; }  // one block: no race
}
