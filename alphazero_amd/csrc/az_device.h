// Device-side data layout and rule kernels' building blocks (gfx950, wave64).
//
// One wavefront owns one game.  The eight most recent boards of the game
// (current + 7 history) sit in LDS as a ring of int8 planes; lane l owns the
// cells l, l+64, l+128, ... of every board.  Scalars of the state are
// wave-uniform.  All functions here are called by every lane of the wave.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define AZ_CP 384   // bytes per board (cells padded)
#define AZ_OP 384   // bytes per action mask
#define AZ_NJ 6     // ceil(361 / 64)
#define AZ_WAVE 64

enum { G_GO = 0, G_REVERSI = 1, G_MNK = 2 };

// int32 scalars of one game state, one 64-byte line per slot
enum {
    S_HEAD = 0, S_CUR, S_KO, S_TURNS, S_NHIST, S_LAST1, S_LAST2, S_END, S_WIN, S_WINSET,
    S_PB, S_PW, S_PRB, S_PRW, S_STONES,
    S_NPREV,                      // positional superko: positions recorded so far (one per stone move)
    S_H0A, S_H0B, S_H1A, S_H1B,   // 128-bit hash of the current position, as four 32-bit words
    S_N = 32
};

// search machine phases of a slot
enum { PH_IDLE = 0, PH_NEED_START = 1, PH_IN_SEARCH = 2, PH_SEARCH_DONE = 3, PH_GAME_DONE = 4 };

#define NODE_ROOT (-1)
#define NOMOVE (-2)

enum {
    C_STEPS = 0, C_EVALS, C_MOVES, C_GAMES, C_LEVELS, C_CHILDREN, C_EDGES, C_RULEMOVES, C_NLEAVES, C_SKIPPED, C_N = 16
};

struct Geo {
    int game, H, W, NC, O, k;
    int superko;   // Go: enforce positional superko (env/go.py:303-330; the reference's GTP games)
};

// Per-slot record that positional superko needs: every move of the game (PASS included) and the hash of the position
// after every stone move.  A playout appends behind the game's own entries and leaves the game's counts untouched, so
// what it wrote is simply overwritten later.
struct SK {
    int16_t *moves;              // [cap]
    unsigned long long *hash;    // [cap][2]
    int cap;
};

struct PS {  // wave-uniform scalars of a (playout) state
    int head, cur, ko, turns, nhist, last1, last2, is_end, winner, winner_set;
    int passes_b, passes_w, pris_b, pris_w, stones;
    int nprev;
    unsigned long long h0, h1;
};

struct __align__(16) WLds {
    int8_t ring[8][AZ_CP];
    int16_t label[AZ_CP];
    int32_t libs[AZ_CP];
    uint8_t mask[AZ_OP];
    int32_t tmp[64];
    double cum[AZ_OP];
};

#define AZ_CTR_BUCKETS 256
struct Dev {
    Geo geo;
    int G;
    int max_playout, transform_on, transform_types, dirichlet_before, max_turn;
    double d_alpha, d_epsilon;
    double c_punt;       // exploration constant of the selection score (search/mcts.py:9)
    int EC, BC, MAXD;
    int rng_replay, record, traj_plies;
    // Several leaves of one tree per round (not in the reference, which has one playout in flight; changes visit
    // counts): the round's select and finish kernels run K times, sub-round j keeping its leaf in row j*G + g of every
    // per-leaf array below, and every edge on a path in flight carries a virtual loss vloss on W until its value arrives.
    int K;
    double vloss;
    int full_policy;     // 0: a full tree arena is an error; 1: the expansion is skipped and counted
    unsigned long long seed;
    // state
    int8_t *ring;        // [G][8][AZ_CP]
    int32_t *sc;         // [G][S_N]
    // machine
    int32_t *phase, *side, *dflag, *pdone, *issued, *automode, *ply, *err;  // issued: playouts started in this search
    // leaves of the current round: rows [K*G]
    int32_t *leaf_valid, *leaf_node, *leaf_block, *leaf_player, *leaf_end, *leaf_winner, *leaf_tid, *leaf_root, *path_len;
    int32_t *path_node;  // [G][MAXD]
    int8_t *path_player; // [G][MAXD]
    uint8_t *leaf_mask;  // [G][AZ_OP]
    int8_t *planes;      // [G][17][NC]
    // search statistics and evaluator outputs are float32, or float64 when f64 is set (an evaluator that returns
    // Python floats / float64 makes every intermediate of the reference's search a double): R = float | double
    int f64;
    void *priors;        // R [G][O]
    void *values;        // R [G]
    // trees [2G]
    int32_t *rootN; void *rootW; void *rootP; int32_t *rootBlock; int32_t *nBlocks; int32_t *nEdges;   // rootW, rootP: R
    // Edges live in pages of (1 << PS) edges that all trees of the engine draw from one pool: a tree addresses its edges
    // 0 .. nEdges-1 as before (children of a node contiguous, blocks in creation order) and pt maps each page of that
    // range to a page of the pool.  EC stays the most edges one tree may hold (the length of its page table).
    int32_t *eN; void *eW; void *eP; int32_t *eChild; int16_t *eAct;       // [pool_pages << PS]; eW, eP: R
    int PS, PTC, pool_pages;   // page shift; page-table entries per tree = ceil(EC / page); pages in the pool
    int32_t *pt;               // [2G][PTC] pool page of each page of a tree (entries below nPages are valid)
    int32_t *nPages;           // [2G] pages a tree holds: max(1, ceil(nEdges / page)); page 0 of tree t is pool page t for good
    int32_t *pool_free;        // [pool_pages] stack of free pool pages
    int32_t *pool_top;         // [0] free pages on the stack; [1] the lowest [0] was when pages were last given back (with
                               // the present [0], the low-water mark since the counters were reset)
    int32_t *bStart, *bCount, *bOwner, *bParent, *bMap, *bShift;             // [2G][BC]
    int32_t *pending;    // [2G] move to re-root with, NOMOVE = none
    // rng
    const int32_t **rt; long long *rt_n, *rt_i;
    const double **re; long long *re_n, *re_i;
    const double **ru; long long *ru_n, *ru_i;
    unsigned long long *rng_ctr;  // [G]
    // trajectories
    int16_t *tr_moves;   // [G][traj_plies]
    int32_t *tr_visits;  // [G][traj_plies][O]
    int32_t *tr_winner;  // [G]
    // counters: AZ_CTR_BUCKETS rows of C_N (a workgroup adds into row blockIdx.x % AZ_CTR_BUCKETS, so thousands of
    // workgroups do not queue on one address), then AZ_CTR_BUCKETS words for the round's leaf count; the host sums
    unsigned long long *ctr;
    // largest number of edges / expanded nodes any tree has held since the counters were reset: AZ_CTR_BUCKETS pairs,
    // a workgroup raises the pair of its bucket (the host takes the maximum)
    int32_t *peak;
    // positional superko (geo.superko): move and position-hash history per slot
    int16_t *sk_moves;             // [G][sk_cap]
    unsigned long long *sk_hash;   // [G][sk_cap][2]
    int sk_cap;
};

__device__ __forceinline__ SK slot_sk(const Dev &D, int g)
{
    SK k;
    k.moves = D.sk_moves ? D.sk_moves + (size_t)g * D.sk_cap : nullptr;
    k.hash = D.sk_hash ? D.sk_hash + (size_t)g * D.sk_cap * 2 : nullptr;
    k.cap = D.sk_cap;
    return k;
}

#define WSYNC() __syncthreads()

// ---------------------------------------------------------------------------
// small helpers
// ---------------------------------------------------------------------------

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

// ---------------------------------------------------------------------------
// paged tree edges
// ---------------------------------------------------------------------------

// pool index of edge v of tree t
__device__ __forceinline__ size_t edge_at(const Dev &D, int t, int v)
{
    return ((size_t)D.pt[(size_t)t * D.PTC + (v >> D.PS)] << D.PS) | (size_t)(v & ((1 << D.PS) - 1));
}

// The children of one node are contiguous in the tree's own numbering and never more than a page (a page holds at least
// as many edges as the game has actions), so they touch at most two pages: both are looked up once.
struct Span {
    size_t b0, b1;
    int pg0, sh, mask;
    __device__ __forceinline__ size_t at(int v) const { return ((v >> sh) == pg0 ? b0 : b1) + (size_t)(v & mask); }
};

__device__ __forceinline__ Span span_of(const Dev &D, int t, int start)
{
    Span s;
    s.sh = D.PS; s.mask = (1 << D.PS) - 1; s.pg0 = start >> D.PS;
    const int32_t *pt = D.pt + (size_t)t * D.PTC;
    s.b0 = (size_t)pt[min(s.pg0, D.PTC - 1)] << D.PS;
    s.b1 = (size_t)pt[min(s.pg0 + 1, D.PTC - 1)] << D.PS;   // (read even when the tree does not hold that page: never used then)
    return s;
}

// Makes tree t hold the pages its first n_edges edges need; called by every lane, lane 0 takes pages off the pool's free
// stack.  Returns 0 when the pool has too few pages left (nothing is taken then).  Only kernels that never give pages
// back may call this: the stack's entries below its top were written by earlier launches.
__device__ __forceinline__ int tree_reserve(const Dev &D, int t, int n_edges)
{
    const int need = (n_edges + (1 << D.PS) - 1) >> D.PS;
    const int have = D.nPages[t];
    if (need <= have) return 1;
    int ok = 1;
    if (lane_id() == 0) {
        const int k = need - have;
        const int top = atomicSub(&D.pool_top[0], k);
        if (top < k) {
            atomicAdd(&D.pool_top[0], k);
            ok = 0;
        } else {
            int32_t *pt = D.pt + (size_t)t * D.PTC;
            for (int i = 0; i < k; i++) pt[have + i] = D.pool_free[top - k + i];
            D.nPages[t] = need;
        }
    }
    ok = __shfl(ok, 0, 64);
    WSYNC();
    return ok;
}

// Tree t keeps its first `keep` pages and hands the others back to the pool.  Its first page is its own for good (pool
// page t): the children of a root always have a place, however empty the pool is.  Only kernels that never take pages
// may call this (the top moves before the entries are written).
__device__ __forceinline__ void tree_release(const Dev &D, int t, int keep)
{
    if (keep < 1) keep = 1;
    const int have = D.nPages[t];
    WSYNC();
    if (have <= keep) return;
    const int k = have - keep;
    int base = 0;
    if (lane_id() == 0) {
        base = atomicAdd(&D.pool_top[0], k);
        // the free count only falls while pages are being taken and only rises here: its low-water mark is what the first
        // release after a stretch of taking sees
        if (base < D.pool_top[1]) atomicMin(&D.pool_top[1], base);
    }
    base = __shfl(base, 0, 64);
    const int32_t *pt = D.pt + (size_t)t * D.PTC;
    for (int i = lane_id(); i < k; i += AZ_WAVE) D.pool_free[base + i] = pt[keep + i];
    if (lane_id() == 0) D.nPages[t] = keep;
    WSYNC();
}


__device__ __forceinline__ unsigned long long mix64(unsigned long long z)
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

__device__ __forceinline__ int wave_sum(int v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__device__ __forceinline__ unsigned long long wave_xor(unsigned long long v)
{
    unsigned lo = (unsigned)v, hi = (unsigned)(v >> 32);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { lo ^= __shfl_xor(lo, o, 64); hi ^= __shfl_xor(hi, o, 64); }
    return ((unsigned long long)hi << 32) | lo;
}

// 128-bit key of a stone of `color` on `cell` (the reference keeps a 64-bit Zobrist table, env/go.py:64-71; what matters
// is that equal positions hash equally and different ones practically never do)
__device__ __forceinline__ void zkey(int color, int cell, unsigned long long &a, unsigned long long &b)
{
    const unsigned long long k = (unsigned long long)(2 * cell + (color == 1));
    a = mix64(0x6A09E667F3BCC909ULL + k * 0x9E3779B97F4A7C15ULL);
    b = mix64(0xBB67AE8584CAA73BULL ^ (k * 0xD6E8FEB86659FD93ULL + a));
}

__device__ __forceinline__ int8_t *cur_board(WLds &L, const PS &s) { return L.ring[s.head & 7]; }
// board_history[t], t = 0 (oldest) .. 6 (newest); t = 7 is the current board
__device__ __forceinline__ int8_t *hist_board(WLds &L, const PS &s, int t) { return L.ring[(s.head + 1 + t) & 7]; }

__device__ __forceinline__ void load_scalars(const Dev &D, int g, PS &s)
{
    const int32_t *sc = D.sc + (size_t)g * S_N;
    s.head = sc[S_HEAD]; s.cur = sc[S_CUR]; s.ko = sc[S_KO]; s.turns = sc[S_TURNS];
    s.nhist = sc[S_NHIST]; s.last1 = sc[S_LAST1]; s.last2 = sc[S_LAST2]; s.is_end = sc[S_END];
    s.winner = sc[S_WIN]; s.winner_set = sc[S_WINSET]; s.passes_b = sc[S_PB]; s.passes_w = sc[S_PW];
    s.pris_b = sc[S_PRB]; s.pris_w = sc[S_PRW]; s.stones = sc[S_STONES];
    s.nprev = sc[S_NPREV];
    s.h0 = ((unsigned long long)(unsigned)sc[S_H0B] << 32) | (unsigned)sc[S_H0A];
    s.h1 = ((unsigned long long)(unsigned)sc[S_H1B] << 32) | (unsigned)sc[S_H1A];
}

__device__ __forceinline__ void store_scalars(const Dev &D, int g, const PS &s)
{
    if (lane_id() == 0) {
        int32_t *sc = D.sc + (size_t)g * S_N;
        sc[S_HEAD] = s.head; sc[S_CUR] = s.cur; sc[S_KO] = s.ko; sc[S_TURNS] = s.turns;
        sc[S_NHIST] = s.nhist; sc[S_LAST1] = s.last1; sc[S_LAST2] = s.last2; sc[S_END] = s.is_end;
        sc[S_WIN] = s.winner; sc[S_WINSET] = s.winner_set; sc[S_PB] = s.passes_b; sc[S_PW] = s.passes_w;
        sc[S_PRB] = s.pris_b; sc[S_PRW] = s.pris_w; sc[S_STONES] = s.stones;
        sc[S_NPREV] = s.nprev;
        sc[S_H0A] = (int)(unsigned)s.h0; sc[S_H0B] = (int)(unsigned)(s.h0 >> 32);
        sc[S_H1A] = (int)(unsigned)s.h1; sc[S_H1B] = (int)(unsigned)(s.h1 >> 32);
    }
}

// whole ring (3 KiB) <-> LDS, 16 bytes per lane per trip
__device__ __forceinline__ void load_ring(const Dev &D, int g, WLds &L)
{
    const int4 *src = reinterpret_cast<const int4 *>(D.ring + (size_t)g * 8 * AZ_CP);
    int4 *dst = reinterpret_cast<int4 *>(&L.ring[0][0]);
    for (int q = lane_id(); q < 8 * AZ_CP / 16; q += AZ_WAVE) dst[q] = src[q];
    WSYNC();
}

__device__ __forceinline__ void store_ring(const Dev &D, int g, WLds &L)
{
    WSYNC();
    int4 *dst = reinterpret_cast<int4 *>(D.ring + (size_t)g * 8 * AZ_CP);
    const int4 *src = reinterpret_cast<const int4 *>(&L.ring[0][0]);
    for (int q = lane_id(); q < 8 * AZ_CP / 16; q += AZ_WAVE) dst[q] = src[q];
}

// board_history.append(board.copy()); board_history.pop(0)
__device__ __forceinline__ void push_history(WLds &L, PS &s)
{
    const int8_t *b = cur_board(L, s);
    int nh = (s.head + 1) & 7;
    int8_t *d = L.ring[nh];
    const int *b4 = reinterpret_cast<const int *>(b);
    int *d4 = reinterpret_cast<int *>(d);
    for (int q = lane_id(); q < AZ_CP / 4; q += AZ_WAVE) d4[q] = b4[q];
    s.head = nh;
    WSYNC();
}

__device__ __forceinline__ void append_move(PS &s, int a)
{
    s.last2 = s.last1;
    s.last1 = a;
    s.nhist++;
}

// ---------------------------------------------------------------------------
// Go: chain labels (min cell index of the chain) and liberty counts
// ---------------------------------------------------------------------------

// The cells a lane owns (lane, lane + 64, ...) with their neighbourhood, gathered once per labelling so that the loops
// below are straight-line code: every LDS read is unconditional (a missing neighbour reads the cell itself) and the
// reads of all 6 cells are in flight together instead of one dependent, branch-guarded read after another.
//   info bits 0-3: neighbour (x-1, x+1, y-1, y+1) exists; bit 4: the cell is on the board;
//   bits 8-9: colour code of the cell, bits 10-17: colour codes of the four neighbours (0 empty, 1 black, 3 white)
struct GoCells {
    int idx[AZ_NJ];
    unsigned info[AZ_NJ];
};
__device__ __forceinline__ int go_nb(const GoCells &gc, int j, int q, int W)
{
    const int d = q == 0 ? -W : q == 1 ? W : q == 2 ? -1 : 1;
    return gc.idx[j] + (((gc.info[j] >> q) & 1u) ? d : 0);
}
__device__ __forceinline__ int go_col(const GoCells &gc, int j) { return (int)((gc.info[j] >> 8) & 3u); }
__device__ __forceinline__ int go_ncol(const GoCells &gc, int j, int q) { return (int)((gc.info[j] >> (10 + 2 * q)) & 3u); }

__device__ __forceinline__ void go_cells(const Geo &G, const int8_t *b, GoCells &gc)
{
    int lane = lane_id();
    // everything below derives from the lane number and the board geometry only, so a caller that labels inside a loop
    // would have all of it hoisted out and kept in registers (some 200 of them); the empty asm makes it per-call work
    asm volatile("" : "+v"(lane));
    const int W = G.W, H = G.H, NC = G.NC;
#pragma unroll
    for (int j = 0; j < AZ_NJ; j++) {
        const int i = lane + j * AZ_WAVE;
        const bool on = i < NC;
        const int ii = on ? i : 0;
        const int x = ii / W, y = ii - x * W;
        gc.idx[j] = ii;
        gc.info[j] = (on ? 16u : 0u) | (on && x > 0 ? 1u : 0u) | (on && x + 1 < H ? 2u : 0u) | (on && y > 0 ? 4u : 0u) |
                     (on && y + 1 < W ? 8u : 0u);
    }
#pragma unroll
    for (int j = 0; j < AZ_NJ; j++) {
        unsigned cc = (unsigned)b[gc.idx[j]] & 3u;
        if (!(gc.info[j] & 16u)) cc = 0;
        unsigned inf = gc.info[j] | (cc << 8);
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const unsigned cn = (unsigned)b[go_nb(gc, j, q, W)] & 3u;
            inf |= (((gc.info[j] >> q) & 1u) ? cn : 0u) << (10 + 2 * q);
        }
        gc.info[j] = inf;
    }
}

// chain labels and liberty counts of the board described by gc (go_cells of the same board)
__device__ __forceinline__ void go_label(const Geo &G, WLds &L, const GoCells &gc)
{
    const int W = G.W;
#pragma unroll
    for (int j = 0; j < AZ_NJ; j++)
        if (gc.info[j] & 16u) {
            L.label[gc.idx[j]] = (int16_t)gc.idx[j];
            L.libs[gc.idx[j]] = 0;
        }
    WSYNC();
    for (;;) {
        int changed = 0;
        int m0[AZ_NJ], m[AZ_NJ];
#pragma unroll
        for (int j = 0; j < AZ_NJ; j++) {
            const int c = go_col(gc, j);
            m0[j] = L.label[gc.idx[j]];
            int mm = m0[j];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int l = L.label[go_nb(gc, j, q, W)];
                if (((gc.info[j] >> q) & 1u) && c != 0 && go_ncol(gc, j, q) == c) mm = min(mm, l);
            }
            m[j] = mm;
        }
#pragma unroll
        for (int j = 0; j < AZ_NJ; j++) m[j] = min(m[j], (int)L.label[m[j]]);
#pragma unroll
        for (int j = 0; j < AZ_NJ; j++)
            if (go_col(gc, j) != 0 && m[j] < m0[j]) { L.label[gc.idx[j]] = (int16_t)m[j]; changed = 1; }
        WSYNC();
        if (!__any(changed)) break;
    }
    // liberties: every empty point adds one to each distinct neighbouring chain
#pragma unroll
    for (int j = 0; j < AZ_NJ; j++) {
        int lq[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int l = L.label[go_nb(gc, j, q, W)];
            const bool stone = ((gc.info[j] >> q) & 1u) && go_ncol(gc, j, q) != 0;
            lq[q] = (gc.info[j] & 16u) && go_col(gc, j) == 0 && stone ? l : -1;
        }
        if (lq[0] >= 0) atomicAdd(&L.libs[lq[0]], 1);
        if (lq[1] >= 0 && lq[1] != lq[0]) atomicAdd(&L.libs[lq[1]], 1);
        if (lq[2] >= 0 && lq[2] != lq[0] && lq[2] != lq[1]) atomicAdd(&L.libs[lq[2]], 1);
        if (lq[3] >= 0 && lq[3] != lq[0] && lq[3] != lq[1] && lq[3] != lq[2]) atomicAdd(&L.libs[lq[3]], 1);
    }
    WSYNC();
}

__device__ __forceinline__ void go_label(const Geo &G, WLds &L, const int8_t *b)
{
    GoCells gc;
    go_cells(G, b, gc);
    go_label(G, L, gc);
}

__device__ __forceinline__ int go_chain_libs(WLds &L, int cell) { return L.libs[L.label[cell]]; }

// ---------------------------------------------------------------------------
// legal mask -> L.mask[0..O), returns number of legal actions
// ---------------------------------------------------------------------------

__device__ int dev_legal_mask(const Geo &G, WLds &L, const PS &s, const SK *sk = nullptr)
{
    const int lane = lane_id();
    const int W = G.W, H = G.H, NC = G.NC;
    const int8_t *b = cur_board(L, s);
    int count = 0;
    if (G.game == G_GO) {
        GoCells gc;
        go_cells(G, b, gc);
        const int me = s.cur & 3, opp = (-s.cur) & 3;
        // an empty point next to an empty point is legal whatever the chains look like; only points with every
        // neighbour occupied need liberty counts, and a board without such points (most of an opening) is not labelled
        int open[AZ_NJ], need = 0;
#pragma unroll
        for (int j = 0; j < AZ_NJ; j++) {
            const bool empty = (gc.info[j] & 16u) && go_col(gc, j) == 0;
            bool o = false;
#pragma unroll
            for (int q = 0; q < 4; q++) o = o || (((gc.info[j] >> q) & 1u) && go_ncol(gc, j, q) == 0);
            open[j] = empty && o;
            need |= empty && !o;
        }
        // positional superko looks only at points the player to move has played before (history[0::2] for Black,
        // [1::2] for White, go.py:314-322): flag them, one pass over the move list
        uint8_t *const played = reinterpret_cast<uint8_t *>(L.cum);
        int sk_any = 0;
        if (G.superko && sk) {
            for (int i = lane; i < AZ_CP; i += AZ_WAVE) played[i] = 0;
            WSYNC();
            const int nh = min(s.nhist, sk->cap);
            for (int i = (s.cur == 1 ? 0 : 1) + 2 * lane; i < nh; i += 2 * AZ_WAVE) {
                const int m = sk->moves[i];
                if (m >= 0 && m < NC) played[m] = 1;
            }
            WSYNC();
#pragma unroll
            for (int j = 0; j < AZ_NJ; j++) sk_any |= (gc.info[j] & 16u) && go_col(gc, j) == 0 && played[gc.idx[j]];
            sk_any = __any(sk_any);
        }
        const bool labelled = __any(need) || sk_any;
        if (labelled) go_label(G, L, gc);
#pragma unroll
        for (int j = 0; j < AZ_NJ; j++) {
            const int i = lane + j * AZ_WAVE;
            int legal = open[j];
            if (labelled) {
                int lb[4];
#pragma unroll
                for (int q = 0; q < 4; q++) lb[q] = L.libs[L.label[go_nb(gc, j, q, W)]];
                if ((gc.info[j] & 16u) && go_col(gc, j) == 0) {
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        if (!((gc.info[j] >> q) & 1u)) continue;
                        const int c = go_ncol(gc, j, q);
                        if (c == me && lb[q] >= 2) legal = 1;
                        if (c == opp && lb[q] == 1) legal = 1;
                    }
                }
            }
            if (i == s.ko) legal = 0;
            if (sk_any) {
                // every otherwise legal point the player has played before is played out: the position it leads to
                // (stone placed, opposing chains whose last liberty it was removed) may not have occurred already
                unsigned long long cand = __ballot(legal && (i < NC) && played[gc.idx[j]]);
                while (cand) {
                    const int src = __ffsll((long long)cand) - 1;
                    cand &= cand - 1;
                    const int a = src + j * AZ_WAVE;
                    const int ax = a / W, ay = a - ax * W;
                    int cap[4] = {-1, -1, -1, -1};
                    if (ax > 0 && b[a - W] == -s.cur && L.libs[L.label[a - W]] == 1) cap[0] = L.label[a - W];
                    if (ax + 1 < H && b[a + W] == -s.cur && L.libs[L.label[a + W]] == 1) cap[1] = L.label[a + W];
                    if (ay > 0 && b[a - 1] == -s.cur && L.libs[L.label[a - 1]] == 1) cap[2] = L.label[a - 1];
                    if (ay + 1 < W && b[a + 1] == -s.cur && L.libs[L.label[a + 1]] == 1) cap[3] = L.label[a + 1];
                    unsigned long long x0 = 0, x1 = 0;
                    for (int c = lane; c < NC; c += AZ_WAVE) {
                        const int lb = L.label[c];
                        if (b[c] == -s.cur && (lb == cap[0] || lb == cap[1] || lb == cap[2] || lb == cap[3])) {
                            unsigned long long k0, k1;
                            zkey(-s.cur, c, k0, k1);
                            x0 ^= k0; x1 ^= k1;
                        }
                    }
                    unsigned long long k0, k1;
                    zkey(s.cur, a, k0, k1);
                    const unsigned long long r0 = s.h0 ^ k0 ^ wave_xor(x0), r1 = s.h1 ^ k1 ^ wave_xor(x1);
                    int hit = 0;
                    const int np = min(s.nprev, sk->cap);
                    for (int q = lane; q < np; q += AZ_WAVE) hit |= sk->hash[2 * q] == r0 && sk->hash[2 * q + 1] == r1;
                    if (__any(hit) && lane == src) legal = 0;
                }
            }
            if (i < NC) L.mask[i] = (uint8_t)legal;
            count += __popcll(__ballot(legal));
        }
        if (lane == 0) L.mask[NC] = 1;  // passing is always legal
        count += 1;
    } else if (G.game == G_REVERSI) {
        for (int i0 = 0; i0 < NC; i0 += AZ_WAVE) {
            int i = i0 + lane;
            int legal = 0;
            if (i < NC && b[i] == 0) {
                int x0 = i / W, y0 = i - x0 * W;
                for (int d = 0; d < 9 && !legal; d++) {
                    int dx = d / 3 - 1, dy = d % 3 - 1;
                    if (dx == 0 && dy == 0) continue;
                    int x = x0 + dx, y = y0 + dy, run = 0;
                    while (x >= 0 && y >= 0 && x < H && y < W && b[x * W + y] == -s.cur) { x += dx; y += dy; run++; }
                    if (run > 0 && x >= 0 && y >= 0 && x < H && y < W && b[x * W + y] == s.cur) legal = 1;
                }
            }
            if (i < NC) L.mask[i] = (uint8_t)legal;
            count += __popcll(__ballot(legal));
        }
        // PASS is legal only when no placement is
        if (lane == 0) L.mask[NC] = (uint8_t)(count == 0);
        if (count == 0) count = 1;
    } else {
        for (int i0 = 0; i0 < NC; i0 += AZ_WAVE) {
            int i = i0 + lane;
            int legal = (i < NC && b[i] == 0);
            if (i < NC) L.mask[i] = (uint8_t)legal;
            count += __popcll(__ballot(legal));
        }
    }
    WSYNC();
    return count;
}

// ---------------------------------------------------------------------------
// do_move for a move already known to be legal
// ---------------------------------------------------------------------------

__device__ void dev_do_move(const Geo &G, WLds &L, PS &s, int a, const SK *sk = nullptr)
{
    const int lane = lane_id();
    const int W = G.W, H = G.H, NC = G.NC;
    const int color = s.cur;
    if (G.game == G_GO) {
        if (G.superko && sk && lane == 0 && s.nhist < sk->cap) sk->moves[s.nhist] = (int16_t)a;
        s.ko = -1;
        push_history(L, s);
        int8_t *b = cur_board(L, s);
        if (a != NC) {
            int x = a / W, y = a - x * W;
            int nb[4], n = 0;
            if (x > 0) nb[n++] = a - W;
            if (x + 1 < H) nb[n++] = a + W;
            if (y > 0) nb[n++] = a - 1;
            if (y + 1 < W) nb[n++] = a + 1;
            int hostile = 0;
            for (int q = 0; q < n; q++) hostile |= b[nb[q]] == -color;
            if (lane == 0) b[a] = (int8_t)color;
            WSYNC();
            unsigned long long rx0 = 0, rx1 = 0;   // keys of the stones this move takes off (superko hash)
            // a stone that touches no opposing stone captures nothing and marks no ko (the move is known to be legal,
            // so it is no suicide either): the chains only have to be labelled for a contact move
            if (hostile) {
            go_label(G, L, b);
            // ko bookkeeping needs the pre-capture neighbourhood of `a`
            int friendly = 0, empty = 0, first = -1;
            for (int q = 0; q < n; q++) {
                int r = nb[q], c = b[r];
                if (c == color) friendly = 1;
                else if (c == 0) empty = 1;
                else if (first < 0 && go_chain_libs(L, r) == 0) first = r;
            }
            int ko = -1;
            if (first >= 0 && !friendly && !empty) {
                // captured chain is a single stone iff it has no same-colour neighbour
                int fx = first / W, fy = first - fx * W, single = 1;
                if (fx > 0 && b[first - W] == -color) single = 0;
                if (fx + 1 < H && b[first + W] == -color) single = 0;
                if (fy > 0 && b[first - 1] == -color) single = 0;
                if (fy + 1 < W && b[first + 1] == -color) single = 0;
                if (single) ko = first;
            }
            // remove every opposing chain left without liberties
            int removed = 0;
            int rem[AZ_NJ];
#pragma unroll
            for (int j = 0; j < AZ_NJ; j++) {
                int i = lane + j * AZ_WAVE;
                rem[j] = (i < NC && b[i] == -color && go_chain_libs(L, i) == 0);
                removed += __popcll(__ballot(rem[j]));
            }
            WSYNC();
#pragma unroll
            for (int j = 0; j < AZ_NJ; j++) {
                int i = lane + j * AZ_WAVE;
                if (rem[j]) {
                    b[i] = 0;
                    if (G.superko) { unsigned long long k0, k1; zkey(-color, i, k0, k1); rx0 ^= k0; rx1 ^= k1; }
                }
            }
            WSYNC();
            if (color == 1) s.pris_w += removed; else s.pris_b += removed;
            s.ko = ko;
            }
            if (G.superko) {
                // current_hash after the move and its captures; previous_hashes.add(current_hash) (go.py:552)
                unsigned long long k0, k1;
                zkey(color, a, k0, k1);
                s.h0 ^= k0 ^ wave_xor(rx0); s.h1 ^= k1 ^ wave_xor(rx1);
                if (sk && lane == 0 && s.nprev < sk->cap) { sk->hash[2 * s.nprev] = s.h0; sk->hash[2 * s.nprev + 1] = s.h1; }
                s.nprev++;
                WSYNC();
            }
        } else {
            if (color == 1) s.passes_b++; else s.passes_w++;
        }
        s.cur = -color;
        append_move(s, a);
        s.turns++;
        if (s.nhist > 1 && s.last1 == NC && s.last2 == NC && s.cur == -1) s.is_end = 1;
    } else if (G.game == G_REVERSI) {
        push_history(L, s);
        int8_t *b = cur_board(L, s);
        append_move(s, a);
        s.turns++;
        if (a != NC) {
            if (lane == 0) b[a] = (int8_t)color;
            WSYNC();
            s.stones++;
            int x0 = a / W, y0 = a - x0 * W;
            if (lane < 9 && lane != 4) {
                int dx = lane / 3 - 1, dy = lane % 3 - 1;
                int x = x0 + dx, y = y0 + dy, run = 0;
                while (x >= 0 && y >= 0 && x < H && y < W && b[x * W + y] == -color) { x += dx; y += dy; run++; }
                if (run > 0 && x >= 0 && y >= 0 && x < H && y < W && b[x * W + y] == color) {
                    for (int q = 0; q < run; q++) { x -= dx; y -= dy; b[x * W + y] = (int8_t)color; }
                }
            }
            WSYNC();
            if (s.stones == NC) s.is_end = 1;
        } else {
            if (s.nhist >= 2 && s.last2 == NC) s.is_end = 1;
        }
        s.cur = -color;
    } else {
        push_history(L, s);
        int8_t *b = cur_board(L, s);
        append_move(s, a);
        if (lane == 0) b[a] = (int8_t)color;
        WSYNC();
        s.turns++;
        int x0 = a / W, y0 = a - x0 * W;
        if (lane < 8) {
            const int dxs[8] = {1, -1, 0, 0, 1, -1, 1, -1};
            const int dys[8] = {0, 0, 1, -1, 1, -1, -1, 1};
            int dx = dxs[lane], dy = dys[lane];
            int x = x0 + dx, y = y0 + dy, run = 0;
            while (x >= 0 && y >= 0 && x < H && y < W && b[x * W + y] == color) { x += dx; y += dy; run++; }
            L.tmp[lane] = run;
        }
        WSYNC();
        int hc = 1 + L.tmp[0] + L.tmp[1], vc = 1 + L.tmp[2] + L.tmp[3];
        int d1 = 1 + L.tmp[4] + L.tmp[5], d2 = 1 + L.tmp[6] + L.tmp[7];
        WSYNC();
        if (hc >= G.k || vc >= G.k || d1 >= G.k || d2 >= G.k) { s.is_end = 1; s.winner = color; s.winner_set = 1; }
        else if (s.turns == NC) { s.is_end = 1; s.winner = 0; s.winner_set = 1; }
        s.cur = -color;
    }
}

// get_winner(): +1 / -1 / 0 ; mnk before the end: 2 (None)
__device__ int dev_winner(const Geo &G, WLds &L, const PS &s)
{
    const int lane = lane_id();
    const int W = G.W, H = G.H, NC = G.NC;
    const int8_t *b = cur_board(L, s);
    if (G.game == G_MNK) return s.winner_set ? s.winner : 2;
    int nb_ = 0, nw_ = 0;
    for (int i0 = 0; i0 < NC; i0 += AZ_WAVE) {
        int i = i0 + lane;
        int isb = 0, isw = 0;
        if (i < NC) {
            int c = b[i];
            if (c == 1) isb = 1;
            else if (c == -1) isw = 1;
            else if (G.game == G_GO) {
                int x = i / W, y = i - x * W, allb = 1, allw = 1;
                if (x > 0) { allb &= b[i - W] == 1; allw &= b[i - W] == -1; }
                if (x + 1 < H) { allb &= b[i + W] == 1; allw &= b[i + W] == -1; }
                if (y > 0) { allb &= b[i - 1] == 1; allw &= b[i - 1] == -1; }
                if (y + 1 < W) { allb &= b[i + 1] == 1; allw &= b[i + 1] == -1; }
                if (allb) isb = 1; else if (allw) isw = 1;
            }
        }
        nb_ += __popcll(__ballot(isb));
        nw_ += __popcll(__ballot(isw));
    }
    if (G.game == G_GO) {
        // komi 7.5 and one point off per own pass, compared in half points
        int sb2 = 2 * (nb_ - s.passes_b), sw2 = 2 * (nw_ - s.passes_w) + 15;
        return sb2 > sw2 ? 1 : (sw2 > sb2 ? -1 : 0);
    }
    return nb_ > nw_ ? 1 : (nb_ < nw_ ? -1 : 0);
}

// source cell of transformed cell i (also: label of the i-th policy entry after
// the reverse transform); square boards when tid != 0
__device__ __forceinline__ int transform_src(int i, int H, int W, int tid)
{
    int x = i / W, y = i - x * W, nx, ny;
    switch (tid & 3) {
    case 1: nx = y; ny = W - 1 - x; break;
    case 2: nx = H - 1 - x; ny = W - 1 - y; break;
    case 3: nx = H - 1 - y; ny = x; break;
    default: nx = x; ny = y; break;
    }
    if (tid >= 4) ny = W - 1 - ny;
    return nx * W + ny;
}

// 17 int8 planes of the state as seen after transform(tid), NCHW
__device__ void dev_features(const Geo &G, WLds &L, const PS &s, int tid, int8_t *out)
{
    const int lane = lane_id();
    const int NC = G.NC;
    for (int i0 = 0; i0 < NC; i0 += AZ_WAVE) {
        int i = i0 + lane;
        if (i >= NC) continue;
        int src = tid ? transform_src(i, G.H, G.W, tid) : i;
#pragma unroll
        for (int t = 0; t < 8; t++) {
            int c = hist_board(L, s, t)[src];
            out[(2 * t) * NC + i] = (int8_t)(c == s.cur);
            out[(2 * t + 1) * NC + i] = (int8_t)(c == -s.cur);
        }
        out[16 * NC + i] = (int8_t)(s.cur == 1);
    }
}
