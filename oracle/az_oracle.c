/*
 * TEST INFRASTRUCTURE - NOT PRODUCT CODE.  See az_oracle.h.
 *
 * Every function cites the reference lines it restates (paths relative to
 * the reference repository root).  Arithmetic that the reference evaluates
 * in numpy float32 (evaluator outputs are float32, see
 * AlphaZero/evaluator/nn_eval_parallel.py:156-158) is evaluated here in C
 * float with contraction disabled (build with -ffp-contract=off).
 */
#include "az_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <stddef.h>
#include <string.h>

/* ------------------------------------------------------------------ */
/* helpers                                                            */
/* ------------------------------------------------------------------ */

static inline int oz_ncell(const OzState *s) { return s->H * s->W; }
static inline int oz_pass(const OzState *s) { return s->H * s->W; }

/* AlphaZero/env/go.py:137-138  neighbour order (x-1,y),(x+1,y),(x,y-1),(x,y+1) */
static int go_nbrs(const OzState *s, int p, int out[4])
{
    int x = p / s->W, y = p % s->W, n = 0;
    if (x - 1 >= 0) out[n++] = (x - 1) * s->W + y;
    if (x + 1 < s->H) out[n++] = (x + 1) * s->W + y;
    if (y - 1 >= 0) out[n++] = x * s->W + (y - 1);
    if (y + 1 < s->W) out[n++] = x * s->W + (y + 1);
    return n;
}

/* Flood fill of the chain holding p.  Returns liberty count; fills group.
 * Equivalent to the reference's cached group_sets / liberty_sets
 * (AlphaZero/env/go.py:169-207,213-234). */
static int go_chain(const OzState *s, int p, int *group, int *ngroup)
{
    uint8_t seen[OZ_MAXC], libseen[OZ_MAXC];
    int nc = oz_ncell(s), head = 0, n = 0, libs = 0, nb[4];
    memset(seen, 0, nc);
    memset(libseen, 0, nc);
    int color = s->board[p];
    group[n++] = p;
    seen[p] = 1;
    while (head < n) {
        int q = group[head++];
        int k = go_nbrs(s, q, nb);
        for (int i = 0; i < k; i++) {
            int r = nb[i];
            if (s->board[r] == 0) {
                if (!libseen[r]) { libseen[r] = 1; libs++; }
            } else if (s->board[r] == color && !seen[r]) {
                seen[r] = 1;
                group[n++] = r;
            }
        }
    }
    *ngroup = n;
    return libs;
}

/* AlphaZero/env/go.py:274-301 */
static int go_is_suicide(const OzState *s, int a)
{
    int nb[4], k = go_nbrs(s, a, nb), group[OZ_MAXC], ng;
    for (int i = 0; i < k; i++)
        if (s->board[nb[i]] == 0) return 0;      /* liberty_sets[x][y] non-empty */
    for (int i = 0; i < k; i++) {
        int c = s->board[nb[i]];
        int libs = go_chain(s, nb[i], group, &ng);
        int other = (libs - 1) > 0;              /* liberties besides `a` */
        if (c == s->cur && other) return 0;
        if (c == -s->cur && !other) return 0;
    }
    return 1;
}

/* Zobrist key of a stone (go.py:64-71 draws its table from RandomState(0); any injective-in-practice keying gives
 * the same legality, which is what is pinned) */
static uint64_t zkey(int color, int cell)
{
    uint64_t z = 0x6A09E667F3BCC909ULL + (uint64_t)(2 * cell + (color == OZ_BLACK)) * 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

static int go_do_move(OzState *s, int a, int color);

/* AlphaZero/env/go.py:303-330.  Only points this player has played before are examined (history[0::2] for Black,
 * [1::2] for White: the slicing assumes alternating colours and no handicap stones, as the reference does). */
static int go_is_positional_superko(const OzState *s, int a)
{
    int seen = 0;
    for (int i = s->cur == OZ_BLACK ? 0 : 1; i < s->nhist && i < OZ_SK_MAX; i += 2)
        if (s->sk_moves[i] == a) { seen = 1; break; }
    if (!seen) return 0;
    static __thread OzState c;   /* too large for the stack of a deep playout recursion */
    oz_copy(&c, s);
    c.superko = 0;
    go_do_move(&c, a, 0);
    for (int i = 0; i < s->nprev && i < OZ_SK_MAX; i++)
        if (s->sk_hash[i] == c.cur_hash) return 1;
    return 0;
}

/* AlphaZero/env/go.py:332-355 (superko is off in self-play: game/gameplay.py:22) */
static int go_is_legal(const OzState *s, int a)
{
    if (a == oz_pass(s)) return 1;
    if (a < 0 || a > oz_pass(s)) return 0;
    if (s->board[a] != 0) return 0;
    if (go_is_suicide(s, a)) return 0;
    if (a == s->ko) return 0;
    if (s->superko && go_is_positional_superko(s, a)) return 0;
    return 1;
}

static void push_history(OzState *s)
{
    /* board_history.append(board.copy()); board_history.pop(0)
       (go.py:520-521, reversi.py:172-173, mnk.py:116-117) */
    int nc = oz_ncell(s);
    for (int t = 0; t + 1 < OZ_NHIST; t++) memcpy(s->hist[t], s->hist[t + 1], nc);
    memcpy(s->hist[OZ_NHIST - 1], s->board, nc);
}

static void append_move(OzState *s, int a)
{
    s->last2 = s->last1;
    s->last1 = a;
    s->nhist++;
}

/* AlphaZero/env/go.py:499-571 */
static int go_do_move(OzState *s, int a, int color)
{
    if (color == 0) color = s->cur;
    int reset = s->cur;
    s->cur = color;
    if (!go_is_legal(s, a)) { s->cur = reset; return -1; }
    s->ko = -1;
    push_history(s);
    if (a != oz_pass(s)) {
        int nb[4], k, group[OZ_MAXC], ng, own[OZ_MAXC], nown;
        s->board[a] = (int8_t)color;
        s->cur_hash ^= zkey(color, a);                       /* _update_hash, go.py:209-211 */
        k = go_nbrs(s, a, nb);
        for (int i = 0; i < k; i++) {
            int r = nb[i];
            if (s->board[r] == -color && go_chain(s, r, group, &ng) == 0) {
                for (int j = 0; j < ng; j++) { s->board[group[j]] = 0; s->cur_hash ^= zkey(-color, group[j]); }
                if (color == OZ_BLACK) s->pris_w += ng; else s->pris_b += ng;
                if (ng == 1) {
                    /* go.py:542-550: evaluated right after THIS capture */
                    int libs = go_chain(s, a, own, &nown);
                    if (libs == 1 && nown == 1) s->ko = r;
                }
            }
        }
        /* previous_hashes.add(current_hash), go.py:552 (kept whether or not superko is enforced: copy() hands it on) */
        if (s->nprev < OZ_SK_MAX) s->sk_hash[s->nprev] = s->cur_hash;
        s->nprev++;
    } else {
        if (color == OZ_BLACK) s->passes_b++;
        if (color == OZ_WHITE) s->passes_w++;
    }
    s->cur = -color;
    if (s->nhist < OZ_SK_MAX) s->sk_moves[s->nhist] = (int16_t)a;
    append_move(s, a);
    s->turns++;
    /* go.py:567-570 */
    if (s->nhist > 1 && s->last1 == oz_pass(s) && s->last2 == oz_pass(s) && s->cur == OZ_WHITE)
        s->is_end = 1;
    return 0;
}

/* AlphaZero/env/go.py:357-374,431-459 */
static int go_winner(const OzState *s)
{
    int nc = oz_ncell(s), nb[4];
    double sw = 0, sb = 0;
    for (int p = 0; p < nc; p++) {
        if (s->board[p] == OZ_WHITE) sw += 1;
        else if (s->board[p] == OZ_BLACK) sb += 1;
    }
    for (int p = 0; p < nc; p++) {
        if (s->board[p] != 0) continue;
        int k = go_nbrs(s, p, nb), allb = 1, allw = 1;
        for (int i = 0; i < k; i++) {
            if (s->board[nb[i]] != OZ_BLACK) allb = 0;
            if (s->board[nb[i]] != OZ_WHITE) allw = 0;
        }
        if (allb) sb += 1; else if (allw) sw += 1;
    }
    sw += s->komi;
    sw -= s->passes_w;
    sb -= s->passes_b;
    if (sb > sw) return OZ_BLACK;
    if (sw > sb) return OZ_WHITE;
    return 0;
}

/* ---- reversi ---- */

/* AlphaZero/env/reversi.py:50-83 */
static int rv_dir(OzState *s, int a, int dx, int dy, int flip)
{
    int x = a / s->W + dx, y = a % s->W + dy;
#define ONB(x, y) ((x) >= 0 && (y) >= 0 && (x) < s->H && (y) < s->W)
    if (!ONB(x, y) || s->board[x * s->W + y] != -s->cur) return 0;
    while (ONB(x, y) && s->board[x * s->W + y] == -s->cur) { x += dx; y += dy; }
    if (!ONB(x, y)) return 0;
    if (s->board[x * s->W + y] == s->cur) {
        if (flip) {
            while (x * s->W + y != a) {
                s->board[x * s->W + y] = (int8_t)s->cur;
                x -= dx; y -= dy;
            }
        }
        return 1;
    }
    return 0;
#undef ONB
}

static int rv_place_legal(const OzState *s, int a)
{
    if (a < 0 || a >= oz_ncell(s)) return 0;
    if (s->board[a] != 0) return 0;
    for (int dx = -1; dx <= 1; dx++)
        for (int dy = -1; dy <= 1; dy++)
            if ((dx || dy) && rv_dir((OzState *)s, a, dx, dy, 0)) return 1;
    return 0;
}

/* AlphaZero/env/reversi.py:101-125 */
static int rv_is_legal(const OzState *s, int a)
{
    if (a == oz_pass(s)) {
        for (int p = 0; p < oz_ncell(s); p++)
            if (rv_place_legal(s, p)) return 0;
        return 1;
    }
    return rv_place_legal(s, a);
}

/* AlphaZero/env/reversi.py:155-205 */
static int rv_do_move(OzState *s, int a, int color)
{
    if (color == 0) color = s->cur;
    int reset = s->cur;
    s->cur = color;
    if (!rv_is_legal(s, a)) { s->cur = reset; return -1; }
    push_history(s);
    append_move(s, a);
    s->turns++;
    if (a != oz_pass(s)) {
        s->board[a] = (int8_t)color;
        s->stones_played++;
        /* the stone is placed before the scans, so a run ends on it */
        for (int di = -1; di <= 1; di++)
            for (int dj = -1; dj <= 1; dj++)
                if (di || dj) rv_dir(s, a, di, dj, 1);
        if (s->stones_played == s->H * s->W) s->is_end = 1;
    } else {
        if (s->nhist >= 2 && s->last2 == oz_pass(s)) s->is_end = 1;
    }
    s->cur = -color;
    return 0;
}

/* AlphaZero/env/reversi.py:138-153 */
static int rv_winner(const OzState *s)
{
    int b = 0, w = 0;
    for (int p = 0; p < oz_ncell(s); p++) {
        if (s->board[p] == OZ_BLACK) b++;
        else if (s->board[p] == OZ_WHITE) w++;
    }
    return b > w ? OZ_BLACK : (b < w ? OZ_WHITE : 0);
}

/* ---- mnk ---- */

/* AlphaZero/env/mnk.py:66-80 (no PASS: the action tuple is unpacked) */
static int mnk_is_legal(const OzState *s, int a)
{
    if (a < 0 || a >= oz_ncell(s)) return 0;
    return s->board[a] == 0;
}

static int mnk_run(const OzState *s, int x, int y, int dx, int dy, int c)
{
    int n = 0;
    x += dx; y += dy;
    while (x >= 0 && y >= 0 && x < s->H && y < s->W && s->board[x * s->W + y] == c) {
        n++; x += dx; y += dy;
    }
    return n;
}

/* AlphaZero/env/mnk.py:99-185 */
static int mnk_do_move(OzState *s, int a, int color)
{
    if (color == 0) color = s->cur;
    int reset = s->cur;
    s->cur = color;
    if (!mnk_is_legal(s, a)) { s->cur = reset; return -1; }
    push_history(s);
    append_move(s, a);
    int x = a / s->W, y = a % s->W;
    s->board[a] = (int8_t)color;
    s->turns++;
    int h = 1 + mnk_run(s, x, y, 1, 0, color) + mnk_run(s, x, y, -1, 0, color);
    int v = 1 + mnk_run(s, x, y, 0, 1, color) + mnk_run(s, x, y, 0, -1, color);
    int d1 = 1 + mnk_run(s, x, y, 1, 1, color) + mnk_run(s, x, y, -1, -1, color);
    int d2 = 1 + mnk_run(s, x, y, 1, -1, color) + mnk_run(s, x, y, -1, 1, color);
    if (h >= s->k || v >= s->k || d1 >= s->k || d2 >= s->k) {
        s->is_end = 1; s->winner = color; s->winner_set = 1;
    } else if (s->turns == s->H * s->W) {
        s->is_end = 1; s->winner = 0; s->winner_set = 1;
    }
    s->cur = -color;
    return 0;
}

/* ------------------------------------------------------------------ */
/* public rules API                                                   */
/* ------------------------------------------------------------------ */

void oz_init(OzState *s, int game, int H, int W, int k)
{
    memset(s, 0, sizeof(*s));
    s->game = game; s->H = H; s->W = W; s->k = k;
    s->O = H * W + (game == OZ_MNK ? 0 : 1);
    s->cur = OZ_BLACK;
    s->ko = -1;
    s->last1 = s->last2 = OZ_NOMOVE;
    s->komi = 7.5;
    if (game == OZ_REVERSI) {
        /* AlphaZero/env/reversi.py:20-23 */
        int h = H / 2;
        s->board[(h - 1) * W + (h - 1)] = OZ_WHITE;
        s->board[h * W + h] = OZ_WHITE;
        s->board[h * W + (h - 1)] = OZ_BLACK;
        s->board[(h - 1) * W + h] = OZ_BLACK;
        s->stones_played = 4;
    }
}

void oz_set_superko(OzState *s, int on) { s->superko = on != 0; }

void oz_copy(OzState *dst, const OzState *src)
{
    /* the superko history is copied as far as it is filled, and only when a state can use it */
    memcpy(dst, src, offsetof(OzState, sk_moves));
    if (src->game == OZ_GO && src->superko) {
        int nm = src->nhist < OZ_SK_MAX ? src->nhist : OZ_SK_MAX, nh = src->nprev < OZ_SK_MAX ? src->nprev : OZ_SK_MAX;
        memcpy(dst->sk_moves, src->sk_moves, sizeof(int16_t) * (size_t)nm);
        memcpy(dst->sk_hash, src->sk_hash, sizeof(uint64_t) * (size_t)nh);
    }
    if (src->game == OZ_GO) {
        /* go.py:242-272: a fresh GameState() is built, so the pass counters
           and the end flag are NOT carried over */
        dst->is_end = 0;
        dst->passes_b = dst->passes_w = 0;
    }
}

int oz_is_legal(const OzState *s, int a)
{
    switch (s->game) {
    case OZ_GO: return go_is_legal(s, a);
    case OZ_REVERSI: return rv_is_legal(s, a);
    default: return mnk_is_legal(s, a);
    }
}

void oz_legal_mask(const OzState *s, uint8_t *mask)
{
    for (int a = 0; a < s->O; a++) mask[a] = (uint8_t)oz_is_legal(s, a);
}

int oz_do_move_color(OzState *s, int a, int color)
{
    switch (s->game) {
    case OZ_GO: return go_do_move(s, a, color);
    case OZ_REVERSI: return rv_do_move(s, a, color);
    default: return mnk_do_move(s, a, color);
    }
}

int oz_do_move(OzState *s, int a) { return oz_do_move_color(s, a, 0); }

int oz_winner(const OzState *s)
{
    switch (s->game) {
    case OZ_GO: return go_winner(s);
    case OZ_REVERSI: return rv_winner(s);
    default: return s->winner_set ? s->winner : 2;
    }
}

/* ------------------------------------------------------------------ */
/* features / transforms                                              */
/* ------------------------------------------------------------------ */

/* np.fliplr then np.rot90(b, id % 4): go.py:585-591 (same in reversi/mnk).
 * Square boards only when id != 0 (mnk.py:199). */
void oz_transform_board(const int8_t *src, int8_t *dst, int H, int W, int tid)
{
    int8_t tmp[OZ_MAXC], tmp2[OZ_MAXC];
    int n = H;
    memcpy(tmp, src, (size_t)H * W);
    if (tid / 4 == 1) {
        for (int i = 0; i < H; i++)
            for (int j = 0; j < W; j++) tmp2[i * W + j] = tmp[i * W + (W - 1 - j)];
        memcpy(tmp, tmp2, (size_t)H * W);
    }
    for (int r = 0; r < tid % 4; r++) {
        /* one counter-clockwise quarter turn: out[i][j] = in[j][n-1-i] */
        for (int i = 0; i < n; i++)
            for (int j = 0; j < n; j++) tmp2[i * n + j] = tmp[j * n + (n - 1 - i)];
        memcpy(tmp, tmp2, (size_t)n * n);
    }
    memcpy(dst, tmp, (size_t)H * W);
}

/* processing/state_converter.py:48-82 on state.copy().transform(tid) */
void oz_features(const OzState *s, int tid, int8_t *planes)
{
    int nc = oz_ncell(s);
    int8_t b[OZ_MAXC];
    for (int t = 0; t < OZ_NHIST + 1; t++) {
        const int8_t *src = t < OZ_NHIST ? s->hist[t] : s->board;
        oz_transform_board(src, b, s->H, s->W, tid);
        for (int p = 0; p < nc; p++) {
            planes[(2 * t) * nc + p] = (int8_t)(b[p] == s->cur);
            planes[(2 * t + 1) * nc + p] = (int8_t)(b[p] == -s->cur);
        }
    }
    for (int p = 0; p < nc; p++) planes[16 * nc + p] = (int8_t)(s->cur == OZ_BLACK);
}

/* processing/state_converter.py:123-180: entry idx of the action list starts
 * as (idx // W, idx % W); rotation labels are rewritten first, then the flip */
int oz_reverse_label(int idx, int H, int W, int tid)
{
    if (idx >= H * W) return idx; /* PASS entry untouched */
    int x = idx / W, y = idx % W, nx, ny;
    switch (tid % 4) {
    case 1: nx = y; ny = W - 1 - x; break;
    case 2: nx = H - 1 - x; ny = W - 1 - y; break;
    case 3: nx = H - 1 - y; ny = x; break;
    default: nx = x; ny = y; break;
    }
    x = nx; y = ny;
    if (tid / 4 == 1) y = W - 1 - y;
    return x * W + y;
}

/* ------------------------------------------------------------------ */
/* RNG streams                                                        */
/* ------------------------------------------------------------------ */

uint64_t oz_splitmix64(uint64_t *x)
{
    uint64_t z = (*x += 0x9E3779B97F4A7C15ULL);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

static double u01(uint64_t *st) { return (double)(oz_splitmix64(st) >> 11) * (1.0 / 9007199254740992.0); }

/* CPython Lib/random.py gammavariate, alpha <= 1 branch (Ahrens-Dieter GS) */
double oz_gamma_small_alpha(double alpha, uint64_t *st)
{
    double x;
    for (;;) {
        double u = u01(st);
        double b = (M_E + alpha) / M_E;
        double p = b * u;
        x = p <= 1.0 ? pow(p, 1.0 / alpha) : -log((b - p) / alpha);
        double u1 = u01(st);
        if (p > 1.0) { if (u1 <= pow(x, alpha - 1.0)) break; }
        else if (u1 <= exp(-x)) break;
    }
    return x;
}

static int rng_tid(OzRng *r, int types)
{
    if (r->tids) {
        if (r->i_tids >= r->n_tids) { r->underflow = 1; return 0; }
        return r->tids[r->i_tids++];
    }
    return (int)(oz_splitmix64(&r->seed) % (uint64_t)types);
}
static double rng_eta(OzRng *r, double alpha)
{
    if (r->etas) {
        if (r->i_etas >= r->n_etas) { r->underflow = 1; return 0; }
        return r->etas[r->i_etas++];
    }
    return oz_gamma_small_alpha(alpha, &r->seed);
}
static double rng_uni(OzRng *r)
{
    if (r->unis) {
        if (r->i_unis >= r->n_unis) { r->underflow = 1; return 0; }
        return r->unis[r->i_unis++];
    }
    return u01(&r->seed);
}

/* ------------------------------------------------------------------ */
/* search (AlphaZero/search/mcts.py)                                  */
/* ------------------------------------------------------------------ */

typedef struct {
    int parent;
    int n;          /* _visit_cnt */
    oz_real w;        /* _total_action_val */
    oz_real p;        /* _prior_prob */
    int first, cnt; /* children, contiguous, insertion order */
    int action;
} OzNode;

struct OzTree {
    OzNode *nodes;
    int64_t cap, used;
    int root;
    int O, transform_types;
    double d_alpha, d_epsilon;
};

/* mcts.py:9: a module global, so a caller may assign it; oz_set_c_punt is that assignment (tests only change it to
 * replay fixtures the reference recorded with another value) */
static double g_c_punt = 5.0;
void oz_set_c_punt(double c) { g_c_punt = c; }

static int tree_alloc(OzTree *t, int n)
{
    if (t->used + n > t->cap) {
        while (t->used + n > t->cap) t->cap *= 2;
        t->nodes = (OzNode *)realloc(t->nodes, (size_t)t->cap * sizeof(OzNode));
    }
    int at = (int)t->used;
    t->used += n;
    return at;
}

static int tree_new_root(OzTree *t)
{
    /* MCTreeNode(None, 1.0): mcts.py:159,346 */
    t->used = 0;
    int r = tree_alloc(t, 1);
    OzNode *nd = &t->nodes[r];
    nd->parent = -1; nd->n = 0; nd->w = 0.0f; nd->p = 1.0f; nd->first = -1; nd->cnt = 0; nd->action = OZ_NOMOVE;
    return r;
}

OzTree *oz_tree_new(int O, int transform_types, double d_alpha, double d_epsilon)
{
    OzTree *t = (OzTree *)calloc(1, sizeof(OzTree));
    t->cap = 4096;
    t->nodes = (OzNode *)malloc((size_t)t->cap * sizeof(OzNode));
    t->O = O; t->transform_types = transform_types;
    t->d_alpha = d_alpha; t->d_epsilon = d_epsilon;
    t->root = tree_new_root(t);
    return t;
}

void oz_tree_free(OzTree *t) { if (t) { free(t->nodes); free(t); } }
int64_t oz_tree_num_nodes(const OzTree *t) { return t->used; }

/* mcts.py:80-103, evaluated left to right in float32 */
static oz_real sel_value(const OzTree *t, const OzNode *c)
{
    const OzNode *par = &t->nodes[c->parent];
    oz_real u = (oz_real)g_c_punt * c->p;
    u = u * (oz_real)sqrt((double)par->n);
    u = u / (oz_real)(1.0 + (double)c->n);
    oz_real q = c->n == 0 ? (oz_real)0.0 : c->w / (oz_real)c->n;
    return q + u;
}

/* mcts.py:54-61: Python max() keeps the first maximal item */
static int tree_select(const OzTree *t, int node)
{
    const OzNode *nd = &t->nodes[node];
    int best = nd->first;
    oz_real bv = sel_value(t, &t->nodes[best]);
    for (int i = 1; i < nd->cnt; i++) {
        oz_real v = sel_value(t, &t->nodes[nd->first + i]);
        if (v > bv) { bv = v; best = nd->first + i; }
    }
    return best;
}

/* Evaluate `s` as MCTSearch does at a leaf (mcts.py:206-217 / 268-280):
 * draw a transform id, evaluate the transformed copy, relabel, keep legal
 * entries in list order.  Returns the candidate count. */
static int eval_candidates(OzTree *t, const OzState *s, oz_eval_fn ev, void *ctx, OzRng *rng,
                           int *acts, oz_real *probs, oz_real *value)
{
    int8_t planes[OZ_NPLANES * OZ_MAXC];
    oz_real pri[OZ_MAXO];
    int tid = 0, n = 0;
    int enable = !(t->transform_types == 0 || t->transform_types == 1);
    if (enable) tid = rng_tid(rng, t->transform_types);
    oz_features(s, tid, planes);
    ev(ctx, planes, s->H, s->W, s->O, pri, value);
    for (int i = 0; i < s->O; i++) {
        int a = enable ? oz_reverse_label(i, s->H, s->W, tid) : i;
        if (oz_is_legal(s, a)) { acts[n] = a; probs[n] = pri[i]; n++; }
    }
    return n;
}

/* mcts.py:30-52 */
static void tree_expand(OzTree *t, int node, const int *acts, const oz_real *probs, int n, oz_real value)
{
    if (t->nodes[node].cnt != 0) return;
    t->nodes[node].w += value;
    int first = tree_alloc(t, n);
    OzNode *nd = &t->nodes[node];
    nd->first = first; nd->cnt = n;
    for (int i = 0; i < n; i++) {
        OzNode *c = &t->nodes[first + i];
        c->parent = node; c->n = 0; c->w = 0.0f; c->p = probs[i];
        c->first = -1; c->cnt = 0; c->action = acts[i];
    }
}

/* mcts.py:173-232 */
static oz_real tree_playout(OzTree *t, OzState *s, int node, oz_eval_fn ev, void *ctx, OzRng *rng)
{
    t->nodes[node].n++;
    if (t->nodes[node].cnt != 0) {
        int c = tree_select(t, node);
        int player = s->cur;
        oz_do_move(s, t->nodes[c].action);
        oz_real v = tree_playout(t, s, c, ev, ctx, rng);
        t->nodes[node].w += (oz_real)(-player) * v;
        return v;
    }
    int acts[OZ_MAXO];
    oz_real probs[OZ_MAXO], value;
    int n = eval_candidates(t, s, ev, ctx, rng, acts, probs, &value);
    if (n != 0 && !s->is_end) {
        tree_expand(t, node, acts, probs, n, (oz_real)(-s->cur) * value);
    } else {
        value = (oz_real)oz_winner(s);
        t->nodes[node].w += (oz_real)(-s->cur) * value;
    }
    return value;
}

void oz_tree_playout(OzTree *t, const OzState *s, oz_eval_fn ev, void *ctx, OzRng *rng)
{
    OzState c;
    oz_copy(&c, s);
    tree_playout(t, &c, t->root, ev, ctx, rng);
}

/* mcts.py:245-296 */
int oz_tree_search(OzTree *t, const OzState *s, int max_playout, int dirichlet,
                   oz_eval_fn ev, void *ctx, OzRng *rng)
{
    int evals = 0;
    t->nodes[t->root].n++;
    if (t->nodes[t->root].cnt == 0) {
        int acts[OZ_MAXO];
        oz_real probs[OZ_MAXO], value;
        int n = eval_candidates(t, s, ev, ctx, rng, acts, probs, &value);
        tree_expand(t, t->root, acts, probs, n, value); /* raw value: mcts.py:283 */
        evals++;
    }
    if (dirichlet) {
        /* mcts.py:285-291; math_helper.py:13-17 (un-normalised gamma draws) */
        OzNode *r = &t->nodes[t->root];
        oz_real keep = (oz_real)(1.0 - t->d_epsilon);
        for (int i = 0; i < r->cnt; i++) {
            double eta = rng_eta(rng, t->d_alpha);
            OzNode *c = &t->nodes[r->first + i];
            oz_real a = keep * c->p;
            c->p = a + (oz_real)(t->d_epsilon * eta);
        }
    }
    for (int i = 0; i < max_playout; i++) {
        OzState c;
        oz_copy(&c, s);
        tree_playout(t, &c, t->root, ev, ctx, rng);
        evals++;
    }
    return evals;
}

/* ------------------------------------------------------------------ */
/* Several leaves per round under a virtual loss.  NOT in the reference (mcts.py:295-296 keeps one playout in
 * flight): this is the definition of the engine's opt-in `leaves_per_round = K > 1` schedule, restated in scalar
 * code so that the engine's trees can be held to something other than themselves.
 *
 * A search = root.visit(); root expansion if the root is a leaf (raw value, then noise) or noise on the existing
 * children; then rounds until max_playout playouts have been issued.  One round:
 *   select, j = 0 .. min(K, playouts left) - 1, one after the other: a playout descends from the root on the
 *     statistics as they stand (including the virtual losses of the round's earlier playouts); every node it enters
 *     is visited (N += 1) at once and every edge it takes loses `vloss` from W; its transform id is drawn before the
 *     descent; the leaf's state, legality and end-of-game status are taken now;
 *   evaluate the round's leaves;
 *   finish, j in the same order: a leaf that an earlier playout of the round has expanded meanwhile is not
 *     expanded again (its value is only backed up); otherwise expand / terminal value as in mcts.py:217-231; the
 *     leaf and every edge of the path get their W update as `W += (-player * v) + vloss` (the root, which never took
 *     a virtual loss, `W += -player * v`).
 * With K = 1 and vloss = 0 this is mcts.py's schedule. */
/* ------------------------------------------------------------------ */

typedef struct {
    int node, depth, tid, terminal, winner, cur, n_legal;
    int path_node[OZ_KMAXD];
    int path_player[OZ_KMAXD];
    uint8_t legal[OZ_MAXO];
    int8_t planes[OZ_NPLANES * OZ_MAXC];
    oz_real pri[OZ_MAXO], value;
} OzLeaf;

int oz_tree_search_k(OzTree *t, const OzState *s, int max_playout, int K, double vloss, int dirichlet,
                     oz_eval_fn ev, void *ctx, OzRng *rng)
{
    if (K < 1) K = 1;
    const oz_real vl = K > 1 ? (oz_real)vloss : (oz_real)0.0;
    const int enable = !(t->transform_types == 0 || t->transform_types == 1);
    int evals = 0;
    t->nodes[t->root].n++;
    if (t->nodes[t->root].cnt == 0) {
        int acts[OZ_MAXO];
        oz_real probs[OZ_MAXO], value;
        int n = eval_candidates(t, s, ev, ctx, rng, acts, probs, &value);
        tree_expand(t, t->root, acts, probs, n, value);
        evals++;
    }
    if (dirichlet) {
        OzNode *r = &t->nodes[t->root];
        oz_real keep = (oz_real)(1.0 - t->d_epsilon);
        for (int i = 0; i < r->cnt; i++) {
            double eta = rng_eta(rng, t->d_alpha);
            OzNode *c = &t->nodes[r->first + i];
            oz_real a = keep * c->p;
            c->p = a + (oz_real)(t->d_epsilon * eta);
        }
    }
    OzLeaf *lv = (OzLeaf *)malloc(sizeof(OzLeaf) * (size_t)K);
    int issued = 0;
    while (issued < max_playout) {
        const int m = max_playout - issued < K ? max_playout - issued : K;
        for (int j = 0; j < m; j++) {               /* select */
            OzLeaf *L = &lv[j];
            OzState c;
            oz_copy(&c, s);
            L->tid = enable ? rng_tid(rng, t->transform_types) : 0;
            int node = t->root;
            t->nodes[node].n++;
            L->depth = 0;
            while (t->nodes[node].cnt != 0) {
                int ch = tree_select(t, node);
                if (L->depth >= OZ_KMAXD) { free(lv); return -2; }
                L->path_node[L->depth] = node;
                L->path_player[L->depth] = c.cur;
                L->depth++;
                oz_do_move(&c, t->nodes[ch].action);
                t->nodes[ch].n++;
                if (K > 1) t->nodes[ch].w -= vl;
                node = ch;
            }
            issued++;
            L->node = node;
            L->cur = c.cur;
            L->n_legal = 0;
            for (int a = 0; a < c.O; a++) { L->legal[a] = (uint8_t)oz_is_legal(&c, a); L->n_legal += L->legal[a]; }
            L->terminal = (L->n_legal == 0) || c.is_end;
            L->winner = L->terminal ? oz_winner(&c) : 0;
            oz_features(&c, L->tid, L->planes);
        }
        for (int j = 0; j < m; j++) {               /* evaluate */
            ev(ctx, lv[j].planes, s->H, s->W, s->O, lv[j].pri, &lv[j].value);
            evals++;
        }
        for (int j = 0; j < m; j++) {               /* finish */
            OzLeaf *L = &lv[j];
            oz_real v = L->value;
            const int is_root = L->node == t->root;
            if (!L->terminal && t->nodes[L->node].cnt != 0) {
                oz_real add = (oz_real)(-L->cur) * v;
                if (is_root) t->nodes[L->node].w += add; else t->nodes[L->node].w += add + vl;
            } else if (!L->terminal) {
                int acts[OZ_MAXO], n = 0;
                oz_real probs[OZ_MAXO];
                for (int i = 0; i < s->O; i++) {
                    int a = enable ? oz_reverse_label(i, s->H, s->W, L->tid) : i;
                    if (L->legal[a]) { acts[n] = a; probs[n] = L->pri[i]; n++; }
                }
                oz_real add = (oz_real)(-L->cur) * v;
                tree_expand(t, L->node, acts, probs, n, is_root ? add : add + vl);
            } else {
                v = (oz_real)L->winner;
                oz_real add = (oz_real)(-L->cur) * v;
                if (is_root) t->nodes[L->node].w += add; else t->nodes[L->node].w += add + vl;
            }
            for (int d = 0; d < L->depth; d++) {
                oz_real add = (oz_real)(-L->path_player[d]) * v;
                if (L->path_node[d] == t->root) t->nodes[L->path_node[d]].w += add;
                else t->nodes[L->path_node[d]].w += add + vl;
            }
        }
    }
    free(lv);
    return evals;
}

int oz_tree_root_stats(const OzTree *t, int32_t *actions, int32_t *visits, oz_real *W, oz_real *P)
{
    const OzNode *r = &t->nodes[t->root];
    for (int i = 0; i < r->cnt; i++) {
        const OzNode *c = &t->nodes[r->first + i];
        if (actions) actions[i] = c->action;
        if (visits) visits[i] = c->n;
        if (W) W[i] = c->w;
        if (P) P[i] = c->p;
    }
    return r->cnt;
}

int oz_tree_root_visits(const OzTree *t) { return t->nodes[t->root].n; }

/* mcts.py:234-243 + math_helper.py:20-23 + CPython random.choices:
 * cum = accumulate(N_i/total); bisect_right(cum, u*(cum[-1]+0.0), 0, n-1) */
int oz_tree_choose(const OzTree *t, double u)
{
    const OzNode *r = &t->nodes[t->root];
    double cum[OZ_MAXO], acc = 0.0;
    long total = 0;
    for (int i = 0; i < r->cnt; i++) total += t->nodes[r->first + i].n;
    for (int i = 0; i < r->cnt; i++) {
        double w = (double)t->nodes[r->first + i].n / (double)total;
        acc = i == 0 ? w : acc + w;
        cum[i] = acc;
    }
    double x = u * (cum[r->cnt - 1] + 0.0);
    int lo = 0, hi = r->cnt - 1;
    while (lo < hi) {
        int mid = (lo + hi) / 2;
        if (x < cum[mid]) hi = mid; else lo = mid + 1;
    }
    return t->nodes[r->first + lo].action;
}

/* mcts.py:316-318 */
int oz_tree_argmax(const OzTree *t)
{
    const OzNode *r = &t->nodes[t->root];
    int best = 0;
    for (int i = 1; i < r->cnt; i++)
        if (t->nodes[r->first + i].n > t->nodes[r->first + best].n) best = i;
    return t->nodes[r->first + best].action;
}

static void copy_subtree(const OzTree *t, int src, OzNode *dst, int64_t *used, int dnode)
{
    const OzNode *s = &t->nodes[src];
    if (s->cnt == 0) { dst[dnode].first = -1; dst[dnode].cnt = 0; return; }
    int first = (int)*used;
    *used += s->cnt;
    dst[dnode].first = first; dst[dnode].cnt = s->cnt;
    for (int i = 0; i < s->cnt; i++) {
        dst[first + i] = t->nodes[s->first + i];
        dst[first + i].parent = dnode;
    }
    for (int i = 0; i < s->cnt; i++) copy_subtree(t, s->first + i, dst, used, first + i);
}

/* mcts.py:336-346 */
void oz_tree_update_with_move(OzTree *t, int action)
{
    const OzNode *r = &t->nodes[t->root];
    int child = -1;
    for (int i = 0; i < r->cnt; i++)
        if (t->nodes[r->first + i].action == action) { child = r->first + i; break; }
    if (child < 0) { t->root = tree_new_root(t); return; }
    OzNode *dst = (OzNode *)malloc((size_t)t->cap * sizeof(OzNode));
    int64_t used = 1;
    dst[0] = t->nodes[child];
    dst[0].parent = -1;
    copy_subtree(t, child, dst, &used, 0);
    free(t->nodes);
    t->nodes = dst;
    t->used = used;
    t->root = 0;
}

int oz_tree_count_expanded(const OzTree *t)
{
    int n = 0;
    for (int64_t i = 0; i < t->used; i++) n += t->nodes[i].cnt != 0;
    return n;
}

/* ------------------------------------------------------------------ */
/* gameplay (AlphaZero/game/gameplay.py:38-94)                        */
/* ------------------------------------------------------------------ */

/* Game(nn_eval_1, nn_eval_2, ...): player_1 (Black, moves first) searches with ev1, player_2 with ev2
 * (gameplay.py:19-20; train/parallel/evaluator.py:81-85 passes two different evaluators).  K > 1 searches with
 * oz_tree_search_k instead of the reference's schedule. */
int oz_play_game_ex(const OzGameCfg *cfg, oz_eval_fn ev1, void *ctx1, oz_eval_fn ev2, void *ctx2, int K, double vloss,
                    OzRng *rng, OzTrajectory *traj, int max_plies)
{
    OzState st;
    oz_init(&st, cfg->game, cfg->H, cfg->W, cfg->k);
    OzTree *tr[2];
    tr[0] = oz_tree_new(st.O, cfg->transform_types, cfg->d_alpha, cfg->d_epsilon);
    tr[1] = oz_tree_new(st.O, cfg->transform_types, cfg->d_alpha, cfg->d_epsilon);
    oz_eval_fn evs[2] = {ev1, ev2};
    void *ctxs[2] = {ctx1, ctx2};
    int cur = 0, ply = 0;
    int32_t acts[OZ_MAXO], vis[OZ_MAXO];
    traj->n_evals = 0;
    while (!(st.is_end || st.turns > cfg->max_turn)) {
        if (ply >= max_plies) break;
        const int noise = st.turns <= cfg->dirichlet_before;
        if (K > 1) traj->n_evals += oz_tree_search_k(tr[cur], &st, cfg->max_playout, K, vloss, noise, evs[cur], ctxs[cur], rng);
        else traj->n_evals += oz_tree_search(tr[cur], &st, cfg->max_playout, noise, evs[cur], ctxs[cur], rng);
        int n = oz_tree_root_stats(tr[cur], acts, vis, NULL, NULL);
        int move = oz_tree_choose(tr[cur], rng_uni(rng));
        if (traj->visits) {
            int32_t *row = traj->visits + (size_t)ply * st.O;
            memset(row, 0, sizeof(int32_t) * st.O);
            for (int i = 0; i < n; i++) row[acts[i]] = vis[i];
        }
        if (traj->moves) traj->moves[ply] = (int16_t)move;
        oz_do_move(&st, move);
        oz_tree_update_with_move(tr[0], move);
        oz_tree_update_with_move(tr[1], move);
        cur ^= 1;
        ply++;
    }
    traj->n_plies = ply;
    traj->winner = oz_winner(&st);
    oz_tree_free(tr[0]);
    oz_tree_free(tr[1]);
    return rng->underflow ? -1 : 0;
}

int oz_play_game(const OzGameCfg *cfg, oz_eval_fn ev, void *ctx, OzRng *rng,
                 OzTrajectory *traj, int max_plies)
{
    return oz_play_game_ex(cfg, ev, ctx, ev, ctx, 1, 0.0, rng, traj, max_plies);
}

/* ------------------------------------------------------------------ */
/* evaluators used by tests / baseline                                */
/* ------------------------------------------------------------------ */

static inline uint64_t mix64(uint64_t z)
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

/* Same arithmetic as oracle/fake_eval.py (numpy): a position hash drives
 * float32 priors in (0, 4/O] and a float32 value in [-0.5, 0.5). */
void oz_fake_eval(void *ctx, const int8_t *planes, int H, int W, int O, oz_real *priors, oz_real *value)
{
    /* ctx: NULL, or a uint64 salt that makes a second, different "network" (0 = the unsalted one) */
    const uint64_t salt = ctx ? *(const uint64_t *)ctx : 0;
    uint64_t h = 0;
    int n = OZ_NPLANES * H * W;
    for (int j = 0; j < n; j++)
        h += (uint64_t)(uint8_t)planes[j] * mix64((uint64_t)(j + 1) * 0x9E3779B97F4A7C15ULL);
    if (salt) h ^= mix64(salt);
    float scale = (float)(4.0 / (double)O);
    for (int i = 0; i < O; i++) {
        uint64_t z = mix64(h + (uint64_t)(i + 1) * 0x9E3779B97F4A7C15ULL);
        float a = (float)((z >> 40) + 1) * (1.0f / 16777216.0f);
        float p = a * a;
        p = p * p;
        priors[i] = p * scale;
    }
    uint64_t z = mix64(h ^ 0xABCDEF0123456789ULL);
    *value = (float)((double)(z >> 40) - 8388608.0) * (1.0f / 16777216.0f);
}

void oz_uniform_eval(void *ctx, const int8_t *planes, int H, int W, int O, oz_real *priors, oz_real *value)
{
    (void)ctx; (void)planes; (void)H; (void)W;
    oz_real p = (oz_real)(1.0 / (double)O);  /* np.full((dims,), 1 / dims) in the evaluator's dtype */
    for (int i = 0; i < O; i++) priors[i] = p;
    *value = (oz_real)0.0;
}
