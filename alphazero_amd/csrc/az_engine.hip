// Host side of the C ABI declared in include/az_engine.h.
#include "../../include/az_engine.h"
#include "az_kernels.h"
#include "az_tower.h"

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

// why the last az_engine_create of the calling thread failed (there is no handle to hang the message on)
static thread_local std::string g_create_error;
#define AZ_EVQ 8

// Staging of az_trajectory_pop_many_static: a fixed array in every GPU's memory instead of a buffer grown with
// hipMalloc.  The name is only a host-side handle on the symbol; the address a kernel or a copy needs is asked for once
// per device (az_engine_create, with that device current) and kept in g_scratch_ptr.
static __device__ __align__(256) unsigned char scratch[AZ_STATIC_POP_BYTES];
#define AZ_MAX_DEVICES 64
static unsigned char *g_scratch_ptr[AZ_MAX_DEVICES];
static std::mutex g_scratch_mutex;
// layout: slots int32[256] | row starts int32[256] | phase, ply, winner int32[256] each | packed moves | packed visits
#define SP_SLOTS 0
#define SP_STARTS 1024
#define SP_META 2048
#define SP_MOVES 8192

#define HIPCHK(call)                                                                          \
    do {                                                                                      \
        hipError_t e_ = (call);                                                               \
        if (e_ != hipSuccess) {                                                               \
            h->error = std::string(#call) + ": " + hipGetErrorString(e_);                     \
            return AZ_ERR_HIP;                                                                \
        }                                                                                     \
    } while (0)

struct az_engine {
    az_config cfg;
    Dev D;
    hipStream_t stream = nullptr;
    // two-network matches: the second network's tower runs beside the first on a stream of its own (a match's half
    // batches rarely fill the chip); created on first use
    hipStream_t stream2 = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    std::vector<void *> allocs;
    std::string error;
    Tower *tower = nullptr;
    // two-network matches: second weight set, which network moves first in each slot, per-network leaf lists
    Tower *tower2 = nullptr;
    bool two_nets = false;
    bool compact = true;  // run the tower only on the leaf rows that need it, when its kernels can take the count from the device
    int32_t *net_first = nullptr, *net_rows = nullptr, *net_cnt = nullptr;  // [G], [2][G], [2]
    unsigned long long *net_tot = nullptr;  // [2] boards each network has evaluated, summed on the device
    unsigned long long *net_tot_spare = nullptr;  // [2] sink for the totals of az_match_leaf_nets' extra split
    int16_t *pop_moves = nullptr;   // staging of az_trajectory_pop_many, grown on demand
    int32_t *pop_visits = nullptr;
    size_t pop_rows = 0;
    // staging
    int32_t *st_a = nullptr, *st_b = nullptr, *st_c = nullptr, *st_d = nullptr, *st_e = nullptr;  // [max(G,64)] each
    uint8_t *st_bytes = nullptr;   // [G * max(O, 17*NC)]
    size_t st_bytes_n = 0;
    // replay streams
    std::vector<void *> rng_bufs;  // 3 per slot
    // timing
    // a ring of event quads (round start / select done / tower done / round done): rounds are launched up to
    // AZ_EVQ ahead of the host, their phase times folded in when their quad is reused or the call ends
    hipEvent_t ev[AZ_EVQ][4];
    int ev_head = 0, ev_pending = 0;
    bool conv_profile = false;      // per-launch conv timing on every round (host waits for each round)
    bool serial_towers = true;      // !az_config.match_parallel_towers: both towers of a match on the engine's one stream
    int prof_every = 0;             // > 0: per-launch conv timing on every prof_every-th round only, rounds stay queued
    bool ev_prof[AZ_EVQ] = {};      // the queued round of this event quad carries the tower's per-launch events
    uint64_t round_index = 0;
    double tower_ms = 0, select_ms = 0, finish_ms = 0;
    uint64_t tower_calls = 0, tower_evals = 0, steps = 0;
    bool started = false;
};

template <typename T>
static T *dalloc(az_engine *h, size_t n, bool zero = true)
{
    void *p = nullptr;
    if (n == 0) n = 1;
    if (hipMalloc(&p, n * sizeof(T)) != hipSuccess) return nullptr;
    if (zero) hipMemset(p, 0, n * sizeof(T));
    h->allocs.push_back(p);
    return reinterpret_cast<T *>(p);
}

static int set_err(az_engine *h, int code, const char *msg)
{
    h->error = msg;
    return code;
}

static const char *err_text(int code)
{
    switch (code) {
    case AZ_ERR_TREE_FULL: return "tree full: tree_edge_cap / tree_block_cap reached, or the shared edge pool (tree_pool_edges) has no page left";
    case AZ_ERR_RNG_UNDERFLOW: return "RNG replay stream underflow";
    case AZ_ERR_DEPTH: return "playout deeper than max_depth";
    case AZ_ERR_STATE: return "search finished without a legal move";
    default: return "device-side error";
    }
}

// copies the per-slot error words back and reports the first one
static int check_slot_errors(az_engine *h)
{
    std::vector<int32_t> e(h->D.G);
    HIPCHK(hipMemcpyAsync(e.data(), h->D.err, sizeof(int32_t) * h->D.G, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    for (int g = 0; g < h->D.G; g++)
        if (e[g] != 0) {
            char buf[160];
            snprintf(buf, sizeof buf, "slot %d: %s", g, err_text(e[g]));
            h->error = buf;
            return e[g];
        }
    return AZ_OK;
}

// Fills the free stack of the edge pool with pages first .. first + n - 1 at positions at .. at + n - 1, highest page
// lowest, so that low pages are handed out first.
__global__ void k_pool_fill(int32_t *__restrict__ stack, int at, int first, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) stack[at + i] = first + n - 1 - i;
}

// Every tree's first page is pool page t, for good.
__global__ void k_pool_home(Dev D)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < 2 * D.G) { D.pt[(size_t)t * D.PTC] = t; D.nPages[t] = 1; }
}

// Raises the limits of every tree to new_ec edges / new_bc expanded nodes and the edge pool to new_pages pages (nothing
// ever shrinks): new arrays, old contents copied to their front (a tree's page table and block table keep their row, a
// pool page keeps its number), the added pages pushed on the free stack, old arrays freed.  All new arrays are
// allocated and filled before anything of the engine changes: a failed allocation or copy leaves trees, limits and pool
// exactly as they were.  The stream is idle on entry and on return.
static int grow_trees(az_engine *h, long long new_ec, long long new_bc, long long new_pages)
{
    Dev &D = h->D;
    if (new_ec < D.EC) new_ec = D.EC;
    if (new_bc < D.BC) new_bc = D.BC;
    if (new_pages < D.pool_pages) new_pages = D.pool_pages;
    const long long page = 1LL << D.PS;
    const long long new_ptc = (new_ec + page - 1) / page;
    if (new_ec > 0x7fffffffLL || new_bc > 0x7fffffffLL || new_pages > 0x7fffffffLL) return set_err(h, AZ_ERR_ARG, "tree caps too large");
    if (new_ec == D.EC && new_bc == D.BC && new_pages == D.pool_pages) return AZ_OK;
    HIPCHK(hipStreamSynchronize(h->stream));
    const size_t T = 2 * (size_t)D.G, rs = D.f64 ? 8 : 4;
    // kind 0: per-tree rows of BC entries; 1: per-tree rows of PTC entries; 2: one row of pool_pages << PS edges;
    // 3: one row of pool_pages entries (the free stack)
    struct Arr { void **p; size_t elem; int kind; };
    Arr arrs[] = {
        {(void **)&D.eN, 4, 2}, {&D.eW, rs, 2}, {&D.eP, rs, 2}, {(void **)&D.eChild, 4, 2}, {(void **)&D.eAct, 2, 2},
        {(void **)&D.bStart, 4, 0}, {(void **)&D.bCount, 4, 0}, {(void **)&D.bOwner, 4, 0},
        {(void **)&D.bParent, 4, 0}, {(void **)&D.bMap, 4, 0}, {(void **)&D.bShift, 4, 0},
        {(void **)&D.pt, 4, 1}, {(void **)&D.pool_free, 4, 3},
    };
    constexpr int NA = sizeof(arrs) / sizeof(arrs[0]);
    void *fresh[NA] = {};
    auto undo = [&]() { for (void *q : fresh) if (q) hipFree(q); };
    for (int i = 0; i < NA; i++) {
        const Arr &a = arrs[i];
        size_t rows = T, oc, nc;
        switch (a.kind) {
        case 0: oc = (size_t)D.BC; nc = (size_t)new_bc; break;
        case 1: oc = (size_t)D.PTC; nc = (size_t)new_ptc; break;
        case 2: rows = 1; oc = (size_t)D.pool_pages << D.PS; nc = (size_t)new_pages << D.PS; break;
        default: rows = 1; oc = (size_t)D.pool_pages; nc = (size_t)new_pages; break;
        }
        if (oc == nc) continue;
        if (hipMalloc(&fresh[i], rows * nc * a.elem) != hipSuccess) {
            (void)hipGetLastError();
            undo();
            return set_err(h, AZ_ERR_HIP, "device allocation failed while enlarging the trees (they are unchanged)");
        }
        const hipError_t e = rows == 1 ? hipMemcpy(fresh[i], *a.p, oc * a.elem, hipMemcpyDeviceToDevice)
                                       : hipMemcpy2D(fresh[i], nc * a.elem, *a.p, oc * a.elem, oc * a.elem, rows, hipMemcpyDeviceToDevice);
        if (e != hipSuccess) {
            undo();
            h->error = std::string("copy while enlarging the trees failed (they are unchanged): ") + hipGetErrorString(e);
            return AZ_ERR_HIP;
        }
    }
    const int added = (int)(new_pages - D.pool_pages);
    int32_t top[2] = {0, 0};
    if (added > 0) {
        const hipError_t e = hipMemcpy(top, D.pool_top, sizeof top, hipMemcpyDeviceToHost);
        if (e != hipSuccess || top[0] < 0 || top[0] > D.pool_pages) {
            undo();
            return set_err(h, AZ_ERR_HIP, "reading the edge pool's free count failed (the trees are unchanged)");
        }
    }
    for (int i = 0; i < NA; i++) {
        if (!fresh[i]) continue;
        for (void *&q : h->allocs)
            if (q == *arrs[i].p) { q = fresh[i]; break; }
        hipFree(*arrs[i].p);
        *arrs[i].p = fresh[i];
    }
    if (added > 0) {
        // the new pages go on top of the free stack; the low-water mark moves with the top, so the peak (pages in use) stays
        hipLaunchKernelGGL(k_pool_fill, dim3((added + 255) / 256), dim3(256), 0, h->stream, D.pool_free, top[0], D.pool_pages, added);
        top[0] += added; top[1] += added;
        HIPCHK(hipMemcpyAsync(D.pool_top, top, sizeof top, hipMemcpyHostToDevice, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
    }
    D.EC = (int)new_ec; D.BC = (int)new_bc; D.PTC = (int)new_ptc; D.pool_pages = (int)new_pages;
    return AZ_OK;
}

static void launch_tail(az_engine *h);
static int check_slot_errors(az_engine *h);

// tree_full_policy 2: a slot stopped by a full tree or an empty pool doubles the trees' limits and the pool, and its pending playout is completed
// (their leaf, path and - for the tower - the same planes are still in place).  selfplay: also play due moves.
static int check_and_grow(az_engine *h, bool selfplay)
{
    int rc = check_slot_errors(h);
    for (int tries = 0; rc == AZ_ERR_TREE_FULL && h->cfg.tree_full_policy == 2 && tries < 8; tries++) {
        if ((rc = grow_trees(h, 2LL * h->D.EC, 2LL * h->D.BC, 2LL * h->D.pool_pages))) return rc;
        h->error.clear();
        hipLaunchKernelGGL(k_clear_tree_full, dim3((h->D.G + 255) / 256), dim3(256), 0, h->stream, h->D);
        if (selfplay) launch_tail(h);
        else if (h->D.f64) hipLaunchKernelGGL(k_finish<double>, dim3(h->D.G), dim3(64), 0, h->stream, h->D, -1);
        else hipLaunchKernelGGL(k_finish<float>, dim3(h->D.G), dim3(64), 0, h->stream, h->D, -1);
        rc = check_slot_errors(h);
    }
    return rc;
}

extern "C" int az_search_grow_trees(az_engine *h, int32_t edge_cap, int32_t block_cap)
{
    if (!h || edge_cap < 0 || block_cap < 0) return AZ_ERR_ARG;
    HIPCHK(hipSetDevice(h->cfg.device));
    const long long ec = std::max<long long>(edge_cap ? edge_cap : 2LL * h->D.EC, h->D.EC);
    // the pool grows by the factor the per-tree limit grows by
    const long long pages = (long long)(((__int128)h->D.pool_pages * ec + h->D.EC - 1) / h->D.EC);
    return grow_trees(h, ec, block_cap ? block_cap : 2LL * h->D.BC, pages);
}

extern "C" int az_search_grow_pool(az_engine *h, int64_t pool_edges)
{
    if (!h || pool_edges < 0) return AZ_ERR_ARG;
    HIPCHK(hipSetDevice(h->cfg.device));
    const long long page = 1LL << h->D.PS, most = 2LL * h->D.G * h->D.PTC;
    long long pages = pool_edges ? (pool_edges + page - 1) / page : 2LL * h->D.pool_pages;
    if (pages > most) pages = most;
    return grow_trees(h, h->D.EC, h->D.BC, pages);
}

extern "C" const char *az_last_error(az_engine *h) { return h ? h->error.c_str() : g_create_error.c_str(); }

extern "C" void az_engine_destroy(az_engine *h)
{
    if (!h) return;
    hipSetDevice(h->cfg.device);
    if (h->stream) hipStreamSynchronize(h->stream);
    if (h->tower) tower_destroy(h->tower);
    if (h->tower2) tower_destroy(h->tower2);
    for (void *p : h->allocs) hipFree(p);
    for (void *p : h->rng_bufs) if (p) hipFree(p);
    if (h->pop_moves) hipFree(h->pop_moves);
    if (h->pop_visits) hipFree(h->pop_visits);
    for (int q = 0; q < AZ_EVQ; q++)
        for (int i = 0; i < 4; i++) if (h->ev[q][i]) hipEventDestroy(h->ev[q][i]);
    if (h->stream2) { hipStreamSynchronize(h->stream2); hipStreamDestroy(h->stream2); }
    if (h->ev_fork) hipEventDestroy(h->ev_fork);
    if (h->ev_join) hipEventDestroy(h->ev_join);
    if (h->stream) hipStreamDestroy(h->stream);
    delete h;
}

extern "C" void az_abi_sizes(int32_t out[3])
{
    if (!out) return;
    out[0] = (int32_t)sizeof(az_config); out[1] = (int32_t)sizeof(az_state_blob); out[2] = (int32_t)sizeof(az_counters);
}

static TowerOptions tower_options(const az_config &c)
{
    TowerOptions o;
    o.graphs = !c.tower_no_graphs; o.duo = !c.tower_no_duo; o.group_boards = c.tower_group_boards;
    return o;
}

extern "C" az_engine *az_engine_create(const az_config *cfg)
{
    if (!cfg) { g_create_error = "null config"; return nullptr; }
    if (cfg->struct_size != sizeof(az_config)) {   // (the first member: safe to read whatever the caller's struct is)
        char buf[160];
        snprintf(buf, sizeof buf, "az_config.struct_size is %u, this library's az_config has %zu bytes: the binding does not match include/az_engine.h",
                 cfg->struct_size, sizeof(az_config));
        g_create_error = buf;
        return nullptr;
    }
    if (!(cfg->c_punt >= 0.0)) { g_create_error = "c_punt must be positive (0 = 5.0)"; return nullptr; }
    const int H = cfg->board_height, W = cfg->board_width, NC = H * W;
    if (cfg->game < 0 || cfg->game > 2 || H < 1 || W < 1 || NC > AZ_MAX_CELLS || cfg->n_slots < 1) {
        g_create_error = "bad game / board size / n_slots";
        return nullptr;
    }
    if (cfg->transform_types > 1 && H != W) {
        g_create_error = "dihedral transforms need a square board (reference asserts this: env/mnk.py:199)";
        return nullptr;
    }
    if (cfg->game == AZ_REVERSI && (H != W || (H & 1))) { g_create_error = "reversi needs an even square board"; return nullptr; }
    if (!(cfg->d_epsilon >= 0.0 && cfg->d_epsilon <= 1.0)) { g_create_error = "d_epsilon must lie in [0, 1]"; return nullptr; }
    if (cfg->tree_full_policy < 0 || cfg->tree_full_policy > 2) { g_create_error = "tree_full_policy must be 0, 1 or 2"; return nullptr; }
    if (!cfg->rng_replay && !(cfg->d_alpha > 0.0)) { g_create_error = "d_alpha must be positive (random.gammavariate raises otherwise)"; return nullptr; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { g_create_error = "no HIP device"; return nullptr; }
    if (hipSetDevice(cfg->device) != hipSuccess) { g_create_error = "hipSetDevice failed"; return nullptr; }
    if (cfg->device >= 0 && cfg->device < AZ_MAX_DEVICES) {
        std::lock_guard<std::mutex> lk(g_scratch_mutex);
        if (!g_scratch_ptr[cfg->device]) {
            void *p = nullptr;
            const hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
            if (e != hipSuccess || !p) {
                g_create_error = std::string("hipGetSymbolAddress(scratch): ") + hipGetErrorString(e);
                return nullptr;
            }
            g_scratch_ptr[cfg->device] = static_cast<unsigned char *>(p);
        }
    }
    az_engine *h = new az_engine();
    h->error.reserve(256);  // messages are set without allocating from then on
    for (int q = 0; q < AZ_EVQ; q++)
        for (int i = 0; i < 4; i++) h->ev[q][i] = nullptr;
    h->cfg = *cfg;
    Dev &D = h->D;
    memset(&D, 0, sizeof(D));
    D.geo.game = cfg->game; D.geo.H = H; D.geo.W = W; D.geo.NC = NC; D.geo.k = cfg->k;
    D.geo.O = NC + (cfg->game == AZ_MNK ? 0 : 1);
    D.geo.superko = (cfg->game == AZ_GO && cfg->enforce_superko) ? 1 : 0;
    const int G = cfg->n_slots, O = D.geo.O;
    D.G = G;
    D.max_playout = cfg->max_playout;
    D.transform_types = cfg->transform_types;
    D.transform_on = !(cfg->transform_types == 0 || cfg->transform_types == 1);
    D.dirichlet_before = cfg->dirichlet_before;
    D.max_turn = cfg->max_turn;
    D.d_alpha = cfg->d_alpha; D.d_epsilon = cfg->d_epsilon;
    D.c_punt = cfg->c_punt > 0.0 ? cfg->c_punt : 5.0;
    h->serial_towers = cfg->match_parallel_towers == 0;
    long long ec = cfg->tree_edge_cap > 0 ? cfg->tree_edge_cap : 3LL * (cfg->max_playout + 1) * O;
    long long bc = cfg->tree_block_cap > 0 ? cfg->tree_block_cap : 8LL * (cfg->max_playout + 1);
    if (ec < O) ec = O;
    if (bc < 2) bc = 2;
    if (ec > 0x7fffffffLL || bc > 0x7fffffffLL) { g_create_error = "tree caps too large"; delete h; return nullptr; }
    D.EC = (int)ec; D.BC = (int)bc;
    {
        // pages of the shared edge pool: a power of two, at least the number of actions (a node's children then touch at
        // most two pages)
        int o2 = 1, ps = 0;
        while (o2 < O) o2 <<= 1;
        long long page = cfg->tree_page_edges > 0 ? cfg->tree_page_edges : std::max(64, 4 * o2);
        if ((page & (page - 1)) != 0 || page < o2 || page > (1 << 24)) {
            g_create_error = "tree_page_edges must be a power of two, at least the number of actions and at most 2^24";
            delete h; return nullptr;
        }
        while ((1LL << ps) < page) ps++;
        D.PS = ps;
        D.PTC = (int)((ec + page - 1) / page);
        const long long T2 = 2LL * cfg->n_slots;
        long long pool_edges = cfg->tree_pool_edges;
        if (pool_edges < 0) { g_create_error = "tree_pool_edges must not be negative"; delete h; return nullptr; }
        long long pages = pool_edges > 0 ? (pool_edges + page - 1) / page
                          : (cfg->tree_edge_cap > 0 ? T2 * D.PTC : (T2 * ec / 2 + page - 1) / page);
        // every tree owns one page for good (a root's children always have a place)
        pages = std::max(pages, T2);
        pages = std::min(pages, T2 * D.PTC);
        if (pages > 0x7fffffffLL) { g_create_error = "tree pool too large"; delete h; return nullptr; }
        D.pool_pages = (int)pages;
    }
    D.MAXD = cfg->max_depth > 0 ? cfg->max_depth : 1024;
    D.rng_replay = cfg->rng_replay; D.seed = cfg->seed;
    D.record = cfg->record_trajectories;
    D.full_policy = cfg->tree_full_policy;
    D.K = cfg->leaves_per_round > 1 ? cfg->leaves_per_round : 1;
    D.vloss = cfg->virtual_loss;
    D.traj_plies = cfg->record_trajectories ? cfg->max_turn + 2 : 1;

    bool ok = hipStreamCreate(&h->stream) == hipSuccess;
    for (int q = 0; q < AZ_EVQ && ok; q++)
        for (int i = 0; i < 4 && ok; i++) ok = hipEventCreate(&h->ev[q][i]) == hipSuccess;
#define A(field, T, n) ok = ok && ((D.field = dalloc<T>(h, (size_t)(n))) != nullptr)
    A(ring, int8_t, (size_t)G * 8 * AZ_CP);
    A(sc, int32_t, (size_t)G * S_N);
    A(phase, int32_t, G); A(side, int32_t, G); A(dflag, int32_t, G); A(pdone, int32_t, G);
    A(automode, int32_t, G); A(ply, int32_t, G); A(err, int32_t, G);
    const size_t LG = (size_t)D.K * G;  // leaf rows per round
    A(issued, int32_t, G);
    A(leaf_valid, int32_t, LG); A(leaf_node, int32_t, LG); A(leaf_block, int32_t, LG); A(leaf_player, int32_t, LG);
    A(leaf_end, int32_t, LG); A(leaf_winner, int32_t, LG); A(leaf_tid, int32_t, LG); A(leaf_root, int32_t, LG);
    A(path_len, int32_t, LG);
    A(path_node, int32_t, LG * D.MAXD);
    A(path_player, int8_t, LG * D.MAXD);
    A(leaf_mask, uint8_t, LG * AZ_OP);
    A(planes, int8_t, LG * 17 * NC + 64);
    D.f64 = cfg->search_f64 ? 1 : 0;
    const size_t rs = D.f64 ? 8 : 4;  // bytes of one search statistic
#define AR(field, n) ok = ok && ((D.field = dalloc<char>(h, (size_t)(n) * rs)) != nullptr)
    AR(priors, LG * O);
    AR(values, LG);
    const size_t T = 2 * (size_t)G;
    A(rootN, int32_t, T); AR(rootW, T); AR(rootP, T); A(rootBlock, int32_t, T);
    A(nBlocks, int32_t, T); A(nEdges, int32_t, T); A(pending, int32_t, T);
    const size_t PE = (size_t)D.pool_pages << D.PS;   // edges in the pool
    A(eN, int32_t, PE); AR(eW, PE); AR(eP, PE);
    A(eChild, int32_t, PE); A(eAct, int16_t, PE);
    A(pt, int32_t, T * D.PTC); A(nPages, int32_t, T); A(pool_free, int32_t, D.pool_pages); A(pool_top, int32_t, 2);
    A(bStart, int32_t, T * D.BC); A(bCount, int32_t, T * D.BC); A(bOwner, int32_t, T * D.BC);
    A(bParent, int32_t, T * D.BC); A(bMap, int32_t, T * D.BC); A(bShift, int32_t, T * D.BC);
    A(rt, const int32_t *, G); A(rt_n, long long, G); A(rt_i, long long, G);
    A(re, const double *, G); A(re_n, long long, G); A(re_i, long long, G);
    A(ru, const double *, G); A(ru_n, long long, G); A(ru_i, long long, G);
    A(rng_ctr, unsigned long long, G);
    A(tr_moves, int16_t, (size_t)G * D.traj_plies);
    A(tr_visits, int32_t, (size_t)G * D.traj_plies * O);
    A(tr_winner, int32_t, G);
    A(ctr, unsigned long long, (size_t)AZ_CTR_BUCKETS * (C_N + 1));
    A(peak, int32_t, (size_t)AZ_CTR_BUCKETS * 2);
    if (D.geo.superko) {
        D.sk_cap = std::max(cfg->max_turn + 2, 1024) + D.MAXD;
        A(sk_moves, int16_t, (size_t)G * D.sk_cap);
        A(sk_hash, unsigned long long, (size_t)G * D.sk_cap * 2);
    }
#undef A
#undef AR
    const size_t stn = std::max(G, 64);
    ok = ok && (h->st_a = dalloc<int32_t>(h, stn)) && (h->st_b = dalloc<int32_t>(h, stn)) &&
         (h->st_c = dalloc<int32_t>(h, stn)) && (h->st_d = dalloc<int32_t>(h, stn)) && (h->st_e = dalloc<int32_t>(h, stn));
    ok = ok && (h->net_first = dalloc<int32_t>(h, G)) && (h->net_rows = dalloc<int32_t>(h, 2 * (size_t)G * std::max(1, (int)cfg->leaves_per_round))) &&
         (h->net_cnt = dalloc<int32_t>(h, 2)) && (h->net_tot = dalloc<unsigned long long>(h, 2)) &&
         (h->net_tot_spare = dalloc<unsigned long long>(h, 2));
    h->st_bytes_n = (size_t)G * std::max(O, 17 * NC);
    ok = ok && (h->st_bytes = dalloc<uint8_t>(h, h->st_bytes_n));
    if (!ok) {
        g_create_error = std::string("device allocation failed: ") + hipGetErrorString(hipGetLastError());
        az_engine_destroy(h);
        return nullptr;
    }
    h->rng_bufs.assign((size_t)3 * G, nullptr);
    if (cfg->num_blocks > 0) {
        std::string terr;
        h->tower = tower_create(H, W, O, cfg->num_blocks, cfg->channels > 0 ? cfg->channels : 256, cfg->tower_dtype, D.K * G,
                                tower_options(*cfg), terr);
        if (!h->tower) { g_create_error = "tower: " + terr; az_engine_destroy(h); return nullptr; }
    }
    {
        // pages 0 .. 2G-1 are the trees' own first pages, the others start on the free stack
        const int T2 = 2 * G, nfree = D.pool_pages - T2;
        const int32_t top[2] = {nfree, nfree};
        hipLaunchKernelGGL(k_pool_home, dim3((T2 + 255) / 256), dim3(256), 0, h->stream, D);
        if (nfree > 0) hipLaunchKernelGGL(k_pool_fill, dim3((nfree + 255) / 256), dim3(256), 0, h->stream, D.pool_free, 0, T2, nfree);
        (void)hipMemcpyAsync(D.pool_top, top, sizeof top, hipMemcpyHostToDevice, h->stream);
        (void)hipStreamSynchronize(h->stream);   // (top is a stack variable)
    }
    hipLaunchKernelGGL(k_reset, dim3(G), dim3(64), 0, h->stream, D, (const int32_t *)nullptr, G, 0);
    if (hipStreamSynchronize(h->stream) != hipSuccess) {
        g_create_error = std::string("reset kernel failed: ") + hipGetErrorString(hipGetLastError());
        az_engine_destroy(h);
        return nullptr;
    }
    return h;
}

extern "C" int az_engine_dims(az_engine *h, int32_t *n_slots, int32_t *cells, int32_t *actions)
{
    if (!h) return AZ_ERR_ARG;
    if (n_slots) *n_slots = h->D.G;
    if (cells) *cells = h->D.geo.NC;
    if (actions) *actions = h->D.geo.O;
    return AZ_OK;
}

static int check_slots(az_engine *h, const int32_t *slots, int n)
{
    if (!h) return AZ_ERR_ARG;
    if (n < 0 || n > h->D.G || (n > 0 && !slots)) return set_err(h, AZ_ERR_ARG, "bad slot list");
    for (int i = 0; i < n; i++)
        if (slots[i] < 0 || slots[i] >= h->D.G) return set_err(h, AZ_ERR_ARG, "slot out of range");
    return AZ_OK;
}

static int up(az_engine *h, int32_t *dst, const int32_t *src, int n)
{
    HIPCHK(hipMemcpyAsync(dst, src, sizeof(int32_t) * n, hipMemcpyHostToDevice, h->stream));
    return AZ_OK;
}

extern "C" int az_games_reset(az_engine *h, const int32_t *slots, int32_t n)
{
    int rc = check_slots(h, slots, n);
    if (rc) return rc;
    if (n == 0) return AZ_OK;
    {   // one workgroup per listed slot gives the slot's tree pages back: a slot listed twice would give them back twice
        std::vector<char> seen(h->D.G, 0);
        for (int i = 0; i < n; i++) {
            if (seen[slots[i]]) return set_err(h, AZ_ERR_ARG, "duplicate slot in one reset");
            seen[slots[i]] = 1;
        }
    }
    HIPCHK(hipSetDevice(h->cfg.device));
    if ((rc = up(h, h->st_a, slots, n))) return rc;
    hipLaunchKernelGGL(k_reset, dim3(n), dim3(64), 0, h->stream, h->D, (const int32_t *)h->st_a, n, 0);
    HIPCHK(hipStreamSynchronize(h->stream));
    return AZ_OK;
}

extern "C" int az_state_copy(az_engine *dst, int32_t ds, az_engine *src, int32_t ss, int32_t ref)
{
    if (!dst || !src) return AZ_ERR_ARG;
    az_engine *h = dst;
    if (ds < 0 || ds >= dst->D.G || ss < 0 || ss >= src->D.G) return set_err(h, AZ_ERR_ARG, "slot out of range");
    if (dst->D.geo.game != src->D.geo.game || dst->D.geo.H != src->D.geo.H || dst->D.geo.W != src->D.geo.W ||
        dst->cfg.device != src->cfg.device)
        return set_err(h, AZ_ERR_ARG, "engines differ in game, board or device");
    if (dst->D.geo.superko) {
        if (!src->D.geo.superko) return set_err(h, AZ_ERR_ARG, "a superko engine needs the move record of a superko source");
        int32_t nh = 0;
        HIPCHK(hipSetDevice(h->cfg.device));
        HIPCHK(hipStreamSynchronize(src->stream));
        HIPCHK(hipMemcpy(&nh, src->D.sc + (size_t)ss * S_N + S_NHIST, 4, hipMemcpyDeviceToHost));
        if (nh + dst->D.MAXD > dst->D.sk_cap) return set_err(h, AZ_ERR_ARG, "the game is too long for the destination's superko record");
    }
    HIPCHK(hipSetDevice(h->cfg.device));
    HIPCHK(hipStreamSynchronize(src->stream));
    hipLaunchKernelGGL(k_state_copy, dim3(1), dim3(64), 0, dst->stream, dst->D, ds, src->D, ss, ref);
    HIPCHK(hipStreamSynchronize(dst->stream));
    return AZ_OK;
}

extern "C" int az_state_get(az_engine *h, int32_t slot, az_state_blob *out)
{
    if (!h || !out || slot < 0 || slot >= h->D.G) return AZ_ERR_ARG;
    HIPCHK(hipSetDevice(h->cfg.device));
    int8_t ring[8][AZ_CP];
    int32_t sc[S_N];
    HIPCHK(hipMemcpyAsync(ring, h->D.ring + (size_t)slot * 8 * AZ_CP, sizeof ring, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipMemcpyAsync(sc, h->D.sc + (size_t)slot * S_N, sizeof sc, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    int head = sc[S_HEAD] & 7;
    memcpy(out->board, ring[head], AZ_CP);
    for (int t = 0; t < 7; t++) memcpy(out->history[t], ring[(head + 1 + t) & 7], AZ_CP);
    out->current_player = sc[S_CUR]; out->ko = sc[S_KO]; out->turns = sc[S_TURNS]; out->n_history = sc[S_NHIST];
    out->last1 = sc[S_LAST1]; out->last2 = sc[S_LAST2]; out->is_end_of_game = sc[S_END];
    out->winner = sc[S_WIN]; out->winner_set = sc[S_WINSET];
    out->passes_black = sc[S_PB]; out->passes_white = sc[S_PW];
    out->num_black_prisoners = sc[S_PRB]; out->num_white_prisoners = sc[S_PRW]; out->stones_played = sc[S_STONES];
    return AZ_OK;
}

extern "C" int az_state_set(az_engine *h, int32_t slot, const az_state_blob *in)
{
    if (!h || !in || slot < 0 || slot >= h->D.G) return AZ_ERR_ARG;
    HIPCHK(hipSetDevice(h->cfg.device));
    int8_t ring[8][AZ_CP];
    int32_t sc[S_N] = {0};
    memcpy(ring[0], in->board, AZ_CP);
    for (int t = 0; t < 7; t++) memcpy(ring[1 + t], in->history[t], AZ_CP);
    sc[S_HEAD] = 0; sc[S_CUR] = in->current_player; sc[S_KO] = in->ko; sc[S_TURNS] = in->turns;
    sc[S_NHIST] = in->n_history; sc[S_LAST1] = in->last1; sc[S_LAST2] = in->last2; sc[S_END] = in->is_end_of_game;
    sc[S_WIN] = in->winner; sc[S_WINSET] = in->winner_set; sc[S_PB] = in->passes_black; sc[S_PW] = in->passes_white;
    sc[S_PRB] = in->num_black_prisoners; sc[S_PRW] = in->num_white_prisoners; sc[S_STONES] = in->stones_played;
    if (h->D.geo.superko) {
        // the blob carries no move list: the slot's own superko record stays valid only if the blob describes the same
        // game at the same length (a caller that edits scalars such as current_player); a fresh game starts it anew
        int32_t old[S_N];
        HIPCHK(hipMemcpyAsync(old, h->D.sc + (size_t)slot * S_N, sizeof old, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
        if (in->n_history == old[S_NHIST]) {
            for (int k = S_NPREV; k <= S_H1B; k++) sc[k] = old[k];
        } else if (in->n_history != 0) {
            return set_err(h, AZ_ERR_ARG, "az_state_set on a superko engine: the blob's history differs from the slot's (no move list to restore)");
        }
    }
    HIPCHK(hipMemcpyAsync(h->D.ring + (size_t)slot * 8 * AZ_CP, ring, sizeof ring, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(h->D.sc + (size_t)slot * S_N, sc, sizeof sc, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return AZ_OK;
}

extern "C" int az_rules_legal_mask(az_engine *h, const int32_t *slots, int32_t n, uint8_t *mask)
{
    int rc = check_slots(h, slots, n);
    if (rc) return rc;
    if (n == 0) return AZ_OK;
    if (!mask) return set_err(h, AZ_ERR_ARG, "null output");
    HIPCHK(hipSetDevice(h->cfg.device));
    if ((rc = up(h, h->st_a, slots, n))) return rc;
    hipLaunchKernelGGL(k_rules_mask, dim3(n), dim3(64), 0, h->stream, h->D, (const int32_t *)h->st_a, n, h->st_bytes);
    HIPCHK(hipMemcpyAsync(mask, h->st_bytes, (size_t)n * h->D.geo.O, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return AZ_OK;
}

extern "C" int az_rules_do_move(az_engine *h, const int32_t *slots, int32_t n, const int32_t *actions,
                                const int32_t *colors, int32_t *status, int32_t *ended)
{
    int rc = check_slots(h, slots, n);
    if (rc) return rc;
    if (n == 0) return AZ_OK;
    if (!actions || !status) return set_err(h, AZ_ERR_ARG, "null argument");
    {
        std::vector<char> seen(h->D.G, 0);
        for (int i = 0; i < n; i++) {
            if (seen[slots[i]]) return set_err(h, AZ_ERR_ARG, "duplicate slot in one do_move batch");
            seen[slots[i]] = 1;
        }
    }
    HIPCHK(hipSetDevice(h->cfg.device));
    if ((rc = up(h, h->st_a, slots, n)) || (rc = up(h, h->st_b, actions, n))) return rc;
    if (colors && (rc = up(h, h->st_c, colors, n))) return rc;
    hipLaunchKernelGGL(k_rules_domove, dim3(n), dim3(64), 0, h->stream, h->D, (const int32_t *)h->st_a, n,
                       (const int32_t *)h->st_b, colors ? (const int32_t *)h->st_c : (const int32_t *)nullptr, h->st_d, h->st_e);
    std::vector<int32_t> en(n);
    HIPCHK(hipMemcpyAsync(status, h->st_d, sizeof(int32_t) * n, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipMemcpyAsync(en.data(), h->st_e, sizeof(int32_t) * n, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    if (ended) memcpy(ended, en.data(), sizeof(int32_t) * n);
    return AZ_OK;
}

extern "C" int az_rules_winner(az_engine *h, const int32_t *slots, int32_t n, int32_t *winner)
{
    int rc = check_slots(h, slots, n);
    if (rc) return rc;
    if (n == 0) return AZ_OK;
    if (!winner) return set_err(h, AZ_ERR_ARG, "null output");
    HIPCHK(hipSetDevice(h->cfg.device));
    if ((rc = up(h, h->st_a, slots, n))) return rc;
    hipLaunchKernelGGL(k_rules_winner, dim3(n), dim3(64), 0, h->stream, h->D, (const int32_t *)h->st_a, n, h->st_d);
    HIPCHK(hipMemcpyAsync(winner, h->st_d, sizeof(int32_t) * n, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return AZ_OK;
}

extern "C" int az_state_transform(az_engine *h, int32_t slot, int32_t tid)
{
    if (!h || slot < 0 || slot >= h->D.G) return AZ_ERR_ARG;
    if (tid < 0 || tid > 7 || (tid != 0 && h->D.geo.H != h->D.geo.W)) return set_err(h, AZ_ERR_ARG, "bad transform id");
    HIPCHK(hipSetDevice(h->cfg.device));
    hipLaunchKernelGGL(k_transform, dim3(1), dim3(64), 0, h->stream, h->D, slot, tid);
    HIPCHK(hipStreamSynchronize(h->stream));
    return AZ_OK;
}

extern "C" int az_features(az_engine *h, const int32_t *slots, int32_t n, const int32_t *tids, int8_t *planes)
{
    int rc = check_slots(h, slots, n);
    if (rc) return rc;
    if (n == 0) return AZ_OK;
    if (!planes) return set_err(h, AZ_ERR_ARG, "null output");
    if (tids)
        for (int i = 0; i < n; i++)
            if (tids[i] < 0 || tids[i] > 7 || (tids[i] != 0 && h->D.geo.H != h->D.geo.W))
                return set_err(h, AZ_ERR_ARG, "bad transform id");
    HIPCHK(hipSetDevice(h->cfg.device));
    if ((rc = up(h, h->st_a, slots, n))) return rc;
    if (tids && (rc = up(h, h->st_b, tids, n))) return rc;
    hipLaunchKernelGGL(k_features, dim3(n), dim3(64), 0, h->stream, h->D, (const int32_t *)h->st_a, n,
                       tids ? (const int32_t *)h->st_b : (const int32_t *)nullptr, (int8_t *)h->st_bytes);
    HIPCHK(hipMemcpyAsync(planes, h->st_bytes, (size_t)n * 17 * h->D.geo.NC, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return AZ_OK;
}

// ---- search -------------------------------------------------------------

extern "C" int az_search_begin(az_engine *h, const int32_t *slots, int32_t n, const int32_t *sides, const int32_t *dirichlet)
{
    int rc = check_slots(h, slots, n);
    if (rc) return rc;
    if (n == 0) return AZ_OK;
    if (!sides || !dirichlet) return set_err(h, AZ_ERR_ARG, "null argument");
    for (int i = 0; i < n; i++)
        if (sides[i] < 0 || sides[i] > 1) return set_err(h, AZ_ERR_ARG, "side must be 0 or 1");
    HIPCHK(hipSetDevice(h->cfg.device));
    if ((rc = up(h, h->st_a, slots, n)) || (rc = up(h, h->st_b, sides, n)) || (rc = up(h, h->st_c, dirichlet, n))) return rc;
    hipLaunchKernelGGL(k_begin, dim3((n + 63) / 64), dim3(64), 0, h->stream, h->D, (const int32_t *)h->st_a, n,
                       (const int32_t *)h->st_b, (const int32_t *)h->st_c);
    HIPCHK(hipStreamSynchronize(h->stream));
    return AZ_OK;
}

static int launch_select(az_engine *h)
{
    HIPCHK(hipMemsetAsync(h->D.ctr + (size_t)AZ_CTR_BUCKETS * C_N, 0, sizeof(unsigned long long) * AZ_CTR_BUCKETS, h->stream));
    if (h->D.K > 1) {  // all sub-rounds of a leaves_per_round > 1 search inside one launch
        if (h->D.f64) hipLaunchKernelGGL(k_select_all<double>, dim3(h->D.G), dim3(64), 0, h->stream, h->D);
        else hipLaunchKernelGGL(k_select_all<float>, dim3(h->D.G), dim3(64), 0, h->stream, h->D);
    } else {
        if (h->D.f64) hipLaunchKernelGGL(k_select<double>, dim3(h->D.G), dim3(64), 0, h->stream, h->D, 0);
        else hipLaunchKernelGGL(k_select<float>, dim3(h->D.G), dim3(64), 0, h->stream, h->D, 0);
    }
    HIPCHK(hipGetLastError());
    return AZ_OK;
}

extern "C" int az_search_step_select(az_engine *h, int32_t *n_leaves)
{
    if (!h) return AZ_ERR_ARG;
    HIPCHK(hipSetDevice(h->cfg.device));
    int rc = launch_select(h);
    if (rc) return rc;
    unsigned long long nl = 0;
    unsigned long long nlb[AZ_CTR_BUCKETS];
    HIPCHK(hipMemcpyAsync(nlb, h->D.ctr + (size_t)AZ_CTR_BUCKETS * C_N, sizeof nlb, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    for (int b = 0; b < AZ_CTR_BUCKETS; b++) nl += nlb[b];
    if (n_leaves) *n_leaves = (int32_t)nl;
    return check_slot_errors(h);
}

extern "C" int az_leaf_batch(az_engine *h, int8_t *planes, uint8_t *valid)
{
    if (!h) return AZ_ERR_ARG;
    HIPCHK(hipSetDevice(h->cfg.device));
    const int G = h->D.K * h->D.G;  // leaf rows: [leaves_per_round][n_slots]
    if (planes)
        HIPCHK(hipMemcpyAsync(planes, h->D.planes, (size_t)G * 17 * h->D.geo.NC, hipMemcpyDeviceToHost, h->stream));
    std::vector<int32_t> v(G);
    HIPCHK(hipMemcpyAsync(v.data(), h->D.leaf_valid, sizeof(int32_t) * G, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    if (valid) for (int g = 0; g < G; g++) valid[g] = (uint8_t)v[g];
    return AZ_OK;
}

extern "C" int az_leaf_outputs(az_engine *h, float *priors, float *values)
{
    if (!h) return AZ_ERR_ARG;
    if (h->D.f64) return set_err(h, AZ_ERR_ARG, "engine searches in float64: its evaluator outputs are the caller's own arrays");
    HIPCHK(hipSetDevice(h->cfg.device));
    const size_t LG = (size_t)h->D.K * h->D.G;
    if (priors) HIPCHK(hipMemcpyAsync(priors, h->D.priors, sizeof(float) * LG * h->D.geo.O, hipMemcpyDeviceToHost, h->stream));
    if (values) HIPCHK(hipMemcpyAsync(values, h->D.values, sizeof(float) * LG, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return AZ_OK;
}

// Two-network rounds without a read-back: both towers take the device-side batch size, and the slot count is large
// enough that the slab kernels (the only ones that read it) are the right choice even for a half-empty batch.
static bool two_nets_async(const az_engine *h)
{
    return h->two_nets && h->tower && h->tower2 && !h->conv_profile &&
           tower_takes_device_count(h->tower, h->D.G) && tower_takes_device_count(h->tower2, h->D.G);
}

// priors / values: host arrays of the engine's statistic type (f64 says which the caller passed), or NULL = run the tower
static int evaluate(az_engine *h, const void *priors, const void *values, int f64 = 0, bool full_batch = false)
{
    const int G = h->D.G, O = h->D.geo.O, LG = h->D.K * h->D.G;
    if ((priors == nullptr) != (values == nullptr)) return set_err(h, AZ_ERR_ARG, "priors and values must both be given or both NULL");
    if (priors) {
        if (f64 != h->D.f64)
            return set_err(h, AZ_ERR_ARG, h->D.f64 ? "engine searches in float64: use the _f64 entry point" : "engine searches in float32: use the float entry point");
        const size_t rs = h->D.f64 ? 8 : 4;
        HIPCHK(hipMemcpyAsync(h->D.priors, priors, rs * (size_t)LG * O, hipMemcpyHostToDevice, h->stream));
        HIPCHK(hipMemcpyAsync(h->D.values, values, rs * LG, hipMemcpyHostToDevice, h->stream));
        return AZ_OK;
    }
    if (h->D.f64) return set_err(h, AZ_ERR_ARG, "a float64 search takes injected priors only: the tower's outputs are float32");
    if (!h->tower || !tower_ready(h->tower)) return set_err(h, AZ_ERR_NO_WEIGHTS, "no tower weights loaded and no priors injected");
    std::string terr;
    if (h->two_nets) {
        if (h->D.K > 1) return set_err(h, AZ_ERR_ARG, "two-network matches run one leaf per tree per round (leaves_per_round = 1)");
        if (!h->tower2 || !tower_ready(h->tower2)) return set_err(h, AZ_ERR_NO_WEIGHTS, "two-network match: network 1 has no weights");
        hipLaunchKernelGGL(k_partition, dim3(1), dim3(1024), 0, h->stream, h->D, (const int32_t *)h->net_first, h->net_rows, h->net_cnt,
                           h->net_tot);
        Tower *tw[2] = {h->tower, h->tower2};
        if (two_nets_async(h)) {
            // the towers read their batch sizes from device memory: nothing comes back to the host, rounds can be queued.
            // The two passes touch disjoint rows of priors / values and each tower has its own activations: the second
            // runs on its own stream between a fork and a join event, so that two half-empty passes share the chip.
            if (!h->stream2 && !h->serial_towers) {
                // (a stream of another priority gets a hardware queue of its own; streams of one priority may share one,
                // and kernels in one queue run one after the other)
                int lo = 0, hi = 0;
                if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess) { lo = hi = 0; (void)hipGetLastError(); }
                if (hipStreamCreateWithPriority(&h->stream2, hipStreamNonBlocking, hi) != hipSuccess ||
                    hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming) != hipSuccess ||
                    hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming) != hipSuccess) {
                    (void)hipGetLastError();
                    if (h->stream2) { hipStreamDestroy(h->stream2); h->stream2 = nullptr; }
                    h->serial_towers = true;
                }
            }
            const bool beside = h->stream2 && !h->serial_towers;
            if (beside) { HIPCHK(hipEventRecord(h->ev_fork, h->stream)); HIPCHK(hipStreamWaitEvent(h->stream2, h->ev_fork, 0)); }
            for (int net = 0; net < 2; net++) {
                if (tower_forward(tw[net], h->D.planes, G, (float *)h->D.priors, (float *)h->D.values, net && beside ? h->stream2 : h->stream,
                                  terr, h->net_rows + (size_t)net * G, h->net_cnt + net) != 0) {
                    h->error = "tower: " + terr;
                    return AZ_ERR_HIP;
                }
                h->tower_calls++;
            }
            if (beside) { HIPCHK(hipEventRecord(h->ev_join, h->stream2)); HIPCHK(hipStreamWaitEvent(h->stream, h->ev_join, 0)); }
            return AZ_OK;
        }
        int32_t cnt[2] = {0, 0};
        HIPCHK(hipMemcpyAsync(cnt, h->net_cnt, sizeof cnt, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
        for (int net = 0; net < 2; net++) {
            if (cnt[net] == 0) continue;
            if (tower_forward(tw[net], h->D.planes, cnt[net], (float *)h->D.priors, (float *)h->D.values, h->stream, terr, h->net_rows + (size_t)net * G) != 0) {
                h->error = "tower: " + terr;
                return AZ_ERR_HIP;
            }
            h->tower_calls++;
        }
        return AZ_OK;
    }
    if (h->compact && !h->conv_profile && !full_batch && tower_takes_device_count(h->tower, LG)) {
        // terminal leaves and idle slots stay out of the batch: the surviving rows are listed on the device and the
        // tower's kernels read their number there (per-launch conv timing measures full batches, so it turns this off)
        hipLaunchKernelGGL(k_compact, dim3(1), dim3(1024), 0, h->stream, h->D, LG, h->net_rows, h->net_cnt, h->net_tot);
        if (tower_forward(h->tower, h->D.planes, LG, (float *)h->D.priors, (float *)h->D.values, h->stream, terr, h->net_rows, h->net_cnt) != 0) {
            h->error = "tower: " + terr;
            return AZ_ERR_HIP;
        }
        h->tower_calls++;
        return AZ_OK;
    }
    if (tower_forward(h->tower, h->D.planes, LG, (float *)h->D.priors, (float *)h->D.values, h->stream, terr) != 0) {
        h->error = "tower: " + terr;
        return AZ_ERR_HIP;
    }
    h->tower_calls++;
    h->tower_evals += LG;
    return AZ_OK;
}

static int step_finish(az_engine *h, const void *priors, const void *values, int f64);

extern "C" int az_search_step_finish(az_engine *h, const float *priors, const float *values) { return step_finish(h, priors, values, 0); }
extern "C" int az_search_step_finish_f64(az_engine *h, const double *priors, const double *values)
{
    if (!priors || !values) return AZ_ERR_ARG;
    return step_finish(h, priors, values, 1);
}

static int step_finish(az_engine *h, const void *priors, const void *values, int f64)
{
    if (!h) return AZ_ERR_ARG;
    HIPCHK(hipSetDevice(h->cfg.device));
    int rc = evaluate(h, priors, values, f64);
    if (rc) return rc;
    if (h->D.f64) hipLaunchKernelGGL(k_finish<double>, dim3(h->D.G), dim3(64), 0, h->stream, h->D, -1);
    else hipLaunchKernelGGL(k_finish<float>, dim3(h->D.G), dim3(64), 0, h->stream, h->D, -1);
    HIPCHK(hipGetLastError());
    h->steps++;
    return check_and_grow(h, false);
}

static int root_stats(az_engine *h, int32_t slot, int32_t side, int32_t *actions, int32_t *visits, void *total_value, void *prior,
                      int32_t *root_visits, int f64);

extern "C" int az_search_root_stats(az_engine *h, int32_t slot, int32_t side, int32_t *actions, int32_t *visits,
                                    float *total_value, float *prior, int32_t *root_visits)
{
    return root_stats(h, slot, side, actions, visits, total_value, prior, root_visits, 0);
}

extern "C" int az_search_root_stats_f64(az_engine *h, int32_t slot, int32_t side, int32_t *actions, int32_t *visits,
                                        double *total_value, double *prior, int32_t *root_visits)
{
    return root_stats(h, slot, side, actions, visits, total_value, prior, root_visits, 1);
}

static int root_stats(az_engine *h, int32_t slot, int32_t side, int32_t *actions, int32_t *visits, void *total_value, void *prior,
                      int32_t *root_visits, int f64)
{
    if (!h || slot < 0 || slot >= h->D.G || side < 0 || side > 1) return AZ_ERR_ARG;
    if ((total_value || prior) && f64 != h->D.f64)
        return set_err(h, AZ_ERR_ARG, h->D.f64 ? "engine searches in float64: use az_search_root_stats_f64" : "engine searches in float32: use az_search_root_stats");
    const size_t rs = h->D.f64 ? 8 : 4;
    HIPCHK(hipSetDevice(h->cfg.device));
    const Dev &D = h->D;
    const int t = 2 * slot + side;
    int32_t rb = 0, rn = 0;
    HIPCHK(hipMemcpyAsync(&rb, D.rootBlock + t, 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipMemcpyAsync(&rn, D.rootN + t, 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    if (root_visits) *root_visits = rn;
    if (rb < 0) return 0;
    int32_t start = 0, n = 0;
    HIPCHK(hipMemcpy(&start, D.bStart + (size_t)t * D.BC + rb, 4, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(&n, D.bCount + (size_t)t * D.BC + rb, 4, hipMemcpyDeviceToHost));
    if (n <= 0) return 0;
    // the children are contiguous in the tree's own numbering; every page of that range is copied from where the tree's
    // page table says it lies in the pool
    const int page = 1 << D.PS, pg0 = start >> D.PS, pg1 = (start + n - 1) >> D.PS;
    std::vector<int32_t> pages(pg1 - pg0 + 1);
    HIPCHK(hipMemcpy(pages.data(), D.pt + (size_t)t * D.PTC + pg0, sizeof(int32_t) * pages.size(), hipMemcpyDeviceToHost));
    std::vector<int16_t> a(actions ? n : 0);
    for (int pg = pg0; pg <= pg1; pg++) {
        const int v0 = std::max(start, pg * page), v1 = std::min(start + n, (pg + 1) * page);   // [v0, v1) of the tree
        const size_t e0 = ((size_t)pages[pg - pg0] << D.PS) + (size_t)(v0 - pg * page), o = (size_t)(v0 - start), k = (size_t)(v1 - v0);
        if (pages[pg - pg0] < 0 || pages[pg - pg0] >= D.pool_pages) return set_err(h, AZ_ERR_HIP, "tree page table holds a page outside the pool");
        if (visits) HIPCHK(hipMemcpy(visits + o, D.eN + e0, sizeof(int32_t) * k, hipMemcpyDeviceToHost));
        if (total_value) HIPCHK(hipMemcpy((char *)total_value + o * rs, (const char *)D.eW + e0 * rs, rs * k, hipMemcpyDeviceToHost));
        if (prior) HIPCHK(hipMemcpy((char *)prior + o * rs, (const char *)D.eP + e0 * rs, rs * k, hipMemcpyDeviceToHost));
        if (actions) HIPCHK(hipMemcpy(a.data() + o, D.eAct + e0, sizeof(int16_t) * k, hipMemcpyDeviceToHost));
    }
    for (int i = 0; i < (int)a.size(); i++) actions[i] = a[i];
    return n;
}

extern "C" int az_search_choose(az_engine *h, int32_t slot, int32_t side, double u, int32_t *move)
{
    if (!h || !move || slot < 0 || slot >= h->D.G || side < 0 || side > 1) return AZ_ERR_ARG;
    HIPCHK(hipSetDevice(h->cfg.device));
    hipLaunchKernelGGL(k_choose, dim3(1), dim3(64), 0, h->stream, h->D, slot, side, u, h->st_d);
    HIPCHK(hipMemcpyAsync(move, h->st_d, 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    if (*move == NOMOVE) return set_err(h, AZ_ERR_STATE, "root has no children");
    return AZ_OK;
}

static int advance(az_engine *h, const int32_t *slots, int n, const int32_t *moves, int side_sel)
{
    int rc = check_slots(h, slots, n);
    if (rc) return rc;
    if (n == 0) return AZ_OK;
    if (!moves) return set_err(h, AZ_ERR_ARG, "null moves");
    HIPCHK(hipSetDevice(h->cfg.device));
    if ((rc = up(h, h->st_a, slots, n)) || (rc = up(h, h->st_b, moves, n))) return rc;
    hipLaunchKernelGGL(k_set_pending, dim3((n + 63) / 64), dim3(64), 0, h->stream, h->D, (const int32_t *)h->st_a, n,
                       (const int32_t *)h->st_b, side_sel);
    if (h->D.f64) hipLaunchKernelGGL(k_reroot<double>, dim3(2 * h->D.G), dim3(64), 0, h->stream, h->D);
    else hipLaunchKernelGGL(k_reroot<float>, dim3(2 * h->D.G), dim3(64), 0, h->stream, h->D);
    HIPCHK(hipStreamSynchronize(h->stream));
    return AZ_OK;
}

extern "C" int az_search_advance_tree(az_engine *h, int32_t slot, int32_t side, int32_t move)
{
    if (!h || side < 0 || side > 1) return AZ_ERR_ARG;
    return advance(h, &slot, 1, &move, side);
}

extern "C" int az_search_advance(az_engine *h, const int32_t *slots, int32_t n, const int32_t *moves)
{
    return advance(h, slots, n, moves, -1);
}

extern "C" int az_search_set_c_punt(az_engine *h, double c_punt)
{
    if (!h) return AZ_ERR_ARG;
    if (!(c_punt > 0.0)) return set_err(h, AZ_ERR_ARG, "c_punt must be positive");
    h->D.c_punt = c_punt;   // the kernels take the engine's parameters by value with every launch
    return AZ_OK;
}

extern "C" int az_search_tree_size(az_engine *h, int32_t slot, int32_t side, int32_t *edges, int32_t *blocks)
{
    if (!h || slot < 0 || slot >= h->D.G || side < 0 || side > 1) return AZ_ERR_ARG;
    HIPCHK(hipSetDevice(h->cfg.device));
    HIPCHK(hipStreamSynchronize(h->stream));
    const int t = 2 * slot + side;
    if (edges) HIPCHK(hipMemcpy(edges, h->D.nEdges + t, 4, hipMemcpyDeviceToHost));
    if (blocks) HIPCHK(hipMemcpy(blocks, h->D.nBlocks + t, 4, hipMemcpyDeviceToHost));
    return AZ_OK;
}

// ---- self-play machine -----------------------------------------------------

extern "C" int az_selfplay_start(az_engine *h)
{
    if (!h) return AZ_ERR_ARG;
    HIPCHK(hipSetDevice(h->cfg.device));
    hipLaunchKernelGGL(k_reset, dim3(h->D.G), dim3(64), 0, h->stream, h->D, (const int32_t *)nullptr, h->D.G, 1);
    HIPCHK(hipStreamSynchronize(h->stream));
    h->started = true;
    return AZ_OK;
}

static void launch_tail(az_engine *h)
{
    if (h->D.f64) hipLaunchKernelGGL(k_finish<double>, dim3(h->D.G), dim3(64), 0, h->stream, h->D, -1);
    else hipLaunchKernelGGL(k_finish<float>, dim3(h->D.G), dim3(64), 0, h->stream, h->D, -1);
    hipLaunchKernelGGL(k_move, dim3(h->D.G), dim3(64), 0, h->stream, h->D);
    if (h->D.f64) hipLaunchKernelGGL(k_reroot<double>, dim3(2 * h->D.G), dim3(64), 0, h->stream, h->D);
    else hipLaunchKernelGGL(k_reroot<float>, dim3(2 * h->D.G), dim3(64), 0, h->stream, h->D);
}

// waits for the oldest queued round and folds its phase times into the counters
static int drain_round(az_engine *h)
{
    hipEvent_t *ev = h->ev[h->ev_head];
    HIPCHK(hipEventSynchronize(ev[3]));
    float a = 0, b = 0, c = 0;
    hipEventElapsedTime(&a, ev[0], ev[1]);
    hipEventElapsedTime(&b, ev[1], ev[2]);
    hipEventElapsedTime(&c, ev[2], ev[3]);
    h->select_ms += a; h->tower_ms += b; h->finish_ms += c;
    if (h->conv_profile || h->ev_prof[h->ev_head]) tower_collect_profile(h->tower);
    h->ev_prof[h->ev_head] = false;
    h->steps++;
    h->ev_head = (h->ev_head + 1) % AZ_EVQ;
    h->ev_pending--;
    return AZ_OK;
}

extern "C" int az_selfplay_run(az_engine *h, int32_t n_rounds)
{
    if (!h || n_rounds < 0) return AZ_ERR_ARG;
    if (!h->started) return set_err(h, AZ_ERR_STATE, "az_selfplay_start has not been called");
    if (!h->tower || !tower_ready(h->tower)) return set_err(h, AZ_ERR_NO_WEIGHTS, "no tower weights loaded");
    HIPCHK(hipSetDevice(h->cfg.device));
    // per-launch conv timing reuses one set of events per forward and the two-network split reads counts back every
    // round, so those modes keep the host in step with the device; otherwise rounds are queued AZ_EVQ deep
    const int depth = (h->conv_profile || (h->two_nets && !two_nets_async(h))) ? 1 : AZ_EVQ;
    for (int r = 0; r < n_rounds; r++) {
        if (h->ev_pending >= depth) { int rc = drain_round(h); if (rc) return rc; }
        const int q = (h->ev_head + h->ev_pending) % AZ_EVQ;
        hipEvent_t *ev = h->ev[q];
        // sampled per-launch timing: this round runs the full batch with events around its conv launches; the rounds
        // are at least AZ_EVQ apart, so the one set of events is never in flight twice (one network only)
        const bool sample = h->prof_every > 0 && !h->two_nets && (h->round_index % (uint64_t)h->prof_every) == 0;
        h->round_index++;
        HIPCHK(hipEventRecord(ev[0], h->stream));
        int rc = launch_select(h);
        if (rc) return rc;
        HIPCHK(hipEventRecord(ev[1], h->stream));
        if (sample) tower_set_profile(h->tower, true);
        rc = evaluate(h, nullptr, nullptr, 0, sample);
        if (sample) { tower_set_profile(h->tower, false); h->ev_prof[q] = true; }
        if (rc) return rc;
        HIPCHK(hipEventRecord(ev[2], h->stream));
        launch_tail(h);
        HIPCHK(hipEventRecord(ev[3], h->stream));
        h->ev_pending++;
        if (depth == 1) { if ((rc = drain_round(h))) return rc; }
    }
    while (h->ev_pending > 0) { int rc = drain_round(h); if (rc) return rc; }
    HIPCHK(hipGetLastError());
    return check_and_grow(h, true);
}

// n_rounds of select -> tower -> finish for the searches begun with az_search_begin, queued on the stream with one
// synchronisation at the end (the loop of az_search_step_select / az_search_step_finish without a host round trip per
// playout).  A slot whose search is complete ignores further rounds.
extern "C" int az_search_run(az_engine *h, int32_t n_rounds)
{
    if (!h || n_rounds < 0) return AZ_ERR_ARG;
    if (!h->tower || !tower_ready(h->tower)) return set_err(h, AZ_ERR_NO_WEIGHTS, "no tower weights loaded");
    HIPCHK(hipSetDevice(h->cfg.device));
    for (int r = 0; r < n_rounds; r++) {
        int rc = launch_select(h);
        if (rc) return rc;
        if ((rc = evaluate(h, nullptr, nullptr))) return rc;
        if (h->D.f64) hipLaunchKernelGGL(k_finish<double>, dim3(h->D.G), dim3(64), 0, h->stream, h->D, -1);
        else hipLaunchKernelGGL(k_finish<float>, dim3(h->D.G), dim3(64), 0, h->stream, h->D, -1);
        h->steps++;
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(h->stream));
    return check_and_grow(h, false);
}

extern "C" int az_selfplay_select(az_engine *h, int32_t *n_leaves)
{
    if (!h) return AZ_ERR_ARG;
    if (!h->started) return set_err(h, AZ_ERR_STATE, "az_selfplay_start has not been called");
    return az_search_step_select(h, n_leaves);
}

static int selfplay_finish(az_engine *h, const void *priors, const void *values, int f64);

extern "C" int az_selfplay_finish(az_engine *h, const float *priors, const float *values) { return selfplay_finish(h, priors, values, 0); }
extern "C" int az_selfplay_finish_f64(az_engine *h, const double *priors, const double *values)
{
    if (!priors || !values) return AZ_ERR_ARG;
    return selfplay_finish(h, priors, values, 1);
}

static int selfplay_finish(az_engine *h, const void *priors, const void *values, int f64)
{
    if (!h) return AZ_ERR_ARG;
    HIPCHK(hipSetDevice(h->cfg.device));
    int rc = evaluate(h, priors, values, f64);
    if (rc) return rc;
    launch_tail(h);
    HIPCHK(hipGetLastError());
    h->steps++;
    return check_and_grow(h, true);
}

extern "C" int az_selfplay_status(az_engine *h, int32_t *plies, int32_t *done)
{
    if (!h) return AZ_ERR_ARG;
    HIPCHK(hipSetDevice(h->cfg.device));
    const int G = h->D.G;
    std::vector<int32_t> ph(G);
    if (plies) HIPCHK(hipMemcpyAsync(plies, h->D.ply, sizeof(int32_t) * G, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipMemcpyAsync(ph.data(), h->D.phase, sizeof(int32_t) * G, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    if (done) for (int g = 0; g < G; g++) done[g] = ph[g] == PH_GAME_DONE;
    return AZ_OK;
}

extern "C" int az_trajectory_max_plies(az_engine *h) { return h ? h->D.traj_plies : AZ_ERR_ARG; }

extern "C" int az_trajectory_pop(az_engine *h, int32_t slot, int16_t *moves, int32_t *visits, int32_t *winner, int32_t restart)
{
    if (!h || slot < 0 || slot >= h->D.G) return AZ_ERR_ARG;
    HIPCHK(hipSetDevice(h->cfg.device));
    const Dev &D = h->D;
    int32_t ph = 0, ply = 0, win = 0;
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipMemcpy(&ph, D.phase + slot, 4, hipMemcpyDeviceToHost));
    if (ph != PH_GAME_DONE) return set_err(h, AZ_ERR_STATE, "game in this slot is not finished");
    HIPCHK(hipMemcpy(&ply, D.ply + slot, 4, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(&win, D.tr_winner + slot, 4, hipMemcpyDeviceToHost));
    int np = std::min(ply, D.traj_plies);
    if (D.record) {
        if (moves) HIPCHK(hipMemcpy(moves, D.tr_moves + (size_t)slot * D.traj_plies, sizeof(int16_t) * np, hipMemcpyDeviceToHost));
        if (visits)
            HIPCHK(hipMemcpy(visits, D.tr_visits + (size_t)slot * D.traj_plies * D.geo.O, sizeof(int32_t) * (size_t)np * D.geo.O,
                             hipMemcpyDeviceToHost));
    }
    if (winner) *winner = win;
    if (restart) {
        int32_t s = slot;
        int rc = up(h, h->st_a, &s, 1);
        if (rc) return rc;
        hipLaunchKernelGGL(k_reset, dim3(1), dim3(64), 0, h->stream, h->D, (const int32_t *)h->st_a, 1, 1);
        HIPCHK(hipStreamSynchronize(h->stream));
    }
    return np;
}

extern "C" int az_trajectory_pop_many(az_engine *h, const int32_t *slots, int32_t n, const int32_t *row_start, int32_t total_rows,
                                      int16_t *moves, int32_t *visits, int32_t *winners, int32_t restart)
{
    int rc = check_slots(h, slots, n);
    if (rc) return rc;
    if (n == 0) return AZ_OK;
    if (!row_start || !moves || !visits || !winners || total_rows < 0) return set_err(h, AZ_ERR_ARG, "null argument");
    const Dev &D = h->D;
    if (!D.record) return set_err(h, AZ_ERR_STATE, "the engine was created without record_trajectories");
    HIPCHK(hipSetDevice(h->cfg.device));
    const int G = D.G;
    std::vector<int32_t> ph(G), ply(G), win(G);
    HIPCHK(hipMemcpyAsync(ph.data(), D.phase, sizeof(int32_t) * G, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipMemcpyAsync(ply.data(), D.ply, sizeof(int32_t) * G, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipMemcpyAsync(win.data(), D.tr_winner, sizeof(int32_t) * G, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    long long expect = 0;
    std::vector<char> seen(G, 0);
    for (int i = 0; i < n; i++) {
        const int s = slots[i];
        if (seen[s]) return set_err(h, AZ_ERR_ARG, "duplicate slot");
        seen[s] = 1;
        if (ph[s] != PH_GAME_DONE) return set_err(h, AZ_ERR_STATE, "a game in the list is not finished");
        if (row_start[i] != expect) return set_err(h, AZ_ERR_ARG, "row_start is not the running sum of the games' plies");
        expect += std::min(ply[s], D.traj_plies);
    }
    if (expect != total_rows) return set_err(h, AZ_ERR_ARG, "total_rows does not match the games' plies");
    if ((size_t)total_rows > h->pop_rows) {
        if (h->pop_moves) hipFree(h->pop_moves);
        if (h->pop_visits) hipFree(h->pop_visits);
        h->pop_moves = nullptr; h->pop_visits = nullptr; h->pop_rows = 0;
        const size_t rows = (size_t)total_rows + (size_t)total_rows / 2 + 64;
        HIPCHK(hipMalloc((void **)&h->pop_moves, rows * sizeof(int16_t)));
        HIPCHK(hipMalloc((void **)&h->pop_visits, rows * D.geo.O * sizeof(int32_t)));
        h->pop_rows = rows;
    }
    if ((rc = up(h, h->st_a, slots, n)) || (rc = up(h, h->st_b, row_start, n))) return rc;
    hipLaunchKernelGGL(k_traj_gather, dim3(n), dim3(256), 0, h->stream, h->D, (const int32_t *)h->st_a, (const int32_t *)h->st_b, n,
                       h->pop_moves, h->pop_visits);
    if (total_rows > 0) {
        HIPCHK(hipMemcpyAsync(moves, h->pop_moves, sizeof(int16_t) * (size_t)total_rows, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipMemcpyAsync(visits, h->pop_visits, sizeof(int32_t) * (size_t)total_rows * D.geo.O, hipMemcpyDeviceToHost, h->stream));
    }
    if (restart) hipLaunchKernelGGL(k_reset, dim3(n), dim3(64), 0, h->stream, h->D, (const int32_t *)h->st_a, n, 1);
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipGetLastError());
    for (int i = 0; i < n; i++) winners[i] = win[slots[i]];
    return AZ_OK;
}

extern "C" int az_trajectory_pop_many_dev(az_engine *h, const int32_t *slots, int32_t n, const int32_t *row_start, int32_t total_rows,
                                          int16_t *moves_dev, int32_t *visits_dev, int32_t *winners, int32_t restart)
{
    int rc = check_slots(h, slots, n);
    if (rc) return rc;
    if (n == 0) return AZ_OK;
    if (!row_start || !winners || total_rows < 0 || (total_rows > 0 && (!moves_dev || !visits_dev))) return set_err(h, AZ_ERR_ARG, "null argument");
    const Dev &D = h->D;
    if (!D.record) return set_err(h, AZ_ERR_STATE, "the engine was created without record_trajectories");
    HIPCHK(hipSetDevice(h->cfg.device));
    const int G = D.G;
    std::vector<int32_t> ph(G), ply(G), win(G);
    HIPCHK(hipMemcpyAsync(ph.data(), D.phase, sizeof(int32_t) * G, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipMemcpyAsync(ply.data(), D.ply, sizeof(int32_t) * G, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipMemcpyAsync(win.data(), D.tr_winner, sizeof(int32_t) * G, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    long long expect = 0;
    std::vector<char> seen(G, 0);
    for (int i = 0; i < n; i++) {
        const int s = slots[i];
        if (seen[s]) return set_err(h, AZ_ERR_ARG, "duplicate slot");
        seen[s] = 1;
        if (ph[s] != PH_GAME_DONE) return set_err(h, AZ_ERR_STATE, "a game in the list is not finished");
        if (row_start[i] != expect) return set_err(h, AZ_ERR_ARG, "row_start is not the running sum of the games' plies");
        expect += std::min(ply[s], D.traj_plies);
    }
    if (expect != total_rows) return set_err(h, AZ_ERR_ARG, "total_rows does not match the games' plies");
    if ((rc = up(h, h->st_a, slots, n)) || (rc = up(h, h->st_b, row_start, n))) return rc;
    hipLaunchKernelGGL(k_traj_gather, dim3(n), dim3(256), 0, h->stream, h->D, (const int32_t *)h->st_a, (const int32_t *)h->st_b, n,
                       moves_dev, visits_dev);
    if (restart) hipLaunchKernelGGL(k_reset, dim3(n), dim3(64), 0, h->stream, h->D, (const int32_t *)h->st_a, n, 1);
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipGetLastError());
    for (int i = 0; i < n; i++) winners[i] = win[slots[i]];
    return AZ_OK;
}

// phase, ply and winner of the listed slots -> meta[0..n), meta[256..), meta[512..)
__global__ void k_traj_meta(Dev D, const int32_t *__restrict__ slots, int n, int32_t *__restrict__ meta)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int g = slots[i];
    meta[i] = D.phase[g];
    meta[AZ_STATIC_POP_SLOTS + i] = D.ply[g];
    meta[2 * AZ_STATIC_POP_SLOTS + i] = D.tr_winner[g];
}

extern "C" int az_trajectory_pop_many_static(az_engine *h, const int32_t *slots, int32_t n, const int32_t *row_start, int32_t total_rows,
                                             int16_t *moves, int32_t *visits, int32_t *winners, int32_t restart)
{
    int rc = check_slots(h, slots, n);
    if (rc) return rc;
    if (n == 0) return AZ_OK;
    if (!row_start || !moves || !visits || !winners || total_rows < 0) return set_err(h, AZ_ERR_ARG, "null argument");
    const Dev &D = h->D;
    if (!D.record) return set_err(h, AZ_ERR_STATE, "the engine was created without record_trajectories");
    if (n > AZ_STATIC_POP_SLOTS) return set_err(h, AZ_ERR_ARG, "az_trajectory_pop_many_static: more than AZ_STATIC_POP_SLOTS (256) slots in one call");
    const size_t O = D.geo.O;
    if ((size_t)total_rows * (2 + 4 * O) > (size_t)AZ_STATIC_POP_BYTES - 16384)
        return set_err(h, AZ_ERR_ARG, "az_trajectory_pop_many_static: the rows exceed the fixed staging array (total_rows * (2 + 4 * O) > AZ_STATIC_POP_BYTES - 16384)");
    if (h->cfg.device < 0 || h->cfg.device >= AZ_MAX_DEVICES || !g_scratch_ptr[h->cfg.device])
        return set_err(h, AZ_ERR_STATE, "no staging array on this device");
    HIPCHK(hipSetDevice(h->cfg.device));
    unsigned char *const sp = g_scratch_ptr[h->cfg.device];   // this GPU's copy of `scratch`
    int32_t *const d_slots = reinterpret_cast<int32_t *>(sp + SP_SLOTS);
    int32_t *const d_starts = reinterpret_cast<int32_t *>(sp + SP_STARTS);
    int32_t *const d_meta = reinterpret_cast<int32_t *>(sp + SP_META);
    int16_t *const d_moves = reinterpret_cast<int16_t *>(sp + SP_MOVES);
    int32_t *const d_visits = reinterpret_cast<int32_t *>(sp + SP_MOVES + (((size_t)total_rows * 2 + 255) & ~(size_t)255));
    HIPCHK(hipMemcpyAsync(d_slots, slots, sizeof(int32_t) * n, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(d_starts, row_start, sizeof(int32_t) * n, hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(k_traj_meta, dim3((n + 255) / 256), dim3(256), 0, h->stream, h->D, (const int32_t *)d_slots, n, d_meta);
    int32_t meta[3 * AZ_STATIC_POP_SLOTS];
    HIPCHK(hipMemcpyAsync(meta, d_meta, sizeof meta, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    long long expect = 0;
    for (int i = 0; i < n; i++) {
        for (int j = 0; j < i; j++)
            if (slots[j] == slots[i]) return set_err(h, AZ_ERR_ARG, "duplicate slot");
        if (meta[i] != PH_GAME_DONE) return set_err(h, AZ_ERR_STATE, "a game in the list is not finished");
        if (row_start[i] != expect) return set_err(h, AZ_ERR_ARG, "row_start is not the running sum of the games' plies");
        expect += std::min(meta[AZ_STATIC_POP_SLOTS + i], D.traj_plies);
    }
    if (expect != total_rows) return set_err(h, AZ_ERR_ARG, "total_rows does not match the games' plies");
    hipLaunchKernelGGL(k_traj_gather, dim3(n), dim3(256), 0, h->stream, h->D, (const int32_t *)d_slots, (const int32_t *)d_starts, n,
                       d_moves, d_visits);
    if (total_rows > 0) {
        HIPCHK(hipMemcpyAsync(moves, d_moves, sizeof(int16_t) * (size_t)total_rows, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipMemcpyAsync(visits, d_visits, sizeof(int32_t) * (size_t)total_rows * O, hipMemcpyDeviceToHost, h->stream));
    }
    if (restart) hipLaunchKernelGGL(k_reset, dim3(n), dim3(64), 0, h->stream, h->D, (const int32_t *)d_slots, n, 1);
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipGetLastError());
    for (int i = 0; i < n; i++) winners[i] = meta[2 * AZ_STATIC_POP_SLOTS + i];
    return AZ_OK;
}

// ---- RNG replay streams ------------------------------------------------------

extern "C" int az_rng_inject(az_engine *h, int32_t slot, const int32_t *tids, int64_t n_tids, const double *etas,
                             int64_t n_etas, const double *unis, int64_t n_unis)
{
    if (!h || slot < 0 || slot >= h->D.G || n_tids < 0 || n_etas < 0 || n_unis < 0) return AZ_ERR_ARG;
    HIPCHK(hipSetDevice(h->cfg.device));
    HIPCHK(hipStreamSynchronize(h->stream));
    const void *src[3] = {tids, etas, unis};
    const size_t bytes[3] = {sizeof(int32_t) * (size_t)n_tids, sizeof(double) * (size_t)n_etas, sizeof(double) * (size_t)n_unis};
    void *dev[3] = {nullptr, nullptr, nullptr};
    for (int k = 0; k < 3; k++) {
        void *&old = h->rng_bufs[(size_t)3 * slot + k];
        if (old) { hipFree(old); old = nullptr; }
        if (bytes[k]) {
            HIPCHK(hipMalloc(&dev[k], bytes[k]));
            old = dev[k];
            HIPCHK(hipMemcpy(dev[k], src[k], bytes[k], hipMemcpyHostToDevice));
        }
    }
    long long zero = 0, n0 = n_tids, n1 = n_etas, n2 = n_unis;
    HIPCHK(hipMemcpy(h->D.rt + slot, &dev[0], sizeof(void *), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(h->D.re + slot, &dev[1], sizeof(void *), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(h->D.ru + slot, &dev[2], sizeof(void *), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(h->D.rt_n + slot, &n0, 8, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(h->D.re_n + slot, &n1, 8, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(h->D.ru_n + slot, &n2, 8, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(h->D.rt_i + slot, &zero, 8, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(h->D.re_i + slot, &zero, 8, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(h->D.ru_i + slot, &zero, 8, hipMemcpyHostToDevice));
    return AZ_OK;
}

extern "C" int az_rng_cursor(az_engine *h, int32_t slot, int64_t *i_tids, int64_t *i_etas, int64_t *i_unis)
{
    if (!h || slot < 0 || slot >= h->D.G) return AZ_ERR_ARG;
    HIPCHK(hipSetDevice(h->cfg.device));
    HIPCHK(hipStreamSynchronize(h->stream));
    long long v[3];
    HIPCHK(hipMemcpy(&v[0], h->D.rt_i + slot, 8, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(&v[1], h->D.re_i + slot, 8, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(&v[2], h->D.ru_i + slot, 8, hipMemcpyDeviceToHost));
    if (i_tids) *i_tids = v[0];
    if (i_etas) *i_etas = v[1];
    if (i_unis) *i_unis = v[2];
    return AZ_OK;
}

// ---- tower ---------------------------------------------------------------------

extern "C" size_t az_tower_blob_floats(az_engine *h) { return (h && h->tower) ? tower_blob_floats(h->tower) : 0; }

extern "C" int az_tower_load_weights(az_engine *h, const float *blob, size_t n)
{
    if (!h || !blob) return AZ_ERR_ARG;
    if (!h->tower) return set_err(h, AZ_ERR_ARG, "engine was created with num_blocks = 0 (no tower)");
    HIPCHK(hipSetDevice(h->cfg.device));
    std::string terr;
    if (tower_load(h->tower, blob, n, h->stream, terr) != 0) { h->error = "tower: " + terr; return AZ_ERR_ARG; }
    return AZ_OK;
}

extern "C" int az_tower_load_weights_net(az_engine *h, int32_t net, const float *blob, size_t n)
{
    if (!h || !blob || net < 0 || net > 1) return AZ_ERR_ARG;
    if (net == 0) return az_tower_load_weights(h, blob, n);
    if (!h->tower) return set_err(h, AZ_ERR_ARG, "engine was created with num_blocks = 0 (no tower)");
    HIPCHK(hipSetDevice(h->cfg.device));
    std::string terr;
    if (!h->tower2) {
        const az_config &c = h->cfg;
        h->tower2 = tower_create(c.board_height, c.board_width, h->D.geo.O, c.num_blocks, c.channels > 0 ? c.channels : 256, c.tower_dtype,
                                 h->D.G, tower_options(c), terr);
        if (!h->tower2) { h->error = "tower: " + terr; return AZ_ERR_HIP; }
    }
    if (tower_load(h->tower2, blob, n, h->stream, terr) != 0) { h->error = "tower: " + terr; return AZ_ERR_ARG; }
    return AZ_OK;
}

extern "C" int az_tower_load_weights_dev(az_engine *h, int32_t net, const float *blob_dev, size_t n)
{
    if (!h || !blob_dev || net < 0 || net > 1) return AZ_ERR_ARG;
    if (!h->tower) return set_err(h, AZ_ERR_ARG, "engine was created with num_blocks = 0 (no tower)");
    HIPCHK(hipSetDevice(h->cfg.device));
    std::string terr;
    if (net == 1 && !h->tower2) {
        const az_config &c = h->cfg;
        h->tower2 = tower_create(c.board_height, c.board_width, h->D.geo.O, c.num_blocks, c.channels > 0 ? c.channels : 256, c.tower_dtype,
                                 h->D.G, tower_options(c), terr);
        if (!h->tower2) { h->error = "tower: " + terr; return AZ_ERR_HIP; }
    }
    if (tower_load_dev(net ? h->tower2 : h->tower, blob_dev, n, h->stream, terr) != 0) { h->error = "tower: " + terr; return AZ_ERR_ARG; }
    return AZ_OK;
}

extern "C" int az_match_assign(az_engine *h, const int32_t *slots, int32_t n, const int32_t *net_of_first)
{
    if (!h || n < 0) return AZ_ERR_ARG;
    HIPCHK(hipSetDevice(h->cfg.device));
    HIPCHK(hipStreamSynchronize(h->stream));
    if (n == 0) { h->two_nets = false; return AZ_OK; }
    int rc = check_slots(h, slots, n);
    if (rc) return rc;
    if (!net_of_first) return set_err(h, AZ_ERR_ARG, "null net_of_first");
    for (int i = 0; i < n; i++) {
        if (net_of_first[i] < 0 || net_of_first[i] > 1) return set_err(h, AZ_ERR_ARG, "network index must be 0 or 1");
        HIPCHK(hipMemcpy(h->net_first + slots[i], net_of_first + i, sizeof(int32_t), hipMemcpyHostToDevice));
    }
    h->two_nets = true;
    return AZ_OK;
}

extern "C" int az_match_leaf_nets(az_engine *h, int32_t *net_of_slot)
{
    if (!h || !net_of_slot) return AZ_ERR_ARG;
    if (!h->two_nets) return set_err(h, AZ_ERR_STATE, "no match assignment is in force (az_match_assign)");
    if (h->D.K > 1) return set_err(h, AZ_ERR_ARG, "two-network matches run one leaf per tree per round (leaves_per_round = 1)");
    HIPCHK(hipSetDevice(h->cfg.device));
    const int G = h->D.G;
    // the same split the tower path launches in evaluate(); the running totals go to a spare pair of words
    hipLaunchKernelGGL(k_partition, dim3(1), dim3(1024), 0, h->stream, h->D, (const int32_t *)h->net_first, h->net_rows, h->net_cnt,
                       h->net_tot_spare);
    std::vector<int32_t> rows(2 * (size_t)G);
    int32_t cnt[2] = {0, 0};
    HIPCHK(hipMemcpyAsync(rows.data(), h->net_rows, sizeof(int32_t) * rows.size(), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipMemcpyAsync(cnt, h->net_cnt, sizeof cnt, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    for (int g = 0; g < G; g++) net_of_slot[g] = -1;
    for (int net = 0; net < 2; net++) {
        if (cnt[net] < 0 || cnt[net] > G) return set_err(h, AZ_ERR_HIP, "leaf partition returned a bad count");
        for (int i = 0; i < cnt[net]; i++) {
            const int g = rows[(size_t)net * G + i];
            if (g < 0 || g >= G || net_of_slot[g] != -1) return set_err(h, AZ_ERR_HIP, "leaf partition lists a slot twice or out of range");
            net_of_slot[g] = net;
        }
    }
    return AZ_OK;
}

extern "C" int az_tower_forward(az_engine *h, const int8_t *planes, int32_t B, float *out_p, float *out_v)
{
    if (!h || !planes || !out_p || !out_v || B < 1) return AZ_ERR_ARG;
    if (!h->tower || !tower_ready(h->tower)) return set_err(h, AZ_ERR_NO_WEIGHTS, "no tower weights loaded");
    HIPCHK(hipSetDevice(h->cfg.device));
    const int G = h->D.G, O = h->D.geo.O, NC = h->D.geo.NC;
    std::string terr;
    for (int b0 = 0; b0 < B; b0 += G) {
        int nb = std::min(G, B - b0);
        HIPCHK(hipMemcpyAsync(h->D.planes, planes + (size_t)b0 * 17 * NC, (size_t)nb * 17 * NC, hipMemcpyHostToDevice, h->stream));
        if (tower_forward(h->tower, h->D.planes, nb, (float *)h->D.priors, (float *)h->D.values, h->stream, terr) != 0) {
            h->error = "tower: " + terr;
            return AZ_ERR_HIP;
        }
        HIPCHK(hipMemcpyAsync(out_p + (size_t)b0 * O, h->D.priors, sizeof(float) * (size_t)nb * O, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipMemcpyAsync(out_v + b0, h->D.values, sizeof(float) * nb, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
    }
    return AZ_OK;
}

extern "C" int az_tower_forward_dev(az_engine *h, const int8_t *planes_dev, int32_t B, float *out_p_dev, float *out_v_dev)
{
    if (!h || !planes_dev || !out_p_dev || !out_v_dev || B < 1) return AZ_ERR_ARG;
    if (!h->tower || !tower_ready(h->tower)) return set_err(h, AZ_ERR_NO_WEIGHTS, "no tower weights loaded");
    HIPCHK(hipSetDevice(h->cfg.device));
    const int G = h->D.G, O = h->D.geo.O, NC = h->D.geo.NC;
    std::string terr;
    for (int b0 = 0; b0 < B; b0 += G) {  // the activation buffers hold n_slots boards
        const int nb = std::min(G, B - b0);
        if (tower_forward(h->tower, planes_dev + (size_t)b0 * 17 * NC, nb, out_p_dev + (size_t)b0 * O, out_v_dev + b0, h->stream, terr) != 0) {
            h->error = "tower: " + terr;
            return AZ_ERR_HIP;
        }
    }
    HIPCHK(hipStreamSynchronize(h->stream));
    return AZ_OK;
}

extern "C" int az_tower_profile(az_engine *h, int32_t on)
{
    if (!h || !h->tower) return AZ_ERR_ARG;
    HIPCHK(hipSetDevice(h->cfg.device));
    HIPCHK(hipStreamSynchronize(h->stream));
    if (on < 0 || (on > 1 && on < AZ_EVQ)) return set_err(h, AZ_ERR_ARG, "az_tower_profile: 0, 1, or a sampling period of at least 8 rounds");
    tower_set_profile(h->tower, on == 1);
    h->conv_profile = on == 1;
    h->prof_every = on > 1 ? on : 0;
    return AZ_OK;
}

extern "C" int az_tower_leaf_compaction(az_engine *h, int32_t on)
{
    if (!h) return AZ_ERR_ARG;
    HIPCHK(hipSetDevice(h->cfg.device));
    HIPCHK(hipStreamSynchronize(h->stream));
    h->compact = on != 0;
    return AZ_OK;
}

extern "C" const char *az_tower_conv_kernel(az_engine *h)
{
    if (!h || !h->tower) return "";
    const int LG = h->D.K * h->D.G;
    return tower_conv_kernel_name(h->tower, LG, tower_takes_device_count(h->tower, LG));
}

extern "C" double az_tower_flops_per_eval(az_engine *h) { return (h && h->tower) ? tower_flops_per_eval(h->tower) : 0.0; }

// ---- measurement -----------------------------------------------------------------

extern "C" int az_counters_get(az_engine *h, az_counters *out)
{
    if (!h || !out) return AZ_ERR_ARG;
    HIPCHK(hipSetDevice(h->cfg.device));
    unsigned long long c[C_N];
    HIPCHK(hipStreamSynchronize(h->stream));
    {
        std::vector<unsigned long long> all((size_t)AZ_CTR_BUCKETS * C_N);
        HIPCHK(hipMemcpy(all.data(), h->D.ctr, all.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        for (int i = 0; i < C_N; i++) c[i] = 0;
        for (int b = 0; b < AZ_CTR_BUCKETS; b++)
            for (int i = 0; i < C_N; i++) c[i] += all[(size_t)b * C_N + i];
    }
    out->steps = h->steps; out->evals = c[C_EVALS]; out->moves = c[C_MOVES]; out->games_finished = c[C_GAMES];
    out->levels = c[C_LEVELS]; out->children_scanned = c[C_CHILDREN]; out->edges_created = c[C_EDGES];
    out->rule_moves = c[C_RULEMOVES]; out->expansions_skipped = c[C_SKIPPED];
    out->tower_ms = h->tower_ms; out->select_ms = h->select_ms; out->finish_ms = h->finish_ms;
    out->tower_calls = h->tower_calls; out->tower_evals = h->tower_evals;
    {
        unsigned long long tot[2];
        HIPCHK(hipMemcpy(tot, h->net_tot, sizeof tot, hipMemcpyDeviceToHost));
        out->tower_evals += tot[0] + tot[1];
    }
    {
        int32_t pk[AZ_CTR_BUCKETS * 2];
        HIPCHK(hipMemcpy(pk, h->D.peak, sizeof pk, hipMemcpyDeviceToHost));
        int32_t pe = 0, pb = 0;
        for (int b = 0; b < AZ_CTR_BUCKETS; b++) { pe = std::max(pe, pk[2 * b]); pb = std::max(pb, pk[2 * b + 1]); }
        out->peak_tree_edges = pe; out->peak_tree_blocks = pb;
        out->tree_edge_cap = h->D.EC; out->tree_block_cap = h->D.BC;
        int32_t top[2];
        HIPCHK(hipMemcpy(top, h->D.pool_top, sizeof top, hipMemcpyDeviceToHost));
        out->tree_page_edges = 1ull << h->D.PS; out->tree_pool_pages = h->D.pool_pages;
        out->tree_pool_pages_in_use = h->D.pool_pages - top[0]; out->tree_pool_pages_peak = h->D.pool_pages - std::min(top[0], top[1]);
    }
    out->conv_ms = 0; out->conv_launches = 0; out->conv_flops_per_launch = 0;
    if (h->tower) {
        unsigned long long n = 0;
        double flops = 0;
        tower_profile_read(h->tower, &out->conv_ms, &n, &flops, false);
        out->conv_launches = n;
        out->conv_flops_per_launch = n ? flops / (double)n : 0.0;
    }
    return AZ_OK;
}

extern "C" int az_counters_reset(az_engine *h)
{
    if (!h) return AZ_ERR_ARG;
    HIPCHK(hipSetDevice(h->cfg.device));
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipMemset(h->D.ctr, 0, sizeof(unsigned long long) * AZ_CTR_BUCKETS * (C_N + 1)));
    h->tower_ms = h->select_ms = h->finish_ms = 0;
    h->tower_calls = h->tower_evals = h->steps = 0;
    HIPCHK(hipMemset(h->net_tot, 0, 2 * sizeof(unsigned long long)));
    HIPCHK(hipMemset(h->D.peak, 0, sizeof(int32_t) * AZ_CTR_BUCKETS * 2));
    HIPCHK(hipMemcpy(h->D.pool_top + 1, h->D.pool_top, sizeof(int32_t), hipMemcpyDeviceToDevice));   // peak restarts at the present use
    if (h->tower) { double a, c; unsigned long long b; tower_profile_read(h->tower, &a, &b, &c, true); }
    return AZ_OK;
}

extern "C" int az_sync(az_engine *h)
{
    if (!h) return AZ_ERR_ARG;
    HIPCHK(hipSetDevice(h->cfg.device));
    HIPCHK(hipStreamSynchronize(h->stream));
    return AZ_OK;
}

#ifdef AZ_STAMP_SELECT
// development build only (tools/select_stamps.py): accumulated phase cycles of k_select's block 0; reading resets them
extern "C" int az_debug_select_stamps(az_engine *h, unsigned long long *out16)
{
    if (!h || !out16) return AZ_ERR_ARG;
    HIPCHK(hipSetDevice(h->cfg.device));
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_sel_stamps), 16 * sizeof(unsigned long long)));
    unsigned long long z[16] = {0};
    HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(g_sel_stamps), z, sizeof z));
    return AZ_OK;
}
#endif
