// Dual-head residual tower, inference only (AlphaZero/network/model.py:39-88 with
// batch-norm folded into the convolutions: network/util.py:18-21, center only).
//
// Activations are NHWC with a one-pixel zero halo per board: [B][(H+2)*(W+2)][C].
// A 3x3 SAME convolution is then an implicit GEMM  D[m][co] = sum_{tap,ci} X[pix(m)+off(tap)][ci] * Wt[co][tap][ci]
// with no boundary tests in the inner loop; the halo is never written.
// bf16 path: MFMA 32x32x16 (fp32 accumulate), fused bias + residual + ReLU epilogue.
// fp32 path: same data flow on the vector ALU, used for the 1e-5 parity mode.
#include "az_tower.h"

#include <hip/hip_bf16.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

#include "az_conv_kernels.h"

// up to this many board points per forward the small-batch convolution is used (16 boards of 19x19)
#ifndef SMALL_BATCH_POINTS
#define SMALL_BATCH_POINTS 4332   // 12 boards of 19x19: above it the channel-split slab kernel is faster
#endif
#ifndef GRAPH_MAX_POINTS
#define GRAPH_MAX_POINTS 46208    // forwards up to 128 boards of 19x19 are replayed as captured graphs
#endif
#ifndef GROUP_ROUNDS
#define GROUP_ROUNDS 2  // rounds of convolution tiles per group of boards (see tower_group_boards)
#endif
#define MAX_GROUPS 64
#ifndef DEVICE_COUNT_MIN_POINTS
#define DEVICE_COUNT_MIN_POINTS 23104  // batches from here on may take their size from device memory
#endif

struct Tower {
    int H, W, HW, PP, O, NB, C, dtype, maxB;
    size_t blob_floats;
    bool ready;
    std::vector<void *> allocs;
    void *w_stem, *w_conv;          // [C][9][CIN0], [2*NB][C][9][C] in the compute dtype
    float *b_stem, *b_conv;         // [C], [2*NB][C]
    float *w_head, *b_head;         // [3][C], [3]
    float *w_pfc, *b_pfc;           // [2*HW][O], [O]
    float *w_v1, *b_v1, *w_v2, *b_v2;  // [HW][256], [256], [256], [1]
    void *X0, *X, *T, *Y;           // activations
    float *hp, *hv;                 // [maxB][2*HW], [maxB][HW]
    // per-launch timing of the residual-tower convolutions (HIP events on the launch stream)
    bool profile;
    std::vector<hipEvent_t> evs;    // 2 per conv launch of one forward
    int evs_used;
    double conv_ms;
    unsigned long long conv_launches;
    // small batches (a game searched alone) are launch-bound: their forward, a fixed chain of about 45 short kernels,
    // is captured once per (batch size, buffers, stream) and replayed as one graph launch
    struct Graph { const void *planes; const void *priors, *values, *rows; int B; hipStream_t st; hipGraphExec_t exec; };
    std::vector<Graph> graphs;
    bool use_graphs;
    bool use_duo;   // two-workgroups-per-CU conv kernel where it applies (off: A/B timing only)
    // large batches go through the tower in groups of boards (see forward_t): group size in boards (0 = automatic,
    // -1 = never), per-group device-side counts, boards of every profiled launch
    int group_boards, cus;
    int32_t *gcnt;
    std::vector<int> ev_boards;
    double conv_flops;
};

// dtype: 0 float32, 1 bfloat16, 2 IEEE half precision; the two 16-bit forms share every code path (format template F)
static inline bool is16(int dtype) { return dtype != 0; }

static inline bf16_t h_f2h16(float f)   // float -> IEEE half, round to nearest even (the compiler's conversion)
{
    const _Float16 h = (_Float16)f;
    bf16_t u;
    memcpy(&u, &h, 2);
    return u;
}

static inline bf16_t h_f2bf(float f)
{
    unsigned int u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (bf16_t)(u >> 16);
}

// ---------------------------------------------------------------------------
// heads.  1x1 convolutions 256->2 (policy) and 256->1 (value) with folded BN + ReLU.
// Output in NCHW-flatten order (model.py:71-72 reshapes an NCHW tensor), i.e. hp[b][c*HW + pix].
// HBM-bound: the last activation tensor is read once with 16-byte loads; LPP lanes share a board point
// (8 bf16 / 4 fp32 channels per lane per load), partial sums are folded with xor-shuffles.
// ---------------------------------------------------------------------------
template <typename T, bool ONEPASS, int F = 0>
__global__ __launch_bounds__(256) void k_head_conv(const T *__restrict__ X, const float *__restrict__ wh, const float *__restrict__ bh,
                                                   float *__restrict__ hp, float *__restrict__ hv, int HW, int Wd, int PP, int C, int LPP,
                                                   const int32_t *__restrict__ cnt)
{
    if (cnt && (int)blockIdx.x >= *cnt) return;
    constexpr int EPL = 16 / (int)sizeof(T);  // channels per 16-byte load
    const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int W2 = Wd + 2;
    const int PPW = 64 / LPP;  // LPP lanes per point (a power of two), PPW points per wave pass
    const int sub = lane / LPP, cl = lane % LPP;
    const float b0 = bh[0], b1 = bh[1], b2 = bh[2];
    float wreg[3][EPL];  // ONEPASS (C == LPP * EPL): this lane's weights stay in registers
    if (ONEPASS) {
#pragma unroll
        for (int k = 0; k < EPL; k++) { wreg[0][k] = wh[cl * EPL + k]; wreg[1][k] = wh[C + cl * EPL + k]; wreg[2][k] = wh[2 * C + cl * EPL + k]; }
    }
    const int npass = (HW + 4 * PPW - 1) / (4 * PPW);
    for (int it = blockIdx.y; it < npass; it += gridDim.y) {
        const int pix = (it * 4 + wave) * PPW + sub;
        const bool ok = pix < HW;
        const int pc = ok ? pix : HW - 1;
        const int y = pc / Wd, x = pc - y * Wd;
        const T *row = X + ((size_t)b * PP + (y + 1) * W2 + (x + 1)) * C;
        float s0 = 0, s1 = 0, s2 = 0;
        for (int c0 = cl * EPL; c0 < C; c0 += LPP * EPL) {
            float v[EPL];
            if constexpr (sizeof(T) == 2) {
                const uint4 u = *reinterpret_cast<const uint4 *>(row + c0);
                const unsigned uu[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
                for (int k = 0; k < 4; k++) { v[2 * k] = lo2f<F>(uu[k]); v[2 * k + 1] = hi2f<F>(uu[k]); }
            } else {
                const float4 f = *reinterpret_cast<const float4 *>(row + c0);
                v[0] = f.x; v[1] = f.y; v[2] = f.z; v[3] = f.w;
            }
#pragma unroll
            for (int k = 0; k < EPL; k++) {
                if (ONEPASS) {
                    s0 = fmaf(v[k], wreg[0][k], s0); s1 = fmaf(v[k], wreg[1][k], s1); s2 = fmaf(v[k], wreg[2][k], s2);
                } else {
                    s0 = fmaf(v[k], wh[c0 + k], s0); s1 = fmaf(v[k], wh[C + c0 + k], s1); s2 = fmaf(v[k], wh[2 * C + c0 + k], s2);
                }
            }
            if (ONEPASS) break;
        }
        for (int o = LPP >> 1; o > 0; o >>= 1) {
            s0 += __shfl_xor(s0, o, 64); s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64);
        }
        if (cl == 0 && ok) {
            hp[(size_t)b * 2 * HW + pix] = fmaxf(s0 + b0, 0.0f);
            hp[(size_t)b * 2 * HW + HW + pix] = fmaxf(s1 + b1, 0.0f);
            hv[(size_t)b * HW + pix] = fmaxf(s2 + b2, 0.0f);
        }
    }
}

// policy FC + softmax, value FC256 + ReLU + FC1 + tanh.  FCB boards per workgroup of 6 waves share every weight
// load; the reduction axis is split over the waves (each lane keeps 6 outputs x FCB boards) so that many
// independent coalesced weight loads are in flight, and the wave partials are folded through LDS.
#define FCB 4
__global__ __launch_bounds__(384) void k_head_fc(const float *__restrict__ hp, const float *__restrict__ hv,
                                                 const float *__restrict__ wp, const float *__restrict__ bp,
                                                 const float *__restrict__ w1, const float *__restrict__ b1,
                                                 const float *__restrict__ w2, const float *__restrict__ b2,
                                                 float *__restrict__ priors, float *__restrict__ values, int HW, int O, int B,
                                                 const int32_t *__restrict__ rows, const int32_t *__restrict__ cnt)
{
    if (cnt) B = *cnt;
    if ((int)blockIdx.x * FCB >= B) return;
    __shared__ float sp[FCB][2 * 361 + 2];
    __shared__ float sv[FCB][361 + 3];
    __shared__ float part[6][FCB][384];
    __shared__ float logit[FCB][384];
    __shared__ float vred[FCB][4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int bb = blockIdx.x * FCB;
    for (int q = 0; q < FCB; q++) {
        const int b = min(bb + q, B - 1);
        for (int i = tid; i < 2 * HW; i += 384) sp[q][i] = hp[(size_t)b * 2 * HW + i];
        for (int i = tid; i < HW; i += 384) sv[q][i] = hv[(size_t)b * HW + i];
    }
    __syncthreads();
    {   // policy logits
        const int J = 2 * HW, per = (J + 5) / 6, j0 = wave * per, j1 = min(J, j0 + per);
        int oo[6];
#pragma unroll
        for (int k = 0; k < 6; k++) oo[k] = min(lane + 64 * k, O - 1);
        float acc[FCB][6] = {};
#pragma unroll 4
        for (int j = j0; j < j1; j++) {
            float wv[6];
#pragma unroll
            for (int k = 0; k < 6; k++) wv[k] = wp[(size_t)j * O + oo[k]];
#pragma unroll
            for (int q = 0; q < FCB; q++) {
                const float sj = sp[q][j];
#pragma unroll
                for (int k = 0; k < 6; k++) acc[q][k] = fmaf(sj, wv[k], acc[q][k]);
            }
        }
#pragma unroll
        for (int q = 0; q < FCB; q++)
#pragma unroll
            for (int k = 0; k < 6; k++) part[wave][q][lane + 64 * k] = acc[q][k];
    }
    __syncthreads();
    if (tid < O) {
        const float bo = bp[tid];
#pragma unroll
        for (int q = 0; q < FCB; q++) {
            float s = 0.0f;
#pragma unroll
            for (int w = 0; w < 6; w++) s += part[w][q][tid];
            logit[q][tid] = s + bo;
        }
    }
    __syncthreads();
    if (wave < FCB && bb + wave < B) {  // softmax of one board per wave
        float lv[6], mx = -INFINITY;
#pragma unroll
        for (int k = 0; k < 6; k++) {
            const int o = lane + 64 * k;
            lv[k] = o < O ? logit[wave][o] : -INFINITY;
            mx = fmaxf(mx, lv[k]);
        }
        for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
        float sum = 0.0f;
#pragma unroll
        for (int k = 0; k < 6; k++) { lv[k] = lane + 64 * k < O ? expf(lv[k] - mx) : 0.0f; sum += lv[k]; }
        for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
        const int orow = rows ? rows[bb + wave] : bb + wave;  // batch board -> row of the caller's arrays
#pragma unroll
        for (int k = 0; k < 6; k++)
            if (lane + 64 * k < O) priors[(size_t)orow * O + lane + 64 * k] = lv[k] / sum;
    }
    __syncthreads();  // `part` is reused below
    {   // value hidden layer: 256 outputs, 4 per lane
        const int per = (HW + 5) / 6, j0 = wave * per, j1 = min(HW, j0 + per);
        float acc[FCB][4] = {};
#pragma unroll 4
        for (int j = j0; j < j1; j++) {
            float wv[4];
#pragma unroll
            for (int k = 0; k < 4; k++) wv[k] = w1[(size_t)j * 256 + lane + 64 * k];
#pragma unroll
            for (int q = 0; q < FCB; q++) {
                const float sj = sv[q][j];
#pragma unroll
                for (int k = 0; k < 4; k++) acc[q][k] = fmaf(sj, wv[k], acc[q][k]);
            }
        }
#pragma unroll
        for (int q = 0; q < FCB; q++)
#pragma unroll
            for (int k = 0; k < 4; k++) part[wave][q][lane + 64 * k] = acc[q][k];
    }
    __syncthreads();
    if (tid < 256) {
        const float b1t = b1[tid], w2t = w2[tid];
#pragma unroll
        for (int q = 0; q < FCB; q++) {
            float h = 0.0f;
#pragma unroll
            for (int w = 0; w < 6; w++) h += part[w][q][tid];
            float v = fmaxf(h + b1t, 0.0f) * w2t;
            for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
            if (lane == 0) vred[q][wave] = v;
        }
    }
    __syncthreads();
    if (tid < FCB && bb + tid < B) values[rows ? rows[bb + tid] : bb + tid] = tanhf(vred[tid][0] + vred[tid][1] + vred[tid][2] + vred[tid][3] + b2[0]);
}

// ---------------------------------------------------------------------------
// host
// ---------------------------------------------------------------------------

template <typename T>
static T *talloc(Tower *t, size_t n)
{
    void *p = nullptr;
    if (hipMalloc(&p, (n ? n : 1) * sizeof(T)) != hipSuccess) return nullptr;
    hipMemset(p, 0, (n ? n : 1) * sizeof(T));
    t->allocs.push_back(p);
    return reinterpret_cast<T *>(p);
}

size_t tower_blob_floats(const Tower *t) { return t->blob_floats; }
bool tower_ready(const Tower *t) { return t->ready; }

double tower_flops_per_eval(const Tower *t)
{
    // 2*H*W*[9*17*C + n_conv*9*C^2 + C*2 + C*1] + 2*[2HW*O + HW*256 + 256]
    const double HW = t->HW, C = t->C;
    return 2.0 * HW * (9.0 * 17.0 * C + (2.0 * t->NB) * 9.0 * C * C + C * 3.0) + 2.0 * (2.0 * HW * t->O + HW * 256.0 + 256.0);
}

double tower_conv_flops_per_launch(const Tower *t, int B) { return 2.0 * (double)B * t->HW * 9.0 * t->C * t->C; }

Tower *tower_create(int H, int W, int O, int num_blocks, int channels, int dtype, int max_batch, const TowerOptions &opt, std::string &err)
{
    if (channels % 128 != 0) { err = "channels must be a multiple of 128"; return nullptr; }
    if (H * W > 361) { err = "board too large"; return nullptr; }
    Tower *t = new Tower();
    t->H = H; t->W = W; t->HW = H * W; t->PP = (H + 2) * (W + 2); t->O = O; t->NB = num_blocks; t->C = channels;
    t->dtype = dtype; t->maxB = max_batch; t->ready = false;
    t->profile = false; t->evs_used = 0; t->conv_ms = 0; t->conv_launches = 0;
    t->use_graphs = opt.graphs; t->use_duo = opt.duo; t->group_boards = opt.group_boards;
    t->conv_flops = 0;
    {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
        t->cus = cus;
    }
    const size_t C = channels, HW = t->HW;
    t->blob_floats = C * 9 * 17 + C + (size_t)num_blocks * 2 * (C * 9 * C + C) + 3 * C + 3 + 2 * HW * O + O + HW * 256 + 256 + 256 + 1;
    if (dtype < 0 || dtype > 2) { err = "tower_dtype must be 0 (float32), 1 (bfloat16) or 2 (half precision)"; return nullptr; }
    const size_t es = is16(dtype) ? 2 : 4;
    bool ok = true;
    ok = ok && (t->w_stem = talloc<char>(t, C * 9 * CIN0 * es));
    ok = ok && (t->w_conv = talloc<char>(t, (size_t)num_blocks * 2 * C * 9 * C * es));
    ok = ok && (t->b_stem = talloc<float>(t, C)) && (t->b_conv = talloc<float>(t, (size_t)num_blocks * 2 * C));
    ok = ok && (t->w_head = talloc<float>(t, 3 * C)) && (t->b_head = talloc<float>(t, 3));
    ok = ok && (t->w_pfc = talloc<float>(t, 2 * HW * O)) && (t->b_pfc = talloc<float>(t, O));
    ok = ok && (t->w_v1 = talloc<float>(t, HW * 256)) && (t->b_v1 = talloc<float>(t, 256));
    ok = ok && (t->w_v2 = talloc<float>(t, 256)) && (t->b_v2 = talloc<float>(t, 1));
    const size_t act = (size_t)max_batch * t->PP;
    // one extra board of slack on both sides is not needed: tap offsets stay inside [0, PP) of a board's own halo
    ok = ok && (t->X0 = talloc<char>(t, act * CIN0 * es));
    ok = ok && (t->X = talloc<char>(t, act * C * es)) && (t->T = talloc<char>(t, act * C * es)) && (t->Y = talloc<char>(t, act * C * es));
    ok = ok && (t->hp = talloc<float>(t, (size_t)max_batch * 2 * HW)) && (t->hv = talloc<float>(t, (size_t)max_batch * HW));
    ok = ok && (t->gcnt = talloc<int32_t>(t, MAX_GROUPS));
    if (!ok) { err = "device allocation failed"; tower_destroy(t); return nullptr; }
    return t;
}

void tower_destroy(Tower *t)
{
    if (!t) return;
    for (void *p : t->allocs) hipFree(p);
    for (hipEvent_t e : t->evs) hipEventDestroy(e);
    for (auto &g : t->graphs) hipGraphExecDestroy(g.exec);
    delete t;
}

static int upload_conv(void *dst, const float *src, size_t n, int dtype, hipStream_t st)
{
    if (is16(dtype)) {
        std::vector<bf16_t> tmp(n);
        if (dtype == 1) for (size_t i = 0; i < n; i++) tmp[i] = h_f2bf(src[i]);
        else for (size_t i = 0; i < n; i++) tmp[i] = h_f2h16(src[i]);
        if (hipMemcpyAsync(dst, tmp.data(), n * 2, hipMemcpyHostToDevice, st) != hipSuccess) return -1;
        return hipStreamSynchronize(st) == hipSuccess ? 0 : -1;
    }
    if (hipMemcpyAsync(dst, src, n * 4, hipMemcpyHostToDevice, st) != hipSuccess) return -1;
    return hipStreamSynchronize(st) == hipSuccess ? 0 : -1;
}

int tower_load(Tower *t, const float *blob, size_t n, hipStream_t st, std::string &err)
{
    if (n != t->blob_floats) {
        char buf[128];
        snprintf(buf, sizeof buf, "weight blob has %zu floats, expected %zu", n, t->blob_floats);
        err = buf;
        return -1;
    }
    const size_t C = t->C, HW = t->HW, O = t->O;
    const float *p = blob;
    int rc = 0;
    {   // stem [C][9][17] -> [C][9][CIN0] zero padded
        std::vector<float> w(C * 9 * CIN0, 0.0f);
        for (size_t co = 0; co < C; co++)
            for (int tap = 0; tap < 9; tap++)
                for (int ci = 0; ci < 17; ci++) w[(co * 9 + tap) * CIN0 + ci] = p[(co * 9 + tap) * 17 + ci];
        rc |= upload_conv(t->w_stem, w.data(), w.size(), t->dtype, st);
        p += C * 9 * 17;
        rc |= upload_conv(t->b_stem, p, C, 0, st);
        p += C;
    }
    for (int l = 0; l < 2 * t->NB; l++) {
        rc |= upload_conv((char *)t->w_conv + (size_t)l * C * 9 * C * (is16(t->dtype) ? 2 : 4), p, C * 9 * C, t->dtype, st);
        p += C * 9 * C;
        rc |= upload_conv(t->b_conv + (size_t)l * C, p, C, 0, st);
        p += C;
    }
    rc |= upload_conv(t->w_head, p, 3 * C, 0, st); p += 3 * C;
    rc |= upload_conv(t->b_head, p, 3, 0, st); p += 3;
    rc |= upload_conv(t->w_pfc, p, 2 * HW * O, 0, st); p += 2 * HW * O;
    rc |= upload_conv(t->b_pfc, p, O, 0, st); p += O;
    rc |= upload_conv(t->w_v1, p, HW * 256, 0, st); p += HW * 256;
    rc |= upload_conv(t->b_v1, p, 256, 0, st); p += 256;
    rc |= upload_conv(t->w_v2, p, 256, 0, st); p += 256;
    rc |= upload_conv(t->b_v2, p, 1, 0, st); p += 1;
    if (rc) { err = "weight upload failed"; return -1; }
    t->ready = true;
    return 0;
}

// Weights from a blob that already lies in device memory.  k_weights_cvt: n floats -> the compute type (bf16 with the same
// round-to-nearest-even, NaN kept quiet, as h_f2bf on the host; float32 copied).  k_stem_pad: the stem's [C][9][17] rows
// padded with zeros to [C][9][CIN0].
// one weight: float -> the compute type, exactly as the host conversion of tower_load does it (weights are not clamped)
template <typename T, int F>
__device__ __forceinline__ T weight_cvt(float f)
{
    if constexpr (sizeof(T) == 4) return f;
    else if constexpr (F == 0) return f2bf(f);
    else return __builtin_bit_cast(bf16_t, (_Float16)f);
}

template <typename T, int F>
__global__ void k_weights_cvt(const float *__restrict__ src, T *__restrict__ dst, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = weight_cvt<T, F>(src[i]);
}

template <typename T, int F>
__global__ void k_stem_pad(const float *__restrict__ src, T *__restrict__ dst, int rows)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * CIN0) return;
    const int r = i / CIN0, ci = i - r * CIN0;
    dst[i] = weight_cvt<T, F>(ci < 17 ? src[(size_t)r * 17 + ci] : 0.0f);
}

template <typename T, int F>
static void load_dev_t(Tower *t, const float *p, hipStream_t st)
{
    const size_t C = t->C, HW = t->HW, O = t->O;
    auto cvt = [&](void *dst, const float *src, size_t n) {
        const int blocks = (int)std::min<size_t>((n + 255) / 256, 4096);
        hipLaunchKernelGGL((k_weights_cvt<T, F>), dim3(blocks), dim3(256), 0, st, src, (T *)dst, n);
    };
    auto cpy = [&](float *dst, const float *src, size_t n) { hipMemcpyAsync(dst, src, n * sizeof(float), hipMemcpyDeviceToDevice, st); };
    hipLaunchKernelGGL((k_stem_pad<T, F>), dim3(((unsigned)(C * 9 * CIN0) + 255) / 256), dim3(256), 0, st, p, (T *)t->w_stem, (int)(C * 9));
    p += C * 9 * 17;
    cpy(t->b_stem, p, C); p += C;
    for (int l = 0; l < 2 * t->NB; l++) {
        cvt((T *)t->w_conv + (size_t)l * C * 9 * C, p, C * 9 * C); p += C * 9 * C;
        cpy(t->b_conv + (size_t)l * C, p, C); p += C;
    }
    cpy(t->w_head, p, 3 * C); p += 3 * C;
    cpy(t->b_head, p, 3); p += 3;
    cpy(t->w_pfc, p, 2 * HW * O); p += 2 * HW * O;
    cpy(t->b_pfc, p, O); p += O;
    cpy(t->w_v1, p, HW * 256); p += HW * 256;
    cpy(t->b_v1, p, 256); p += 256;
    cpy(t->w_v2, p, 256); p += 256;
    cpy(t->b_v2, p, 1);
}

int tower_load_dev(Tower *t, const float *dev_blob, size_t n, hipStream_t st, std::string &err)
{
    if (n != t->blob_floats) {
        char buf[128];
        snprintf(buf, sizeof buf, "weight blob has %zu floats, expected %zu", n, t->blob_floats);
        err = buf;
        return -1;
    }
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, dev_blob) != hipSuccess || at.type != hipMemoryTypeDevice) {
        (void)hipGetLastError();
        err = "blob_dev is not a device pointer";
        return -1;
    }
    if (t->dtype == 1) load_dev_t<bf16_t, 0>(t, dev_blob, st);
    else if (t->dtype == 2) load_dev_t<bf16_t, 1>(t, dev_blob, st);
    else load_dev_t<float, 0>(t, dev_blob, st);
    if (hipGetLastError() != hipSuccess || hipStreamSynchronize(st) != hipSuccess) { err = "device-side weight load failed"; return -1; }
    t->ready = true;
    return 0;
}

template <int F>
static void launch_conv16(const Tower *t, const bf16_t *X, const bf16_t *Wt, const float *bias, const bf16_t *R, bf16_t *Y, int B,
                          int cin, hipStream_t st, const int32_t *cnt)
{
    // batches that fill the chip with whole 256-point tiles, boards whose slab fits 3 x 128 positions (19x19): the form
    // with two 4-wave workgroups per compute unit (each 256 points x 128 channels)
    const bool duo = t->use_duo && cin == 256 && t->C == 256 && conv_channel_split(B * t->HW) == 1;
    if (cnt) {  // device-side batch size: only the slab kernels take it (tower_takes_device_count() said they apply)
        const int Mc = B * t->HW;
        if (cin == CIN0) launch_v5_stem<F>(X, Wt, bias, Y, Mc, t->HW, t->W, t->PP, st, cnt);
        else if (!(duo && launch_v6<F>(X, Wt, bias, R, Y, Mc, t->HW, t->W, t->PP, st, cnt))) launch_v5<F>(X, Wt, bias, R, Y, Mc, t->HW, t->W, t->PP, st, cnt);
        return;
    }
    const int M = B * t->HW;
    dim3 grid((M + 127) / 128, t->C / 128), block(256);
    // a few positions only: spread them over many small workgroups instead of a handful of 256-point tiles
    if (M <= SMALL_BATCH_POINTS && cin % 32 == 0 && t->C % 64 == 0) {
        if (M <= 722) launch_small<8, 2, 9, F>  /* one or two 19x19 boards */(X, Wt, bias, R, Y, M, t->HW, t->W, t->PP, cin, t->C, st);
        else launch_small<4, 4, 3, F>(X, Wt, bias, R, Y, M, t->HW, t->W, t->PP, cin, t->C, st);
        return;
    }
    if (cin == 256 && t->C == 256) {
        if (duo && launch_v6<F>(X, Wt, bias, R, Y, M, t->HW, t->W, t->PP, st)) return;
        // slab kernel when a 256-pixel tile spans at most 512 padded positions (boards of 8x8 and larger)
        if (launch_v5<F>(X, Wt, bias, R, Y, M, t->HW, t->W, t->PP, st)) return;
        dim3 g2((M + 255) / 256), b2(512);
        if (R) hipLaunchKernelGGL((k_conv3x3_c256_bf16<true, F>), g2, b2, 0, st, X, Wt, bias, R, Y, M, t->HW, t->W, t->PP);
        else hipLaunchKernelGGL((k_conv3x3_c256_bf16<false, F>), g2, b2, 0, st, X, Wt, bias, R, Y, M, t->HW, t->W, t->PP);
        return;
    }
    if (cin == CIN0 && CIN0 == 32 && t->C == 256 && !R && launch_v5_stem<F>(X, Wt, bias, Y, M, t->HW, t->W, t->PP, st)) return;
    if (cin == CIN0)
        hipLaunchKernelGGL((k_conv3x3_bf16<32, false, F>), grid, block, 0, st, X, Wt, bias, R, Y, M, t->HW, t->W, t->PP, cin, t->C);
    else if (R)
        hipLaunchKernelGGL((k_conv3x3_bf16<64, true, F>), grid, block, 0, st, X, Wt, bias, R, Y, M, t->HW, t->W, t->PP, cin, t->C);
    else
        hipLaunchKernelGGL((k_conv3x3_bf16<64, false, F>), grid, block, 0, st, X, Wt, bias, R, Y, M, t->HW, t->W, t->PP, cin, t->C);
}

static void launch_conv32(const Tower *t, const float *X, const float *Wt, const float *bias, const float *R, float *Y, int B, int cin,
                          hipStream_t st)
{
    const int M = B * t->HW;
    dim3 grid((M + 63) / 64, t->C / 64), block(256);
    if (R)
        hipLaunchKernelGGL((k_conv3x3_f32<true>), grid, block, 0, st, X, Wt, bias, R, Y, M, t->HW, t->W, t->PP, cin, t->C);
    else
        hipLaunchKernelGGL((k_conv3x3_f32<false>), grid, block, 0, st, X, Wt, bias, R, Y, M, t->HW, t->W, t->PP, cin, t->C);
}

template <typename T, int F>
static void launch_conv(const Tower *t, const T *X, const T *Wt, const float *bias, const T *R, T *Y, int B, int cin, hipStream_t st,
                        const int32_t *cnt = nullptr)
{
    if constexpr (sizeof(T) == 2) launch_conv16<F>(t, X, Wt, bias, R, Y, B, cin, st, cnt);
    else launch_conv32(t, X, Wt, bias, R, Y, B, cin, st);
}

// The tower in three stages, so that a large batch can take the middle one in groups of boards:
//   tower_planes  int8 planes of `B` boards -> the 32-channel NHWC input (rows / count as the caller gave them)
//   tower_layers  stem, residual blocks and the heads' 1x1 convolutions for the `B` boards that start at board `first` of the
//                 input and head buffers; the 256-channel activations always start at the buffers' beginning.  `alias`: the
//                 second convolution of a block writes its output over its residual input (each output element depends on the
//                 residual at that element only and nobody else reads it), so two activation buffers are live, not three.
//   tower_head_fc policy / value fully connected layers of `B` boards into the caller's rows
template <typename T, int F>
static void tower_planes(Tower *t, const int8_t *planes, int B, hipStream_t st, const int32_t *rows, const int32_t *cnt)
{
    const int ysplit = B < 64 ? 8 : 1;  // few boards: several workgroups per board in the memory-bound kernels
    hipLaunchKernelGGL((k_planes_to_nhwc<T, F>), dim3(B, ysplit), dim3(256), 0, st, planes, (T *)t->X0, t->H, t->W, rows, cnt);
}

template <typename T, int F>
static void tower_layers(Tower *t, int B, size_t first, hipStream_t st, const int32_t *cnt, bool alias)
{
    const size_t C = t->C;
    T *X0 = (T *)t->X0 + first * t->PP * CIN0, *X = (T *)t->X, *Tm = (T *)t->T, *Y = alias ? X : (T *)t->Y;
    float *hp = t->hp + first * 2 * t->HW, *hv = t->hv + first * t->HW;
    launch_conv<T, F>(t, X0, (const T *)t->w_stem, t->b_stem, (const T *)nullptr, X, B, CIN0, st, cnt);
    for (int l = 0; l < t->NB; l++) {
        const T *w1 = (const T *)t->w_conv + (size_t)(2 * l) * C * 9 * C;
        const T *w2 = (const T *)t->w_conv + (size_t)(2 * l + 1) * C * 9 * C;
        const bool prof = t->profile && (size_t)t->evs_used + 4 <= t->evs.size();
        if (prof) { t->ev_boards[t->evs_used / 2] = B; hipEventRecord(t->evs[t->evs_used++], st); }
        launch_conv<T, F>(t, X, w1, t->b_conv + (size_t)(2 * l) * C, (const T *)nullptr, Tm, B, (int)C, st, cnt);
        if (prof) { hipEventRecord(t->evs[t->evs_used++], st); t->ev_boards[t->evs_used / 2] = B; hipEventRecord(t->evs[t->evs_used++], st); }
        launch_conv<T, F>(t, Tm, w2, t->b_conv + (size_t)(2 * l + 1) * C, X, Y, B, (int)C, st, cnt);
        if (prof) hipEventRecord(t->evs[t->evs_used++], st);
        T *tmp = X; X = Y; Y = tmp;
    }
    const int chunks = (int)C / (16 / (int)sizeof(T));  // 16-byte pieces per board point
    int lpp = 64;
    while (chunks % lpp) lpp >>= 1;                     // largest power of two <= 64 dividing it
    // one workgroup per board reads 185 KB (19x19) at a time; below about 4 workgroups per compute unit the boards are cut
    const int ysplit = B < 64 ? 8 : (B < 4 * t->cus ? 4 : 1);
    if (chunks == lpp)
        hipLaunchKernelGGL((k_head_conv<T, true, F>), dim3(B, ysplit), dim3(256), 0, st, (const T *)X, t->w_head, t->b_head, hp, hv, t->HW,
                           t->W, t->PP, (int)C, lpp, cnt);
    else
        hipLaunchKernelGGL((k_head_conv<T, false, F>), dim3(B, ysplit), dim3(256), 0, st, (const T *)X, t->w_head, t->b_head, hp, hv, t->HW,
                           t->W, t->PP, (int)C, lpp, cnt);
}

static void tower_head_fc(Tower *t, int B, float *priors, float *values, hipStream_t st, const int32_t *rows, const int32_t *cnt)
{
    hipLaunchKernelGGL(k_head_fc, dim3((B + FCB - 1) / FCB), dim3(384), 0, st, t->hp, t->hv, t->w_pfc, t->b_pfc, t->w_v1, t->b_v1, t->w_v2,
                       t->b_v2, priors, values, t->HW, t->O, B, rows, cnt);
}

// Boards per group for a batch of B boards (0: the whole batch in one pass).  The 256-channel bf16 tower moves three
// activation tensors per convolution (input, residual, output) and a batch of thousands of boards streams all of them
// through HBM in every layer.  A group of boards small enough that its two live activation buffers stay in the 256 MB
// memory-side cache goes through ALL layers before the next group starts, over the same two buffers: from the second
// layer on the convolutions find their operands in that cache.  A group is what fills GROUP_ROUNDS whole rounds of 256-point
// tiles on the chip's compute units (a tile more would cost a whole extra round: 363 boards of 19x19 on 256 units, 82 MB
// per buffer); the last group of a batch takes what is left.
static int tower_group_boards(const Tower *t, int B)
{
    if (t->group_boards < 0 || !is16(t->dtype) || t->C != 256) return 0;
    int g = t->group_boards;
    if (g == 0) {
        g = GROUP_ROUNDS * t->cus * 256 / t->HW;
        if (B < 2 * g) return 0;   // too small to be worth splitting
    }
    if (g < 1 || B <= g) return 0;
    if ((B + g - 1) / g > MAX_GROUPS) g = (B + MAX_GROUPS - 1) / MAX_GROUPS;
    return g;
}

__global__ void k_group_counts(const int32_t *__restrict__ cnt, int32_t *__restrict__ gcnt, int g, int n)
{
    const int i = threadIdx.x;
    if (i < n) gcnt[i] = min(max(*cnt - i * g, 0), g);
}

template <typename T, int F>
static int forward_t(Tower *t, const int8_t *planes, int B, float *priors, float *values, hipStream_t st, const int32_t *rows,
                     const int32_t *cnt)
{
    const int g = tower_group_boards(t, B);
    tower_planes<T, F>(t, planes, B, st, rows, cnt);
    if (g == 0) {
        tower_layers<T, F>(t, B, 0, st, cnt, false);
    } else {
        // the i-th group holds boards i*g ... of the batch (of the compacted batch when the caller lists rows)
        const int n = (B + g - 1) / g;
        if (cnt) hipLaunchKernelGGL(k_group_counts, dim3(1), dim3(MAX_GROUPS), 0, st, cnt, t->gcnt, g, n);
        for (int i = 0; i < n; i++) {
            const int b0 = i * g;
            tower_layers<T, F>(t, B - b0 < g ? B - b0 : g, (size_t)b0, st, cnt ? t->gcnt + i : nullptr, true);
        }
    }
    tower_head_fc(t, B, priors, values, st, rows, cnt);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

// Which convolution kernel launch_conv picks for the residual layers of a batch of B boards (the tower's groups of boards
// taken into account): the name a profiler shows, for labelling measurements.
const char *tower_conv_kernel_name(const Tower *t, int B, bool device_count)
{
    if (!is16(t->dtype)) return "k_conv3x3_f32 (vector ALU, float32 operands, float64 accumulation)";
    // the 16-bit kernels are one template; its last parameter is the operand format (0: bf16, 1: IEEE half precision)
    const bool h = t->dtype == 2;
    const int g = tower_group_boards(t, B);
    const int M = (g ? g : B) * t->HW;
    if (!device_count && M <= SMALL_BATCH_POINTS && t->C % 64 == 0)
        return h ? "k_conv3x3_bf16_small<..., 1> (32 points x 32/64 channels per workgroup, reduction axis split over the waves, f16 operands)"
                 : "k_conv3x3_bf16_small (32 points x 32/64 channels per workgroup, reduction axis split over the waves)";
    if (t->C != 256) return h ? "k_conv3x3_bf16<..., 1> (per-tap gather, f16 MFMA 16x16x32)" : "k_conv3x3_bf16 (per-tap gather, bf16 MFMA 16x16x32)";
    const int span = conv_slab_span(t->HW, t->W, t->PP);
    if (t->use_duo && conv_channel_split(M) == 1 && span <= 27 * 16)
        return h ? "k_conv3x3_c256_bf16_duo<..., 1> (3x3 256->256 implicit GEMM, f16 MFMA 16x16x32, two 4-wave workgroups per CU)"
                 : "k_conv3x3_c256_bf16_duo (3x3 256->256 implicit GEMM, bf16 MFMA 16x16x32, two 4-wave workgroups per CU)";
    if (span <= 4 * 128)
        return h ? "k_conv3x3_c256_bf16_slab<..., 1> (3x3 256->256 implicit GEMM, f16 MFMA 16x16x32, one 8-wave workgroup per CU)"
                 : "k_conv3x3_c256_bf16_slab (3x3 256->256 implicit GEMM, bf16 MFMA 16x16x32, one 8-wave workgroup per CU)";
    return h ? "k_conv3x3_c256_bf16<..., 1> (per-tap gather of both operands, f16 MFMA 16x16x32)"
             : "k_conv3x3_c256_bf16 (per-tap gather of both operands, bf16 MFMA 16x16x32)";
}

static int forward_any(Tower *t, const int8_t *planes, int B, float *priors, float *values, hipStream_t st, const int32_t *rows,
                       const int32_t *cnt)
{
    if (t->dtype == 1) return forward_t<bf16_t, 0>(t, planes, B, priors, values, st, rows, cnt);
    if (t->dtype == 2) return forward_t<bf16_t, 1>(t, planes, B, priors, values, st, rows, cnt);
    return forward_t<float, 0>(t, planes, B, priors, values, st, rows, cnt);
}

bool tower_takes_device_count(const Tower *t, int B)
{
    return is16(t->dtype) && t->C == 256 && CIN0 == 32 && conv_slab_span(t->HW, t->W, t->PP) <= 4 * 128 && B >= 1 && B <= t->maxB &&
           (size_t)B * t->HW > (size_t)DEVICE_COUNT_MIN_POINTS;  // below that a half-empty batch belongs to the kernels for few boards
}

int tower_forward(Tower *t, const int8_t *planes_dev, int B, float *priors_dev, float *values_dev, hipStream_t st, std::string &err,
                  const int32_t *rows_dev, const int32_t *count_dev)
{
    if (!t->ready) { err = "weights not loaded"; return -1; }
    if (B < 1 || B > t->maxB) { err = "batch exceeds the tower's activation buffers"; return -1; }
    if (count_dev && !tower_takes_device_count(t, B)) { err = "this tower shape needs the batch size on the host"; return -1; }
    if (t->profile) t->evs_used = 0;  // (a sampled forward's events wait for tower_collect_profile across later forwards)
    const bool graphable = t->use_graphs && !t->profile && !count_dev && (size_t)B * t->HW <= GRAPH_MAX_POINTS && tower_group_boards(t, B) == 0;
    if (graphable) {
        for (auto &g : t->graphs)
            if (g.B == B && g.planes == planes_dev && g.priors == priors_dev && g.values == values_dev && g.rows == rows_dev && g.st == st) {
                if (hipGraphLaunch(g.exec, st) == hipSuccess) return 0;
                break;  // fall back to plain launches below
            }
        if (t->graphs.size() < 16 && hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed) == hipSuccess) {
            const int crc = forward_any(t, planes_dev, B, priors_dev, values_dev, st, rows_dev, nullptr);
            hipGraph_t graph = nullptr;
            hipGraphExec_t exec = nullptr;
            const hipError_t e = hipStreamEndCapture(st, &graph);
            const bool ok = crc == 0 && e == hipSuccess && graph && hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) == hipSuccess;
            if (graph) hipGraphDestroy(graph);
            if (ok && hipGraphLaunch(exec, st) == hipSuccess) {
                t->graphs.push_back({planes_dev, priors_dev, values_dev, rows_dev, B, st, exec});
                return 0;
            }
            if (exec) hipGraphExecDestroy(exec);
            (void)hipGetLastError();
            t->use_graphs = false;  // capture is not available here: plain launches from now on
        }
    }
    int rc = forward_any(t, planes_dev, B, priors_dev, values_dev, st, rows_dev, count_dev);
    if (rc) err = std::string("kernel launch failed: ") + hipGetErrorString(hipGetLastError());
    return rc;
}

void tower_set_profile(Tower *t, bool on)
{
    t->profile = on;
    if (on && t->evs.empty()) {
        const int g = tower_group_boards(t, t->maxB);
        const int groups = g ? (t->maxB + g - 1) / g : 1;
        t->evs.resize((size_t)4 * t->NB * groups);   // forwards of fewer, larger groups need fewer
        t->ev_boards.assign(t->evs.size() / 2, 0);
        for (auto &e : t->evs) hipEventCreate(&e);
    }
}

// call after the launch stream has been synchronised: folds the last forward's event pairs
void tower_collect_profile(Tower *t)
{
    for (int i = 0; i + 1 < t->evs_used; i += 2) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, t->evs[i], t->evs[i + 1]) == hipSuccess) {
            t->conv_ms += ms; t->conv_launches++; t->conv_flops += tower_conv_flops_per_launch(t, t->ev_boards[i / 2]);
        }
    }
    t->evs_used = 0;
}

void tower_profile_read(Tower *t, double *conv_ms, unsigned long long *launches, double *flops, bool reset)
{
    *conv_ms = t->conv_ms; *launches = t->conv_launches; *flops = t->conv_flops;
    if (reset) { t->conv_ms = 0; t->conv_launches = 0; t->conv_flops = 0; }
}
