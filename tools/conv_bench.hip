// Dev harness: correctness + timing of the tower's 3x3 256->256 bf16 convolution kernels.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/conv_bench.hip -o tools/conv_bench
//   tools/conv_bench [boards=1024] [iters=20] [variant=0] [resid=1] [check=1] [channel_split=0]
#include "conv_ablation_kernels.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

static bf16_t h_bf(float f) { unsigned u; memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (bf16_t)(u >> 16); }
static float h_f(bf16_t b) { unsigned u = ((unsigned)b) << 16; float f; memcpy(&f, &u, 4); return f; }

// naive reference for sampled outputs: one thread per (sample, cout)
__global__ void k_ref(const bf16_t *X, const bf16_t *Wt, const float *bias, const bf16_t *R, float *out, const int *rows, int nrows,
                      int HW, int Wd, int PP, int resid)
{
    int s = blockIdx.x, co = threadIdx.x;
    int m = rows[s];
    int b = m / HW, pix = m - b * HW, y = pix / Wd, x = pix - y * Wd, W2 = Wd + 2;
    long base = (long)b * PP + (y + 1) * W2 + (x + 1);
    float acc = 0.f;
    for (int tap = 0; tap < 9; tap++) {
        long p = base + (tap / 3 - 1) * W2 + (tap % 3 - 1);
        for (int ci = 0; ci < 256; ci++) acc += bf2f(X[p * 256 + ci]) * bf2f(Wt[((long)co * 9 + tap) * 256 + ci]);
    }
    acc += bias[co];
    if (resid) acc += bf2f(R[base * 256 + co]);
    out[(long)s * 256 + co] = fmaxf(acc, 0.f);
}

int main(int argc, char **argv)
{
    int B = argc > 1 ? atoi(argv[1]) : 1024, iters = argc > 2 ? atoi(argv[2]) : 20, variant = argc > 3 ? atoi(argv[3]) : 0;
    int resid = argc > 4 ? atoi(argv[4]) : 1, check = argc > 5 ? atoi(argv[5]) : 1;
    int cs = argc > 6 ? atoi(argv[6]) : 0;  // channel split of the slab kernel (1, 2, 4; 0 = automatic)
    const int H = 19, Wd = 19, HW = 361, PP = 441, C = 256;
    const int M = B * HW;
    size_t act = (size_t)B * PP * C;
    std::vector<bf16_t> hX(act, 0), hR(act, 0), hW((size_t)C * 9 * C);
    std::vector<float> hb(C);
    srand(1);
    const bool relu_data = getenv("RELU") != nullptr;
    auto rnd = []() { return (float)rand() / RAND_MAX * 2.f - 1.f; };
    for (int b = 0; b < B; b++)
        for (int y = 0; y < H; y++)
            for (int x = 0; x < Wd; x++) {
                size_t o = ((size_t)b * PP + (y + 1) * (Wd + 2) + (x + 1)) * C;
                // RELU=1: operands like the tower's real ones (outputs of a ReLU: about half zeros, the rest positive)
                if (b < 64 || check == 0) for (int c = 0; c < C; c++) {
                    float a = rnd(), r = rnd();
                    if (relu_data) { a = a > 0 ? a : 0.f; r = r > 0 ? r : 0.f; }
                    hX[o + c] = h_bf(a); hR[o + c] = h_bf(r);
                }
                else { size_t src = ((size_t)(b % 64) * PP + (y + 1) * (Wd + 2) + (x + 1)) * C; memcpy(&hX[o], &hX[src], C * 2); memcpy(&hR[o], &hR[src], C * 2); }
            }
    for (auto &w : hW) w = h_bf(rnd() * 0.03f);
    for (auto &b : hb) b = rnd() * 0.1f;
    bf16_t *X, *R, *Y, *Wt; float *bias;
    CK(hipMalloc(&X, act * 2)); CK(hipMalloc(&R, act * 2)); CK(hipMalloc(&Y, act * 2)); CK(hipMalloc(&Wt, hW.size() * 2)); CK(hipMalloc(&bias, C * 4));
    CK(hipMemcpy(X, hX.data(), act * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(R, hR.data(), act * 2, hipMemcpyHostToDevice));
    CK(hipMemset(Y, 0, act * 2));
    CK(hipMemcpy(Wt, hW.data(), hW.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(bias, hb.data(), C * 4, hipMemcpyHostToDevice));
    auto launch = [&]() {
        dim3 g((M + 255) / 256), blk(512);
        switch (variant) {
        default:
            if (resid) hipLaunchKernelGGL((k_conv3x3_c256_bf16<true>), g, blk, 0, 0, X, Wt, bias, R, Y, M, HW, Wd, PP);
            else hipLaunchKernelGGL((k_conv3x3_c256_bf16<false>), g, blk, 0, 0, X, Wt, bias, (const bf16_t *)nullptr, Y, M, HW, Wd, PP);
            break;
        case 6:
            if (!launch_v6(X, Wt, bias, resid ? R : nullptr, Y, M, HW, Wd, PP, 0, nullptr)) { printf("duo kernel not applicable\n"); exit(1); }
            break;
        case 5:
            if (!launch_v5(X, Wt, bias, resid ? R : nullptr, Y, M, HW, Wd, PP, 0, nullptr, cs)) { printf("slab kernel not applicable\n"); exit(1); }
            break;
        }
    };
    launch();
    CK(hipDeviceSynchronize());
    if (check) {
        const int NS = 2048;
        std::vector<int> rows(NS);
        for (int i = 0; i < NS; i++) rows[i] = (i < 512) ? i : (i < 1024 ? M - 1 - (i - 512) : (int)((long)rand() * 7919 % M));
        int *drows; float *dout;
        CK(hipMalloc(&drows, NS * 4)); CK(hipMalloc(&dout, (size_t)NS * 256 * 4));
        CK(hipMemcpy(drows, rows.data(), NS * 4, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_ref, dim3(NS), dim3(256), 0, 0, X, Wt, bias, R, dout, drows, NS, HW, Wd, PP, resid);
        std::vector<float> ref((size_t)NS * 256);
        CK(hipMemcpy(ref.data(), dout, ref.size() * 4, hipMemcpyDeviceToHost));
        std::vector<bf16_t> hY(act);
        CK(hipMemcpy(hY.data(), Y, act * 2, hipMemcpyDeviceToHost));
        double maxerr = 0; long bad = 0;
        for (int i = 0; i < NS; i++) {
            int m = rows[i], b = m / HW, pix = m % HW, y = pix / Wd, x = pix % Wd;
            size_t o = ((size_t)b * PP + (y + 1) * (Wd + 2) + (x + 1)) * C;
            for (int c = 0; c < C; c++) {
                double e = fabs(h_f(hY[o + c]) - ref[(size_t)i * 256 + c]);
                double tol = 0.02 + 0.01 * fabs(ref[(size_t)i * 256 + c]);
                if (e > maxerr) maxerr = e;
                if (e > tol) bad++;
            }
        }
        // halo must stay zero
        long halo_bad = 0;
        for (int b = 0; b < (B < 8 ? B : 8); b++)
            for (int p = 0; p < PP; p++) {
                int yy = p / (Wd + 2), xx = p % (Wd + 2);
                if (yy == 0 || yy == H + 1 || xx == 0 || xx == Wd + 1)
                    for (int c = 0; c < C; c++) if (hY[((size_t)b * PP + p) * C + c] != 0) halo_bad++;
            }
        printf("check: max_abs_err %.4f  out_of_tol %ld  halo_nonzero %ld\n", maxerr, bad, halo_bad);
    }
#ifdef STAMP
    {
        launch();
        CK(hipDeviceSynchronize());
        static unsigned long long st[2][512];
        CK(hipMemcpyFromSymbol(st, HIP_SYMBOL(g_stamps), sizeof st));
        printf("interval durations (cycles), group0 / group1 stamps after each barrier:\n");
        if (getenv("VERBOSE")) for (int i = 1; i < 60; i++) printf("%3d: g0 %6lld  g1 %6lld   (g1-g0 skew %lld)\n", i, (long long)(st[0][i] - st[0][i - 1]), (long long)(st[1][i] - st[1][i - 1]), (long long)(st[1][i] - st[0][i]));
        printf("total g0 %lld cycles over 146 barriers\n", (long long)(st[0][145] - st[0][0]));
        {
            double rt = (double)(st[0][501] - st[0][500]) * 10.0;  // 100 MHz ticks -> ns
            double cyc = (double)(st[0][503] - st[0][502]);
            printf("K loop of one workgroup: %.0f shader cycles in %.0f ns -> %.3f GHz, %.1f cycles per 32-MFMA interval\n", cyc, rt, cyc / rt, cyc / 145.0);
            for (int w = 0; w < 2; w++) {
                const unsigned long long *p = &st[1][500 + 4 * w];
                printf("workgroup %s wall clock: prologue %.2f us, K loop %.2f us, epilogue (stores issued) %.2f us\n", w ? "late" : "early",
                       (p[1] - p[0]) * 0.01, (p[2] - p[1]) * 0.01, (p[3] - p[2]) * 0.01);
            }
        }
    }
#endif
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; i++) launch();
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < iters; i++) launch();
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= iters;
    double fl = 2.0 * M * 9.0 * C * C;
    printf("variant %d resid %d B %d: %.4f ms  %.1f TFLOP/s  (%.1f%% of 2500)\n", variant, resid, B, ms, fl / ms / 1e9, fl / ms / 1e9 / 25.0);
    if (getenv("CHUNK") && variant == 6) {
        // the same work launched as chunks of CHUNK boards: over the whole buffers (every chunk touches new memory) and over the
        // first chunk's memory again and again (activations stay in the memory-side cache if they fit)
        const int cb = atoi(getenv("CHUNK")), nch = B / cb;
        const size_t coff = (size_t)cb * PP * C;
        for (int same = 0; same < 2; same++) {
            auto pass = [&]() {
                for (int c = 0; c < nch; c++) {
                    size_t o = same ? 0 : c * coff;
                    launch_v6(X + o, Wt, bias, resid ? R + o : nullptr, Y + o, cb * HW, HW, Wd, PP, 0, nullptr);
                }
            };
            pass();
            CK(hipEventRecord(e0, 0));
            for (int i = 0; i < iters; i++) pass();
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1));
            ms /= iters;
            double flc = 2.0 * nch * cb * HW * 9.0 * C * C;
            printf("chunks of %d boards (%d launches, %.0f MB per buffer), %s memory: %.4f ms  %.1f TFLOP/s\n", cb, nch, coff * 2 / 1e6,
                   same ? "same" : "distinct", ms, flc / ms / 1e9);
        }
    }
    if (getenv("TWO") && variant == 5 && B % 2 == 0) {
        // the same work as two dependent chains of half-batch launches on two streams
        hipStream_t s2[2];
        CK(hipStreamCreate(&s2[0])); CK(hipStreamCreate(&s2[1]));
        const int Mh = M / 2;
        const size_t half = (size_t)(B / 2) * PP * C;
        auto chain = [&](int n) {
            for (int i = 0; i < n; i++)
                for (int h = 0; h < 2; h++)
                    launch_v5(X + h * half, Wt, bias, resid ? R + h * half : nullptr, Y + h * half, Mh, HW, Wd, PP, s2[h]);
        };
        chain(3);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, s2[0]));
        CK(hipStreamWaitEvent(s2[1], e0, 0));
        chain(iters);
        hipEvent_t j;
        CK(hipEventCreate(&j));
        CK(hipEventRecord(j, s2[1]));
        CK(hipStreamWaitEvent(s2[0], j, 0));
        CK(hipEventRecord(e1, s2[0]));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        ms /= iters;
        printf("two streams x half batch: %.4f ms per layer  %.1f TFLOP/s\n", ms, fl / ms / 1e9);
    }
    return 0;
}
