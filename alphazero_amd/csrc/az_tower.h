// Dual-head residual tower (inference only) - internal interface used by az_engine.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include <string>

struct Tower;

// measurement switches (az_config.tower_no_graphs / tower_no_duo / tower_group_boards); the defaults are the production setting
struct TowerOptions {
    bool graphs = true;     // small-batch forwards replayed as captured HIP graphs
    bool duo = true;        // two-workgroups-per-CU convolution kernel where it applies
    int group_boards = 0;   // boards per group of a large batch: 0 automatic, -1 never, N forced
};

Tower *tower_create(int H, int W, int O, int num_blocks, int channels, int dtype, int max_batch, const TowerOptions &opt, std::string &err);
void tower_destroy(Tower *t);
size_t tower_blob_floats(const Tower *t);
// host float32 blob (layout: alphazero_amd/network/model.py) -> device weights in the compute dtype
int tower_load(Tower *t, const float *host_blob, size_t n_floats, hipStream_t st, std::string &err);
// the same blob in device memory: padded / converted / copied by kernels on `st`; returns once `st` has finished
int tower_load_dev(Tower *t, const float *dev_blob, size_t n_floats, hipStream_t st, std::string &err);
bool tower_ready(const Tower *t);
// planes_dev int8 (*,17,H,W); priors_dev (*,O), values_dev (*) float32.  Board b of the batch of B is row
// rows_dev[b] of those arrays (rows_dev == nullptr: row b), so a subset of a larger leaf batch can be evaluated in place.
// count_dev (optional, only when tower_takes_device_count): the real number of boards lives in device memory and B is
// the capacity the grids are sized for, so the caller does not have to read the count back before launching.
int tower_forward(Tower *t, const int8_t *planes_dev, int B, float *priors_dev, float *values_dev, hipStream_t st,
                  std::string &err, const int32_t *rows_dev = nullptr, const int32_t *count_dev = nullptr);
bool tower_takes_device_count(const Tower *t, int B);
double tower_flops_per_eval(const Tower *t);
// names/durations of the dominant kernel for the roofline line
double tower_conv_flops_per_launch(const Tower *t, int B);
const char *tower_conv_kernel_name(const Tower *t, int B, bool device_count);
// per-launch HIP-event timing of the residual-tower 3x3 convolutions
void tower_set_profile(Tower *t, bool on);
void tower_collect_profile(Tower *t);
void tower_profile_read(Tower *t, double *conv_ms, unsigned long long *launches, double *flops, bool reset);
