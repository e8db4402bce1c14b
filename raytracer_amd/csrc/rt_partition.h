// rt_partition.h — the column partition of a frame over n shares (multi-GPU, rt_render_multi, rt_params.x_begin/x_step)
// and where a share's compact output lands in the whole frame.  Plain integer arithmetic, usable on the host (the
// C ABI, the CPU test tests/host_cpp/partition_check.cpp) and on the device (place_columns_kernel).
//
// The frame is `Array2D.create width height` (Scene.fs:28): element (px, py) at px*height + py.  Share r of n renders
// the columns px = r, r+n, r+2n, ... < width into a compact [ncols][height] block, element (k, py) at k*height + py.
#pragma once
#ifdef __HIPCC__
#define RT_HD __host__ __device__
#else
#define RT_HD
#endif

namespace rt {

// number of columns share r of n renders
RT_HD inline int share_columns(int width, int r, int n) { return r < width ? (width - r + n - 1) / n : 0; }
// frame index of compact element (k, py) of share r of n
RT_HD inline long long frame_index(int r, int n, int k, int height, int py) { return (long long)(r + k * n) * height + py; }

}  // namespace rt
