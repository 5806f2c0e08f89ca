// Compile-time geometry shared by host packing and the screen kernel.
#pragma once
namespace mf {
constexpr int SCREEN_U = 2;            // uint4 loads in flight per lane per chunk
constexpr int SCREEN_BLOCK = 1024;     // threads per screen workgroup
constexpr int SCREEN_CU_NUM = 7, SCREEN_CU_DEN = 8;   // stride-16 screen: workgroups per CU of the device (see screen_grid_for)
constexpr unsigned KB_CO_LOG2W = 12;   // k-mer bit table of a co-resident exact kernel: at most 1 << 12 words (16 KiB)
constexpr int EXACT_MAX_GRID = 2048;   // most workgroups of the exact / finish / protein kernels (one (pass, candidate) partial pair each)
}
