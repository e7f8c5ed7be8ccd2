// Entry points of the round-3 experiments that lost to the block tiles and were moved out of the product library
// (DESIGN.md §4 "stream-K, built, measured, not adopted"; §10 finding i for the 4-wave tile). `make -C treedetection_amd/csrc experimental`
// compiles them (objects only) so they keep building; nothing in libtreedet_hip.so references them.
#pragma once
#include "../common.h"

td_status conv_w4_launch(const ConvArgs& a, bool out_f32, hipStream_t stream);      // former tile id 28 (conv_w4.hip)
// stream-K form of the fp16 small-map layers (conv_streamk.hip; former tile ids 21 / 22): variant 0 = 128 x 128 tiles / 4 waves / 512
// resident blocks, 1 = 256 x 128 / 8 waves / 256 blocks. a.sk_ws: conv_sk_workspace_floats() floats; a.sk_cnt: conv_sk_max_tiles() zeroed ints.
int conv_sk_grid(int bm, int bn);
size_t conv_sk_workspace_floats(void);
int conv_sk_max_tiles(void);
td_status conv_sk_launch(const ConvArgs& a, int precision, int variant, hipStream_t stream);
