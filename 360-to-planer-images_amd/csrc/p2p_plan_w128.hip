// The 128 x 16 tile shape (512-thread workgroups, LDS buffers of 1408 items): p2p_plan.hip once more, in namespace
// p2p::w128 (p2p_device.h: tile shapes).
#undef P2P_TILE_W
#undef P2P_TILE_ROWS
#undef P2P_BLOCK
#undef P2P_CAP
#undef P2P_SLOTS
#undef P2P_WAVES
#undef P2P_SHAPE_NS
#define P2P_TILE_W 128
#define P2P_TILE_ROWS 16
#define P2P_BLOCK 512
#define P2P_CAP 1408
#define P2P_SLOTS 3
#define P2P_SHAPE_NS w128
#include "p2p_plan.hip"
