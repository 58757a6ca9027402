// The 128 x 16 tile shape (512-thread workgroups, LDS buffers of 1408 items): p2p_views.hip once more, in namespace
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
// the main kernel's store goes out BEHIND the next pair's loads in this shape (draw_tight): its launches stream 4-18 GB of
// views to HBM, and a pair's wait for its pieces then does not include the acknowledgement of the store issued just
// before them -- config 4 6.22 -> 6.15 / 6.13 -> 6.08 ms, config 3's 64 panoramas 6.02 -> 6.00 / 6.11 -> 6.06; the 64-wide
// shape, whose views mostly stay in the Infinity Cache, loses 0.3-0.7 % with it (profiles/r04_store_after_loads.txt)
#ifndef P2P_STORE_BEFORE_LOADS
#define P2P_STORE_AFTER_LOADS 1
#endif
#include "p2p_views.hip"
