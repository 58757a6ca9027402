// The band shape: 64 x 16 tiles and 256-thread workgroups like w64, but LDS buffers of 960 items (30 KB per workgroup,
// five per CU) and four item slots per thread -- the shape of jobs that are drawn from source-band tiles (p2p_device.h).
// A band tile of a minifying view set is limited by its rectangle, not by its groups: with 704 items the reference
// CLI's default set is 5943 tiles of 125 groups, with 960 items 4400 of 170, and a tile's set-up is half of what a
// band workgroup does there (4 pairs).  CLI default set 47.5 -> 46.0 us; config 2's per-view tiles lose 4 % in this
// shape, so it is not theirs.  p2p_views.hip once more, in namespace p2p::w64b.
#undef P2P_CAP
#undef P2P_SLOTS
#undef P2P_WAVES
#undef P2P_BAND_WAVES
#undef P2P_SHAPE_NS
#define P2P_CAP 960
#define P2P_SLOTS 4
#define P2P_WAVES 5
#define P2P_BAND_WAVES 5
#define P2P_SHAPE_NS w64b
#include "p2p_views.hip"
