// The band shape (p2p_views_band.hip): p2p_plan.hip once more, in namespace p2p::w64b.  No float pixel path in this shape.
#undef P2P_CAP
#undef P2P_SLOTS
#undef P2P_WAVES
#undef P2P_SHAPE_NS
#define P2P_CAP 960
#define P2P_SLOTS 4
#define P2P_WAVES 5
#define P2P_SHAPE_NS w64b
#define P2P_SHAPE_NO_FLOAT 1
#include "p2p_plan.hip"
