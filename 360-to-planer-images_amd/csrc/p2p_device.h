// p2p_device.h -- structures and launchers shared by the device code (p2p_views / p2p_maps / p2p_remap / p2p_float .hip) and the
// host side of the C ABI (p2p_host.cpp).  Not part of the public ABI (that is include/p2p_hip.h).
#ifndef P2P_DEVICE_H
#define P2P_DEVICE_H

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

namespace p2p {

constexpr int TILE_W = 32;          // output tile of one workgroup
constexpr int TILE_H = 16;
constexpr int VIEWS_BLOCK = 256;
constexpr int VIEWS_PXT = TILE_W * TILE_H / VIEWS_BLOCK;  // output pixels per thread (rows ROWSTEP apart)
constexpr int VIEWS_SLOTS = 2;      // 16-byte footprint items one thread produces per (panorama, yaw) pair
constexpr int LDS_ITEMS_CAP = VIEWS_SLOTS * VIEWS_BLOCK;  // items (4 rot pixels each) per LDS buffer

// how the yaw map of one yaw angle acts on columns (see yaw_desc_kernel)
struct YawDesc {
    int s;        // rot column c reads source column (c + s) mod pw
    int mode;     // 0: one two-tap weight f for every column; 1: per-column weights (f4tab); 2: not a shift
    int f;        // mode 0: the weight, 0..32
    int c_clamp;  // mode 0: the one column whose weight differs (clipped to pw-1), or -1
};

// scalars of precompute_pitch_mapping (P:114-175) that NumPy evaluates once, on the host, in
// float64 and then uses as float32
struct MapGeom {
    float half_w, half_h;  // W / 2.0, H / 2.0          (P:129-130)
    float focal;           // 0.5 * W / tan(FOV / 2)    (P:119, cast at P:131)
    float pw_f, ph_f;      // panorama size as float32  (P:167-173)
};

struct PitchConst {
    float c, s;  // cos / sin of the pitch angle, float64 -> float32 (R_pitch dtype, P:142-149)
};

struct ViewsParams {
    const uint8_t* src;      // [n_panos] panoramas, each ph rows of src_pitch bytes (BGR interleaved)
    size_t pano_stride;      // bytes between panoramas
    int src_pitch;           // bytes between rows (>= 3 * pw)
    int pw, ph;
    const uint32_t* ytab;    // [n_yaw][pw] packed yaw-table entries: 3*ix | fx << 20
    const YawDesc* ydesc;    // [n_yaw]
    const uint32_t* f4tab;   // [n_yaw][pw] weights of columns c..c+3 (mod pw), one byte each
    int n_yaw, n_pitch, n_panos;
    int pairs_per_block;     // (panorama, yaw) pairs looped over by one workgroup
    uint32_t n_yaw_magic;    // ceil(2^32 / n_yaw): pair / n_yaw == umulhi(pair, magic) while pair * n_yaw < 2^32
    const PitchConst* pitch; // [n_pitch]
    const float* mapU;       // [n_pitch][oh][ow] caller maps (HOST_MAPS) or nullptr
    const float* mapV;
    MapGeom geom;
    int ow, oh;
    uint8_t* out;            // [n_panos][n_yaw][n_pitch][oh][ow][3]
    int32_t* coords;         // optional [n_pitch][oh][ow][2] dump of (sx, sy)
    int border;              // stage-2 border mode (0 = BORDER_CONSTANT 0, the reference's current tool)
    uint8_t pitch_order[64]; // blockIdx.y -> pitch index, heaviest view first (n_pitch <= 64 per job)
    uint2* plan;             // sub-tiles of the tiles whose footprint outgrows the LDS buffers: x0 | y0 << 15 |
                             // (16 wide) << 30, pitch index; written by the plan pass, read by the sub-tile pass
    uint32_t* plan_count;
    uint8_t* plan_flag;      // [n_pitch][tiles]: 1 = the tile is in the plan (the main pass skips it)
    int plan_n;              // entries in plan (host copy)
    int plan_gx;             // sub-tile workgroups per view row of the grid: 8 * ceil(plan_n / n_pitch / 8)
    int use_plan;            // the main pass leaves the listed tiles to the sub-tile pass
    float centre;            // float pixel path only: 0 = the reference's sampling convention, 0.5 = pixel centres
};

struct RemapParams {
    const uint8_t* src;
    int sw, sh, src_pitch;
    const float* U;
    const float* V;
    uint8_t* dst;
    int ow, oh;
    int border;
    uint8_t cval[4];
    const short* ctab;  // INTER_CUBIC only: [32 * 32][4][4] fixed-point weights (cubic_tab_kernel)
};

hipError_t launch_yaw_tables(uint32_t* packed, float* rows, int pw, int n_yaw, const double* yaw_rad,
                             hipStream_t st);
hipError_t launch_yaw_pack(uint32_t* packed, const float* rows, size_t n, hipStream_t st);
hipError_t launch_yaw_desc(YawDesc* desc, uint32_t* f4tab, const uint32_t* packed, int pw, int n_yaw,
                           hipStream_t st);
hipError_t launch_rot_map(float* U, float* V, int ow, int oh, const MapGeom& g, const float* R9, hipStream_t st);
hipError_t launch_pitch_map(float* U, float* V, int ow, int oh, const MapGeom& g, float c, float s,
                            hipStream_t st);
hipError_t launch_remap_views(const ViewsParams& P, int mapsrc, int mode, hipStream_t st);
hipError_t launch_remap_maps(const RemapParams& P, int cn, int interpolation, hipStream_t st);
hipError_t launch_cubic_tab(short* tab, hipStream_t st);
hipError_t launch_float_views(const ViewsParams& P, const double* yaw_rad, bool half, hipStream_t st);
// diagnostic build only (-DP2P_STAMPS): per-phase s_memtime sums of remap_views_kernel's pair loop
hipError_t read_stamps(unsigned long long* out16, bool reset);

}  // namespace p2p
#endif
