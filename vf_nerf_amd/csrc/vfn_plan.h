// vfn_plan.h — host-side layer plan shared by the weight packer and the fused MLP kernels.
//
// A network (vfn_net_geom, include/vfn.h) is lowered to a list of "hidden" layers that each
// produce up to 8 column tiles of 32 (MFMA 32x32x2 f32) into the 64x256 activation tile, plus one
// 16-column "head" (MFMA 16x16x4 f32) for the 3 output channels.  K is consumed in blocks of 8
// (two lane halves x four k's = one 16-byte fragment per lane), first from the activation tile
// ("act", previous layer / features) then from the auxiliary tile ("aux": positional encoding for the
// VF net; [p, PE(d), n] for the rendering net).
//
// Packed weights, per hidden layer:   tile[nt][kb][lane 0..63][j 0..3]  =  W'[n][k]
//        n = 32*nt + (lane & 31),  k = 8*kb + 4*(lane >> 5) + j          (k indexes act ++ aux)
//   followed by bias'[32*n_tiles].
// Head:  w[kb16][lane][j] = W'[n = lane & 15][k = 16*kb16 + 4*(lane >> 4) + j], then bias'[16].
// W', bias' = Linear with eval-mode BatchNorm (eps 1e-5) folded in, and 1/sqrt(2) for the skip layer.
//
// Backward pack (dX = dY * W'), per hidden layer that has an "act" input:
//   tileT[kt][nb][lane][j] = W'[n = 8*nb + 4*(lane >> 5) + j][k = 32*kt + (lane & 31)]
//   kt < nkb_act/4 (output = act columns of the layer input), nb < 4*n_tiles (reduction over n).
#pragma once
#include <stdint.h>
#include "../../include/vfn.h"

#define VFN_TM 64            // rows (points) per workgroup
#define VFN_AUX_K 40         // usable width of the aux tile (floats)

struct VfnLayerPlan {
    uint32_t w_off;      // float offset of tile[0][0][0][0] in the packed buffer
    uint32_t b_off;      // float offset of bias'
    uint16_t nkb_act;    // K blocks (of 8) taken from the act tile
    uint16_t nkb_aux;    // K blocks taken from the aux tile
    uint16_t n_tiles;    // 32-column output tiles (1..8)
    uint16_t ref_layer;  // index of the reference Linear
    uint32_t bw_off;     // float offset of the transposed ("backward") tiles in the bwd pack; VFN_NO_BWD if none
};
#define VFN_NO_BWD 0xffffffffu

struct VfnNetPlan {
    int32_t n_hidden;            // number of hidden (tile) layers
    int32_t feat_layer;          // 1: the last entry of hidden[] is the VF feature block of the final Linear (tanh)
    int32_t multires;
    int32_t pe_dim;              // 3 + 6*multires
    uint32_t head_w_off;
    uint32_t head_b_off;
    uint32_t head_nkb16;         // K/16 blocks of the head (always 16: K = 256)
    uint32_t total_floats;
    uint32_t total_bwd_floats;   // size of the backward pack
    VfnLayerPlan hidden[VFN_MAX_LAYERS];
};

// Returns VFN_OK or a negative status and fills `plan`.  `err` receives a message (may be NULL).
int vfn_make_plan(int net_kind, const vfn_net_geom* g, VfnNetPlan* plan, char* err, int errlen);
