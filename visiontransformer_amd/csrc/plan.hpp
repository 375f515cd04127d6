// Shapes, configuration checks and the parameter-arena layout shared by the inference and training
// entry points of libvitseg.
#pragma once
#include "kernels.hpp"

namespace vitseg {
namespace plan {

constexpr int MID = 256;
constexpr size_t ALIGN_F = 64;  // arena tensors start on 256-byte boundaries

inline size_t up(size_t v, size_t a) { return (v + a - 1) / a * a; }

struct Shape {
    int C, P, D, L, A, S, I, Cin, g, Np, N, Kp;
};

inline int check_config(const vitseg_config* c, Shape* s) {
    VITSEG_CHECK_ARG(c != nullptr, VITSEG_EINVAL, "config is null");
    VITSEG_CHECK_ARG(c->num_classes >= 1 && c->num_classes <= 255, VITSEG_ESHAPE, "num_classes %d out of [1,255]",
                     c->num_classes);
    VITSEG_CHECK_ARG(c->patch_size >= 4 && c->patch_size % 4 == 0, VITSEG_ESHAPE,
                     "patch_size %d must be a positive multiple of 4", c->patch_size);
    VITSEG_CHECK_ARG(c->image_size > 0 && c->image_size % c->patch_size == 0, VITSEG_ESHAPE,
                     "image_size %d is not a multiple of patch_size %d", c->image_size, c->patch_size);
    VITSEG_CHECK_ARG(c->num_heads > 0 && c->hidden_size == 64 * c->num_heads, VITSEG_ESHAPE,
                     "hidden_size %d / num_heads %d: this build needs head_dim 64", c->hidden_size, c->num_heads);
    VITSEG_CHECK_ARG(c->hidden_size <= 2048, VITSEG_ESHAPE, "hidden_size %d > 2048", c->hidden_size);
    VITSEG_CHECK_ARG(c->intermediate_size > 0 && c->intermediate_size % 4 == 0, VITSEG_ESHAPE,
                     "intermediate_size %d must be a multiple of 4", c->intermediate_size);
    VITSEG_CHECK_ARG(c->num_layers >= 1, VITSEG_ESHAPE, "num_layers %d", c->num_layers);
    VITSEG_CHECK_ARG(c->num_channels == 3, VITSEG_ESHAPE, "num_channels %d (reference: 3)", c->num_channels);
    s->C = c->num_classes;
    s->P = c->patch_size;
    s->D = c->hidden_size;
    s->L = c->num_layers;
    s->A = c->num_heads;
    s->S = c->image_size;
    s->I = c->intermediate_size;
    s->Cin = c->num_channels;
    s->g = s->S / s->P;
    s->Np = s->g * s->g;
    s->N = s->Np + 1;
    s->Kp = s->Cin * s->P * s->P;
    return VITSEG_OK;
}

inline size_t tensor_numel(const Shape& s, int t) {
    const size_t D = s.D, I = s.I;
    switch (t) {
        case VITSEG_T_CLS: return D;
        case VITSEG_T_POS: return (size_t)s.N * D;
        case VITSEG_T_PATCH_W: return D * s.Kp;
        case VITSEG_T_PATCH_B: return D;
        case VITSEG_T_LN1_W: case VITSEG_T_LN1_B: case VITSEG_T_LN2_W: case VITSEG_T_LN2_B: return D;
        case VITSEG_T_WQKV: return 3 * D * D;
        case VITSEG_T_BQKV: return 3 * D;
        case VITSEG_T_WO: return D * D;
        case VITSEG_T_BO: return D;
        case VITSEG_T_W1: return I * D;
        case VITSEG_T_B1: return I;
        case VITSEG_T_W2: return D * I;
        case VITSEG_T_B2: return D;
        case VITSEG_T_LNF_W: case VITSEG_T_LNF_B: return D;
        case VITSEG_T_HEAD0_W: return (size_t)MID * 9 * D;
        case VITSEG_T_HEAD0_B: return MID;
        case VITSEG_T_HEAD2_W: return (size_t)s.C * MID;
        case VITSEG_T_HEAD2_B: return s.C;
    }
    return 0;
}

inline bool per_layer(int t) { return t >= VITSEG_T_LN1_W && t <= VITSEG_T_B2; }

// Arena order = forward order: embeddings, layer 0 .. L-1, final norm, head.
struct Layout {
    size_t pre[4];                   // CLS, POS, PATCH_W, PATCH_B
    size_t layer0;                   // offset of layer 0
    size_t layer_stride;             // floats per layer
    size_t in_layer[VITSEG_T_B2 + 1];  // offset inside a layer, indexed by tensor id
    size_t post[VITSEG_T_COUNT];     // LNF.., indexed by tensor id
    size_t total;
};

inline Layout make_layout(const Shape& s) {
    Layout l{};
    size_t off = 0;
    for (int t = VITSEG_T_CLS; t <= VITSEG_T_PATCH_B; ++t) {
        l.pre[t] = off;
        off += up(tensor_numel(s, t), ALIGN_F);
    }
    l.layer0 = off;
    size_t lo = 0;
    for (int t = VITSEG_T_LN1_W; t <= VITSEG_T_B2; ++t) {
        l.in_layer[t] = lo;
        lo += up(tensor_numel(s, t), ALIGN_F);
    }
    l.layer_stride = lo;
    off += lo * s.L;
    for (int t = VITSEG_T_LNF_W; t < VITSEG_T_COUNT; ++t) {
        l.post[t] = off;
        off += up(tensor_numel(s, t), ALIGN_F);
    }
    l.total = off;
    return l;
}

inline size_t tensor_offset(const Layout& l, int t, int layer) {
    if (t <= VITSEG_T_PATCH_B) return l.pre[t];
    if (per_layer(t)) return l.layer0 + (size_t)layer * l.layer_stride + l.in_layer[t];
    return l.post[t];
}


}  // namespace plan
}  // namespace vitseg
