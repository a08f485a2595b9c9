// Native step executor for MiddleConvNet (model.py:202-281): the whole forward and the whole backward of
// the 3 Conv3d + 17 Conv2d + 3 ConvTranspose2d + heads stack as ONE C-ABI call each.
//
// Why: a train step is ~450 kernel launches of 2-2000 us.  Issued one by one from Python (ctypes +
// torch.empty/zeros per launch) the host needs ~9 ms per step — as long as the GPU needs to execute them.
// Here the layer table, the geometry of every launch, and a bump-allocated workspace arena (caller
// allocated, one buffer for the whole step, zeroed regions grouped so ONE memset per pass replaces ~100
// fill launches) live in C++; a launch costs a few microseconds of host time and Python makes two calls.
// The arena keeps the activations between vn_net_forward and vn_net_backward (the autograd "saved
// tensors").  No state lives in the library: the plan is a pure function of (config, K).
//
// The per-layer call sequences are exactly those of voxelnet_amd/engine.py (the Python reference
// orchestration, still used by the per-layer tests and the bf16x3 mode).
#include "common.h"
#include <new>
#include <string.h>

namespace {

constexpr int NL = 23;   // conv layers with BatchNorm, in execution order (net.layer_table)

struct Spec {
    int dim, cin, cout, k[3], s[3], p[3];
    bool transposed;
    int cin_fold;
};

struct Rows {            // channels-last rows (B,D,H,W,C) with element strides
    char *ptr;
    int dtype;           // VN_F32 / VN_BF16
    int B, D, H, W, C;
    int64_t sB, sD, sH, sW;
    int64_t M() const { return (int64_t)B * D * H * W; }
};

Rows dense_rows(void *ptr, int dtype, int B, int D, int H, int W, int C, int64_t width = 0) {
    if (!width) width = C;
    return Rows{static_cast<char *>(ptr), dtype, B, D, H, W, C, (int64_t)D * H * W * width, (int64_t)H * W * width,
                (int64_t)W * width, width};
}

void layer_table(int block1_stride, Spec *t) {
    int i = 0;
    auto c3 = [&](int ci, int co, int sd, int pd) { t[i++] = Spec{3, ci, co, {3, 3, 3}, {sd, 1, 1}, {pd, 1, 1}, false, 1}; };
    auto c2 = [&](int ci, int co, int s, int fold = 1) { t[i++] = Spec{2, ci, co, {1, 3, 3}, {1, s, s}, {0, 1, 1}, false, fold}; };
    auto d2 = [&](int ci, int co, int k, int s, int p) { t[i++] = Spec{2, ci, co, {1, k, k}, {1, s, s}, {0, p, p}, true, 1}; };
    c3(128, 64, 2, 1); c3(64, 64, 1, 0); c3(64, 64, 2, 1);                       // model.py:207-209
    c2(128, 128, block1_stride, 2); for (int j = 0; j < 4; ++j) c2(128, 128, 1);  // block1 (model.py:212-227)
    d2(128, 256, 3, 1, 1);                                                         // deconv1 (229)
    c2(128, 128, 2); for (int j = 0; j < 5; ++j) c2(128, 128, 1);                  // block2 (231-238)
    d2(128, 256, 2, 2, 0);                                                         // deconv2 (240)
    c2(128, 256, 2); for (int j = 0; j < 5; ++j) c2(256, 256, 1);                  // block3 (242-249)
    d2(256, 256, 4, 4, 0);                                                         // deconv3 (251)
}
// indices into the table
constexpr int L_M2 = 2, L_B1 = 3, L_D1 = 8, L_B2 = 9, L_D2 = 15, L_B3 = 16, L_D3 = 22;

void out_dims(const Spec &sp, const int in[3], int out[3]) {
    for (int a = 0; a < 3; ++a)
        out[a] = sp.transposed ? (in[a] - 1) * sp.s[a] - 2 * sp.p[a] + sp.k[a] : (in[a] + 2 * sp.p[a] - sp.k[a]) / sp.s[a] + 1;
}

struct Arena {
    char *base;
    size_t off, cap;
    void *take(size_t bytes) {
        void *r = base ? base + off : nullptr;
        off += vn_align(bytes, 256);
        return r;
    }
};

// everything the two passes need, laid out by one deterministic walk (plan == allocation)
struct Plan {
    Spec spec[NL];
    int in_dims[NL][3], odims[NL][3];
    int esz, adt;                // operand element size / activation (MFMA operand) dtype
    int ydt, gdt, cdt;           // storage dtype of the conv outputs y / of the data gradients dx / of the concat gradient
                                 // (bf16 mode: bf16 unless cfg->grad_storage promotes them to fp32)
    // per layer
    void *wp_f[NL], *wp_d[NL];   // packed weights (forward / data-gradient orientation)
    Rows y[NL], a[NL];           // conv output (pre-BN), activation
    float *slab[NL]; int64_t slab_rows[NL];
    double *fsums[NL];           // forward bn_stats sums (deconvs)
    float *stats[NL];
    // backward
    float *bslab[NL]; int64_t bslab_rows[NL];
    float *coef[NL];
    Rows dy[NL], dx[NL];
    float *dwp[NL]; size_t dwp_bytes[NL];   // row-chunk partials of the weight gradient (vn_conv_wgrad_partials)
    // heads
    bool x3;          // cfg->mode == 2 ("fp32x3"): fp32 storage, the convolutions' and weight gradients' products as three bf16 MFMAs
    bool x3_store;    // fp32x3: activations / gradients that feed convolutions and weight gradients are stored split (VN_F32X3S)
    bool m2_passes;   // x3_store: middle_layer.2's weight gradient as three in-place bf16 passes of the nine-tap patch kernel
                      // (its geometry class only: images >= 128 x 128; smaller grids take the split-operand row form)
    bool round_act;   // fp32 mode diagnostic (grad_storage & 16): activations rounded to bf16 VALUES, everything else exact fp32
    bool exact_heads; Rows d_rows32;   // grad_storage & 8: the fp32 logit gradient beside the bf16 one
    void *hwp_f, *hwp_d; Rows hy, cat, d_rows, d_cat; float *hdwp, *hcs; size_t hdwp_bytes; void *hcs_ws; size_t hcs_ws_bytes;
    // sparse first layer
    int64_t *alist; int32_t *acount; int64_t acap; void *aws; size_t aws_bytes;
    float *dtot; void *dtot_ws; size_t dtot_ws_bytes; bool list_bwd;   // first layer's BatchNorm backward from the active sites only
    void *drows; bool sparse_w1;   // middle_layer.1's weight gradient: [acap][64] rows a0 - const at the active sites + a rank-1 term
    int32_t *igrid; float *rbP;   // rulebook: voxel index grid over the input cells, P[v][tap][64]
    // zeroed regions
    char *zf_begin, *zf_end, *zb_begin, *zb_end;
    size_t bytes;
    int hf, wf;
};

vnConv geom(const Rows &src, const int row_dims[3], int Cs, int Cr, const int k[3], const int mul[3], const int tmul[3],
            const int pad[3], const int div[3], const int64_t ostr[4]) {
    vnConv g;
    memset(&g, 0, sizeof(g));
    g.dtype = src.dtype;
    g.B = src.B; g.Ds = src.D; g.Hs = src.H; g.Ws = src.W;
    g.Dr = row_dims[0]; g.Hr = row_dims[1]; g.Wr = row_dims[2];
    g.Cs = Cs; g.src_wrap = 0; g.Cr = Cr;
    g.kD = k[0]; g.kH = k[1]; g.kW = k[2];
    g.mulD = mul[0]; g.mulH = mul[1]; g.mulW = mul[2];
    g.tmulD = tmul[0]; g.tmulH = tmul[1]; g.tmulW = tmul[2];
    g.padD = pad[0]; g.padH = pad[1]; g.padW = pad[2];
    g.divD = div[0]; g.divH = div[1]; g.divW = div[2];
    g.src_sB = src.sB; g.src_sD = src.sD; g.src_sH = src.sH; g.src_sW = src.sW;
    g.out_sB = ostr[0]; g.out_sD = ostr[1]; g.out_sH = ostr[2]; g.out_sW = ostr[3];
    return g;
}
const int ONE[3] = {1, 1, 1}, NEG[3] = {-1, -1, -1};
struct Plan;
int32_t wdt(const Plan &P);
const vnConv *cx(const Plan &P, vnConv &g);   // the geometry as the conv / weight-gradient entry points get it (fp32x3 mode: VN_F32X3)

vnConv fwd_geom(const Spec &sp, const Rows &x, const int od[3], const Rows &out) {
    const int64_t os[4] = {out.sB, out.sD, out.sH, out.sW};
    if (sp.transposed) {
        const int np[3] = {-sp.p[0], -sp.p[1], -sp.p[2]};
        return geom(x, od, sp.cin, sp.cout, sp.k, ONE, NEG, np, sp.s, os);
    }
    return geom(x, od, sp.cin, sp.cout, sp.k, sp.s, ONE, sp.p, ONE, os);
}

// weight-gradient launch geometry of layer l (dense form): x = the layer's input activation rows
vnConv wgrad_geom(const Plan &P, int l, const Rows &x) {
    const Spec &sp = P.spec[l];
    const Rows &dy = P.dy[l];
    if (sp.transposed) {
        const int64_t rs[4] = {x.sB, x.sD, x.sH, x.sW};
        return geom(dy, P.in_dims[l], sp.cout, sp.cin, sp.k, sp.s, ONE, sp.p, ONE, rs);
    }
    const int64_t rs[4] = {dy.sB, dy.sD, dy.sH, dy.sW};
    return geom(x, P.odims[l], sp.cin, sp.cout, sp.k, sp.s, ONE, sp.p, ONE, rs);
}

const vnConv *cx(const Plan &P, vnConv &g) {
    if (P.x3 && g.dtype == VN_F32) g.dtype = VN_F32X3;
    return &g;
}
// dtype the packed conv weights are made in: the storage dtype; fp32x3 mode: split once by the pack launch (VN_F32X3)
int32_t wdt(const Plan &P) { return P.x3 ? (int32_t)VN_F32X3 : (int32_t)P.adt; }
// fp32x3, middle_layer.2's weight gradient: in fp32 the row form re-stages both operands once per tap (27 x: 2.4 ms); the
// operands are cast to [hi|lo] bf16 rows once (row width 2 C) and the bf16 patch kernel runs three times over them —
// hi.hi, lo(src).hi, hi(src).lo(rows) — into consecutive partial slabs that the unpack sums like row chunks
vnConv x3_wgrad_geom(const Plan &P, int l, void *src_hl, void *rows_hl, int src_lo, int rows_lo, int B_plan = 0) {
    const Spec &sp = P.spec[l];
    const int B = B_plan > 0 ? B_plan : P.dy[l].B;      // (B_plan: while the plan is still being laid out)
    Rows xs = dense_rows(src_hl, VN_BF16, B, P.in_dims[l][0], P.in_dims[l][1], P.in_dims[l][2], sp.cin, 2 * sp.cin);
    Rows ds = dense_rows(rows_hl, VN_BF16, B, P.odims[l][0], P.odims[l][1], P.odims[l][2], sp.cout, 2 * sp.cout);
    if (xs.ptr) xs.ptr += (size_t)(src_lo ? sp.cin : 0) * 2;
    if (ds.ptr) ds.ptr += (size_t)(rows_lo ? sp.cout : 0) * 2;
    const int64_t rs[4] = {ds.sB, ds.sD, ds.sH, ds.sW};
    return geom(xs, P.odims[l], sp.cin, sp.cout, sp.k, sp.s, ONE, sp.p, ONE, rs);
}
int m0_bn_knob();
bool make_plan(const vnNetConfig *c, int64_t K, char *base, Plan *P) {
    // depth: every D whose three Conv3d layers (model.py:207-209: stride 2 / pad 1, stride 1 / no pad, stride 2 / pad 1)
    // end at depth 2, the BEV fold of model.py:262 — D = 9 ... 12 (the reference's grids are all D = 10; round 4 lifted the
    // D == 10 restriction: nothing below depends on it, the walk checks the folded depth)
    if (!c || c->B <= 0 || c->D < 9 || c->D > 12 || c->H <= 0 || c->W <= 0 || (c->H & 7) || (c->W & 7)) return false;
    if (c->mode < 0 || c->mode > 2) return false;
    if (c->block1_stride != 1 && c->block1_stride != 2) return false;
    memset(P, 0, sizeof(*P));
    layer_table(c->block1_stride, P->spec);
    const bool f32 = c->mode != 0;      // modes 1 (fp32) and 2 (fp32x3) store everything in fp32
    P->x3 = c->mode == 2;
    P->x3_store = P->x3;      // (round 4's in-kernel splits of every operand left the library with their knob: DESIGN_HISTORY.md)
    P->m2_passes = false;     // (set below, once the dims are known)
    P->esz = f32 ? 4 : 2;
    P->adt = f32 ? VN_F32 : VN_BF16;
    if (c->grad_storage & ~31) return false;
    const int gs = f32 ? 0 : (c->grad_storage & 15);
    P->round_act = f32 && (c->grad_storage & 16);
    P->ydt = (gs & 4) ? VN_F32 : P->adt;
    P->gdt = (gs & 2) ? VN_F32 : P->adt;
    P->cdt = (gs & 1) ? VN_F32 : P->adt;
    P->exact_heads = f32 ? false : (gs & 8) != 0;
    const int B = c->B;
    Arena A{base, 0, 0};
    auto rows_new = [&](int dtype, const int d[3], int C, int64_t width = 0) {
        if (!width) width = C;
        void *p = A.take((size_t)B * d[0] * d[1] * d[2] * width * (dtype == VN_BF16 ? 2 : 4));
        return dense_rows(p, dtype, B, d[0], d[1], d[2], C, width);
    };
    // ---- dims walk
    int cur[3] = {c->D, c->H, c->W};
    int x1[3] = {0, 0, 0}, x2[3] = {0, 0, 0};
    for (int l = 0; l < NL; ++l) {
        int in[3] = {cur[0], cur[1], cur[2]};
        if (l == L_B1) { in[0] = 1; }                       // BEV: (B,2,H,W,64) -> (B,1,H,W,128)
        if (l == L_B2 || l == L_D1) { in[0] = x1[0]; in[1] = x1[1]; in[2] = x1[2]; }
        if (l == L_B3 || l == L_D2) { in[0] = x2[0]; in[1] = x2[1]; in[2] = x2[2]; }
        memcpy(P->in_dims[l], in, sizeof(in));
        out_dims(P->spec[l], in, P->odims[l]);
        if (!P->spec[l].transposed) memcpy(cur, P->odims[l], sizeof(cur));
        if (l == L_D1 - 1) memcpy(x1, cur, sizeof(cur));
        if (l == L_D2 - 1) memcpy(x2, cur, sizeof(cur));
        if (l == L_M2 && P->odims[l][0] != 2) return false;
    }
    P->hf = P->odims[L_D1][1];
    P->wf = P->odims[L_D1][2];
    for (int l : {L_D2, L_D3})
        if (P->odims[l][1] != P->hf || P->odims[l][2] != P->wf) return false;
    if (P->x3_store) {
        const vnConv gb = x3_wgrad_geom(*P, L_M2, nullptr, nullptr, 0, 0, c->B);      // (P->dy is not laid out yet)
        P->m2_passes = vn_conv_wgrad_plan_id(&gb, 0, 0) == 200;
    }
    // ---- packed weights
    for (int l = 0; l < NL; ++l) {
        const Spec &sp = P->spec[l];
        const size_t n = (size_t)sp.k[0] * sp.k[1] * sp.k[2] * sp.cin * sp.cout * P->esz;
        P->wp_f[l] = A.take(n);
        P->wp_d[l] = A.take(n);
    }
    P->hwp_f = A.take((size_t)16 * 768 * P->esz);
    P->hwp_d = A.take((size_t)16 * 768 * P->esz);
    // ---- forward-zeroed region (deconv bn sums)
    P->zf_begin = base ? base + A.off : nullptr;
    for (int l = 0; l < NL; ++l) P->fsums[l] = P->spec[l].transposed ? (double *)A.take(2 * 256 * sizeof(double)) : nullptr;
    P->zf_end = base ? base + A.off : nullptr;
    // ---- concat buffer and activations
    const int fm[3] = {1, P->hf, P->wf};
    P->cat = rows_new(P->adt, fm, 768);
    for (int l = 0; l < NL; ++l) {
        const Spec &sp = P->spec[l];
        P->y[l] = rows_new(P->ydt, P->odims[l], sp.cout);
        P->stats[l] = (float *)A.take(4 * 256 * sizeof(float));
        const int64_t M = P->y[l].M();
        P->slab_rows[l] = 0;
        P->slab[l] = nullptr;
        {
            // the conv kernel picks its tiling (= slab granularity) from the launch geometry
            const Rows xin = dense_rows(nullptr, P->adt, B, P->in_dims[l][0], P->in_dims[l][1], P->in_dims[l][2], sp.cin);
            const vnConv sg = fwd_geom(sp, xin, P->odims[l], P->y[l]);
            P->slab_rows[l] = vn_conv_stats_slab_rows(&sg);
            (void)M;
            P->slab[l] = (float *)A.take((size_t)P->slab_rows[l] * 2 * sp.cout * sizeof(float));
        }
        if (sp.transposed) {   // activation = channel slice of the concat: cat([d3,d2,d1]) (model.py:271-273)
            const int off = l == L_D3 ? 0 : (l == L_D2 ? 256 : 512);
            Rows r = P->cat;
            r.ptr += (size_t)off * P->esz;
            r.C = 256;
            P->a[l] = r;
        } else if (l == L_M2) {   // BEV fold (model.py:262): stored channel d*64 + c
            const int bd[3] = {1, P->odims[l][1], P->odims[l][2]};
            P->a[l] = rows_new(P->adt, bd, 128);
        } else {
            P->a[l] = rows_new(P->adt, P->odims[l], sp.cout);
        }
        // fp32x3: the activations that only convolutions and weight gradients read are STORED split (VN_F32X3S: per 8
        // channels their hi bf16 parts, then their lo parts — same bytes as fp32), written that way by the BatchNorm apply:
        // no split work is left in the kernels that consume them.  Not the first layer's output (its sparse routes read it
        // as numbers) and not the deconvs' (the concat feeds the 16-column heads)
        if (P->x3_store && l >= 1 && !sp.transposed) P->a[l].dtype = VN_F32X3S;
    }
    P->hy = rows_new(VN_F32, fm, 16);
    // ---- sparse first layer
    P->acap = 0;
    if (c->sparse_first) {
        int64_t per = 1;
        for (int a = 0; a < 3; ++a) per *= (P->spec[0].k[a] + P->spec[0].s[a] - 1) / P->spec[0].s[a];
        int64_t cap = K * per;
        const int64_t M0 = P->y[0].M();
        if (cap > M0) cap = M0;
        if (cap < 1) cap = 1;
        P->acap = cap;
        P->alist = (int64_t *)A.take((size_t)cap * 4 * sizeof(int64_t));
        P->acount = (int32_t *)A.take(256);
        const int od[3] = {P->odims[0][0], P->odims[0][1], P->odims[0][2]};
        Rows dummy = dense_rows(nullptr, P->adt, B, c->D, c->H, c->W, 128);
        vnConv g = fwd_geom(P->spec[0], dummy, od, P->y[0]);
        P->aws_bytes = vn_active_sites_workspace_bytes(&g);
        P->aws = A.take(P->aws_bytes);
        P->igrid = (int32_t *)A.take(sizeof(int32_t) * (size_t)B * c->D * c->H * c->W);
        P->rbP = (float *)A.take(sizeof(float) * (size_t)(K > 0 ? K : 1) * 27 * 64);
        P->slab_rows[0] = vn_rulebook_slab_rows(cap);
        P->slab[0] = (float *)A.take((size_t)P->slab_rows[0] * 2 * P->spec[0].cout * sizeof(float));   // (its own size)
        P->dtot = (float *)A.take(256 * sizeof(float));
        P->dtot_ws_bytes = vn_dgrad_total_workspace_bytes(P->spec[1].cout);
        P->dtot_ws = A.take(P->dtot_ws_bytes);
    }
    {   // middle_layer.1's data gradient at middle_layer.0's active sites only + box sums for the BatchNorm totals: needs
        // its 3x3 taps with stride 1 / padding 1 in H/W and every depth tap in range (stride 1, no padding in D)
        const Spec &s1 = P->spec[1];
        P->list_bwd = c->sparse_first && !P->round_act && (m0_bn_knob() & 8) && !s1.transposed && s1.k[1] == 3 && s1.k[2] == 3 && s1.k[0] <= 3 &&
                      s1.s[0] == 1 && s1.s[1] == 1 && s1.s[2] == 1 && s1.p[0] == 0 && s1.p[1] == 1 && s1.p[2] == 1 &&
                      s1.cin == P->spec[0].cout && s1.cin == 64 && s1.cout <= 256 &&
                      // worth it while the active sites are a minority (their number is only known on the device: the list's
                      // capacity K * 18 is the bound): the row-list launch re-gathers every tap, the dense kernel stages
                      // halo patches — at 160k voxels (BASELINE configs[4]) the dense route is 3 % faster
                      P->acap * 10 <= P->y[0].M() * 3;
        P->sparse_w1 = P->list_bwd && (m0_bn_knob() & 16);
        P->drows = P->sparse_w1 ? A.take((size_t)P->acap * 64 * P->esz) : nullptr;
    }
    // ---- backward buffers
    P->zb_begin = base ? base + A.off : nullptr;
    P->hcs = (float *)A.take(64 * sizeof(float));
    P->hcs_ws_bytes = vn_col_sums_workspace_bytes((int64_t)B * P->hf * P->wf, 16);
    P->hcs_ws = A.take(P->hcs_ws_bytes);
    P->zb_end = base ? base + A.off : nullptr;
    for (int l = 0; l < NL; ++l) {
        const Spec &sp = P->spec[l];
        P->coef[l] = (float *)A.take(3 * 256 * sizeof(float));
        P->bslab_rows[l] = vn_bn_bwd_slab_rows(P->y[l].M(), sp.cout);
        size_t bslab_floats = (size_t)P->bslab_rows[l] * 2 * sp.cout;
        if (l == 0 && P->list_bwd) {   // the list-based reduce writes THREE columns per slab row (vn_bn_bwd_reduce_list)
            const size_t lf = (size_t)vn_bn_bwd_list_slab_rows(P->acap, sp.cout) * 3 * sp.cout;
            if (lf > bslab_floats) bslab_floats = lf;
        }
        P->bslab[l] = (float *)A.take(bslab_floats * sizeof(float));
        P->dy[l] = rows_new(P->adt, P->odims[l], sp.cout);
        // (fp32x3: dy of every layer whose consumers read it as a convolution / weight-gradient operand is stored split by the
        //  BatchNorm backward apply.  Layer 1 on its sparse route too: its data gradient is a row-list launch, its weight
        //  gradient reads dy against the split-stored rows of vn_act_delta_rows, the box sums add hi + lo.  Not layer 0 (its
        //  rows operand is the fp32 voxel features) and not layer 1 on the dense route (its source is layer 0's fp32 output))
        if (P->x3_store && (l >= 2 || (l == 1 && P->sparse_w1))) P->dy[l].dtype = VN_F32X3S;
        // data gradient buffer of the layer's input (shared where two consumers accumulate)
        P->dx[l] = Rows{};
    }
    // dx buffers: one per distinct layer input
    for (int l = 1; l < NL; ++l) {
        if (l == L_D1 || l == L_D2) continue;             // written into the block output's dx (accumulate)
        const Spec &sp = P->spec[l];
        P->dx[l] = rows_new(P->gdt, P->in_dims[l], sp.cin);
    }
    P->dx[L_D1] = P->dx[L_B2];
    P->dx[L_D2] = P->dx[L_B3];
    if (!c->sparse_first) P->dx[0] = rows_new(P->adt, P->in_dims[0], 128);
    P->d_rows = rows_new(P->adt, fm, 16);
    P->d_cat = rows_new(P->cdt, fm, 768);
    P->d_rows32 = P->exact_heads ? rows_new(VN_F32, fm, 16) : Rows{};
    {   // weight-gradient partials: one slab per layer, summed by the batched unpack at the end of a segment
        auto ask = [&](const int rd[3], int Cs, int Cr, const int k[3], int64_t n_rows) {
            vnConv q{};
            q.dtype = wdt(*P);               // (the operand dtype of the launch: the chunk count depends on it)
            q.B = B; q.Dr = rd[0]; q.Hr = rd[1]; q.Wr = rd[2]; q.Cs = Cs; q.Cr = Cr;
            q.kD = k[0]; q.kH = k[1]; q.kW = k[2];
            return vn_conv_wgrad_workspace_bytes(&q, 0, n_rows);
        };
        for (int l = 0; l < NL; ++l) {
            const Spec &sp = P->spec[l];
            size_t b;
            if (l == 0 && c->sparse_first) {
                b = ask(P->in_dims[0], sp.cout, sp.cin, sp.k, K > 0 ? K : 1);
            } else if (l == 1 && P->sparse_w1) {
                b = ask(P->in_dims[1], sp.cout, sp.cin, sp.k, P->acap);
            } else if (l == L_M2 && P->x3_store && P->m2_passes) {
                // three passes of the bf16 nine-tap patch kernel over the split-stored operands in place
                const vnConv gb = x3_wgrad_geom(*P, l, nullptr, nullptr, 0, 0);
                b = 3 * vn_conv_wgrad_workspace_bytes(&gb, 0, 0);
            } else {   // the real launch geometry: the kernel variant (and its chunking) is chosen from it
                const int src_l = l == 0 ? -1 : (l == L_D1 || l == L_B2) ? L_D1 - 1 : (l == L_D2 || l == L_B3) ? L_D2 - 1 : l - 1;
                const Rows xin = dense_rows(nullptr, src_l < 0 ? P->adt : P->a[src_l].dtype, B, P->in_dims[l][0], P->in_dims[l][1],
                                            P->in_dims[l][2], sp.cin);
                vnConv gw = wgrad_geom(*P, l, xin);
                b = vn_conv_wgrad_workspace_bytes(cx(*P, gw), 0, 0);
            }
            P->dwp_bytes[l] = b;
            P->dwp[l] = (float *)A.take(b);
        }
        const int k1[3] = {1, 1, 1};
        P->hdwp_bytes = ask(fm, 768, 16, k1, 0);
        P->hdwp = (float *)A.take(P->hdwp_bytes);
    }
    P->bytes = A.off;
    return true;
}

#define RT(call) do { int rc_ = (call); if (rc_ != VN_OK) return rc_; } while (0)
// diagnostic VN_DUP (bit = VN_T_* kind): issue every launch of that kind TWICE — the step-time difference is the in-step
// marginal cost of those launches on their dependency chain (only meaningful for the idempotent BatchNorm passes: kinds 3-6)
static int dup_mask() {
    static const int v = vn_knob("VN_DUP", 0);
    return v;
}
// -DVN_DIAG_SKIP builds only (tools/skip_costs.sh; never the product library): VN_SKIP (bit = VN_T_* kind) drops every
// launch of that kind — the results are wrong, the step-time difference is what that family costs INSIDE the step
// (its place on the two streams, its HBM traffic beside the other stream), which the summed kernel times do not say
#ifdef VN_DIAG_SKIP
static int skip_mask() {
    static const int v = vn_knob("VN_SKIP", 0);
    return v;
}
#define VN_SKIPPED(kind) (skip_mask() & (1 << (kind)))
#else
#define VN_SKIPPED(kind) 0
#endif
// a launch of the executor, bracketed by timing events on ITS stream when the context is in timing mode
#define RTT(kind, layer, flops, bytes, st, call) do {                                                           \
        vnTimeSlot *ts_ = nullptr;                                                                              \
        if (net->timing && !(ts_ = net->time_begin((kind), (layer), (double)(flops), (double)(bytes), vn_stream(st)))) \
            return VN_EINVAL;                                                                                   \
        if (!VN_SKIPPED(kind)) {                                                                                \
            const int rc_ = (call);                                                                             \
            if (rc_ != VN_OK) {                                                                                 \
                if (ts_) --net->t_used;   /* e1 was never recorded: drop the slot, vn_net_timing_read would fail on it */ \
                return rc_;                                                                                     \
            }                                                                                                   \
        }                                                                                                       \
        if (dup_mask() & (1 << (kind))) RT(call);   /* diagnostic VN_DUP: the launch twice (all BatchNorm kinds are idempotent) */ \
        if (ts_) VN_HIP(hipEventRecord(ts_->e1, vn_stream(st)));                                                \
    } while (0)

enum { T_CONV_FWD = VN_T_CONV_FWD, T_CONV_DGRAD = VN_T_CONV_DGRAD, T_WGRAD = VN_T_WGRAD, T_BN_APPLY = VN_T_BN_APPLY,
       T_BN_BWD_REDUCE = VN_T_BN_BWD_REDUCE, T_BN_BWD_APPLY = VN_T_BN_BWD_APPLY, T_BN_FINALIZE = VN_T_BN_FINALIZE,
       T_UNPACK = VN_T_UNPACK, T_PACK = VN_T_PACK, T_FIRST = VN_T_FIRST, T_MISC = VN_T_MISC };

// Algorithmic work of a layer: 2 x MACs of the reference's layer definition; the data gradient and the
// weight gradient of a layer count the same FLOPs as its forward.  rows = output sites (input sites for a deconv).
double layer_flops(const Spec &sp, const int in_dims[3], const int odims[3], int B) {
    const int *d = sp.transposed ? in_dims : odims;
    return 2.0 * B * d[0] * d[1] * d[2] * (double)sp.cin * sp.cout * sp.k[0] * sp.k[1] * sp.k[2];
}
double rows_bytes(const Rows &r) { return (double)r.M() * r.C * (r.dtype == VN_BF16 ? 2 : 4); }
// algorithmic bytes of the gradient unpack: every weight gradient read once and written once (the row-chunk partials it
// also sums are an implementation artefact: not counted)
double unpack_bytes(const vnUnpackJob *j, int n) {
    double b = 0.0;
    for (int i = 0; i < n; ++i) b += 8.0 * j[i].c_out * j[i].c_in * j[i].taps;
    return b;
}

}  // namespace

// Executor context (vn_net_create / vn_net_destroy): the HIP events the two-stream schedule needs.  Caller-owned, one
// per executor instance (an RPN3D module on one device); the library itself keeps no mutable state.
//   bucket_ev[b][0/1]: "gradients of parameter group b are final" of the most recent single-call backward with
//     cfg->bucket_events — [0] recorded on the main stream (BatchNorm gradients, zeroed conv biases), [1] on the side
//     stream (weight gradients unpacked); vn_net_wait_bucket makes a communication stream wait for both.
//   ring: timing-less events for the fork / join of the side stream (re-recording an event that an earlier
//     hipStreamWaitEvent has consumed is well defined: the wait captured the record that preceded it).
// Calls that share a context must not run concurrently on the host (they never do: one step = one thread).
//   timing (vn_net_timing_begin / _read): optional HIP-event pairs around every launch of the executor, on the stream
//     the kernel is launched on — bench.py's live per-kernel measurement of the SAME native path it times.
struct vnTimeSlot {
    hipEvent_t e0, e1;
    int32_t kind, layer;
    double flops, bytes;
};
struct vnNet {
    hipEvent_t bucket_ev[4][2];
    // vn_net_prepare on a stream of its own records [0] behind what the FIRST layer needs (its packed weights, the site
    // list, the index grid) and [1] behind the rest of the weight packing; vn_net_forward (cfg->prepared) waits for [0]
    // at its start and for [1] in front of the second layer, so the ~65 us of packing are off the start of the step
    hipEvent_t prep_ev[2];
    bool prep_recorded;
    bool prep_first_done;   // vn_net_prepare phase 1 (cfg->prepared == 1) has been issued: phase 2 issues only the rest
    // what phase 1 was issued FOR: phase 2 and the vn_net_forward that follows must name the same step (round-3 advisor:
    // hidden state without a check let a stale first phase stand in for a later step's)
    const void *prep_ws, *prep_coord;
    int64_t prep_K;
    int32_t prep_dims[9];   // B, D, H, W + training, mode, block1_stride, sparse_first, grad_storage (round-4 advisor: a prepare
                            // for an eval / other-mode step must not stand in for a training step on the same arena)
    // the data-gradient orientation of the packed weights is first read by the BACKWARD: vn_net_prepare leaves those jobs
    // here and vn_net_forward issues them on the side stream behind deconv2, beside block3's small images (CUs to spare)
    // instead of beside the first layers (where the launch cost 0.03 ms of step time)
    vnPackJob deferred_pack[NL + 1];
    int n_deferred;
    // vn_net_step's loss pass has already written the heads' gradient rows (vn_rpn_loss_fwd_bwd_rows) into this arena's
    // d_rows: the vn_net_backward that follows skips its vn_heads_bwd launch (consumed there)
    const void *heads_rows_ready;
    hipEvent_t ring[64];
    unsigned next;
    hipEvent_t next_event() { return ring[next++ & 63]; }
    bool timing;
    int32_t t_cap, t_made, t_used;
    vnTimeSlot *slots;
    // start of a timed launch: the slot whose e1 the caller records after the launch, or NULL (table full / HIP error)
    vnTimeSlot *time_begin(int kind, int layer, double flops, double bytes, hipStream_t st) {
        if (t_used >= t_cap) return nullptr;
        vnTimeSlot &t = slots[t_used];
        if (t_used >= t_made) {
            if (hipEventCreate(&t.e0) != hipSuccess || hipEventCreate(&t.e1) != hipSuccess) return nullptr;
            t_made = t_used + 1;
        }
        t.kind = kind; t.layer = layer; t.flops = flops; t.bytes = bytes;
        if (hipEventRecord(t.e0, st) != hipSuccess) return nullptr;
        ++t_used;
        return &t;
    }
};

extern "C" int vn_net_create(vnNet **out) {
    VN_CHECK_ARG(out);
    *out = nullptr;
    vnNet *n = new (std::nothrow) vnNet();
    if (!n) return VN_EINVAL;
    memset(n, 0, sizeof(*n));
    hipError_t err = hipSuccess;
    for (int b = 0; b < 4 && err == hipSuccess; ++b)
        for (int w = 0; w < 2 && err == hipSuccess; ++w) err = hipEventCreateWithFlags(&n->bucket_ev[b][w], hipEventDisableTiming);
    for (int i = 0; i < 64 && err == hipSuccess; ++i) err = hipEventCreateWithFlags(&n->ring[i], hipEventDisableTiming);
    for (int i = 0; i < 2 && err == hipSuccess; ++i) err = hipEventCreateWithFlags(&n->prep_ev[i], hipEventDisableTiming);
    if (err != hipSuccess) {
        vn_net_destroy(n);
        return (int)err;
    }
    *out = n;
    return VN_OK;
}

extern "C" int vn_net_destroy(vnNet *n) {
    if (!n) return VN_OK;
    for (int b = 0; b < 4; ++b)
        for (int w = 0; w < 2; ++w)
            if (n->bucket_ev[b][w]) (void)hipEventDestroy(n->bucket_ev[b][w]);
    for (int i = 0; i < 64; ++i)
        if (n->ring[i]) (void)hipEventDestroy(n->ring[i]);
    for (int i = 0; i < 2; ++i)
        if (n->prep_ev[i]) (void)hipEventDestroy(n->prep_ev[i]);
    for (int i = 0; i < n->t_made; ++i) {
        (void)hipEventDestroy(n->slots[i].e0);
        (void)hipEventDestroy(n->slots[i].e1);
    }
    delete[] n->slots;
    delete n;
    return VN_OK;
}

// From now on every launch of vn_net_prepare / vn_net_forward / vn_net_backward on this context is bracketed by two
// timing events on its stream, up to max_records launches (further calls fail with VN_EINVAL rather than drop records).
extern "C" int vn_net_timing_begin(vnNet *n, int32_t max_records) {
    VN_CHECK_ARG(n && max_records > 0 && max_records <= (1 << 20));
    if (max_records > n->t_cap) {
        vnTimeSlot *s = new (std::nothrow) vnTimeSlot[max_records];
        if (!s) return VN_EINVAL;
        memset(s, 0, sizeof(vnTimeSlot) * (size_t)max_records);
        if (n->slots) memcpy(s, n->slots, sizeof(vnTimeSlot) * (size_t)n->t_made);
        delete[] n->slots;
        n->slots = s;
        n->t_cap = max_records;
    }
    n->t_used = 0;
    n->timing = true;
    return VN_OK;
}

// Stops the timing, waits for the recorded events (the caller has queued nothing behind them that could deadlock) and
// returns one record per launch in issue order: duration in milliseconds between the two events of the launch.
extern "C" int vn_net_timing_read(vnNet *n, vnTimingRecord *out, int32_t cap, int32_t *count) {
    VN_CHECK_ARG(n && count && (out || cap == 0));
    n->timing = false;
    const int m = n->t_used < cap ? n->t_used : cap;
    for (int i = 0; i < m; ++i) {
        vnTimeSlot &t = n->slots[i];
        VN_HIP(hipEventSynchronize(t.e1));
        float ms = 0.f, start = 0.f;
        VN_HIP(hipEventElapsedTime(&ms, t.e0, t.e1));
        // start of the launch's bracket relative to the first record's (events of different streams share one clock): a
        // two-stream timeline of the step (tools/step_timeline.py)
        if (i > 0 && hipEventElapsedTime(&start, n->slots[0].e0, t.e0) != hipSuccess) start = 0.f;
        out[i] = vnTimingRecord{t.kind, t.layer, ms, start, t.flops, t.bytes};
    }
    *count = n->t_used;
    n->t_used = 0;
    return VN_OK;
}

namespace {

// zero up to NL small fp32 vectors in one launch (the conv-bias gradients: exactly 0 before a train-mode BatchNorm)
struct ZeroJobs {
    int32_t n;
    int32_t len[NL];
    float *ptr[NL];
};
__global__ void __launch_bounds__(256) k_zero_many(const ZeroJobs z) {
    float *p = z.ptr[blockIdx.x];
    for (int i = threadIdx.x; i < z.len[blockIdx.x]; i += 256) p[i] = 0.f;
}
int zero_many(const ZeroJobs &z, hipStream_t st) {
    k_zero_many<<<z.n, 256, 0, st>>>(z);
    VN_LAUNCH_STATUS();
    return VN_OK;
}

// First middle layer (tuning aid VN_M0_BN, bits; default 31 = all on): 1 flagged forward apply (k_bn_apply<true>: the ~90 %
// of rows without a flag are written without reading y), 2 flagged backward reduce, 4 list-based backward apply, 8 the
// BatchNorm backward from the activation gradient at the active sites only (middle_layer.1's data gradient as a row-list
// launch + box sums), 16 middle_layer.1's weight gradient from the active sites' rows a0 - const + a rank-1 term
int m0_bn_knob() {
    static const int v = vn_knob("VN_M0_BN", 31);
    return v;
}

int heads_stream_on() {   // tuning aid VN_HEADS_STREAM=0: the heads through k_gather_gemm as before
    static const int v = vn_knob("VN_HEADS_STREAM", 1);
    return v;
}
int fuse_bwd_reduce_on() {   // tuning aid VN_FUSE_BWD_REDUCE=0: every BatchNorm backward reduction as its own launch
    static const int v = vn_knob("VN_FUSE_BWD_REDUCE", 1);
    return v;
}
int bn_apply_rows(const Rows &y, const float *stats, const Rows &a, int C, int relu, vnStream st) {
    return vn_bn_apply(y.ptr, (vnDtype)y.dtype, y.sW, y.M(), C, stats, relu, a.ptr, (vnDtype)a.dtype, a.sW, 0, st);
}

}  // namespace

extern "C" size_t vn_net_workspace_bytes(const vnNetConfig *cfg, int64_t K) {
    Plan P;
    if (K < 0 || !make_plan(cfg, K, nullptr, &P)) return 0;
    return P.bytes;
}

// Everything of the forward that does not depend on the voxel features: weight packing (all layers, both
// orientations) and, for the sparse first layer, the active-site list, the voxel index grid and the bias fill of its
// output.  vn_net_forward does it itself unless cfg->prepared says vn_net_prepare already did (on another stream,
// beside the VFE forward).
static int net_prepare(vnNet *net, const vnNetConfig *cfg, const Plan &P, const vnLayerParams *L, const float *heads_w,
                       const int64_t *coord, int64_t K, vnStream stream, bool own_stream) {
    const int training = cfg->training;
    // own_stream (vn_net_prepare): the first layer's forward weights are packed by a launch of their own, in front of the
    // site list; everything else is packed behind it (see vnNet::prep_ev)
    auto first_needs = [&]() -> int {
        if (cfg->sparse_first) {
            const Spec &sp = P.spec[0];
            const Rows &y = P.y[0];
            const Rows xin = dense_rows(nullptr, P.adt, cfg->B, cfg->D, cfg->H, cfg->W, 128);
            vnConv g = fwd_geom(sp, xin, P.odims[0], y);
            // y holds the conv bias at the ~90 % of sites no occupied voxel reaches.  When every consumer of y walks the flags /
            // the active-site list (flagged forward apply; list-based BatchNorm backward: P.list_bwd), those rows are never
            // read and the 180 MB fill is skipped (round 3)
            const bool y_dense_readers = !(training ? (P.list_bwd && (m0_bn_knob() & 1)) : (m0_bn_knob() & 1));
            if (y_dense_readers)
                RTT(T_FIRST, 0, 0.0, rows_bytes(y), stream, vn_fill_rows(y.ptr, (vnDtype)y.dtype, y.M(), sp.cout, sp.cout, L[0].bias, stream));
            RTT(T_FIRST, 0, 0.0, 0.0, stream, vn_active_sites(coord, K, &g, P.aws, P.aws_bytes, P.alist, P.acap, P.acount, stream));
            RTT(T_FIRST, 0, 0.0, 4.0 * cfg->B * cfg->D * cfg->H * cfg->W, stream,
                vn_voxel_index_grid(coord, K, cfg->B, cfg->D, cfg->H, cfg->W, P.igrid, stream));
        }
        return VN_OK;
    };
    // own_stream: cfg->prepared names the phase of this call — 0 everything in one call, 1 only the first layer's needs
    // (event 0; heads_w unused), 2 only the rest (event 1; needs a phase 1 for the same workspace / coord / K / grid)
    const int phase = own_stream ? cfg->prepared : 0;
    if (phase < 0 || phase > 2 || (phase != 1 && !heads_w)) return VN_EINVAL;
    const int32_t dims[9] = {cfg->B, cfg->D, cfg->H, cfg->W, cfg->training, cfg->mode, cfg->block1_stride, cfg->sparse_first, cfg->grad_storage};
    const bool rest_only = phase == 2;
    if (rest_only && !(net->prep_first_done && net->prep_ws == P.wp_f[0] && net->prep_coord == coord && net->prep_K == K &&
                       !memcmp(net->prep_dims, dims, sizeof(dims)))) {
        net->prep_first_done = false;
        return VN_EINVAL;      // no (or another step's) first phase
    }
    net->prep_first_done = false;
    if (own_stream && !rest_only) {
        const Spec &sp = P.spec[0];
        vnPackJob j0{L[0].weight, P.wp_f[0], sp.cout, sp.cin, sp.k[0] * sp.k[1] * sp.k[2], sp.transposed ? 2 : 0, 0, sp.cin_fold, wdt(P), 0};
        RTT(T_PACK, 0, 0.0, (double)j0.c_out * j0.c_in * j0.taps * (4 + P.esz), stream, vn_pack_weights_batch(&j0, 1, stream));
        RT(first_needs());
        VN_HIP(hipEventRecord(net->prep_ev[0], vn_stream(stream)));
        net->prep_ws = P.wp_f[0]; net->prep_coord = coord; net->prep_K = K;
        memcpy(net->prep_dims, dims, sizeof(dims));
        if (phase == 1) {   // the caller has more to queue on this stream (e.g. the heads' concatenation) before phase 2
            net->prep_first_done = true;
            return VN_OK;
        }
    }
    {   // every layer's weights -> MFMA operand layout, forward and (training) data-gradient orientation: one launch
        vnPackJob jobs[2 * NL + 2];
        int nj = 0;
        net->n_deferred = 0;
        for (int l = 0; l < NL; ++l) {
            const Spec &sp = P.spec[l];
            const int taps = sp.k[0] * sp.k[1] * sp.k[2];
            if (!(own_stream && l == 0))
                jobs[nj++] = vnPackJob{L[l].weight, P.wp_f[l], sp.cout, sp.cin, taps, sp.transposed ? 2 : 0, 0, sp.cin_fold, wdt(P), 0};
            if (training) {
                const bool first_sparse = l == 0 && cfg->sparse_first;
                const vnPackJob jd{L[l].weight, P.wp_d[l], sp.cout, sp.cin, taps, sp.transposed ? 3 : 1, 0,
                                   first_sparse ? 1 : sp.cin_fold, wdt(P), 0};
                if (own_stream) net->deferred_pack[net->n_deferred++] = jd;
                else jobs[nj++] = jd;
            }
        }
        jobs[nj++] = vnPackJob{heads_w, P.hwp_f, 16, 768, 1, 0, 0, 1, wdt(P), 0};
        if (training) {
            const vnPackJob jd{heads_w, P.hwp_d, 16, 768, 1, 1, 0, 1, wdt(P), 0};
            if (own_stream) net->deferred_pack[net->n_deferred++] = jd;
            else jobs[nj++] = jd;
        }
        double pbytes = 0.0;
        for (int j = 0; j < nj; ++j) pbytes += (double)jobs[j].c_out * jobs[j].c_in * jobs[j].taps * (4 + P.esz);
        RTT(T_PACK, -1, 0.0, pbytes, stream, vn_pack_weights_batch(jobs, nj, stream));
    }
    if (own_stream) {
        VN_HIP(hipEventRecord(net->prep_ev[1], vn_stream(stream)));
        net->prep_recorded = true;
    } else {
        RT(first_needs());
    }
    return VN_OK;
}

extern "C" int vn_net_prepare(vnNet *net, const vnNetConfig *cfg, const vnLayerParams *L, const float *heads_w,
                              const int64_t *coord, int64_t K, void *workspace, size_t workspace_bytes, vnStream stream) {
    VN_CHECK_ARG(net && cfg && L && workspace && K >= 0 && (!cfg->sparse_first || coord));
    Plan P;
    if (!make_plan(cfg, K, static_cast<char *>(workspace), &P)) return VN_EUNSUPPORTED;
    if (workspace_bytes < P.bytes) return VN_EWORKSPACE;
    const int rc = net_prepare(net, cfg, P, L, heads_w, coord, K, stream, true);
    if (rc != VN_OK) {     // no half-issued protocol state survives an error
        net->prep_first_done = false;
        net->prep_recorded = false;
        net->n_deferred = 0;
    }
    return rc;
}

extern "C" int vn_net_forward(vnNet *net, const vnNetConfig *cfg, const vnLayerParams *L, const float *heads_w,
                              const float *heads_b, const void *dense, const int64_t *coord, const void *vw_rows, int64_t K,
                              void *workspace, size_t workspace_bytes, float *prob, float *reg, vnStream stream,
                              vnStream side_stream) {
    VN_CHECK_ARG(net && cfg && L && heads_w && heads_b && workspace && prob && reg && K >= 0);
    VN_CHECK_ARG(cfg->sparse_first ? (coord && vw_rows) : dense != nullptr);
    Plan P;
    if (!make_plan(cfg, K, static_cast<char *>(workspace), &P)) return VN_EUNSUPPORTED;
    if (workspace_bytes < P.bytes) return VN_EWORKSPACE;
    hipStream_t hs = vn_stream(stream);
    const int training = cfg->training;
    const float mom = 0.1f, eps = 1e-5f;
    const int relu_fl = P.round_act ? 3 : 1;    // (bit 1: the activation is rounded to the nearest bf16 value — diagnostic)
    // (no memset: every statistics buffer of the forward is a per-workgroup slab written with plain stores)
    if (!cfg->prepared) RT(net_prepare(net, cfg, P, L, heads_w, coord, K, stream, false));
    if (cfg->prepared && !net->prep_recorded) return VN_EINVAL;   // "prepared" without a vn_net_prepare on this context (or already consumed)
    const bool prep_wait = cfg->prepared != 0;
    if (prep_wait) {        // the prepared step must be THIS one: same arena, voxel coordinates, K and grid
        const int32_t dims[9] = {cfg->B, cfg->D, cfg->H, cfg->W, cfg->training, cfg->mode, cfg->block1_stride, cfg->sparse_first, cfg->grad_storage};
        if (net->prep_ws != P.wp_f[0] || net->prep_K != K || (cfg->sparse_first && net->prep_coord != coord) ||
            memcmp(net->prep_dims, dims, sizeof(dims))) {
            net->prep_recorded = false;
            net->n_deferred = 0;
            return VN_EINVAL;
        }
    }
    net->prep_recorded = false;     // consumed: a later forward needs its own vn_net_prepare
    net->heads_rows_ready = nullptr;   // (only vn_net_step, behind THIS forward, may announce rows for the backward)
    if (prep_wait) VN_HIP(hipStreamWaitEvent(hs, net->prep_ev[0], 0));
    Rows x = dense_rows(const_cast<void *>(dense), P.adt, cfg->B, cfg->D, cfg->H, cfg->W, 128);
    Rows x1{}, x2{};
    // deconv1 / deconv2 only feed the concat: with a side stream they run beside block2 / block3 (whose 100x88 and
    // 50x44 images leave most CUs idle) and are joined before the heads
    hipStream_t ss = side_stream ? vn_stream(side_stream) : hs;
    const vnStream main_stream = stream;
    for (int l = 0; l < NL; ++l) {
        const Spec &sp = P.spec[l];
        const bool on_side = ss != hs && (l == L_D1 || l == L_D2);
        if (on_side) {   // fork: the block output this deconv reads exists at this point of the main stream
            hipEvent_t e = net->next_event();
            if (!e) return VN_EINVAL;
            VN_HIP(hipEventRecord(e, hs));
            VN_HIP(hipStreamWaitEvent(ss, e, 0));
        }
        if (l == 1 && prep_wait) VN_HIP(hipStreamWaitEvent(hs, net->prep_ev[1], 0));   // the other layers' packed weights
        const vnStream stream = on_side ? side_stream : main_stream;   // (shadows the parameter inside the loop body)
        if (l == L_D1) x = x1;                 // deconv1 and block2 both read the block1 output
        if (l == L_B2) x = x1;
        if (l == L_D2 || l == L_B3) x = x2;
        const Rows &y = P.y[l];
        const int64_t M = y.M();
        vnConv g = fwd_geom(sp, x, P.odims[l], y);
        float *slab = training ? P.slab[l] : nullptr;
        if (l == 0 && cfg->sparse_first) {
            // (bias fill of y, active-site list, voxel index grid: net_prepare)
            // rulebook: P[v][tap] = W[tap] . x[v] as ONE dense GEMM over the K voxel rows ([27*64][128] packed weights),
            // then every active site adds the P rows of its occupied source cells (no dense grid, no zero work)
            if (K > 0) {
                vnConv q;
                memset(&q, 0, sizeof(q));
                q.dtype = P.adt; q.B = 1; q.Ds = q.Hs = 1; q.Ws = (int32_t)K; q.Dr = q.Hr = 1; q.Wr = (int32_t)K;
                q.Cs = sp.cin; q.Cr = 27 * sp.cout;
                q.kD = q.kH = q.kW = 1; q.mulD = q.mulH = q.mulW = 1; q.tmulD = q.tmulH = q.tmulW = 1;
                q.divD = q.divH = q.divW = 1;
                q.src_sB = q.src_sD = q.src_sH = K * sp.cin; q.src_sW = sp.cin;
                q.out_sB = q.out_sD = q.out_sH = K * q.Cr; q.out_sW = q.Cr;
                RTT(T_CONV_FWD, l, 2.0 * K * sp.cin * 27.0 * sp.cout, (double)K * (sp.cin * P.esz + 27.0 * sp.cout * 4), stream,
                    vn_conv_gather_gemm(vw_rows, P.wp_f[l], nullptr, P.rbP, VN_F32, cx(P, q), 0, nullptr, stream));
            }
            RTT(T_FIRST, l, 0.0, 0.0, stream,
                vn_rulebook_combine(P.rbP, P.igrid, P.alist, P.acap, P.acount, &g, L[l].bias, y.ptr, (vnDtype)y.dtype, slab,
                                    stream));
        } else {
            RTT(T_CONV_FWD, l, layer_flops(sp, P.in_dims[l], P.odims[l], cfg->B), rows_bytes(x) + rows_bytes(y), stream,
                vn_conv_gather_gemm(x.ptr, P.wp_f[l], L[l].bias, y.ptr, (vnDtype)y.dtype, cx(P, g), 0, slab, stream));
        }
        const Rows &a = P.a[l];
        if (slab) {
            RTT(T_BN_FINALIZE, l, 0.0, 8.0 * P.slab_rows[l] * sp.cout, stream,
                vn_bn_finalize_slab(slab, P.slab_rows[l], M, sp.cout, L[l].bias, L[l].gamma, L[l].beta, L[l].running_mean,
                                    L[l].running_var, mom, eps, P.stats[l], stream));
        } else {
            if (training)
                RT(vn_bn_stats(y.ptr, (vnDtype)y.dtype, M, sp.cout, y.sW, 1, L[l].bias, P.fsums[l], stream));
            RT(vn_bn_finalize(training ? P.fsums[l] : nullptr, M, sp.cout, 1, L[l].bias, L[l].gamma, L[l].beta,
                              L[l].running_mean, L[l].running_var, training, mom, eps, P.stats[l], stream));
        }
        if (l == L_M2) {   // BEV fold: channel d*64 + c of the (B,1,H,W,128) activation
            RTT(T_BN_APPLY, l, 0.0, 2.0 * rows_bytes(y), stream,
                vn_bn_apply_bev(y.ptr, (vnDtype)y.dtype, M, 64, (int64_t)P.odims[l][1] * P.odims[l][2], P.stats[l], relu_fl, a.ptr,
                                (vnDtype)a.dtype, 128, stream));
        } else if (l == 0 && cfg->sparse_first && (m0_bn_knob() & 1)) {
            // ~90 % of the first layer's sites hold the bias (no occupied voxel in reach): their y is not read
            RTT(T_BN_APPLY, l, 0.0, 1.1 * rows_bytes(y), stream,
                vn_bn_apply_flagged(y.ptr, (vnDtype)y.dtype, y.sW, M, sp.cout, P.stats[l], relu_fl, a.ptr, (vnDtype)a.dtype, a.sW,
                                    static_cast<const uint8_t *>(P.aws), L[l].bias, stream));
        } else {
            RTT(T_BN_APPLY, l, 0.0, 2.0 * rows_bytes(y), stream, bn_apply_rows(y, P.stats[l], a, sp.cout, relu_fl, stream));
        }
        if (!sp.transposed) x = a;
        if (l == L_D1 - 1) x1 = a;
        if (l == L_D2 - 1) x2 = a;
        if (l == L_D2 && net->n_deferred > 0) {   // vn_net_prepare's data-gradient packs: behind deconv2, on its stream
            double pb = 0.0;
            for (int j = 0; j < net->n_deferred; ++j)
                pb += (double)net->deferred_pack[j].c_out * net->deferred_pack[j].c_in * net->deferred_pack[j].taps * (4 + P.esz);
            RTT(T_PACK, -1, 0.0, pb, stream, vn_pack_weights_batch(net->deferred_pack, net->n_deferred, stream));
            net->n_deferred = 0;
        }
    }
    if (ss != hs) {   // join before the heads read the concat
        hipEvent_t e = net->next_event();
        if (!e) return VN_EINVAL;
        VN_HIP(hipEventRecord(e, ss));
        VN_HIP(hipStreamWaitEvent(hs, e, 0));
    }
    // heads: one N=16 GEMM over the 768-channel concat + sigmoid on the first two channels (model.py:276-281)
    {
        Spec hs16{2, 768, 16, {1, 1, 1}, {1, 1, 1}, {0, 0, 0}, false, 1};
        const int od[3] = {1, P.hf, P.wf};
        vnConv g = fwd_geom(hs16, P.cat, od, P.hy);
        const int64_t S = (int64_t)P.hf * P.wf;
        if (P.adt == VN_BF16 && heads_stream_on()) {
            RTT(T_CONV_FWD, NL, 2.0 * cfg->B * S * 768 * 16, rows_bytes(P.cat) + rows_bytes(P.hy), stream,
                vn_heads_fwd(P.cat.ptr, P.cat.sW, P.hwp_f, heads_b, cfg->B, S, prob, reg, stream));
        } else {
        RTT(T_CONV_FWD, NL, 2.0 * cfg->B * S * 768 * 16, rows_bytes(P.cat) + rows_bytes(P.hy), stream,
            vn_conv_gather_gemm(P.cat.ptr, P.hwp_f, heads_b, P.hy.ptr, VN_F32, cx(P, g), 0, nullptr, stream));
        RTT(T_MISC, NL, 0.0, 2.0 * rows_bytes(P.hy), stream,
            vn_heads_to_nchw(reinterpret_cast<const float *>(P.hy.ptr), cfg->B, S, prob, reg, stream));
        }
    }
    return VN_OK;
}

extern "C" int vn_net_backward(vnNet *net, const vnNetConfig *cfg, const vnLayerParams *L, const float *heads_w,
                               const float *d_prob, const float *d_reg, const float *prob, const void *dense, const int64_t *coord,
                               const void *vw_rows, int64_t K, void *workspace, size_t workspace_bytes,
                               const vnLayerGrads *G, float *d_heads_w, float *d_heads_b, void *d_input,
                               int32_t seg_begin, int32_t seg_end, vnStream stream, vnStream side_stream) {
    VN_CHECK_ARG(net && cfg && L && heads_w && d_prob && d_reg && prob && workspace && G && d_heads_w && d_heads_b);
    if (!cfg->training) return VN_EUNSUPPORTED;   // train-mode BatchNorm backward only; the forward made no dgrad packs
    VN_CHECK_ARG(cfg->sparse_first ? (coord && vw_rows && d_input) : dense != nullptr);
    Plan P;
    if (!make_plan(cfg, K, static_cast<char *>(workspace), &P)) return VN_EUNSUPPORTED;
    if (workspace_bytes < P.bytes) return VN_EWORKSPACE;
    hipStream_t hs = vn_stream(stream);
    // backward steps: 0 = heads, 1..23 = the BatchNorm layers in backward order; [seg_begin, seg_end) runs now
    VN_CHECK_ARG(seg_begin >= 0 && seg_end <= NL + 1 && seg_begin < seg_end);
    // (no memset: the backward sums are per-workgroup slabs / partial rows written with plain stores)
    const int B = cfg->B;
    const int64_t S = (int64_t)P.hf * P.wf;
    // timing mode only: the number of active first-layer sites (device-resident; the FLOPs of the row-list launches executed
    // in this call are reported with it).  A blocking read — never in a timed step.
    double active_rows = 0.0;
    if (net->timing && P.list_bwd) {
        int32_t cnt = 0;
        VN_HIP(hipMemcpy(&cnt, P.acount, sizeof(cnt), hipMemcpyDeviceToHost));
        active_rows = cnt < 0 ? 0.0 : (cnt > P.acap ? (double)P.acap : (double)cnt);
    }
    // packed fp32 weight gradients -> torch layouts: collected here, one launch at the end of the segment.
    // (the data-gradient operand packs were made by vn_net_forward, cfg->training)
    vnUnpackJob unpack[NL + 1];
    int nu = 0;
    int rank1_job = -1;     // index of middle_layer.1's unpack job when its weight gradient took the sparse route
    // (after the unpack that holds that job: + const (x) box sums of dy, on the same stream)
    auto after_unpack = [&](int first, int end, vnStream st) -> int {
        if (rank1_job < first || rank1_job >= end) return VN_OK;
        const Spec &s1 = P.spec[1];
        RTT(T_UNPACK, 1, 0.0, 0.0, st,
            vn_wgrad_const_add(G[1].weight, P.dtot_ws, P.dtot_ws_bytes, s1.cout, s1.cin, s1.k[0], P.stats[0], L[0].bias,
                               (vnDtype)P.y[0].dtype, (vnDtype)P.a[0].dtype, 1, st));
        return VN_OK;
    };
    ZeroJobs zj{};
    // The weight gradient of a layer and its data gradient are independent; on the small late layers either one is
    // 60-140 workgroups on 256 CUs.  With a side stream the weight-gradient launches run beside the main stream's
    // data-gradient / BatchNorm-backward launches: fork when dy exists, join before the segment's unpack.
    hipStream_t ws = side_stream ? vn_stream(side_stream) : hs;
    const vnStream wstream = side_stream ? side_stream : stream;
    auto fork = [&]() -> int {
        if (ws == hs) return VN_OK;
        hipEvent_t e = net->next_event();
        if (!e) return VN_EINVAL;
        VN_HIP(hipEventRecord(e, hs));
        VN_HIP(hipStreamWaitEvent(ws, e, 0));
        return VN_OK;
    };
    bool heads_forked = false;     // the side stream already waits for the main stream's state after the heads
    // ---- heads
    if (seg_begin == 0) {
        Spec hs16{2, 768, 16, {1, 1, 1}, {1, 1, 1}, {0, 0, 0}, false, 1};
        if (net->heads_rows_ready != P.d_rows.ptr)
            RTT(T_MISC, NL, 0.0, 0.0, stream, vn_heads_bwd(d_prob, d_reg, prob, B, S, P.d_rows.ptr, (vnDtype)P.adt, 16, 0, stream));
        net->heads_rows_ready = nullptr;
        const int od[3] = {1, P.hf, P.wf};
        const int64_t rs[4] = {P.d_rows.sB, P.d_rows.sD, P.d_rows.sH, P.d_rows.sW};
        vnConv gw = geom(P.cat, od, 768, 16, hs16.k, ONE, ONE, hs16.p, ONE, rs);
        const int64_t os[4] = {P.d_cat.sB, P.d_cat.sD, P.d_cat.sH, P.d_cat.sW};
        vnConv gd = geom(P.d_rows, od, 16, 768, hs16.k, ONE, NEG, hs16.p, ONE, os);
        // the heads' weight gradient goes to the side stream behind ONE fork that also serves the early deconv
        // branches (which need the data gradient): one event record less on the main stream (483 vs 480 pc/s)
        if (P.exact_heads) {
            RTT(T_MISC, NL, 0.0, 0.0, stream, vn_heads_bwd(d_prob, d_reg, prob, B, S, P.d_rows32.ptr, VN_F32, 16, 0, stream));
            RTT(T_CONV_DGRAD, NL, 2.0 * B * S * 768 * 16, rows_bytes(P.d_rows32) + rows_bytes(P.d_cat), stream,
                vn_heads_dgrad_f32(reinterpret_cast<const float *>(P.d_rows32.ptr), P.d_rows32.sW, heads_w, P.d_cat.ptr,
                                   (vnDtype)P.cdt, P.d_cat.sW, (int64_t)B * S, stream));
        } else if (P.adt == VN_BF16 && P.cdt == VN_BF16 && heads_stream_on())
            RTT(T_CONV_DGRAD, NL, 2.0 * B * S * 768 * 16, rows_bytes(P.d_rows) + rows_bytes(P.d_cat), stream,
                vn_heads_dgrad(P.d_rows.ptr, P.d_rows.sW, P.hwp_d, P.d_cat.ptr, P.d_cat.sW, (int64_t)B * S, stream));
        else
        RTT(T_CONV_DGRAD, NL, 2.0 * B * S * 768 * 16, rows_bytes(P.d_rows) + rows_bytes(P.d_cat), stream,
            vn_conv_gather_gemm(P.d_rows.ptr, P.hwp_d, nullptr, P.d_cat.ptr, (vnDtype)P.cdt, cx(P, gd), 0, nullptr, stream));
        RT(fork());
        heads_forked = true;
        // the heads' bias gradients (column sums of d_rows: two small launches) are nobody's input: behind the fork, on the
        // side stream (the main stream when there is none), not between the loss and the first data gradient
        int32_t hch = 1;
        RTT(T_MISC, NL, 0.0, 0.0, wstream,
            vn_col_sums(P.d_rows.ptr, (vnDtype)P.adt, 16, B * S, 16, d_heads_b, P.hcs_ws, P.hcs_ws_bytes, wstream));
        RTT(T_WGRAD, NL, 2.0 * B * S * 768 * 16, rows_bytes(P.d_rows) + rows_bytes(P.cat), wstream,
            vn_conv_wgrad_partials(P.cat.ptr, P.d_rows.ptr, cx(P, gw), 0, nullptr, 0, P.hdwp, P.hdwp_bytes, &hch, wstream));
        unpack[nu++] = vnUnpackJob{P.hdwp, d_heads_w, 16, 768, 1, 0, 1, hch, (int64_t)16 * 768};
    }
    auto cat_slice = [&](int off) {
        Rows r = P.d_cat;
        r.ptr += (size_t)off * (P.cdt == VN_F32 ? 4 : 2);
        r.C = 256;
        return r;
    };
    // input activation of every layer (as in the forward)
    auto input_of = [&](int l) -> Rows {
        if (l == 0) return dense_rows(const_cast<void *>(dense), P.adt, B, cfg->D, cfg->H, cfg->W, 128);
        if (l == L_D1 || l == L_B2) return P.a[L_D1 - 1];
        if (l == L_D2 || l == L_B3) return P.a[L_D2 - 1];
        return P.a[l - 1];
    };
    // backward order: deconv3, block3 (5..0), deconv2, block2, deconv1, block1, middle 2,1,0
    int order[NL], n = 0;
    order[n++] = L_D3; for (int l = L_D3 - 1; l >= L_B3; --l) order[n++] = l;
    order[n++] = L_D2; for (int l = L_D2 - 1; l >= L_B2; --l) order[n++] = l;
    order[n++] = L_D1; for (int l = L_D1 - 1; l >= L_B1; --l) order[n++] = l;
    order[n++] = 2; order[n++] = 1; order[n++] = 0;
    // weight gradient of layer l on the side stream (its dy exists in stream order of whoever calls this)
    auto launch_wgrad = [&](int l, vnStream wstream) -> int {
        const Spec &sp = P.spec[l];
        const int taps = sp.k[0] * sp.k[1] * sp.k[2];
        const int C = sp.cout;
        const Rows &dy = P.dy[l];
        int32_t wch = 1;
        const int64_t dw_elems = (int64_t)taps * sp.cin * sp.cout;
        if (l == 0 && cfg->sparse_first) {
            const int np[3] = {-sp.p[0], -sp.p[1], -sp.p[2]};
            const int64_t rs[4] = {0, 0, 0, 128};
            vnConv gw = geom(dy, P.in_dims[0], C, sp.cin, sp.k, ONE, NEG, np, sp.s, rs);
            RTT(T_WGRAD, l, 2.0 * K * sp.cin * 27.0 * C, 0.0, wstream,
                vn_conv_wgrad_partials(dy.ptr, vw_rows, cx(P, gw), 0, coord, K, P.dwp[l], P.dwp_bytes[l], &wch, wstream));
            unpack[nu++] = vnUnpackJob{P.dwp[l], G[l].weight, sp.cin, C, taps, 2, 1, wch, dw_elems};
            return VN_OK;
        }
        if (l == 1 && P.sparse_w1) {
            // a0 = const + (rows at the first layer's active sites): the rows' part as a row-list weight gradient (10 % of the
            // sites), the constant's part = const (x) box sums of dy, added after the unpack (rank1_job)
            const Rows &a0 = P.a[0];
            RTT(T_MISC, l, 0.0, 0.0, wstream,
                vn_act_delta_rows(a0.ptr, (vnDtype)a0.dtype, sp.cin, P.odims[0][0], P.odims[0][1], P.odims[0][2], P.stats[0],
                                  L[0].bias, (vnDtype)P.y[0].dtype, 1, P.alist, P.acount, P.acap, P.drows,
                                  dy.dtype == VN_F32X3S ? VN_F32X3S : (vnDtype)a0.dtype, wstream));
            const int np[3] = {-sp.p[0], -sp.p[1], -sp.p[2]};
            const int64_t rs[4] = {0, 0, 0, sp.cin};
            vnConv gl = geom(dy, P.in_dims[l], C, sp.cin, sp.k, ONE, NEG, np, sp.s, rs);
            RTT(T_WGRAD, l, 2.0 * active_rows * taps * sp.cin * C, rows_bytes(dy) + 2.0 * active_rows * sp.cin * P.esz, wstream,
                vn_conv_wgrad_partials_counted(dy.ptr, P.drows, cx(P, gl), P.alist, P.acap, P.acount, P.dwp[l], P.dwp_bytes[l], &wch,
                                               wstream));
            rank1_job = nu;
            unpack[nu++] = vnUnpackJob{P.dwp[l], G[l].weight, sp.cin, C, taps, 2, 1, wch, dw_elems};
            return VN_OK;
        }
        const Rows x = input_of(l);
        if (l == L_M2 && P.x3 && P.x3_store && P.m2_passes) {
            // three bf16 passes of the nine-tap patch kernel over the split-stored operands, in place
            vnConv gs = wgrad_geom(P, l, x);
            if (gs.dtype != VN_F32X3S || dy.dtype != VN_F32X3S) return VN_EINVAL;
            const size_t pass_bytes = P.dwp_bytes[l] / 3;
            int32_t total = 0;
            for (int pass = 0; pass < 3; ++pass) {
                int32_t ch = 1;
                float *slabs = P.dwp[l] + (size_t)total * dw_elems;
                RTT(T_WGRAD, l, layer_flops(sp, P.in_dims[l], P.odims[l], B) / 3.0, (rows_bytes(x) + rows_bytes(dy)) / 3.0, wstream,
                    vn_conv_wgrad_partials_split_pass(x.ptr, dy.ptr, &gs, pass, slabs, pass_bytes, &ch, wstream));
                total += ch;
            }
            unpack[nu++] = vnUnpackJob{P.dwp[l], G[l].weight, C, sp.cin, taps, 0, sp.cin_fold, total, dw_elems};
            return VN_OK;
        }
        vnConv gw = wgrad_geom(P, l, x);
        if (sp.transposed) {
            RTT(T_WGRAD, l, layer_flops(sp, P.in_dims[l], P.odims[l], B), rows_bytes(x) + rows_bytes(dy), wstream,
                vn_conv_wgrad_partials(dy.ptr, x.ptr, cx(P, gw), 0, nullptr, 0, P.dwp[l], P.dwp_bytes[l], &wch, wstream));
            unpack[nu++] = vnUnpackJob{P.dwp[l], G[l].weight, sp.cin, C, taps, 0, 1, wch, dw_elems};
        } else {
            RTT(T_WGRAD, l, layer_flops(sp, P.in_dims[l], P.odims[l], B), rows_bytes(x) + rows_bytes(dy), wstream,
                vn_conv_wgrad_partials(x.ptr, dy.ptr, cx(P, gw), 0, nullptr, 0, P.dwp[l], P.dwp_bytes[l], &wch, wstream));
            unpack[nu++] = vnUnpackJob{P.dwp[l], G[l].weight, C, sp.cin, taps, 0, sp.cin_fold, wch, dw_elems};
        }
        return VN_OK;
    };
    // Main-stream layers queue their weight gradient; flush() forks ONCE (an event record on the main stream costs a
    // ~6 us bubble there: measured) and issues the queued ones on the side stream.  Flushed after every block chain.
    int pending[NL], npend = 0;
    auto flush = [&]() -> int {
        if (npend == 0) return VN_OK;
        RT(fork());
        for (int i = 0; i < npend; ++i) {
            RT(launch_wgrad(pending[i], wstream));
        }
        npend = 0;
        return VN_OK;
    };
    // Tail balance of the deferred-join single call (tuning aids VN_M0_MAIN / VN_EARLY_UNPACK, default on): the side
    // stream ends with the two long Conv3d weight gradients, the main stream with the short first-layer data gradient.
    // So the FIRST layer's weight gradient runs on the main stream (behind its data gradient), and everything up to
    // block1 is unpacked on the side stream before it waits for middle_layer.2's dy (a ~180 us idle gap there): only
    // the three Conv3d gradients are left for the final unpack.
    const bool tail_balance = ws != hs && cfg->defer_join && !cfg->bucket_events && seg_begin == 0 && seg_end == NL + 1;
    const bool bucket_mode = ws != hs && cfg->bucket_events && seg_begin == 0 && seg_end == NL + 1;
    // ... while that weight gradient is short: it is a K x 27-tap row-list launch, 38 us at the car's 12 k voxels but 0.45 ms at
    // the dense config's 160 k — there it sat in front of the first layer's data gradient and the whole VFE backward (the
    // step's last millisecond) while the side stream idled (round 5, `tools/step_timeline.py --dense`): above 40 k voxels it
    // goes to the side stream like every other weight gradient
    const bool m0_on_main = (tail_balance || bucket_mode) && K <= 40000;
    int u_early = 0;
    hipEvent_t box_ev = nullptr;       // recorded on the side stream behind the box sums of middle_layer.1's dy
    int64_t fused_rows[NL] = {0};      // > 0: layer's BatchNorm-backward slab was written by the data gradient above it (rows)
    const bool single_call = seg_begin == 0 && seg_end == NL + 1;
    // one layer of the backward: BatchNorm backward and data gradient on `ls`, weight gradient on the side stream
    auto do_layer = [&](int l, vnStream ls, bool on_side, bool accumulate) -> int {
        const Spec &sp = P.spec[l];
        const int taps = sp.k[0] * sp.k[1] * sp.k[2];
        const int C = sp.cout;
        const Rows &y = P.y[l];
        const int64_t M = y.M();
        // gradient w.r.t. this layer's activation
        Rows da;
        if (l == L_D3) da = cat_slice(0);
        else if (l == L_D2) da = cat_slice(256);
        else if (l == L_D1) da = cat_slice(512);
        else if (l == L_D3 - 1) da = P.dx[L_D3];
        else if (l == L_D2 - 1) da = P.dx[L_B3];   // block3.0's dx, deconv2 accumulated into it
        else if (l == L_D1 - 1) da = P.dx[L_B2];
        else da = P.dx[l + 1];
        const Rows &dy = P.dy[l];
        if (l == L_M2) {   // da is the BEV gradient (B,1,H,W,128): channel d*64+c
            const int64_t hw = (int64_t)P.odims[l][1] * P.odims[l][2];
            RTT(T_BN_BWD_REDUCE, l, 0.0, 2.0 * rows_bytes(y), ls,
                vn_bn_bwd_reduce_slab_bev(da.ptr, (vnDtype)da.dtype, 128, y.ptr, (vnDtype)y.dtype, M, C, hw, P.stats[l], 1,
                                          P.bslab[l], ls));
            RTT(T_BN_FINALIZE, l, 0.0, 8.0 * P.bslab_rows[l] * C, ls,
                vn_bn_bwd_finalize_slab(P.bslab[l], P.bslab_rows[l], M, C, L[l].gamma, P.stats[l], P.coef[l], G[l].gamma,
                                        G[l].beta, ls));
            RTT(T_BN_BWD_APPLY, l, 0.0, 3.0 * rows_bytes(y), ls,
                vn_bn_bwd_apply_bev(da.ptr, (vnDtype)da.dtype, 128, y.ptr, (vnDtype)y.dtype, M, C, hw, P.stats[l], P.coef[l], 1,
                                    dy.ptr, (vnDtype)dy.dtype, ls));
        } else {
            if (l == 0 && P.list_bwd) {
                // da = middle_layer.1's data gradient at the active sites only ([acap][C] rows in list order) + its total
                const Rows &dac = P.dx[1];
                const int64_t lrows = vn_bn_bwd_list_slab_rows(P.acap, C);
                RTT(T_BN_BWD_REDUCE, l, 0.0, 0.0, ls,
                    vn_bn_bwd_reduce_list(dac.ptr, (vnDtype)dac.dtype, y.ptr, (vnDtype)y.dtype, C, P.odims[0][0], P.odims[0][1],
                                          P.odims[0][2], P.stats[l], 1, P.bslab[l], P.alist, P.acount, P.acap, ls));
                if (box_ev) VN_HIP(hipStreamWaitEvent(vn_stream(ls), box_ev, 0));    // the totals from the side stream
                RTT(T_BN_FINALIZE, l, 0.0, 12.0 * lrows * C, ls,
                    vn_bn_bwd_finalize_list(P.bslab[l], lrows, M, C, L[l].gamma, P.stats[l], P.dtot, L[l].bias, (vnDtype)y.dtype, 1,
                                            P.coef[l], G[l].gamma, G[l].beta, ls));
                RTT(T_BN_BWD_APPLY, l, 0.0, 0.0, ls,
                    vn_bn_bwd_apply_list_rows(dac.ptr, (vnDtype)dac.dtype, y.ptr, (vnDtype)y.dtype, C, P.odims[0][0], P.odims[0][1],
                                              P.odims[0][2], P.stats[l], P.coef[l], 1, dy.ptr, (vnDtype)dy.dtype, P.alist,
                                              P.acount, P.acap, ls));
            } else {
            if (l == 0 && cfg->sparse_first && (m0_bn_knob() & 2))   // y is the bias at the ~90 % inactive sites: read at the flagged rows only
                RTT(T_BN_BWD_REDUCE, l, 0.0, 1.1 * rows_bytes(y), ls,
                    vn_bn_bwd_reduce_slab_flagged(da.ptr, (vnDtype)da.dtype, da.sW, y.ptr, (vnDtype)y.dtype, y.sW, M, C,
                                                  P.stats[l], 1, P.bslab[l], static_cast<const uint8_t *>(P.aws), L[l].bias, ls));
            else if (fused_rows[l] == 0)      // (else: the sums came out of the data-gradient launch of layer l + 1)
            RTT(T_BN_BWD_REDUCE, l, 0.0, 2.0 * rows_bytes(y), ls,
                vn_bn_bwd_reduce_slab(da.ptr, (vnDtype)da.dtype, da.sW, y.ptr, (vnDtype)y.dtype, y.sW, M, C, P.stats[l], 1,
                                      P.bslab[l], ls));
            const int64_t brows = fused_rows[l] > 0 ? fused_rows[l] : P.bslab_rows[l];
            RTT(T_BN_FINALIZE, l, 0.0, 8.0 * brows * C, ls,
                vn_bn_bwd_finalize_slab(P.bslab[l], brows, M, C, L[l].gamma, P.stats[l], P.coef[l], G[l].gamma,
                                        G[l].beta, ls));
            if (l == 0 && cfg->sparse_first && !(m0_bn_knob() & 4))
                RTT(T_BN_BWD_APPLY, l, 0.0, 0.0, ls,
                    vn_bn_bwd_apply_flagged(da.ptr, (vnDtype)da.dtype, da.sW, y.ptr, (vnDtype)y.dtype, y.sW, M, C, P.stats[l],
                                            P.coef[l], 1, dy.ptr, (vnDtype)dy.dtype, dy.sW,
                                            static_cast<const uint8_t *>(P.aws), ls));
            else if (l == 0 && cfg->sparse_first)   // dy is only gathered at the active sites (flags: the forward's vn_active_sites)
                RTT(T_BN_BWD_APPLY, l, 0.0, 0.0, ls,     // (bytes depend on the number of active sites, known on the device only)
                    vn_bn_bwd_apply_list(da.ptr, (vnDtype)da.dtype, y.ptr, (vnDtype)y.dtype, C, P.odims[0][0], P.odims[0][1],
                                         P.odims[0][2], P.stats[l], P.coef[l], 1, dy.ptr, (vnDtype)dy.dtype, P.alist, P.acount,
                                         P.acap, ls));
            else
                RTT(T_BN_BWD_APPLY, l, 0.0, 3.0 * rows_bytes(y), ls,
                    vn_bn_bwd_apply(da.ptr, (vnDtype)da.dtype, da.sW, y.ptr, (vnDtype)y.dtype, y.sW, M, C, P.stats[l],
                                    P.coef[l], 1, dy.ptr, (vnDtype)dy.dtype, dy.sW, 0, ls));
            }
        }
        zj.ptr[zj.n] = G[l].bias; zj.len[zj.n] = C; ++zj.n;          // bias before a train-mode BN: gradient exactly 0
        const int np[3] = {-sp.p[0], -sp.p[1], -sp.p[2]};
        // weight gradient: at once when this layer runs on the side stream itself, else queued for the next flush
        if (on_side) RT(launch_wgrad(l, wstream));
        else if (l == 0 && m0_on_main) RT(launch_wgrad(l, stream));
        else {
            pending[npend++] = l;
            // The three Conv3d layers on their DENSE backward route (no list-based BatchNorm backward: the dense config's 160 k
            // voxels): their weight gradients are the longest of the step (0.45-0.73 ms at batch 4) and only need dy — fork
            // HERE, in front of the data gradient, not behind it: the side stream sat idle for the 0.28-0.41 ms of each of
            // these data gradients and then finished the step alone (tools/step_timeline.py --dense, round 5).  The car /
            // pedestrian steps (list route) keep their schedule: there the early fork measured nothing (round 4).
            if (l <= L_M2 && !P.list_bwd && ws != hs && single_call) RT(flush());
        }
        if (l == 0 && cfg->sparse_first) {
            const int64_t rs[4] = {0, 0, 0, 128};
            vnConv gw = geom(dy, P.in_dims[0], C, sp.cin, sp.k, ONE, NEG, np, sp.s, rs);
            RTT(T_CONV_DGRAD, l, 2.0 * K * sp.cin * 27.0 * C, 0.0, ls,
                vn_conv_gather_gemm_rows(dy.ptr, P.wp_d[l], nullptr, d_input, VN_F32, cx(P, gw), coord, K, nullptr, 1, nullptr, ls));
            return VN_OK;
        }
        if (l == 1 && P.list_bwd) {
            if (!L[l].weight) return VN_EINVAL;
            // the layer below is constant outside its active sites: its BatchNorm backward needs this data gradient at those
            // sites only (a row-list launch into [acap][cin] rows) and its sum over all sites (box sums of dy)
            // (the box sums are only read by the first layer's finalize, ~120 us down this stream, and by the weight
            //  gradient's constant part on the side stream: with a side stream they run there, beside the row-list launch)
            const bool box_on_side = ws != hs && !on_side && single_call;
            if (box_on_side) RT(fork());
            const vnStream bs = box_on_side ? wstream : ls;
            RTT(T_MISC, l, 0.0, rows_bytes(dy), bs,
                vn_dgrad_total(dy.ptr, (vnDtype)dy.dtype, B, P.odims[l][0], P.odims[l][1], P.odims[l][2], C, sp.cin, sp.k[0],
                               L[l].weight, 0, P.dtot_ws, P.dtot_ws_bytes, P.dtot, bs));
            if (box_on_side) {
                box_ev = net->next_event();
                if (!box_ev) return VN_EINVAL;
                VN_HIP(hipEventRecord(box_ev, ws));
            }
            const int64_t rs[4] = {0, 0, 0, sp.cin};
            vnConv gl = geom(dy, P.in_dims[l], C, sp.cin, sp.k, ONE, NEG, np, sp.s, rs);
            RTT(T_CONV_DGRAD, l, 2.0 * active_rows * taps * sp.cin * C, rows_bytes(dy) + 2.0 * active_rows * sp.cin * P.esz, ls,
                vn_conv_gather_gemm_rows(dy.ptr, P.wp_d[l], nullptr, P.dx[l].ptr, (vnDtype)P.dx[l].dtype, cx(P, gl), P.alist, P.acap,
                                         P.acount, 1, nullptr, ls));
            return VN_OK;
        }
        // data gradient
        Rows dx = P.dx[l];
        if (l == 0) {
            if (!d_input) return VN_OK;
            dx = dense_rows(d_input, P.adt, B, cfg->D, cfg->H, cfg->W, 128);
        }
        const int64_t os[4] = {dx.sB, dx.sD, dx.sH, dx.sW};
        vnConv gd = sp.transposed ? geom(dy, P.in_dims[l], C, sp.cin, sp.k, sp.s, ONE, sp.p, ONE, os)
                                  : geom(dy, P.in_dims[l], C, sp.cin, sp.k, ONE, NEG, np, sp.s, os);
        // small-image 3x3 layers: the BatchNorm-backward sums of the layer below come out of this launch's epilogue (its
        // own reduce launch — 4-8 us plus the gap between two dependent launches — is skipped in do_layer(l - 1))
        const bool below_plain = l >= 2 && l - 1 != L_M2 && l - 1 != L_D1 && l - 1 != L_D2 && l - 1 != L_D3 && l != L_D1 &&
                                 l != L_D2 && l != L_D3 && !P.spec[l - 1].transposed;
        if (fuse_bwd_reduce_on() && single_call && !accumulate && !on_side && below_plain && vn_conv_plan_id(cx(P, gd)) == 123 &&
            P.y[l - 1].sB == dx.sB && P.y[l - 1].sD == dx.sD && P.y[l - 1].sH == dx.sH && P.y[l - 1].sW == dx.sW &&
            vn_conv_stats_slab_rows(cx(P, gd)) <= P.bslab_rows[l - 1]) {
            RTT(T_CONV_DGRAD, l, layer_flops(sp, P.in_dims[l], P.odims[l], B), rows_bytes(dy) + 2.0 * rows_bytes(dx), ls,
                vn_conv_dgrad_bn_bwd(dy.ptr, P.wp_d[l], dx.ptr, (vnDtype)dx.dtype, cx(P, gd), P.y[l - 1].ptr, (vnDtype)P.y[l - 1].dtype,
                                     P.stats[l - 1], P.bslab[l - 1], ls));
            fused_rows[l - 1] = vn_conv_stats_slab_rows(cx(P, gd));
            return VN_OK;
        }
        RTT(T_CONV_DGRAD, l, layer_flops(sp, P.in_dims[l], P.odims[l], B), rows_bytes(dy) + rows_bytes(dx), ls,
            vn_conv_gather_gemm(dy.ptr, P.wp_d[l], nullptr, dx.ptr, (vnDtype)dx.dtype, cx(P, gd), accumulate ? 1 : 0, nullptr, ls));
        return VN_OK;
    };
    // Single-call backward with a side stream: deconv2 / deconv1 only depend on the heads' data gradient, so their
    // whole backward is issued first, on the side stream, and overlaps the block3 / block2 chains (50x44 and 100x88
    // images: under-filled launches).  Their data gradient then WRITES the shared buffer and the strided block3.0 /
    // block2.0 data gradient accumulates into it (instead of the other way round).
    static const int early_on = vn_knob("VN_EARLY_DECONV", 1);   // tuning aid
    const bool early = early_on && ws != hs && seg_begin == 0 && seg_end == NL + 1;
    hipEvent_t ev_d2 = nullptr, ev_d1 = nullptr;
    if (early) {
        if (!heads_forked) RT(fork());
        RT(do_layer(L_D2, wstream, true, false));
        ev_d2 = net->next_event();
        if (!ev_d2) return VN_EINVAL;
        VN_HIP(hipEventRecord(ev_d2, ws));
        RT(do_layer(L_D1, wstream, true, false));
        ev_d1 = net->next_event();
        if (!ev_d1) return VN_EINVAL;
        VN_HIP(hipEventRecord(ev_d1, ws));
    }
    // cfg->bucket_events (single call, side stream): at the end of each parameter group (= DDP bucket: heads+deconv3+
    // block3 | deconv2+block2+deconv1 | block1 | middle_layer) the group's weight gradients are unpacked on the SIDE
    // stream and two events mark "final" — the main stream never waits for the side stream (vn_net_wait_bucket).
    const bool bucket_ev = cfg->bucket_events && ws != hs && seg_begin == 0 && seg_end == NL + 1;
    int u_done = 0, z_done = 0;
    auto bucket_done = [&](int b) -> int {
        RT(flush());
            if (b == 3 && m0_on_main) RT(fork());     // the first layer's partials come from the main stream
        RTT(T_UNPACK, -1, 0.0, unpack_bytes(unpack + u_done, nu - u_done), wstream, vn_unpack_wgrads_batch(unpack + u_done, nu - u_done, wstream));
        RT(after_unpack(u_done, nu, wstream));
        u_done = nu;
        if (zj.n > z_done) {   // (bias gradients in front of a train-mode BatchNorm: zero; off the main chain)
            ZeroJobs part{};
            part.n = zj.n - z_done;
            for (int i = 0; i < part.n; ++i) { part.ptr[i] = zj.ptr[z_done + i]; part.len[i] = zj.len[z_done + i]; }
            RTT(T_MISC, -1, 0.0, 0.0, wstream, zero_many(part, ws));
            z_done = zj.n;
        }
        VN_HIP(hipEventRecord(net->bucket_ev[b][1], ws));
        VN_HIP(hipEventRecord(net->bucket_ev[b][0], hs));   // (the BatchNorm gradients of the group: written on the main stream)
        return VN_OK;
    };
    for (int oi = 0; oi < NL; ++oi) {
        if (oi + 1 < seg_begin || oi + 1 >= seg_end) continue;
        const int l = order[oi];
        if (!(early && (l == L_D2 || l == L_D1))) {
            bool accumulate = (l == L_D1 || l == L_D2);
            if (early && l == L_B3) { VN_HIP(hipStreamWaitEvent(hs, ev_d2, 0)); accumulate = true; }
            if (early && l == L_B2) { VN_HIP(hipStreamWaitEvent(hs, ev_d1, 0)); accumulate = true; }
            RT(do_layer(l, stream, false, accumulate));
            // flush at the end of every block chain and after each Conv3d (their weight gradients are the long ones)
            // flush (fork + issue the queued weight gradients on the side stream): after EVERY block1 layer (their weight
            // gradients are 60 us each: started early they fill the side stream while the chain goes on; measured 480 vs
            // 473 pc/s against one flush per block), at the end of the block2 / block3 chains (per-layer flushes there
            // measure nothing: 470) and after each Conv3d
            if ((l >= L_B1 && l < L_D1) || l == L_B2 || l == L_B3 || l <= L_M2) RT(flush());
            if (l == L_B1 && tail_balance) {
                RTT(T_UNPACK, -1, 0.0, unpack_bytes(unpack, nu), wstream, vn_unpack_wgrads_batch(unpack, nu, wstream));
                RT(after_unpack(0, nu, wstream));
                u_early = nu;
            }
        }
        if (bucket_ev) {   // group ends (backward order): ... block3.0 | ... deconv1 | ... block1.0 | ... middle_layer.0
            if (l == L_B3) RT(bucket_done(0));
            else if (l == L_D1) RT(bucket_done(1));
            else if (l == L_B1) RT(bucket_done(2));
            else if (l == 0) RT(bucket_done(3));
        }
    }
    RT(flush());
    if (bucket_ev) return VN_OK;   // everything unpacked / zeroed per group; the caller joins the side stream
    if (ws != hs && cfg->defer_join && seg_end == NL + 1) {
        // last segment, join deferred to the caller: the unpack follows the weight gradients on the side stream
        if (m0_on_main) RT(fork());     // the first layer's partials come from the main stream
        RTT(T_UNPACK, -1, 0.0, unpack_bytes(unpack + u_early, nu - u_early), wstream, vn_unpack_wgrads_batch(unpack + u_early, nu - u_early, wstream));
        RT(after_unpack(u_early, nu, wstream));
        if (zj.n > 0) {
            RTT(T_MISC, -1, 0.0, 0.0, wstream, zero_many(zj, ws));
        }
        return VN_OK;
    }
    if (ws != hs) {   // join: the segment's weight-gradient partials are complete before they are summed / unpacked
        hipEvent_t e = net->next_event();
        if (!e) return VN_EINVAL;
        VN_HIP(hipEventRecord(e, ws));
        VN_HIP(hipStreamWaitEvent(hs, e, 0));
    }
    RTT(T_UNPACK, -1, 0.0, unpack_bytes(unpack, nu), stream, vn_unpack_wgrads_batch(unpack, nu, stream));
    RT(after_unpack(0, nu, stream));
    if (zj.n > 0) {
        RTT(T_MISC, -1, 0.0, 0.0, stream, zero_many(zj, hs));
    }
    return VN_OK;
}

extern "C" int vn_net_wait_bucket(vnNet *net, int32_t bucket, vnStream stream) {
    VN_CHECK_ARG(net && bucket >= 0 && bucket < 4);
    VN_HIP(hipStreamWaitEvent(vn_stream(stream), net->bucket_ev[bucket][0], 0));
    VN_HIP(hipStreamWaitEvent(vn_stream(stream), net->bucket_ev[bucket][1], 0));
    return VN_OK;
}

// ---- one call per train step ------------------------------------------------------------------------------------------
namespace {
// heads_w (16,768) / heads_b (16) <- prob_conv (2 rows) then reg_conv (14 rows): what the two torch.cat of the host path do
__global__ void __launch_bounds__(256) k_heads_cat(const float *__restrict__ pw, const float *__restrict__ rw,
                                                   const float *__restrict__ pb, const float *__restrict__ rb,
                                                   float *__restrict__ w, float *__restrict__ b) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < 16 * 768) w[i] = i < 2 * 768 ? pw[i] : rw[i - 2 * 768];
    if (i < 16) b[i] = i < 2 ? pb[i] : rb[i - 2];
}
__global__ void __launch_bounds__(64) k_tick(int64_t *const *__restrict__ ctrs, int n) {
    if ((int)threadIdx.x < n) *ctrs[threadIdx.x] += 1;
}
}  // namespace

extern "C" int vn_net_step(vnNet *net, const vnNetConfig *cfg, const vnStep *s) {
    VN_CHECK_ARG(net && cfg && s);
    VN_CHECK_ARG(s->feature && s->coord && s->K > 0 && s->T > 0 && s->vfe_ws && s->voxelwise && s->vfe_stats && s->vw_rows &&
                 s->d_voxelwise);
    VN_CHECK_ARG(s->prob_w && s->prob_b && s->reg_w && s->reg_b && s->heads_w && s->heads_b && s->d_heads_w && s->d_heads_b);
    VN_CHECK_ARG(s->layers && s->grads && s->ws && s->prob && s->reg && s->d_prob && s->d_reg);
    VN_CHECK_ARG(s->pos && s->neg && s->targets && s->loss_ws && s->loss5 && s->g_loss);
    VN_CHECK_ARG(s->n_chunks >= 0 && (s->n_chunks == 0 || (s->chunks && s->opt_ws)));
    VN_CHECK_ARG(s->side_stream && s->side_stream != s->stream);
    VN_CHECK_ARG(s->n_bn_counters >= 0 && s->n_bn_counters <= 64);
    if (!cfg->sparse_first || !cfg->training || (cfg->grad_storage & 16)) return VN_EUNSUPPORTED;
    if (cfg->mode == 0 ? s->vw_rows == (void *)s->voxelwise : s->vw_rows != (void *)s->voxelwise) return VN_EINVAL;
    hipStream_t hs = vn_stream(s->stream), ss = vn_stream(s->side_stream);
    const int32_t hf = cfg->H / cfg->block1_stride, wf = cfg->W / cfg->block1_stride;
    vnNetConfig c = *cfg;
    // side stream: what does not depend on the voxel features, the first layer's needs first
    hipEvent_t ev = net->next_event();
    VN_HIP(hipEventRecord(ev, hs));
    VN_HIP(hipStreamWaitEvent(ss, ev, 0));
    if (s->bn_counters && s->n_bn_counters > 0) {       // nothing in the step reads them
        k_tick<<<1, 64, 0, ss>>>(s->bn_counters, s->n_bn_counters);
        VN_HIP(hipGetLastError());
    }
    c.prepared = 1;
    RT(vn_net_prepare(net, &c, s->layers, nullptr, s->coord, s->K, s->ws, s->ws_bytes, s->side_stream));
    k_heads_cat<<<(16 * 768 + 255) / 256, 256, 0, ss>>>(s->prob_w, s->reg_w, s->prob_b, s->reg_b, s->heads_w, s->heads_b);
    VN_HIP(hipGetLastError());
    c.prepared = 2;
    RT(vn_net_prepare(net, &c, s->layers, s->heads_w, s->coord, s->K, s->ws, s->ws_bytes, s->side_stream));
    // the loss's per-sample normalisers depend on the target maps only: here, beside the encoder, not between the heads and
    // their backward
    if (s->targets_stream && s->targets_stream != s->side_stream) {
        ev = net->next_event();
        VN_HIP(hipEventRecord(ev, vn_stream(s->targets_stream)));
        VN_HIP(hipStreamWaitEvent(ss, ev, 0));
    }
    RT(vn_rpn_loss_norm(s->pos, s->neg, cfg->B, hf, wf, s->loss_ws, s->loss_ws_bytes, s->side_stream));
    hipEvent_t ev_norm = net->next_event();
    VN_HIP(hipEventRecord(ev_norm, ss));
    // main stream: encoder, network, loss
    if (cfg->mode == 0)   // bf16 mode: the encoder's last pass writes the bf16 rows too (no cast launch in front of the network)
        RT(vn_vfe_fwd_rows(s->feature, s->K, s->T, &s->vfe, 1, s->bn_momentum, s->bn_eps, s->voxelwise, s->vw_rows, s->vfe_stats,
                           s->vfe_ws, s->vfe_ws_bytes, s->stream));
    else
        RT(vn_vfe_fwd(s->feature, s->K, s->T, &s->vfe, 1, s->bn_momentum, s->bn_eps, s->voxelwise, s->vfe_stats, s->vfe_ws,
                      s->vfe_ws_bytes, s->stream));
    c.prepared = 1;
    RT(vn_net_forward(net, &c, s->layers, s->heads_w, s->heads_b, nullptr, s->coord, s->vw_rows, s->K, s->ws, s->ws_bytes,
                      s->prob, s->reg, s->stream, s->side_stream));
    // the loss: ONE launch between the heads and their backward (sums and gradients in one pass); its five output scalars are
    // finished on the side stream (nobody's input)
    VN_HIP(hipStreamWaitEvent(hs, ev_norm, 0));
    {   // ... and the same pass leaves the heads' gradient rows (what vn_heads_bwd makes of d_prob / d_reg / prob) in the arena
        Plan P;
        if (!make_plan(&c, s->K, static_cast<char *>(s->ws), &P)) return VN_EUNSUPPORTED;
        if (s->ws_bytes < P.bytes) return VN_EWORKSPACE;
        RT(vn_rpn_loss_fwd_bwd_rows(s->prob, s->reg, s->pos, s->neg, s->targets, cfg->B, hf, wf, s->alpha, s->beta, s->sigma,
                                    s->loss_ws, s->loss_ws_bytes, s->g_loss, s->d_prob, s->d_reg, P.d_rows.ptr, (vnDtype)P.adt, 16, 0,
                                    s->stream));
        net->heads_rows_ready = P.d_rows.ptr;
    }
    ev = net->next_event();
    VN_HIP(hipEventRecord(ev, hs));
    VN_HIP(hipStreamWaitEvent(ss, ev, 0));
    RT(vn_rpn_loss_finalize(s->loss_ws, s->loss_ws_bytes, cfg->B, hf, wf, s->alpha, s->beta, s->loss5, s->side_stream));
    // backward: the last weight gradients and their unpack run on the side stream beside the encoder's backward
    c.defer_join = 1;
    RT(vn_net_backward(net, &c, s->layers, s->heads_w, s->d_prob, s->d_reg, s->prob, nullptr, s->coord, s->vw_rows, s->K, s->ws,
                       s->ws_bytes, s->grads, s->d_heads_w, s->d_heads_b, s->d_voxelwise, 0, 24, s->stream, s->side_stream));
    RT(vn_vfe_bwd(s->feature, s->K, s->T, &s->vfe, s->vfe_stats, s->d_voxelwise, &s->vfe_grads, s->vfe_ws, s->vfe_ws_bytes, 1,
                  s->stream));
    ev = net->next_event();
    VN_HIP(hipEventRecord(ev, ss));
    VN_HIP(hipStreamWaitEvent(hs, ev, 0));
    if (s->n_chunks > 0)
        RT(vn_clip_sgd(s->chunks, s->n_chunks, s->max_norm, s->lr, s->scale_grads, s->opt_ws, s->opt_ws_bytes, s->total_norm,
                       s->stream));
    return VN_OK;
}
