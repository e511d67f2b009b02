// Image observations for discrete envs (row I1): ImageMultiDiscrete.generate_image,
// /root/reference/mdp_playground/spaces/image_multi_discrete.py:129-288, called from
// rl_toy_env.py:2095-2096 (step) and :2347-2350 (reset).
//
// One wavefront per env image.  The wave draws the transform variates from the env's image-space
// PCG64 stream in the reference's order (scale: random(); shift: integers() x2; rotate:
// integers(360); flip: integers(2) [+ integers(2)]) and then writes the uint8[W][H][1]
// observation with one dword (4 pixels) per lane per store.  A pixel is produced by walking the
// reference's pipeline backwards:  obs[x][y] = final[y][x]  (the .T at :264-266)
//   final = flip(rot)                                   (:257-262)
//   rot[y][x] = src[ys][xs], (xs, ys) = Pillow's NEAREST affine map in 16.16 fixed point
//               (Image.rotate -> ImagingTransformAffine "affine_fixed"; exact transposes for
//               0/90/180/270 on square images)           (:247-254)
//   src = polygon raster: a host-made template (Pillow ImageDraw.polygon at a canonical centre,
//         one per state x radius x vertex-rounding class) translated to the drawn centre (:186-245)
// Templates (<= 1.7 KB each for R = 20) are staged in LDS; the 6 fixed-point coefficients of the
// drawn angle come from a 360-row table made on the host.
#include "mdpp_internal.hpp"
#include "mdpp_rng.hpp"

namespace mdpp {

struct ImageArgs {
    int32_t N, W, H, S;
    int32_t has_scale, has_shift, has_rotate, has_flip, sh_quant, ro_quant;
    int32_t r0, r_min, r_max, tpl, n_radii, n_cls_x, n_cls_y, autoreset;
    double log_min_r, log_max_r;
    const uint8_t *tpl_data;   // [S][n_radii][n_cls_x][n_cls_y][tpl][tpl], indexed [ty][tx]
    const int16_t *cls_x;      // [S][n_radii][W]
    const int16_t *cls_y;      // [S][n_radii][H]
    const int32_t *rot;        // [360][6] = a0 a1 a2 a3 a4 a5
    ulonglong2 *rng_s, *rng_inc;
    uint2 *rng_half;           // {has_uint32, uinteger}
};

struct Xform { int R, cx, cy, angle, flip; };

__device__ __forceinline__ int floordiv_i(int a, int b) {
    int q = a / b;
    return ((a % b != 0) && ((a < 0) != (b < 0))) ? q - 1 : q;
}

__device__ Xform draw_xform(const ImageArgs &a, Pcg64 &g, Half32 &h) {
    Xform x;
    x.R = a.r0;
    x.cx = a.W / 2; x.cy = a.H / 2;              // int(width / 2)
    if (a.has_scale) {
        double ls = a.log_min_r + np_random(g) * (a.log_max_r - a.log_min_r);
        x.R = (int)exp(ls);
    }
    if (a.has_shift) {
        double mw = a.W / 2.0 - x.R, mh = a.H / 2.0 - x.R;
        int aw = np_integers(g, h, (int)(-mw + 1), (int)mw);
        int ah = np_integers(g, h, (int)(-mh + 1), (int)mh);
        x.cx += floordiv_i(aw, a.sh_quant) * a.sh_quant;
        x.cy += floordiv_i(ah, a.sh_quant) * a.sh_quant;
    }
    x.angle = 0;
    if (a.has_rotate) {
        int r = np_integers(g, h, 0, 360);
        x.angle = floordiv_i(r, a.ro_quant) * a.ro_quant;
    }
    x.flip = 0;
    if (a.has_flip) {
        if (np_integers(g, h, 0, 2) == 0) x.flip = (np_integers(g, h, 0, 2) == 0) ? 1 : 2;
    }
    return x;
}

// One wavefront per env image (4 images per 256-thread workgroup): no workgroup barrier, the
// transform draw is done redundantly by all 64 lanes from wave-uniform addresses (same cost as
// one lane), the template is staged into the wave's own slice of LDS.
//
// The whole final->source pixel map (transpose, flip, rotation) is ONE integer affine map per
// image: the host table holds Pillow's 16.16 coefficients for every angle (exact integer rows for
// 0/90/180/270 on square images, where Pillow transposes instead), the flip is folded into them
// here.  Pixels outside the polygon's bounding circle (about 3/4 of an 84x84 image at R = 20) are
// written as zeros without evaluating the map.
__device__ __forceinline__ void render(const ImageArgs &a, const Xform &t, int state, uint8_t *lds_tpl,
                                       uint8_t *__restrict__ out, int lane) {
    const int ri = t.R - a.r_min;
    const size_t sr = (size_t)state * a.n_radii + ri;
    const int cx_cls = a.cls_x[sr * a.W + t.cx], cy_cls = a.cls_y[sr * a.H + t.cy];
    const int tsz = a.tpl * a.tpl;
    const uint8_t *gt = a.tpl_data + ((sr * a.n_cls_x + cx_cls) * a.n_cls_y + cy_cls) * (size_t)tsz;
    // template -> this wave's LDS slice (wave-local: LDS ops of one wave complete in order)
    for (int k = lane * 4; k < tsz; k += 64 * 4) {
        uint32_t w = 0;
#pragma unroll
        for (int b = 0; b < 4; b++) if (k + b < tsz) w |= (uint32_t)gt[k + b] << (8 * b);
        *(uint32_t *)(lds_tpl + k) = w;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    // source = A * (fx, fy) + b in 16.16, with (fx, fy) = flip(x, y) folded in
    int a0 = a.rot[t.angle * 6 + 0], a1 = a.rot[t.angle * 6 + 1], a2 = a.rot[t.angle * 6 + 2];
    int a3 = a.rot[t.angle * 6 + 3], a4 = a.rot[t.angle * 6 + 4], a5 = a.rot[t.angle * 6 + 5];
    if (t.flip == 1) { a2 += a0 * (a.W - 1); a5 += a3 * (a.W - 1); a0 = -a0; a3 = -a3; }
    if (t.flip == 2) { a2 += a1 * (a.H - 1); a5 += a4 * (a.H - 1); a1 = -a1; a4 = -a4; }
    // centre of the polygon in final-image coordinates: invert the 2x2 part (a rotation, so the
    // inverse is the transpose up to the 16.16 scale); one pixel of slack covers the rounding
    const float fa0 = a0 * (1.0f / 65536.0f), fa1 = a1 * (1.0f / 65536.0f);
    const float fa3 = a3 * (1.0f / 65536.0f), fa4 = a4 * (1.0f / 65536.0f);
    const float sx = (float)t.cx + 0.5f - a2 * (1.0f / 65536.0f), sy = (float)t.cy + 0.5f - a5 * (1.0f / 65536.0f);
    const float det = fa0 * fa4 - fa1 * fa3;
    const float fcx = (fa4 * sx - fa1 * sy) / det, fcy = (fa0 * sy - fa3 * sx) / det;
    const float rad = (float)t.R + 3.0f, rad2 = rad * rad;
    const int half = a.tpl / 2, ox = t.cx - half, oy = t.cy - half;
    const int total = a.W * a.H;
    auto pixel = [&](int x, int y) -> uint32_t {
        const int xs = (a2 + a0 * x + a1 * y) >> 16, ys = (a5 + a3 * x + a4 * y) >> 16;
        const uint32_t tx = (uint32_t)(xs - ox), ty = (uint32_t)(ys - oy);
        const bool in = (uint32_t)xs < (uint32_t)a.W && (uint32_t)ys < (uint32_t)a.H &&
                        tx < (uint32_t)a.tpl && ty < (uint32_t)a.tpl;
        return in ? (uint32_t)lds_tpl[in ? ty * a.tpl + tx : 0] : 0u;
    };
    if ((a.H & 3) == 0) {
        // dword q covers pixels (x, 4*yq .. 4*yq+3); q advances by 64 per iteration, carried in
        // (x, yq) without a division per dword
        const int HQ = a.H >> 2, nq = total >> 2;
        int x = lane / HQ, yq = lane - x * HQ;
        const int dx = 64 / HQ, dy = 64 - dx * HQ;
        for (int q = lane; q < nq; q += 64) {
            uint32_t word = 0;
            const float ddx = (float)x - fcx, ddy = (float)(4 * yq) + 1.5f - fcy;
            const bool near = ddx * ddx + ddy * ddy <= rad2 + 3.0f * rad + 2.25f; // any of the 4 pixels within rad
            if (__builtin_amdgcn_ballot_w64(near) != 0) {
                if (near) {
#pragma unroll
                    for (int b = 0; b < 4; b++) word |= pixel(x, 4 * yq + b) << (8 * b);
                }
            }
            *(uint32_t *)(out + 4 * (size_t)q) = word;
            x += dx; yq += dy;
            if (yq >= HQ) { yq -= HQ; x += 1; }
        }
    } else {
        for (int p = lane; p < total; p += 64) {
            int x = p / a.H, y = p - x * a.H;
            out[p] = (uint8_t)pixel(x, y);
        }
    }
}

// state_out: the state whose image goes to img_out; for envs that were reset in this step
// (autoreset && (term|trunc)) the terminal state (state_final) is rendered first, consuming the
// draws the reference's step() made before its reset() (image goes to img_final if given).
__global__ __launch_bounds__(kBlock) void k_image_obs(ImageArgs a, const int32_t *__restrict__ state_out,
                                                      const int32_t *__restrict__ state_final,
                                                      const uint8_t *__restrict__ term,
                                                      const uint8_t *__restrict__ trunc,
                                                      const uint8_t *__restrict__ mask,
                                                      uint8_t *__restrict__ img_out,
                                                      uint8_t *__restrict__ img_final) {
    extern __shared__ __align__(16) uint8_t lds_all[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int slice = (a.tpl * a.tpl + 15) & ~15;
    // wave-uniform env index (readfirstlane makes the uniformity visible to the compiler, so the
    // stream state and the transform live in SGPRs)
    const int i = __builtin_amdgcn_readfirstlane(blockIdx.x * (kBlock / 64) + wave);
    if (i >= a.N) return;
    if (mask && !mask[i]) return;
    uint8_t *lds_tpl = lds_all + wave * slice;
    const bool two = a.autoreset && term && (term[i] | trunc[i]);
    Pcg64 g;
    g.load(a.rng_s, a.rng_inc, i);
    const uint2 hh = a.rng_half[i];
    Half32 h{hh.x, hh.y};
    const Xform x0 = draw_xform(a, g, h);
    Xform x1 = x0;
    if (two) x1 = draw_xform(a, g, h);
    if (lane == 0) {
        g.store(a.rng_s, i);
        a.rng_half[i] = make_uint2(h.has32, h.u32);
    }
    const size_t isz = (size_t)a.W * a.H;
    if (two) {
        if (img_final) render(a, x0, state_final[i], lds_tpl, img_final + (size_t)i * isz, lane);
        render(a, x1, state_out[i], lds_tpl, img_out + (size_t)i * isz, lane);
    } else {
        render(a, x0, state_out[i], lds_tpl, img_out + (size_t)i * isz, lane);
    }
}

int launch_image_obs(mdpp_env *h, const int32_t *state_out, const int32_t *state_final,
                     const uint8_t *term, const uint8_t *trunc, const uint8_t *mask,
                     uint8_t *img_out, uint8_t *img_final, hipStream_t s) {
    const mdpp_config &c = h->cfg;
    ImageArgs a;
    a.N = c.num_envs; a.W = c.img_w; a.H = c.img_h; a.S = c.S;
    a.has_scale = c.img_has_scale; a.has_shift = c.img_has_shift; a.has_rotate = c.img_has_rotate;
    a.has_flip = c.img_has_flip; a.sh_quant = c.img_sh_quant > 0 ? c.img_sh_quant : 1;
    a.ro_quant = c.img_ro_quant > 0 ? c.img_ro_quant : 1;
    a.r0 = c.img_r0; a.r_min = c.img_r_min; a.r_max = c.img_r_max; a.tpl = c.img_tpl_size;
    a.n_radii = h->img_n_radii; a.n_cls_x = h->img_n_cls_x; a.n_cls_y = h->img_n_cls_y;
    a.autoreset = c.autoreset;
    a.log_min_r = c.img_log_min_r; a.log_max_r = c.img_log_max_r;
    a.tpl_data = (const uint8_t *)h->d_img_tpl; a.cls_x = (const int16_t *)h->d_img_clsx;
    a.cls_y = (const int16_t *)h->d_img_clsy; a.rot = (const int32_t *)h->d_img_rot;
    a.rng_s = (ulonglong2 *)h->d_rng_s[MDPP_STREAM_IMAGE];
    a.rng_inc = (ulonglong2 *)h->d_rng_inc[MDPP_STREAM_IMAGE];
    a.rng_half = (uint2 *)h->d_rng_half;
    const int per_block = kBlock / 64;
    const size_t lds = (size_t)per_block * (((size_t)a.tpl * a.tpl + 15) & ~(size_t)15);
    if (lds > 64 * 1024) { h->err = "k_image_obs: polygon template too large for LDS"; return MDPP_EUNSUPPORTED; }
    hipLaunchKernelGGL(k_image_obs, dim3((a.N + per_block - 1) / per_block), dim3(kBlock), lds, s, a,
                       state_out, state_final, term, trunc, mask, img_out, img_final);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { h->err = std::string("k_image_obs launch: ") + hipGetErrorString(e); return MDPP_EHIP; }
    return MDPP_OK;
}

} // namespace mdpp
