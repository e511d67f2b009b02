// Image observations for discrete envs (row I1): ImageMultiDiscrete.generate_image,
// /root/reference/mdp_playground/spaces/image_multi_discrete.py:129-288, called from
// rl_toy_env.py:2095-2096 (step) and :2347-2350 (reset).
//
// One workgroup per env image.  Thread 0 draws the transform variates from the env's image-space
// PCG64 stream in the reference's order (scale: random(); shift: integers() x2; rotate:
// integers(360); flip: integers(2) [+ integers(2)]), the block then writes the uint8[W][H][1]
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

__device__ __forceinline__ uint32_t pixel(const ImageArgs &a, const Xform &t, const uint8_t *tp,
                                          const int32_t *rc, int x, int y) {
    // obs[x][y] = final[y][x]; undo the flip
    int fx = (t.flip == 1) ? a.W - 1 - x : x;
    int fy = (t.flip == 2) ? a.H - 1 - y : y;
    int xs, ys;
    if (t.angle == 0) { xs = fx; ys = fy; }
    else if (t.angle == 180) { xs = a.W - 1 - fx; ys = a.H - 1 - fy; }
    else if (t.angle == 90 && a.W == a.H) { xs = a.W - 1 - fy; ys = fx; }
    else if (t.angle == 270 && a.W == a.H) { xs = fy; ys = a.H - 1 - fx; }
    else {
        xs = (rc[2] + rc[0] * fx + rc[1] * fy) >> 16;
        ys = (rc[5] + rc[3] * fx + rc[4] * fy) >> 16;
        if (xs < 0 || xs >= a.W || ys < 0 || ys >= a.H) return 0;
    }
    const int half = a.tpl / 2;
    int tx = xs - t.cx + half, ty = ys - t.cy + half;
    if (tx < 0 || tx >= a.tpl || ty < 0 || ty >= a.tpl) return 0;
    return tp[ty * a.tpl + tx];
}

__device__ void render(const ImageArgs &a, const Xform &t, int state, uint8_t *lds_tpl,
                       uint8_t *__restrict__ out) {
    const int tid = threadIdx.x;
    const int ri = t.R - a.r_min;
    const size_t sr = (size_t)state * a.n_radii + ri;
    const int cx_cls = a.cls_x[sr * a.W + t.cx], cy_cls = a.cls_y[sr * a.H + t.cy];
    const uint8_t *gt = a.tpl_data + ((sr * a.n_cls_x + cx_cls) * a.n_cls_y + cy_cls) * (size_t)(a.tpl * a.tpl);
    __syncthreads(); // previous render (if any) is done with lds_tpl
    for (int k = tid; k < a.tpl * a.tpl; k += blockDim.x) lds_tpl[k] = gt[k];
    __syncthreads();
    const int32_t *rc = a.rot + t.angle * 6;
    const int total = a.W * a.H;
    for (int p0 = tid * 4; p0 < total; p0 += blockDim.x * 4) {
        uint32_t word = 0;
#pragma unroll
        for (int b = 0; b < 4; b++) {
            int p = p0 + b;
            if (p < total) {
                int x = p / a.H, y = p - x * a.H;
                word |= pixel(a, t, lds_tpl, rc, x, y) << (8 * b);
            }
        }
        if (p0 + 3 < total) *(uint32_t *)(out + p0) = word;
        else for (int b = 0; p0 + b < total; b++) out[p0 + b] = (uint8_t)(word >> (8 * b));
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
    extern __shared__ __align__(16) uint8_t lds_tpl[];
    __shared__ Xform xf[2];
    const int i = blockIdx.x;
    if (mask && !mask[i]) return;
    const bool two = a.autoreset && term && (term[i] | trunc[i]);
    if (threadIdx.x == 0) {
        Pcg64 g;
        g.load(a.rng_s, a.rng_inc, i);
        uint2 hh = a.rng_half[i];
        Half32 h{hh.x, hh.y};
        xf[0] = draw_xform(a, g, h);
        if (two) xf[1] = draw_xform(a, g, h);
        g.store(a.rng_s, i);
        a.rng_half[i] = make_uint2(h.has32, h.u32);
    }
    __syncthreads();
    const size_t isz = (size_t)a.W * a.H;
    if (two) {
        if (img_final) render(a, xf[0], state_final[i], lds_tpl, img_final + (size_t)i * isz);
        render(a, xf[1], state_out[i], lds_tpl, img_out + (size_t)i * isz);
    } else {
        render(a, xf[0], state_out[i], lds_tpl, img_out + (size_t)i * isz);
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
    const size_t lds = ((size_t)a.tpl * a.tpl + 15) & ~(size_t)15;
    hipLaunchKernelGGL(k_image_obs, dim3(a.N), dim3(kBlock), lds, s, a, state_out, state_final, term,
                       trunc, mask, img_out, img_final);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { h->err = std::string("k_image_obs launch: ") + hipGetErrorString(e); return MDPP_EHIP; }
    return MDPP_OK;
}

} // namespace mdpp
