// Image observations for discrete envs (row I1): ImageMultiDiscrete.generate_image,
// /root/reference/mdp_playground/spaces/image_multi_discrete.py:129-288, called from
// rl_toy_env.py:2095-2096 (step) and :2347-2350 (reset).
//
// Two kernels per batch of K steps x N envs images:
//  k_image_draw   one LANE per env: draws the transform variates of the env's images from its
//                 image-space PCG64 stream in the reference's order (scale: random(); shift:
//                 integers() x2; rotate: integers(360); flip: integers(2) [+ integers(2)]), for
//                 every step of the batch, and resolves everything that is per image and scalar
//                 into a 64-byte record: the fixed-point map, the template, the bounding box.
//                 (Doing this inside the render kernel put ~400 scalar instructions per wave on
//                 the CU's single scalar unit: 6 us of a 20 us launch, and vector loads of the
//                 per-image scalars wait on vmcnt together with the previous image's stores;
//                 profiles/archive/r01_ablation_image_kernel.txt.)
//  k_image_obs*   one WAVEFRONT per image, a pure rasteriser: reads the record with scalar loads
//                 and writes the uint8[W][H][1] observation.  A pixel is produced by walking the
//                 reference's pipeline backwards:
//   obs[x][y] = final[y][x]  (the .T at :264-266)
//   final = flip(rot)                                   (:257-262)
//   rot[y][x] = src[ys][xs], (xs, ys) = Pillow's NEAREST affine map in 16.16 fixed point
//               (Image.rotate -> ImagingTransformAffine "affine_fixed"; exact transposes for
//               0/90/180/270 on square images)           (:247-254)
//   src = polygon raster: a host-made template (Pillow ImageDraw.polygon at a canonical centre,
//         one per state x radius x vertex-rounding class) translated to the drawn centre (:186-245)
// The whole final->source pixel map (transpose, flip, rotation) is ONE integer affine map per
// image: the host table holds Pillow's 16.16 coefficients for every angle (exact integer rows for
// 0/90/180/270 on square images, where Pillow transposes instead), the flip is folded in by the
// draw kernel.  Templates (<= 1.9 KB each for R = 20) are staged in LDS.
#include "mdpp_internal.hpp"
#include "mdpp_rng.hpp"
#include <cstdlib>
#include <type_traits>

namespace mdpp {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x8 __attribute__((ext_vector_type(8)));

// Workgroup b runs on XCD b % 8 (round-robin dispatch).  With MDPP_IMG_XCD every XCD renders one contiguous eighth of a launch's
// pictures -- its L2 then writes back ONE sequential range instead of every eighth 28 KiB piece of the output: cfg4 5 970-6 210 ->
// 5 600 us per launch (0.60-0.62 -> 0.66 of HBM; round 6, tools/ablate.py x0 / x1 on one lease).
#ifndef MDPP_IMG_XCD
#define MDPP_IMG_XCD 1
#endif
__device__ __forceinline__ uint32_t img_xcd_block() {
    return (MDPP_IMG_XCD && (gridDim.x & 7u) == 0u) ? (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3) : blockIdx.x;
}
constexpr int kImgPad = 8;               // zero border of the fast renderer's templates, pixels
constexpr int kImgColDw = 1536;          // dwords of LDS image columns per wave (6 KiB)

// One image to rasterise (written by k_image_draw, read with s_load by the renderers).
struct ImgRec {
    int32_t a[6];        // source = A (x, y) + b in 16.16: xs = (a2 + a0 x + a1 y) >> 16, ys = (a5 + a3 x + a4 y) >> 16
    uint32_t cxy;        // polygon centre in the source image: cx | cy << 16
    uint32_t meta;       // R | two << 10 | skip << 11 | template index << 12
    float fcx, fcy;      // polygon centre in final-image coordinates
    uint32_t xr, qr;     // fast renderer: box of the bounding circle, X0 | X1 << 16 columns, Q0 | Q1 << 16 dword rows
    uint32_t pad[4];
};
static_assert(sizeof(ImgRec) == 64, "ImgRec is read as two s_load_dwordx8");

struct ImageArgs {
    int32_t N, W, H, S;        // S: states the templates cover (max over the sub-spaces)
    int32_t SUB;               // images per observation: 2 with an irrelevant sub-space (one image per sub-space,
                               // concatenated along x, get_image_representation :272-288), else 1
    int32_t has_scale, has_shift, has_rotate, has_flip, sh_quant, ro_quant;
    int32_t r0, r_min, r_max, tpl, n_radii, n_cls_x, n_cls_y, autoreset;
    double log_min_r, log_max_r;
    const uint8_t *tpl_data;   // [S][n_radii][n_cls_x][n_cls_y][tpl][tpl], indexed [ty][tx]
    const uint8_t *tplp_data;  // fast renderer: the same templates inside a kImgPad-wide zero
                               // border, [..][tplp rows][64 B]; tplp = tpl + 2 * kImgPad <= 64
    int32_t tplp;
    int32_t colb;              // fast renderer: bytes of an LDS row that belong to one wave -- 64 (four waves share a row; tplp <= 64) or 128
                               // (two waves: templates up to 128 wide, k_image_obs_wide -- the scale transform's radii)
    const int16_t *cls_x;      // [S][n_radii][W]
    const int16_t *cls_y;      // [S][n_radii][H]
    const int32_t *rot;        // [360][6] = a0 a1 a2 a3 a4 a5
    ulonglong2 *rng_s, *rng_inc;
    uint2 *rng_half;           // {has_uint32, uinteger}
    // Philox streams (round 3): the transforms of tick t come from stream (seed, global env id, t, MDPP_STREAM_IMAGE) in the
    // order the reference draws them (the step's images, then reset()'s where the step ended the episode); an explicit
    // reset() from stream kPhiloxResetImageStream keyed by the reset count
    int32_t coldw;             // fast renderer: dwords of image columns per wave in LDS (host: the widest box + slack)
    const uint32_t *near_tab;  // fast renderer: the near dwords of a polygon as (dx, dq) int8 pairs, [order][cy mod 4][lane][8] (mdpp_capi.hip), or null
    uint32_t *work_ctr;        // fast renderer: [2][kImgCtrs x 32] work counters of this batch's two render launches, one per group of waves,
                               // 128 B apart (zeroed by k_image_draw)
    int32_t philox, is_reset;
    uint64_t philox_seed, ptick;
    const uint64_t *dtick;     // launches captured into a HIP graph: device word added to ptick at run time (tick_now)
    int64_t env_id_offset;
    ImgRec *rec0, *rec1;       // [M]: the image that goes to img_out / the terminal observation of a
                               // step that ends in a reset (img_final)
};

struct Xform { int R, cx, cy, angle, flip; };

__device__ __forceinline__ int floordiv_i(int a, int b) {
    int q = a / b;
    return ((a % b != 0) && ((a < 0) != (b < 0))) ? q - 1 : q;
}

// Bounds of the shift draws (:172-181) when the radius is fixed (no scale transform): hoisted out
// of the per-env serial chain, which is latency-bound (one lane per env).
struct ShiftBounds { int lo_w, hi_w, lo_h, hi_h; };
__device__ __forceinline__ ShiftBounds shift_bounds(const ImageArgs &a, int R) {
    const double mw = a.W / 2.0 - R, mh = a.H / 2.0 - R;
    return ShiftBounds{(int)(-mw + 1), (int)mw, (int)(-mh + 1), (int)mh};   // Generator.integers truncates toward 0
}

template <class G>
__device__ __forceinline__ Xform draw_xform(const ImageArgs &a, const ShiftBounds &fixed, G &g, Half32 &h) {
    Xform x;
    x.R = a.r0;
    x.cx = a.W / 2; x.cy = a.H / 2;              // int(width / 2)
    if (a.has_scale) {
        double ls = a.log_min_r + np_random(g) * (a.log_max_r - a.log_min_r);
        x.R = (int)exp(ls);
    }
    if (a.has_shift) {
        ShiftBounds b = fixed;
        if (a.has_scale) b = shift_bounds(a, x.R);
        const int aw = np_integers(g, h, b.lo_w, b.hi_w);
        const int ah = np_integers(g, h, b.lo_h, b.hi_h);
        x.cx += a.sh_quant == 1 ? aw : floordiv_i(aw, a.sh_quant) * a.sh_quant;
        x.cy += a.sh_quant == 1 ? ah : floordiv_i(ah, a.sh_quant) * a.sh_quant;
    }
    x.angle = 0;
    if (a.has_rotate) {
        const int r = np_integers(g, h, 0, 360);
        x.angle = a.ro_quant == 1 ? r : floordiv_i(r, a.ro_quant) * a.ro_quant;
    }
    x.flip = 0;
    if (a.has_flip) {
        if (np_integers(g, h, 0, 2) == 0) x.flip = (np_integers(g, h, 0, 2) == 0) ? 1 : 2;
    }
    return x;
}

// Records are read through the constant address space: a wave-uniform address there is always a
// scalar load (lgkmcnt), so fetching the next image's record never waits for this image's stores.
struct RecRegs { u32x8 lo; u32x4 hi; };

// The twelve words of an image's record (ImgRec's layout).
__device__ __forceinline__ RecRegs rec_words(const ImageArgs &a, const Xform &t, int state, bool two) {
    // (all three table reads at once, unconditionally -- the class tables exist for every handle: a load behind a branch waits
    //  for the loads before it, four round trips in a row where one does)
    const size_t sr = (size_t)state * a.n_radii + (t.R - a.r_min);
    const int clx = a.cls_x[sr * a.W + t.cx], cly = a.cls_y[sr * a.H + t.cy];
    int a0 = a.rot[t.angle * 6 + 0], a1 = a.rot[t.angle * 6 + 1], a2 = a.rot[t.angle * 6 + 2];
    int a3 = a.rot[t.angle * 6 + 3], a4 = a.rot[t.angle * 6 + 4], a5 = a.rot[t.angle * 6 + 5];
    const int cx_cls = a.n_cls_x > 1 ? clx : 0, cy_cls = a.n_cls_y > 1 ? cly : 0;
    const uint32_t tix = (uint32_t)((sr * a.n_cls_x + cx_cls) * a.n_cls_y + cy_cls);
    // source = A * (fx, fy) + b with (fx, fy) = flip(x, y) folded in
    if (t.flip == 1) { a2 += a0 * (a.W - 1); a5 += a3 * (a.W - 1); a0 = -a0; a3 = -a3; }
    if (t.flip == 2) { a2 += a1 * (a.H - 1); a5 += a4 * (a.H - 1); a1 = -a1; a4 = -a4; }
    const uint32_t cxy = (uint32_t)t.cx | ((uint32_t)t.cy << 16);
    const uint32_t meta = (uint32_t)t.R | (two ? 1u << 10 : 0u) | (tix << 12);
    // centre of the polygon in final-image coordinates: invert the 2x2 part (a rotation, so the
    // inverse is the transpose up to the 16.16 scale); the renderers allow a pixel of slack
    const float fa0 = a0 * (1.0f / 65536.0f), fa1 = a1 * (1.0f / 65536.0f);
    const float fa3 = a3 * (1.0f / 65536.0f), fa4 = a4 * (1.0f / 65536.0f);
    const float sx = (float)t.cx + 0.5f - a2 * (1.0f / 65536.0f), sy = (float)t.cy + 0.5f - a5 * (1.0f / 65536.0f);
    const float det = fa0 * fa4 - fa1 * fa3;
    const float fcx = (fa4 * sx - fa1 * sy) / det, fcy = (fa0 * sy - fa3 * sx) / det;
    // bounding box of the "near" circle (radius R + 4.5 around the centre: every dword with a
    // pixel within R + 3) in (column, dword-row) units, a pixel of slack
    const float rr = (float)t.R + 4.5f;
    const int HQ = a.H >> 2;
    const int X0 = max(0, (int)floorf(fcx - rr) - 1), X1 = min(a.W, (int)floorf(fcx + rr) + 2);
    const int Q0 = max(0, (int)floorf((fcy - rr - 1.5f) * 0.25f) - 1);
    const int Q1 = min(HQ, (int)floorf((fcy + rr - 1.5f) * 0.25f) + 2);
    const uint32_t xr = (uint32_t)X0 | ((uint32_t)max(X1, X0) << 16);
    const uint32_t qr = (uint32_t)Q0 | ((uint32_t)max(Q1, Q0) << 16);
    RecRegs r;
    r.lo = u32x8{(uint32_t)a0, (uint32_t)a1, (uint32_t)a2, (uint32_t)a3, (uint32_t)a4, (uint32_t)a5, cxy, meta};
    r.hi = u32x4{__float_as_uint(fcx), __float_as_uint(fcy), xr, qr};
    return r;
}

__device__ __forceinline__ void make_rec(const ImageArgs &a, const Xform &t, int state, bool two, ImgRec *out) {
    const RecRegs r = rec_words(a, t, state, two);
    u32x4 *o = (u32x4 *)out;
    o[0] = u32x4{r.lo[0], r.lo[1], r.lo[2], r.lo[3]};
    o[1] = u32x4{r.lo[4], r.lo[5], r.lo[6], r.lo[7]};
    o[2] = r.hi;
}

__device__ __forceinline__ uint2 xf_pack(const Xform &x) {
    return make_uint2((uint32_t)x.cx | ((uint32_t)x.cy << 16),
                      (uint32_t)x.angle | ((uint32_t)x.flip << 9) | ((uint32_t)x.R << 11));
}
__device__ __forceinline__ Xform xf_unpack(uint2 v) {
    Xform x;
    x.cx = (int)(v.x & 0xFFFFu); x.cy = (int)(v.x >> 16);
    x.angle = (int)(v.y & 0x1FFu); x.flip = (int)((v.y >> 9) & 3u); x.R = (int)(v.y >> 11);
    return x;
}

#ifndef MDPP_IMG_CTRS
#define MDPP_IMG_CTRS 32
#endif
constexpr int kImgCtrs = MDPP_IMG_CTRS;   // work counters of the fast renderer (k_image_obs_fast; at most 64: mdpp_capi.hip sizes the array)
constexpr int kImgChunk = 64;            // most env steps per batch (mdpp_env::img_chunk; one bit per step in `twos`)

// One lane per env, K <= kImgChunk steps of a batch (time-major [K][N] arrays): the serial part.
// An env that is reset in a step (autoreset && (term | trunc)) draws twice there, like the
// reference's step() then reset(): first the terminal observation's transform (rec1:
// state_final), then the new episode's (rec0).
// REC: build the records right here (K = 1: one launch less); otherwise only leave the drawn
// transforms in the records' pad words for k_image_rec, so that the per-image table lookups and
// stores run one lane per IMAGE instead of serially per env (60 us -> a few us for 16 steps).
constexpr uint32_t kPhiloxResetImageStream = 11;   // an explicit reset()'s image transforms (keyed by the reset count)
template <bool REC, bool PHILOX = false>
__global__ __launch_bounds__(kBlock) void k_image_draw(ImageArgs a, int K, const int32_t *__restrict__ state_out,
                                                       const int32_t *__restrict__ state_final,
                                                       const uint8_t *__restrict__ term,
                                                       const uint8_t *__restrict__ trunc,
                                                       const uint8_t *__restrict__ mask) {
    const long i = (long)blockIdx.x * kBlock + threadIdx.x;
    if (a.work_ctr)                                      // (the renderer of this batch runs after this kernel)
        for (long k = i; k < 2L * kImgCtrs * 32; k += (long)gridDim.x * kBlock) a.work_ctr[k] = 0u;
    if (i >= a.N) return;
    const int SUB = a.SUB;
    if (mask && !mask[i]) {
        for (int k = 0; k < K; k++)
            for (int q = 0; q < SUB; q++)
                a.rec0[((long)k * a.N + i) * SUB + q].meta = a.rec1[((long)k * a.N + i) * SUB + q].meta = 1u << 11; // skip
        return;
    }
    // all the reset flags of the batch in one round trip
    uint64_t twos = 0;
    if (a.autoreset && term) {
#pragma unroll
        for (int k = 0; k < kImgChunk; k++)
            if (k < K) twos |= (uint64_t)((term[(long)k * a.N + i] | trunc[(long)k * a.N + i]) != 0) << k;
    }
    typename std::conditional<PHILOX, Philox, Pcg64>::type g;
    Half32 h{0u, 0u};
    if constexpr (!PHILOX) {
        g.load(a.rng_s, a.rng_inc, i);
        const uint2 hh = a.rng_half[i];
        h = Half32{hh.x, hh.y};
    }
    const ShiftBounds sb = shift_bounds(a, a.r0);
    for (int k = 0; k < K; k++) {
        const long j0 = ((long)k * a.N + i) * SUB;
        const bool two = (twos >> k) & 1ull;
        if constexpr (PHILOX) {                  // this tick's stream; its buffered 32-bit half starts empty
            g.init(a.philox_seed, (uint64_t)(a.env_id_offset + i), tick_now(a) + (uint64_t)k,
                   a.is_reset ? kPhiloxResetImageStream : (uint32_t)MDPP_STREAM_IMAGE);
            h = Half32{0u, 0u};
        }
        // the step's observation first (one image per sub-space, relevant then irrelevant), then --
        // where the step ended the episode -- the images of reset()'s observation, in the same order
        Xform x0[2], x1[2];
        for (int q = 0; q < SUB; q++) x0[q] = draw_xform(a, sb, g, h);
        for (int q = 0; q < SUB; q++) { x1[q] = x0[q]; if (two) x1[q] = draw_xform(a, sb, g, h); }
        for (int q = 0; q < SUB; q++) {
            const long j = j0 + q;
            if (REC) {
                if (two) {
                    make_rec(a, x0[q], state_final[j], false, &a.rec1[j]);
                    make_rec(a, x1[q], state_out[j], true, &a.rec0[j]);
                } else {
                    make_rec(a, x0[q], state_out[j], false, &a.rec0[j]);
                    a.rec1[j].meta = 1u << 11;                                      // skip
                }
            } else {
                const uint2 p0 = xf_pack(x0[q]), p1 = xf_pack(x1[q]);
                *(u32x4 *)a.rec0[j].pad = u32x4{p0.x, p0.y, p1.x, p1.y | (two ? 0x80000000u : 0u)};
            }
        }
    }
    if constexpr (!PHILOX) {
        g.store(a.rng_s, i);
        a.rng_half[i] = make_uint2(h.has32, h.u32);
    }
}

// One lane per image: transforms (pad words of rec0) -> records.  mask is nullptr here (K > 1).
__global__ __launch_bounds__(kBlock) void k_image_rec(ImageArgs a, long M, const int32_t *__restrict__ state_out,
                                                      const int32_t *__restrict__ state_final) {
    const long j = (long)blockIdx.x * kBlock + threadIdx.x;
    if (j >= M) return;
    const u32x4 p = *(const u32x4 *)a.rec0[j].pad;
    const bool two = p.w >> 31;
    const Xform x0 = xf_unpack(make_uint2(p.x, p.y)), x1 = xf_unpack(make_uint2(p.z, p.w & 0x7FFFFFFFu));
    if (two) {
        make_rec(a, x0, state_final[j], false, &a.rec1[j]);
        make_rec(a, x1, state_out[j], true, &a.rec0[j]);
    } else {
        make_rec(a, x0, state_out[j], false, &a.rec0[j]);
        a.rec1[j].meta = 1u << 11;                                                  // skip
    }
}

__device__ __forceinline__ RecRegs load_rec(const ImgRec *p) {
    typedef const __attribute__((address_space(4))) u32x8 *cptr8;
    typedef const __attribute__((address_space(4))) u32x4 *cptr4;
    return RecRegs{*(cptr8)(uintptr_t)p, *((cptr4)(uintptr_t)p + 2)};
}

// ---- general renderer ---------------------------------------------------------------------------
// Any size; four range tests per pixel, one dword (4 pixels) per lane per store.  Pixels outside
// the polygon's bounding circle (about 3/4 of an 84x84 image at R = 20) are written as zeros
// without evaluating the map.
__device__ __forceinline__ void render(const ImageArgs &a, const RecRegs &r, uint8_t *lds_tpl,
                                       uint8_t *__restrict__ out, int lane) {
    const int a0 = (int)r.lo[0], a1 = (int)r.lo[1], a2 = (int)r.lo[2], a3 = (int)r.lo[3], a4 = (int)r.lo[4],
              a5 = (int)r.lo[5];
    const int cx = (int)(r.lo[6] & 0xFFFFu), cy = (int)(r.lo[6] >> 16), R = (int)(r.lo[7] & 0x3FFu);
    const float fcx = __uint_as_float(r.hi[0]), fcy = __uint_as_float(r.hi[1]);
    const int tsz = a.tpl * a.tpl;
    const uint8_t *gt = a.tpl_data + (size_t)(r.lo[7] >> 12) * (size_t)tsz;
    // template -> this wave's LDS slice (wave-local: LDS ops of one wave complete in order)
    for (int k = lane * 4; k < tsz; k += 64 * 4) {
        uint32_t w = 0;
#pragma unroll
        for (int b = 0; b < 4; b++) if (k + b < tsz) w |= (uint32_t)gt[k + b] << (8 * b);
        *(uint32_t *)(lds_tpl + k) = w;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    const float rad = (float)R + 3.0f, rad2 = rad * rad;
    const int half = a.tpl / 2, ox = cx - half, oy = cy - half;
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

// M = K * N images (time-major like the step outputs): rec[j] -> img[j].  Launched on rec0 for the
// observations and, if the caller wants them, on rec1 for the terminal observations of steps that
// ended in a reset (other records there say "skip").
__global__ __launch_bounds__(kBlock) void k_image_obs(ImageArgs a, long M, const ImgRec *__restrict__ rec,
                                                      uint8_t *__restrict__ img) {
    extern __shared__ __align__(16) uint8_t lds_all[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int slice = (a.tpl * a.tpl + 15) & ~15;
    const long j = __builtin_amdgcn_readfirstlane((int)((long)blockIdx.x * (kBlock / 64) + wave));
    if (j >= M) return;
    const RecRegs r = load_rec(rec + j);
    if (r.lo[7] & (1u << 11)) return;
    render(a, r, lds_all + wave * slice, img + (size_t)j * ((size_t)a.W * a.H), lane);
}

// ---- fast renderer (conditions checked on the host, mdpp_capi.hip: img_fast_ok) ----------------
// Measured on the general renderer (profiles/archive/r01_ablation_image_kernel.txt): the per-pixel map and
// the four range tests dominate, and stores that leave holes in a cache line are several times
// slower than full-line stores.  So:
//  * the four waves of a workgroup share 64 LDS rows of 256 B; wave w owns byte columns
//    [64 w, 64 w + 64) of every row and holds its image's template there inside a zero border of
//    kImgPad pixels.  The LDS address of source pixel (xs, ys) is then ONE v_perm_b32 of the two
//    16.16 accumulators (byte 2 of each = the integer part) and no range test is needed: a dword
//    whose 4 pixels can touch the polygon's bounding circle ("near", the criterion of the general
//    renderer) maps, through the isometry, to within R + 8 of the template centre, i.e. inside
//    the border; the polygon never leaves the image (host-checked), so Pillow's "source outside
//    the image -> 0" rule can only hit zero template pixels;
//  * only the near dwords of the bounding box of that circle are evaluated; they go to a
//    wave-private LDS copy of the image columns the box spans (zero elsewhere);
//  * the image is then written front to back with 16-byte stores, 1 KiB contiguous per wave
//    instruction: from the LDS columns where the box is, zeros elsewhere;
//  * waves are persistent: each walks images j, j + (waves in the grid), ... and has the next
//    image's record (scalar loads) and template (4 dwordx4 per lane) in flight while it evaluates
//    the current one; the stores of an image drain while the next one is evaluated.
#ifndef MDPP_IMG_ST_AUX
#define MDPP_IMG_ST_AUX 0
#endif
struct TplRegs { u32x4 v[4]; };          // a padded template (<= 64 rows x 64 B), 4 chunks per lane
// LDS accesses by INTEGER address (round 4): indexing the dynamic LDS array costs a v_add_u32 of its (link-time) base per access --
// four per evaluated dword here; an address-space-3 pointer made from an integer is used as it is
typedef __attribute__((address_space(3))) const uint8_t *lds_u8p;
typedef __attribute__((address_space(3))) uint32_t *lds_u32p;
#ifndef MDPP_IMG_LEAN_LOOP
#define MDPP_IMG_LEAN_LOOP 1       // the trimmed evaluation loop (integer LDS addresses, integer near test in half pixels, byte-address columns)
#endif

__device__ __forceinline__ TplRegs load_tpl(const ImageArgs &a, uint32_t tix, int lane) {
    const u32x4 *gt = (const u32x4 *)(a.tplp_data + (size_t)tix * ((size_t)a.tplp * 64));
    const int nchunk = a.tplp * 4;
    TplRegs r;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int c = lane + 64 * j;
        r.v[j] = gt[c < nchunk ? c : 0];
    }
    return r;
}

// The lane's eight entries of the near table for an image (record r): order by the walk direction (render_fast_eval), phase by
// the centre row rounded to whole pixels.
__device__ __forceinline__ u32x4 load_near(const ImageArgs &a, const RecRegs &r, int lane) {
    if (!a.near_tab) return u32x4{0u, 0u, 0u, 0u};
    const int a0 = (int)r.lo[0], a1 = (int)r.lo[1];
    const uint32_t order = (a0 < 0 ? -a0 : a0) >= (a1 < 0 ? -a1 : a1) ? 0u : 1u;
    int cyi = (int)rintf(__uint_as_float(r.hi[1]));
    asm volatile("v_readfirstlane_b32 %0, %1" : "=s"(cyi) : "v"(cyi));
    return ((const u32x4 *)a.near_tab)[((order * 4u + ((uint32_t)cyi & 3u)) * 64u) + (uint32_t)lane];
}

__device__ __forceinline__ void stage_tpl(const ImageArgs &a, const TplRegs &tp, uint8_t *lds, int wave, int lane) {
#ifndef MDPP_IMG_ABL_NOTPL
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int c = lane + 64 * j;
        if (c < a.tplp * 4) *(u32x4 *)(lds + (c >> 2) * 256 + wave * 64 + (c & 3) * 16) = tp.v[j];
    }
#endif
}

// Evaluation half: the template of this image is already in the wave's LDS columns (stage_tpl);
// leaves the image columns of the bounding box in lds_col and returns their chunk range.
struct ColRange { int C0, C1; };
template <int COLB = 64>
__device__ __forceinline__ ColRange render_fast_eval(const ImageArgs &a, const RecRegs &r, const uint8_t *lds,
                                                     uint32_t *lds_col, int wave, int lane, const u32x4 near) {
    const int a0 = (int)r.lo[0], a1 = (int)r.lo[1], a2 = (int)r.lo[2], a3 = (int)r.lo[3], a4 = (int)r.lo[4],
              a5 = (int)r.lo[5];
    const int cx = (int)(r.lo[6] & 0xFFFFu), cy = (int)(r.lo[6] >> 16), R = (int)(r.lo[7] & 0x3FFu);
    const float fcx = __uint_as_float(r.hi[0]), fcy = __uint_as_float(r.hi[1]);
    const int X0 = (int)(r.hi[2] & 0xFFFFu), X1 = (int)(r.hi[2] >> 16);
    const int Q0 = (int)(r.hi[3] & 0xFFFFu), Q1 = (int)(r.hi[3] >> 16);
    const float rr = (float)R + 4.5f, rr2 = rr * rr;          // near radius of the general renderer
    (void)rr2;
    // ... in integers (MDPP_IMG_LEAN_LOOP): doubled coordinates against the centre rounded to half pixels (off by <= 0.36 px), radius
    // R + 5 -- a superset of the float test's dwords inside the same bounding box, and every pixel of a dword it lets through
    // still lies within R + 5.36 + 1.5 of the centre: inside the zero border (R + 8) after the map's rounding
    int ncx2, ncy2;                              // (scalar registers; through asm: the compiler folds a readfirstlane of a uniform value away
    {                                            //  and keeps the value in a vector register, which costs the loop two instructions per test)
        const int vx = -(int)rintf(2.0f * fcx), vy = 3 - (int)rintf(2.0f * fcy);
        asm volatile("v_readfirstlane_b32 %0, %1" : "=s"(ncx2) : "v"(vx));
        asm volatile("v_readfirstlane_b32 %0, %1" : "=s"(ncy2) : "v"(vy));
    }
    const int rr2i = (2 * R + 10) * (2 * R + 10);
    const int HQ = a.H >> 2;
    const int bw = X1 - X0, bhq = Q1 - Q0;
    // LDS columns hold image dwords [B0, B1), 16-byte chunks [C0, C1)
    const int C0 = (X0 * HQ) >> 2, C1 = bw > 0 ? (X1 * HQ + 3) >> 2 : C0, B0 = C0 << 2;
    for (int c = lane; c < C1 - C0; c += 64) *(u32x4 *)(lds_col + 4 * c) = u32x4{0u, 0u, 0u, 0u};
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    if (bw > 0 && bhq > 0) {
        // accumulators carry the template-local source coordinates in 16.16:
        // bx >> 16 = xs - (cx - half_p) + 64 wave, by >> 16 = ys - (cy - half_p)
        const int half_p = a.tplp >> 1;
        const int A2 = a2 - ((cx - half_p) << 16) + ((wave * COLB) << 16);
        int A5 = a5 - ((cy - half_p) << 16);
        // (integer LDS addresses: the template rows start at LDS offset lbase -- 0 for this kernel, which has no static LDS; a
        //  multiple of 256 folds into the row accumulator)
        const uint32_t lbase = (uint32_t)(uintptr_t)(lds_u8p)lds;
        if (MDPP_IMG_LEAN_LOOP) {
            if (__builtin_expect((lbase & 255u) != 0u || lbase > 0xC000u, 0)) __builtin_trap();
            A5 += (int)(lbase >> 8) << 16;
        }
        const uint32_t cbase = (uint32_t)(uintptr_t)(lds_u32p)lds_col - 4u * (uint32_t)B0, HQ4 = 4u * (uint32_t)HQ;
        if (MDPP_IMG_LEAN_LOOP && a.near_tab) {
            // Table-driven enumeration (round 4): the lane's eight (dx, dq) entries, relative to the centre rounded to whole
            // pixels; an entry outside the record's box (or the padding entry) fails two unsigned compares.  No idle lanes
            // inside a bounding box, no near test, no walk: 8 rounds of 64 dwords for R = 20 where the box took 10.
            int cxi = (int)rintf(fcx), cyi = (int)rintf(fcy);
            asm volatile("v_readfirstlane_b32 %0, %1" : "=s"(cxi) : "v"(cxi));
            asm volatile("v_readfirstlane_b32 %0, %1" : "=s"(cyi) : "v"(cyi));
            const int cX = cxi - X0, cQ = (cyi >> 2) - Q0;
            // x = xr + X0, y = yr + 4 Q0: the box origin folded into the (scalar) constants
            const int A2t = A2 + a0 * X0 + a1 * 4 * Q0, A5t = A5 + a3 * X0 + a4 * 4 * Q0;
            const uint32_t cb2 = cbase + (uint32_t)X0 * HQ4 + 4u * (uint32_t)Q0;
#pragma unroll
            for (int it = 0; it < 8; it++) {
                const uint32_t ew = (it >> 1) == 0 ? near.x : (it >> 1) == 1 ? near.y : (it >> 1) == 2 ? near.z : near.w;
                const int xr = __builtin_amdgcn_sbfe((int)ew, (it & 1) * 16, 8) + cX;
                const int qr = __builtin_amdgcn_sbfe((int)ew, (it & 1) * 16 + 8, 8) + cQ;
                if ((uint32_t)xr < (uint32_t)bw && (uint32_t)qr < (uint32_t)bhq) {
                    const int yr = qr << 2;
                    const int bx = A2t + __mul24(a0, xr) + __mul24(a1, yr);
                    const int by = A5t + __mul24(a3, xr) + __mul24(a4, yr);
                    uint32_t px[4];
#pragma unroll
                    for (int b = 0; b < 4; b++) {
                        const uint32_t addr = __builtin_amdgcn_perm((uint32_t)(by + b * a4), (uint32_t)(bx + b * a1), 0x0c0c0602u);
                        px[b] = *(lds_u8p)(uintptr_t)addr;
                    }
                    const uint32_t word = (px[0] | (px[1] << 8)) | ((px[2] | (px[3] << 8)) << 16);
                    *(lds_u32p)(uintptr_t)(__umul24((uint32_t)xr, HQ4) + (uint32_t)yr + cb2) = word;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            return ColRange{C0, C1};
        }
        const int nb = bw * bhq;
        // Which way the 64 lanes of an iteration walk the box (round 4).  A template row is 256 B = all 64 LDS banks once, so
        // source pixels in ONE COLUMN of the template sit in one bank: at rotations near 0 / 180 degrees lanes that walk DOWN
        // an image column read down a template column (13-way bank conflicts on every ds_read_u8: 16.3 us per step of 8 192
        // images with angles {0, 180} against 11.3 with {90, 270}); near 90 / 270 the same happens to lanes that walk ALONG an
        // image row.  So the walk follows the map: x fastest where |d xs / d x| >= |d xs / d y|, y fastest otherwise -- the
        // lanes of an iteration then read along template rows either way (wave-uniform choice, no per-pixel cost).
#ifndef MDPP_IMG_WALK
#define MDPP_IMG_WALK 1
#endif
        const bool xfast = MDPP_IMG_WALK && (a0 < 0 ? -a0 : a0) >= (a1 < 0 ? -a1 : a1);
        const int span = xfast ? bw : bhq;                          // lanes per line of the walk
        const uint32_t inv = 65536u / (uint32_t)span + 1u;         // k / span == (k * inv) >> 16 here (k < 2^16 / span)
        const int kq = (int)(((uint32_t)lane * inv) >> 16), kr = lane - kq * span;
        int x = X0 + (xfast ? kr : kq), y = 4 * (Q0 + (xfast ? kq : kr));        // y = first pixel row of the dword
        const int d64 = 64 / span, r64 = 64 - d64 * span;
        const int yend = 4 * Q1, ywrap = 4 * bhq;
        for (int k = lane; k < nb; k += 64) {
#ifndef MDPP_IMG_ABL_ZERO
#if MDPP_IMG_LEAN_LOOP
            const int dx2 = 2 * x + ncx2, dy2 = 2 * y + ncy2;        // (v_lshl_add_u32 with a scalar)
            if (dx2 * dx2 + dy2 * dy2 <= rr2i) {          // (plain multiplies: __mul24 sign-extends its operands from 24 bits first, two shifts each)
                const int bx = A2 + __mul24(a0, x) + __mul24(a1, y);      // |a_i| <= 2^16, x, y < 2^23
                const int by = A5 + __mul24(a3, x) + __mul24(a4, y);
                uint32_t px[4];
#pragma unroll
                for (int b = 0; b < 4; b++) {
                    // byte 2 of each accumulator = its integer part (< 256): address = uy * 256 + ux
                    const uint32_t addr = __builtin_amdgcn_perm((uint32_t)(by + b * a4), (uint32_t)(bx + b * a1), 0x0c0c0602u);
                    px[b] = *(lds_u8p)(uintptr_t)addr;
                }
                const uint32_t word = (px[0] | (px[1] << 8)) | ((px[2] | (px[3] << 8)) << 16);
                // (y is a multiple of 4: the dword's byte address in the columns is x * 4 HQ + y)
                *(lds_u32p)(uintptr_t)(__umul24((uint32_t)x, HQ4) + (uint32_t)y + cbase) = word;
            }
#else
            const float ddx = (float)x - fcx, ddy = (float)y + 1.5f - fcy;
            if (ddx * ddx + ddy * ddy <= rr2) {
                const int bx = A2 + __mul24(a0, x) + __mul24(a1, y);      // |a_i| <= 2^16, x, y < 2^23
                const int by = A5 + __mul24(a3, x) + __mul24(a4, y);
                uint32_t word = 0;
#pragma unroll
                for (int b = 0; b < 4; b++) {
                    // byte 2 of each accumulator = its integer part (< 256): address = uy * 256 + ux
                    const uint32_t addr = __builtin_amdgcn_perm((uint32_t)(by + b * a4), (uint32_t)(bx + b * a1), 0x0c0c0602u);
                    word |= (uint32_t)lds[addr] << (8 * b);
                }
                lds_col[__mul24(x, HQ) + (y >> 2) - B0] = word;
            }
#endif
#endif
            if (xfast) {
                y += 4 * d64; x += r64;
                if (x >= X1) { x -= bw; y += 4; }
            } else {
                x += d64; y += 4 * r64;
                if (y >= yend) { y -= ywrap; x += 1; }
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    return ColRange{C0, C1};
}

// Store half: the image front to back, 16 B per lane per store.  NST > 0: exactly NST (=
// ceil(W H / 1024)) wave-stores, unrolled.
template <int NST>
__device__ __forceinline__ void render_fast_store(const ImageArgs &a, const ColRange cr, const uint32_t *lds_col,
                                                  uint8_t *__restrict__ out, int lane) {
    const auto r_out = __builtin_amdgcn_make_buffer_rsrc((void *)out, 0, a.W * a.H, 0x00020000);
    const int nchunk = (a.W * a.H) >> 4;
#if MDPP_IMG_LEAN_LOOP
    // (round 4) every lane reads unconditionally: chunks outside the box come from the wave's ZERO CHUNK (the last 16 bytes of its
    // column buffer, cleared once per kernel) -- one v_cndmask_b32 on the address instead of an exec-mask branch around each read,
    // so the seven LDS reads of an image are in flight together instead of one wait per store
    typedef __attribute__((address_space(3))) const u32x4 *lds_cu128p;
    const uint32_t cb = (uint32_t)(uintptr_t)(lds_u32p)const_cast<uint32_t *>(lds_col), zaddr = cb + 4u * (uint32_t)(a.coldw - 4);
    const uint32_t nin = (uint32_t)(cr.C1 - cr.C0);
    auto put = [&](int c) {
        const uint32_t rel = (uint32_t)(c - cr.C0);
        const u32x4 v = *(lds_cu128p)(uintptr_t)(rel < nin ? cb + 16u * rel : zaddr);
#ifdef MDPP_IMG_ABL_NOSTORE
        if (v.x == 0x12345678u)
#endif
        __builtin_amdgcn_raw_buffer_store_b128(v, r_out, c * 16, 0, MDPP_IMG_ST_AUX);   // beyond the descriptor: dropped
    };
#else
    auto put = [&](int c) {
        u32x4 v = u32x4{0u, 0u, 0u, 0u};
        if ((uint32_t)(c - cr.C0) < (uint32_t)(cr.C1 - cr.C0)) v = *(const u32x4 *)(lds_col + 4 * (c - cr.C0));
#ifdef MDPP_IMG_ABL_NOSTORE
        if (v.x == 0x12345678u)
#endif
        __builtin_amdgcn_raw_buffer_store_b128(v, r_out, c * 16, 0, MDPP_IMG_ST_AUX);   // beyond the descriptor: dropped
    };
#endif
    if (NST > 0) {
#pragma unroll
        for (int u = 0; u < NST; u++) put(lane + 64 * u);
    } else {
        for (int c = lane; c < nchunk; c += 64) put(c);
    }
}

// Per image: [next record: s_load] [next template: 4 dwordx4 per lane in flight] evaluate ->
// stage the next template (its loads have had the whole evaluation to land) -> store.  The only
// vmcnt wait of an iteration sits after the evaluation, so the 16-byte stores of one image drain
// under the evaluation of the next.
template <int NST>
__global__ __launch_bounds__(kBlock) void k_image_obs_fast(ImageArgs a, long M, const ImgRec *__restrict__ rec,
                                                           uint8_t *__restrict__ img, uint32_t *__restrict__ ctr) {
    // LDS, sized by the launch (round 3): tplp template rows of 256 B, then coldw dwords of image columns per wave -- for
    // 84 x 84 images at R = 20 that is 14.25 + 4 x 4.4 KiB = 31.75 KiB, FIVE workgroups per CU where the fixed 16 + 24 KiB
    // of rounds 1-2 allowed four: more waves whose evaluation and store phases interleave
    extern __shared__ __align__(16) uint8_t lds[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    uint32_t *const lds_col = (uint32_t *)(lds + (size_t)a.tplp * 256) + (size_t)wave * a.coldw;
    if (MDPP_IMG_LEAN_LOOP && lane < 4) lds_col[a.coldw - 4 + lane] = 0u;      // the wave's zero chunk (render_fast_store)
    // Work distribution (round 3): a wave CLAIMS its next image from a counter instead of walking j, j + (waves in the grid),
    // ...: the next batch's state / draw / record kernels run beside this kernel on a few CUs, images differ in cost, and with
    // a static split the slowest wave sets the kernel's duration (a 32-step render: 427 us alone, 465-475 us beside them).
    // cfg4 on one box: static 7 770 us per launch; claims of 1 / 2 / 4 / 8 images 6 960 / 7 380 / 7 620 / 7 720 (consecutive
    // images in one wave cost more than the coarser balance saves); 8 / 16 / 32 / 64 counters 7 110 / 6 930 / 6 910 / 6 950.
    // Round 5: ONE picture per wave, then the workgroup ENDS (MDPP_IMG_DYNAMIC = 0 with a grid of one wave per picture): the
    // hardware's dispatcher hands out the pictures in address order, so the chip's write front is the dispatch order -- the
    // form in which plain fills run 20 % faster than on a persistent grid (mdpp_probe_hbm, tools/bench_store.hip).  It gives
    // up the next picture's prefetch under the current one's evaluation (20 waves per CU hide the two round trips instead) and
    // needs no claim counters: cfg4 6 530-6 550 -> 6 300 us per launch on one lease (0.568 -> 0.589); two / four pictures per
    // wave 6 460 / 6 640; non-temporal stores 6 384.  The claiming loop stays below for MDPP_IMG_DYNAMIC = 1.
#ifndef MDPP_IMG_DYNAMIC
#define MDPP_IMG_DYNAMIC 0
#endif
#ifndef MDPP_IMG_CLAIM
#define MDPP_IMG_CLAIM 1
#endif
    constexpr long kImgClaim = MDPP_IMG_CLAIM;
    const size_t isz = (size_t)a.W * a.H;
#if MDPP_IMG_DYNAMIC
    // One counter per group of waves (wave id mod kImgCtrs: its waves sit on CUs all over the chip), each over its own
    // contiguous slice of the images: a single counter saturates -- 65 536 same-address atomics per batch took as long as the
    // rendering itself.  The claim's result stays in a vector register until it is needed, a whole chunk later.
    const long wid = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * (kBlock / 64) + wave));
    const long grp = wid % kImgCtrs;
    const long per = (M + kImgCtrs - 1) / kImgCtrs;
    const long g0 = grp * per;
    M = g0 + per < M ? g0 + per : M;             // (from here on: the end of this group's slice)
    uint32_t *const gctr = ctr + grp * 32;
    uint32_t pend = 0;
    auto claim_issue = [&]() __attribute__((always_inline)) { if (lane == 0) pend = atomicAdd(gctr, (uint32_t)kImgClaim); };
    auto claim_take = [&]() __attribute__((always_inline)) -> long {
        return g0 + (long)(uint32_t)__builtin_amdgcn_readfirstlane((int)pend);
    };
    claim_issue();
    long base = claim_take();
    if (base >= M) return;
    claim_issue();                               // (one chunk ahead)
    long j = base;
    RecRegs cur = load_rec(rec + j);
    u32x4 near_cur = load_near(a, cur, lane);
    stage_tpl(a, load_tpl(a, cur.lo[7] >> 12, lane), lds, wave, lane);
    for (;;) {
        const bool last_of_chunk = j + 1 >= base + kImgClaim || j + 1 >= M;
        long jn = j + 1;
        if (last_of_chunk) {                     // (wave-uniform)
            jn = claim_take();
            base = jn;
            if (jn < M) claim_issue();
        }
        const bool more = jn < M;
        const bool skip = cur.lo[7] & (1u << 11);
        const RecRegs nxt = load_rec(rec + (more ? jn : j));
        const TplRegs tp = load_tpl(a, nxt.lo[7] >> 12, lane);
        const u32x4 near_nxt = load_near(a, nxt, lane);
        ColRange cr{0, 0};
        if (!skip) cr = render_fast_eval(a, cur, lds, lds_col, wave, lane, near_cur);
        stage_tpl(a, tp, lds, wave, lane);
        if (!skip) render_fast_store<NST>(a, cr, lds_col, img + (size_t)j * isz, lane);
        if (!more) break;
        j = jn; cur = nxt; near_cur = near_nxt;
    }
#else
    const int nw = (int)gridDim.x * (kBlock / 64);
    const uint32_t bx = img_xcd_block();
    long j = __builtin_amdgcn_readfirstlane((int)(bx * (kBlock / 64) + wave));
    if (j >= M) return;
    RecRegs cur = load_rec(rec + j);
    u32x4 near_cur = load_near(a, cur, lane);
    stage_tpl(a, load_tpl(a, cur.lo[7] >> 12, lane), lds, wave, lane);
    // (one-shot grid: the loop body runs once.  Its loads "for the next picture" then re-read this picture's record and template
    //  -- and a straight-line form without them measured 6 148 us per launch against this one's 5 946, same lease: kept)
    for (;;) {
        const long jn = j + nw;
        const bool more = jn < M;
        const bool skip = cur.lo[7] & (1u << 11);
        const RecRegs nxt = load_rec(rec + (more ? jn : j));
        const TplRegs tp = load_tpl(a, nxt.lo[7] >> 12, lane);
        const u32x4 near_nxt = load_near(a, nxt, lane);
        ColRange cr{0, 0};
        if (!skip) cr = render_fast_eval(a, cur, lds, lds_col, wave, lane, near_cur);
        stage_tpl(a, tp, lds, wave, lane);
        if (!skip) render_fast_store<NST>(a, cr, lds_col, img + (size_t)j * isz, lane);
        if (!more) break;
        j = jn; cur = nxt; near_cur = near_nxt;
    }
#endif
}

// ---- wide templates (round 5) --------------------------------------------------------------------
// The scale transform draws a radius per picture; the reference's own sweeps use image_scale_range = (0.5, 2) on 100 x 100
// pictures: radii 9 ... 41, templates 85 wide, 101 inside the zero border -- past the 64-byte LDS columns of k_image_obs_fast, so
// those handles ran the general renderer (four range tests per pixel, dword stores).  Here TWO waves per workgroup own 128 B
// of every 256-byte LDS row each: the address of a source pixel is still one v_perm_b32 of the two accumulators (column <
// 256).  One picture per wave, one shot.  Differences from k_image_obs_fast, all to keep LDS per wave at the template alone
// (12.6 KiB at tplp = 101; a column buffer for R = 40 would add 9.3 and halve the waves per CU):
//  * only the template rows the map can reach (within R + 9 of the centre) are staged, straight from memory;
//  * no image columns in LDS: a lane owns the 16-byte chunks lane, lane + 64, ... of the picture, evaluates the near dwords
//    among its chunk's four (the same integer near test as render_fast_eval's box walk) and stores the chunk from registers
//    -- the same front-to-back 1 KiB-per-instruction store pattern; chunks far from the polygon cost one test.
constexpr int kWideTplRegs = 16;          // 16-byte template chunks per lane: up to 128 rows of 128 B
struct WideTpl { u32x4 v[kWideTplRegs]; };
// The template rows a picture's map can reach, global memory -> registers (loads in flight), and registers -> the wave's LDS columns.
__device__ __forceinline__ void wide_tpl_load(const ImageArgs &a, const RecRegs &r, int lane, WideTpl &t) {
    const int R = (int)(r.lo[7] & 0x3FFu), half_p = a.tplp >> 1;
    const u32x4 *gt = (const u32x4 *)(a.tplp_data + (size_t)(r.lo[7] >> 12) * ((size_t)a.tplp * 128));
    const int r0 = max(0, half_p - R - 9), r1 = min(a.tplp, half_p + R + 10);
#pragma unroll
    for (int q = 0; q < kWideTplRegs; q++) {
        const int c = r0 * 8 + lane + 64 * q;
        t.v[q] = gt[c < r1 * 8 ? c : r0 * 8];
    }
}
__device__ __forceinline__ void wide_tpl_stage(const ImageArgs &a, const RecRegs &r, uint8_t *lds, int wave, int lane, const WideTpl &t) {
    const int R = (int)(r.lo[7] & 0x3FFu), half_p = a.tplp >> 1;
    const int r0 = max(0, half_p - R - 9), r1 = min(a.tplp, half_p + R + 10);
#pragma unroll
    for (int q = 0; q < kWideTplRegs; q++) {
        const int c = r0 * 8 + lane + 64 * q;
        if (c < r1 * 8) *(u32x4 *)(lds + (c >> 3) * 256 + wave * 128 + (c & 7) * 16) = t.v[q];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
}
template <bool STAGE = true>
__device__ __forceinline__ void render_wide(const ImageArgs &a, const RecRegs &r, uint8_t *lds, int wave, int lane,
                                            uint8_t *__restrict__ out) {
    const int a0 = (int)r.lo[0], a1 = (int)r.lo[1], a2 = (int)r.lo[2], a3 = (int)r.lo[3], a4 = (int)r.lo[4],
              a5 = (int)r.lo[5];
    const int cx = (int)(r.lo[6] & 0xFFFFu), cy = (int)(r.lo[6] >> 16), R = (int)(r.lo[7] & 0x3FFu);
    const float fcx = __uint_as_float(r.hi[0]), fcy = __uint_as_float(r.hi[1]);
    const int half_p = a.tplp >> 1;
    if constexpr (STAGE) {   // template rows [half_p - R - 9, half_p + R + 9]: everything a near dword can map to (R + 8 around the centre)
        const u32x4 *gt = (const u32x4 *)(a.tplp_data + (size_t)(r.lo[7] >> 12) * ((size_t)a.tplp * 128));
        const int r0 = max(0, half_p - R - 9), r1 = min(a.tplp, half_p + R + 10);
        for (int c = r0 * 8 + lane; c < r1 * 8; c += 64) *(u32x4 *)(lds + (c >> 3) * 256 + wave * 128 + (c & 7) * 16) = gt[c];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    int ncx2, ncy2;                              // (scalar registers, see render_fast_eval)
    {
        const int vx = -(int)rintf(2.0f * fcx), vy = 3 - (int)rintf(2.0f * fcy);
        asm volatile("v_readfirstlane_b32 %0, %1" : "=s"(ncx2) : "v"(vx));
        asm volatile("v_readfirstlane_b32 %0, %1" : "=s"(ncy2) : "v"(vy));
    }
    const int rr2i = (2 * R + 10) * (2 * R + 10);             // a dword is near: centre within R + 5 (doubled coordinates)
    const int rc2i = (2 * R + 28) * (2 * R + 28);             // a chunk (16 rows of one column) can hold a near dword: within R + 14
    const uint32_t lbase = (uint32_t)(uintptr_t)(lds_u8p)lds;
    if (__builtin_expect((lbase & 255u) != 0u || lbase > 0x8000u, 0)) __builtin_trap();
    const int A2 = a2 - ((cx - half_p) << 16) + ((wave * 128) << 16);
    const int A5 = a5 - ((cy - half_p) << 16) + ((int)(lbase >> 8) << 16);
    const int HQ = a.H >> 2;
    const uint32_t inv = (1u << 24) / (uint32_t)HQ + 1u;      // q / HQ == (q * inv) >> 24 for q HQ < 2^24 (host-checked)
    const int nchunk = (a.W * a.H) >> 4;
    const auto r_out = __builtin_amdgcn_make_buffer_rsrc((void *)out, 0, a.W * a.H, 0x00020000);
    for (int c0 = 0; c0 < nchunk; c0 += 64) {
        const int c = c0 + lane;
        int x = (int)(((uint32_t)(4 * c) * inv) >> 24), yq = 4 * c - x * HQ;
        u32x4 v = u32x4{0u, 0u, 0u, 0u};
        // the chunk's run of 16 rows: centre at y + 7.5 in column x; a run that wraps into the next column is taken as near
        const int dxc = 2 * x + ncx2, dyc = 8 * yq + 12 + ncy2;
        const bool maybe = yq + 4 > HQ || dxc * dxc + dyc * dyc <= rc2i;
        if (maybe) {
#pragma unroll
            for (int d = 0; d < 4; d++) {
                const int y = 4 * yq;
                const int dx2 = 2 * x + ncx2, dy2 = 2 * y + ncy2;
                if (dx2 * dx2 + dy2 * dy2 <= rr2i) {
                    const int bx = A2 + __mul24(a0, x) + __mul24(a1, y);      // |a_i| <= 2^16, x, y < 2^23
                    const int by = A5 + __mul24(a3, x) + __mul24(a4, y);
                    uint32_t px[4];
#pragma unroll
                    for (int b = 0; b < 4; b++) {
                        // byte 2 of each accumulator = its integer part (< 256): address = uy * 256 + ux
                        const uint32_t addr = __builtin_amdgcn_perm((uint32_t)(by + b * a4), (uint32_t)(bx + b * a1), 0x0c0c0602u);
                        px[b] = *(lds_u8p)(uintptr_t)addr;
                    }
                    v[d] = (px[0] | (px[1] << 8)) | ((px[2] | (px[3] << 8)) << 16);
                }
                yq += 1;
                if (yq == HQ) { yq = 0; x += 1; }
            }
        }
        __builtin_amdgcn_raw_buffer_store_b128(v, r_out, c * 16, 0, 0);          // beyond the descriptor: dropped
    }
}

constexpr int kWideBlock = 128;
__global__ __launch_bounds__(kWideBlock) void k_image_obs_wide(ImageArgs a, long M, const ImgRec *__restrict__ rec,
                                                               uint8_t *__restrict__ img) {
    extern __shared__ __align__(16) uint8_t lds[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
#ifndef MDPP_IMG_XCD_WIDE
#define MDPP_IMG_XCD_WIDE MDPP_IMG_XCD
#endif
    // MDPP_IMG_WIDE_PAIR (round 6): a wave renders TWO consecutive pictures.  A picture starts with two dependent round trips -- its
    // record, then the template rows the record names -- which three waves per SIMD (LDS: 12.6 KiB of template per wave) do not hide;
    // both records are loaded up front, and the second picture's template rows are in flight (in registers) while the first
    // picture is evaluated and stored: img100_all 1 835 -> 1 567-1 590 us per 64-step launch (0.358 -> 0.415-0.419 of HBM; a rolling
    // form with three / four / eight pictures per wave: 0.399 / 0.400-0.408 / 0.390 -- tools/ablate.py p0 / p1 / p2 / p4, one lease).
#ifndef MDPP_IMG_WIDE_PAIR
#define MDPP_IMG_WIDE_PAIR 1
#endif
    const long wid = __builtin_amdgcn_readfirstlane((int)((MDPP_IMG_XCD_WIDE ? img_xcd_block() : blockIdx.x) * (kWideBlock / 64) + wave));
    const size_t isz = (size_t)a.W * a.H;
#if MDPP_IMG_WIDE_PAIR == 1
    const long j0 = 2 * wid, j1 = j0 + 1;
    if (j0 >= M) return;
    const bool has1 = j1 < M;
    const RecRegs r0 = load_rec(rec + j0), r1 = load_rec(rec + (has1 ? j1 : j0));
    const bool go0 = !(r0.lo[7] & (1u << 11)), go1 = has1 && !(r1.lo[7] & (1u << 11));
    WideTpl t;
    if (go0) { wide_tpl_load(a, r0, lane, t); wide_tpl_stage(a, r0, lds, wave, lane, t); }
    if (go1) wide_tpl_load(a, r1, lane, t);                     // (in flight under the first picture)
    if (go0) render_wide<false>(a, r0, lds, wave, lane, img + (size_t)j0 * isz);
    if (go1) {
        wide_tpl_stage(a, r1, lds, wave, lane, t);
        render_wide<false>(a, r1, lds, wave, lane, img + (size_t)j1 * isz);
    }
#else
    const long j = wid;
    if (j >= M) return;
    const RecRegs r = load_rec(rec + j);
    if (r.lo[7] & (1u << 11)) return;
    render_wide(a, r, lds, wave, lane, img + (size_t)j * isz);
#endif
}

// The launch arguments every image kernel of a batch of K steps shares (buf: the scratch set of a pipelined rollout).
static ImageArgs image_args(mdpp_env *h, int K, bool is_reset, int buf) {
    const mdpp_config &c = h->cfg;
    ImageArgs a;
    a.N = c.num_envs; a.W = c.img_w; a.H = c.img_h;
    a.S = (c.irrelevant && c.S_irr > c.S) ? c.S_irr : c.S;
    a.SUB = c.irrelevant ? 2 : 1;
    a.has_scale = c.img_has_scale; a.has_shift = c.img_has_shift; a.has_rotate = c.img_has_rotate;
    a.has_flip = c.img_has_flip; a.sh_quant = c.img_sh_quant > 0 ? c.img_sh_quant : 1;
    a.ro_quant = c.img_ro_quant > 0 ? c.img_ro_quant : 1;
    a.r0 = c.img_r0; a.r_min = c.img_r_min; a.r_max = c.img_r_max; a.tpl = c.img_tpl_size;
    a.n_radii = h->img_n_radii; a.n_cls_x = h->img_n_cls_x; a.n_cls_y = h->img_n_cls_y;
    a.autoreset = c.autoreset == MDPP_AUTORESET_SAME_STEP;   // (next-step: a reset call draws one observation like any step)
    a.log_min_r = c.img_log_min_r; a.log_max_r = c.img_log_max_r;
    a.tpl_data = (const uint8_t *)h->d_img_tpl; a.cls_x = (const int16_t *)h->d_img_clsx;
    a.cls_y = (const int16_t *)h->d_img_clsy; a.rot = (const int32_t *)h->d_img_rot;
    a.rng_s = (ulonglong2 *)h->d_rng_s[MDPP_STREAM_IMAGE];
    a.rng_inc = (ulonglong2 *)h->d_rng_inc[MDPP_STREAM_IMAGE];
    a.rng_half = (uint2 *)h->d_rng_half;
    a.tplp_data = (const uint8_t *)h->d_img_tplp; a.tplp = c.img_tpl_size + 2 * kImgPad;
    a.colb = h->img_colb;
    a.near_tab = ((h->opts & MDPP_OPT_NO_IMG_NEARTAB) || a.colb != 64) ? nullptr : (const uint32_t *)h->d_img_near;
    a.philox = c.rng_mode == MDPP_RNG_PHILOX; a.philox_seed = c.philox_seed; a.env_id_offset = c.env_id_offset;
    // (the state kernel / reset kernel of this batch ran just before and has advanced the handle's counters)
    a.is_reset = is_reset;
    a.ptick = a.is_reset ? h->reset_tick - 1 : h->tick - (uint64_t)K;
    a.dtick = (h->graph_capture && !a.is_reset) ? (const uint64_t *)h->d_tick_off : nullptr;   // (launches being captured into a HIP graph)
    a.rec0 = (ImgRec *)h->d_img_rec + (size_t)buf * 2 * h->img_chunk * c.num_envs * a.SUB;
    a.rec1 = a.rec0 + (size_t)h->img_chunk * c.num_envs * a.SUB;
    a.work_ctr = (uint32_t *)h->d_img_ctr + (size_t)buf * 2 * kImgCtrs * 32;
    a.coldw = 0;
    return a;
}

// LDS of a fast-renderer workgroup: the template rows + four column buffers wide enough for the widest box (rec_words: at most
// 2 R + 13 columns) plus chunk-alignment slack
static size_t image_fast_lds(const mdpp_env *h, ImageArgs &a) {
    const int span_dw = (2 * h->cfg.img_r_max + 13) * (h->cfg.img_h / 4) + 8;
    a.coldw = ((span_dw + 3) & ~3) + 4;                  // (+ the zero chunk of render_fast_store)
    if (a.coldw > kImgColDw) a.coldw = kImgColDw;
    if (a.colb != 64) return (size_t)a.tplp * 256;      // k_image_obs_wide: the template rows alone
    return (size_t)a.tplp * 256 + (size_t)(kBlock / 64) * a.coldw * 4;
}

// ---- one step of every env: draw + record + render in ONE kernel (round 5) ----------------------
// mdpp_step() on an image handle was four launches -- the state kernel, k_image_draw<REC> (one lane per env), the renderer
// over rec0 and again over rec1 (where all but the reset envs' records say "skip") -- 40 us for 8 192 envs of BASELINE cfg4,
// whose 57.8 MB of pictures take 11 us to store.  Here ONE WAVE per env does the serial part itself, every lane the same
// arithmetic on the same values (the generator's state through wave-uniform loads; 64 lanes of one wave cost what one lane
// does): the transform draws in the reference's order -- the step's observation, then reset()'s where the step ended the
// episode --, the record (in registers, moved to scalar registers), and the picture(s) with the fast renderer's evaluation and
// store halves.  Lane 0 writes the generator back.  An irrelevant sub-space's second picture (same generator, drawn right after
// the first) is rendered by the same wave, after the first.  The general renderer's handles keep the four launches.
template <int NST, bool PHILOX, bool WIDE = false>
__global__ __launch_bounds__(kBlock) void k_image_step1(ImageArgs a, const int32_t *__restrict__ state_out,
                                                        const int32_t *__restrict__ state_final,
                                                        const uint8_t *__restrict__ term, const uint8_t *__restrict__ trunc,
                                                        uint8_t *__restrict__ img_out, uint8_t *__restrict__ img_final) {
    extern __shared__ __align__(16) uint8_t lds[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    uint32_t *const lds_col = (uint32_t *)(lds + (size_t)a.tplp * 256) + (size_t)wave * a.coldw;     // (WIDE: no image columns in LDS)
    if (!WIDE && MDPP_IMG_LEAN_LOOP && lane < 4) lds_col[a.coldw - 4 + lane] = 0u;      // the wave's zero chunk (render_fast_store)
#ifndef MDPP_IMG_XCD_STEP1
#define MDPP_IMG_XCD_STEP1 MDPP_IMG_XCD
#endif
    const long i = __builtin_amdgcn_readfirstlane((int)((MDPP_IMG_XCD_STEP1 ? img_xcd_block() : blockIdx.x) * ((WIDE ? kWideBlock : kBlock) / 64) + wave));
    if (i >= a.N) return;
    // Everything the wave needs from memory before it can draw, in ONE batch of scalar loads: the generator, the state, and the
    // step's two flag bytes as the dwords they sit in (there is no scalar byte load, and a vector load -- or a load behind a
    // branch -- would be waited for before the next one is issued: a round trip each at the head of every wave).
    typedef const __attribute__((address_space(4))) uint32_t *cptr32;
    typename std::conditional<PHILOX, Philox, Pcg64>::type g;
    Half32 h{0u, 0u};
    if constexpr (PHILOX) {
        g.init(a.philox_seed, (uint64_t)(a.env_id_offset + i), tick_now(a), (uint32_t)MDPP_STREAM_IMAGE);
    } else {
        g.load(a.rng_s, a.rng_inc, i);
        const uint2 hh = a.rng_half[i];
        h = Half32{hh.x, hh.y};
    }
    const int SUB = a.SUB;                           // pictures per env: 2 with an irrelevant sub-space (drawn and rendered one after the other)
    const int s_out0 = state_out[i * SUB], s_out1 = state_out[i * SUB + SUB - 1];
    const uintptr_t pt = (uintptr_t)term + (uintptr_t)i, pu = (uintptr_t)trunc + (uintptr_t)i;
    const uint32_t wt = *(cptr32)(pt & ~(uintptr_t)3), wu = *(cptr32)(pu & ~(uintptr_t)3);
    const uint32_t flags = ((wt >> (8 * (int)(pt & 3))) | (wu >> (8 * (int)(pu & 3)))) & 0xFFu;
    const bool two = (a.autoreset != 0) & (flags != 0);
    const int s_fin0 = state_final[i * SUB], s_fin1 = state_final[i * SUB + SUB - 1];   // (read only where two: scratch the state kernel fills for every env)
    const ShiftBounds sb = shift_bounds(a, a.r0);
#ifdef MDPP_S1I_ABL_NODRAW                       // (timing only, tools/ablate_step1.py: a made-up transform, no generator)
    Xform x00{a.r0, a.W / 2 + (int)(i % 7) - 3, a.H / 2 + (int)(i % 5) - 2, (int)(i % 360), 0};
    Xform x01 = x00, x10 = x00, x11 = x00;
    (void)sb;
#else
    // the reference's order: the step's observation (one picture per sub-space, relevant then irrelevant), then reset()'s
    const Xform x00 = draw_xform(a, sb, g, h);
    Xform x01 = x00;
    if (SUB == 2) x01 = draw_xform(a, sb, g, h);
    Xform x10 = x00, x11 = x01;
    if (two) {
        x10 = draw_xform(a, sb, g, h);
        if (SUB == 2) x11 = draw_xform(a, sb, g, h);
    }
#endif
    if constexpr (!PHILOX) {
        if (lane == 0) {
            g.store(a.rng_s, i);
            a.rng_half[i] = make_uint2(h.has32, h.u32);
        }
    }
    const size_t isz = (size_t)a.W * a.H;
    auto picture = [&](const Xform &x, int state, bool second, uint8_t *out) __attribute__((always_inline)) {
        RecRegs r = rec_words(a, x, state, second);
#pragma unroll
        for (int k = 0; k < 8; k++) r.lo[k] = (uint32_t)__builtin_amdgcn_readfirstlane((int)r.lo[k]);
#pragma unroll
        for (int k = 0; k < 4; k++) r.hi[k] = (uint32_t)__builtin_amdgcn_readfirstlane((int)r.hi[k]);
        if constexpr (WIDE) {
            render_wide(a, r, lds, wave, lane, out);
        } else {
            const TplRegs tp = load_tpl(a, r.lo[7] >> 12, lane);
            const u32x4 near = load_near(a, r, lane);
            stage_tpl(a, tp, lds, wave, lane);
            const ColRange cr = render_fast_eval(a, r, lds, lds_col, wave, lane, near);
            render_fast_store<NST>(a, cr, lds_col, out, lane);
        }
    };
#ifdef MDPP_S1I_ABL_NOPIC                        // (timing only: the serial head of the wave alone)
    if (x00.cx == 12345 && lane == 1) img_out[i] = (uint8_t)(x10.cy + x11.cy + x01.cy + s_out0 + s_fin0 + s_out1 + s_fin1);
    return;
#endif
#pragma unroll 1
    for (int q = 0; q < SUB; q++) {
        const size_t j = (size_t)i * SUB + q;
        if (two && img_final) picture(q ? x01 : x00, q ? s_fin1 : s_fin0, false, img_final + j * isz);
        picture(q ? x11 : x10, q ? s_out1 : s_out0, two, img_out + j * isz);
    }
}

// The fused step above, if this handle can use it (the caller then skips launch_image_obs): 0 = not taken, 1 = launched,
// < 0 = error.  state_out / state_final / term / trunc: what the state kernel of this step has just written (on s).
int launch_image_step1(mdpp_env *h, const int32_t *state_out, const int32_t *state_final, const uint8_t *term,
                       const uint8_t *trunc, uint8_t *img_out, uint8_t *img_final, hipStream_t s) {
    const mdpp_config &c = h->cfg;
    if (!h->img_fast_ok || (h->opts & (MDPP_OPT_NO_IMGFAST | MDPP_OPT_NO_STEP1)) || !img_out) return 0;
    (void)c;
    ImageArgs a = image_args(h, 1, false, 0);
    const size_t lds_bytes = image_fast_lds(h, a);
    if (a.colb != 64) {                          // wide templates: two waves per workgroup (render_wide)
        const dim3 wgrid((unsigned)((a.N + kWideBlock / 64 - 1) / (kWideBlock / 64)));
        if (a.philox) hipLaunchKernelGGL((k_image_step1<0, true, true>), wgrid, dim3(kWideBlock), lds_bytes, s, a, state_out, state_final,
                                         term, trunc, img_out, img_final);
        else hipLaunchKernelGGL((k_image_step1<0, false, true>), wgrid, dim3(kWideBlock), lds_bytes, s, a, state_out, state_final,
                                term, trunc, img_out, img_final);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) { h->err = std::string("k_image_step1 launch: ") + hipGetErrorString(e); return MDPP_EHIP; }
        return 1;
    }
    const dim3 grid((unsigned)((a.N + kBlock / 64 - 1) / (kBlock / 64)));
    const int nst = (int)(((size_t)a.W * a.H / 16 + 63) / 64);
#define MDPP_IMG_S1(NST_)                                                                                                     \
    do {                                                                                                                      \
        if (a.philox) hipLaunchKernelGGL((k_image_step1<NST_, true>), grid, dim3(kBlock), lds_bytes, s, a, state_out, state_final, \
                                         term, trunc, img_out, img_final);                                                    \
        else hipLaunchKernelGGL((k_image_step1<NST_, false>), grid, dim3(kBlock), lds_bytes, s, a, state_out, state_final,   \
                                term, trunc, img_out, img_final);                                                             \
    } while (0)
    if (nst == 7) MDPP_IMG_S1(7);          // 84 x 84
    else if (nst == 4) MDPP_IMG_S1(4);     // 64 x 64
    else MDPP_IMG_S1(0);
#undef MDPP_IMG_S1
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { h->err = std::string("k_image_step1 launch: ") + hipGetErrorString(e); return MDPP_EHIP; }
    return 1;
}

// K steps x N envs, arrays time-major; mask (reset only, K = 1) selects envs.  img_out == nullptr:
// draw only (the reference's reset()/step() consume the variates whether or not anyone looks).
int launch_image_obs(mdpp_env *h, int K, const int32_t *state_out, const int32_t *state_final,
                     const uint8_t *term, const uint8_t *trunc, const uint8_t *mask,
                     uint8_t *img_out, uint8_t *img_final, hipStream_t s, int phase, int buf) {
    if (K < 1 || K > h->img_chunk || K > kImgChunk || (mask && K != 1)) {
        h->err = "launch_image_obs: K outside the record scratch"; return MDPP_EINVAL;
    }
    ImageArgs a = image_args(h, K, term == nullptr, buf);
    static_assert(kBlock == 256, "render_fast packs four 64-byte template columns into a 256-byte LDS row");
    if (!(phase & 1)) {
        // records of this batch were made earlier (side stream)
    } else if (K == 1) {
        if (a.philox) hipLaunchKernelGGL((k_image_draw<true, true>), dim3((a.N + kBlock - 1) / kBlock), dim3(kBlock), 0, s, a, K, state_out,
                                         state_final, term, trunc, mask);
        else hipLaunchKernelGGL((k_image_draw<true, false>), dim3((a.N + kBlock - 1) / kBlock), dim3(kBlock), 0, s, a, K, state_out,
                                state_final, term, trunc, mask);
    } else {
        const long M = (long)K * a.N * a.SUB;
        if (a.philox) hipLaunchKernelGGL((k_image_draw<false, true>), dim3((a.N + kBlock - 1) / kBlock), dim3(kBlock), 0, s, a, K, state_out,
                                         state_final, term, trunc, mask);
        else hipLaunchKernelGGL((k_image_draw<false, false>), dim3((a.N + kBlock - 1) / kBlock), dim3(kBlock), 0, s, a, K, state_out,
                                state_final, term, trunc, mask);
        hipLaunchKernelGGL(k_image_rec, dim3((unsigned)((M + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, a, M,
                           state_out, state_final);
    }
    if (img_out && (phase & 2)) {
        const int per_block = kBlock / 64;
        const long M = (long)K * a.N * a.SUB;
        const unsigned nblk = (unsigned)((M + per_block - 1) / per_block);
        if (h->img_fast_ok && !(h->opts & MDPP_OPT_NO_IMGFAST) && h->img_colb == 128) {
            const size_t lds_bytes = image_fast_lds(h, a);                  // (host-checked: <= 64 KiB)
            const long per_wg = (long)(kWideBlock / 64) * (MDPP_IMG_WIDE_PAIR ? 2 : 1);            // pictures per workgroup
            const dim3 grid((unsigned)((M + per_wg - 1) / per_wg));
            hipLaunchKernelGGL(k_image_obs_wide, grid, dim3(kWideBlock), lds_bytes, s, a, M, a.rec0, img_out);
            if (img_final) hipLaunchKernelGGL(k_image_obs_wide, grid, dim3(kWideBlock), lds_bytes, s, a, M, a.rec1, img_final);
        } else if (h->img_fast_ok && !(h->opts & MDPP_OPT_NO_IMGFAST)) {
            const size_t lds_bytes = image_fast_lds(h, a);
            unsigned per_cu = (unsigned)((160u * 1024u) / ((lds_bytes + 511) & ~(size_t)511));
            per_cu = per_cu < 1u ? 1u : (per_cu > 8u ? 8u : per_cu);
#ifdef MDPP_IMG_WG_PER_CU
            per_cu = MDPP_IMG_WG_PER_CU;
#endif
            // (the comment below: rounds 1-2 had 40 KiB per workgroup, 4 resident workgroups per CU)
            // (phase 2 = the pipelined rollout: a few CUs keep a slot free, so that the next batch's state
            // kernel, which needs a little LDS, can run beside this one)
            // a few slots (rounds 1-2, 40 KiB of LDS per workgroup, four per CU: as many as the state kernel has workgroups --
            // 16 gave no overlap at all, 32 and 64 the same +13 %; round 3, LDS sized by the launch, five per CU for cfg4 and
            // batches of 32 steps: 7 385 / 7 672 us per launch with 32 reserved slots, 7 227 / 7 551 with 8, 7 289 / 7 660
            // with none, on two boxes)
            unsigned reserve = 8u;
#ifdef MDPP_IMG_RESERVE
            reserve = MDPP_IMG_RESERVE;
#endif
            const unsigned resident = per_cu * (unsigned)h->num_cus - (phase == 2 ? reserve : 0u);
#ifndef MDPP_IMG_ONESHOT
#define MDPP_IMG_ONESHOT 1         /* pictures per wave of the one-shot grid (0: the persistent grid of rounds 1-4, with MDPP_IMG_DYNAMIC = 1) */
#endif
#if MDPP_IMG_ONESHOT
            const dim3 grid((nblk + MDPP_IMG_ONESHOT - 1) / MDPP_IMG_ONESHOT);
            (void)resident;
#else
            const dim3 grid(nblk < resident ? nblk : resident);
#endif
            const int nst = (int)(((size_t)a.W * a.H / 16 + 63) / 64);
            for (int pass = 0; pass < (img_final ? 2 : 1); pass++) {
                const ImgRec *rec = pass ? a.rec1 : a.rec0;
                uint8_t *img = pass ? img_final : img_out;
                uint32_t *ctr = a.work_ctr + pass * kImgCtrs * 32;
                if (nst == 7) hipLaunchKernelGGL(k_image_obs_fast<7>, grid, dim3(kBlock), lds_bytes, s, a, M, rec, img, ctr);   // 84 x 84
                else if (nst == 4) hipLaunchKernelGGL(k_image_obs_fast<4>, grid, dim3(kBlock), lds_bytes, s, a, M, rec, img, ctr); // 64 x 64
                else hipLaunchKernelGGL(k_image_obs_fast<0>, grid, dim3(kBlock), lds_bytes, s, a, M, rec, img, ctr);
            }
        } else {
            const size_t lds = (size_t)per_block * (((size_t)a.tpl * a.tpl + 15) & ~(size_t)15);
            if (lds > 64 * 1024) { h->err = "k_image_obs: polygon template too large for LDS"; return MDPP_EUNSUPPORTED; }
            hipLaunchKernelGGL(k_image_obs, dim3(nblk), dim3(kBlock), lds, s, a, M, a.rec0, img_out);
            if (img_final) hipLaunchKernelGGL(k_image_obs, dim3(nblk), dim3(kBlock), lds, s, a, M, a.rec1, img_final);
        }
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { h->err = std::string("k_image_obs launch: ") + hipGetErrorString(e); return MDPP_EHIP; }
    return MDPP_OK;
}

const char *image_obs_kernel_name(const mdpp_env *h, int K) {
    if (!(h->img_fast_ok && !(h->opts & MDPP_OPT_NO_IMGFAST))) return "k_image_obs";
    if (h->img_colb == 128) return (K == 1 && !(h->opts & MDPP_OPT_NO_STEP1)) ? "k_image_step1<WIDE=1>" : "k_image_obs_wide";
    const int nst = (int)(((size_t)h->cfg.img_w * h->cfg.img_h / 16 + 63) / 64);
    if (K == 1 && !(h->opts & MDPP_OPT_NO_STEP1))     // (launch_image_step1's conditions)
        return nst == 7 ? "k_image_step1<NST=7>" : nst == 4 ? "k_image_step1<NST=4>" : "k_image_step1<NST=0>";
    return nst == 7 ? "k_image_obs_fast<NST=7>" : nst == 4 ? "k_image_obs_fast<NST=4>" : "k_image_obs_fast<NST=0>";
}

} // namespace mdpp
