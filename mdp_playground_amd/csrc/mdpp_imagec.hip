// ImageContinuous observations of continuous and grid envs:
// /root/reference/mdp_playground/spaces/image_continuous.py:116-277, called from
// rl_toy_env.py:2095-2096 (step) and :2347-2350 (reset).
//
// Per 2-D sub-space one W x H RGB picture: background (208,208,208); the terminal hypercubes as
// black rectangles (inclusive corner pixels from convert_to_pixel, :176-188); the target as a green
// disc and the agent as a blue disc, both Pillow ellipses with the integer bounding box centre +- R
// (:190-207), i.e. one fixed (2R+1)^2 raster (made on the host with Pillow, uploaded as row
// bitmasks) at an integer position; the irrelevant sub-space's picture shows only its agent disc;
// the pictures are concatenated along the first axis of the [x][y][channel] observation (:209-250).
// convert_to_pixel (:252-277): (v - min) / (max - min) in float32, times the image size in
// float64, truncated toward zero.  No random draws.
// Grid envs (draw_grid, :139-207): the same picture plus white grid lines (a per-config bit mask
// made on the host with Pillow's draw.line from the reference's end points), terminal CELLS as
// rectangles convert_to_pixel(cell) .. convert_to_pixel(cell + 1), discs at cell + 0.5, and
// convert_to_pixel in float64 throughout: int(v / g * size).
//
// One wavefront per image.  Phase 1 paints 2-bit colour codes {0 background, 1 terminal, 2 target,
// 3 agent} of the n_sub * W * H pixels into LDS with ds_or / ds_and (shapes are a few hundred
// pixels); phase 2 streams the picture out: a lane takes 16 consecutive pixels (one dword of codes),
// expands them to 48 bytes and writes three 16-byte stores, so a wave instruction covers 3 KiB
// contiguous; groups that are all background skip the expansion.  Write-bound: 3 W H n_sub bytes
// per observation and nothing else.
#include <string.h>

#include "mdpp_internal.hpp"

namespace mdpp {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

struct ImageCArgs {
    int32_t N, D, W, H, R, n_sub, n_boxes, autoreset;
    float smax;
    float target[2];
    const float *boxes;         // [n_boxes]{lo0, lo1, hi0, hi1}: the hypercubes' first two relevant dimensions (grid: {cell0, cell1, -, -})
    uint32_t disc_rows[32];     // bit dx of row dy: pixel (dx, dy) of the (2R+1)^2 disc raster
    // grid envs
    int32_t G, shape[4], gtarget[2];
    const uint16_t *lines;      // [ceil(n_sub W H / 16)] grid-line bits of 16 consecutive pixels
};

__device__ __forceinline__ int ic_px(float v, float smax, int size) {
    const float lo = -smax, hi = smax;
    const float f = (v - lo) / (hi - lo);
    return (int)((double)f * (double)size);
}

__device__ __forceinline__ int ig_px(double v, int g, int size) { return (int)((v - 0.0) / (double)(g - 0) * (double)size); }

// states [M][D] float32, or [M][G] int32 cells for GRID (time-major batches of the step kernel's
// observations); flags (nullable): only images with term | trunc set are rendered (the terminal
// observations of reset steps).
template <bool GRID>
__global__ __launch_bounds__(kBlock) void k_imagec_obs(ImageCArgs a, long M, const void *__restrict__ states_v,
                                                       const uint8_t *__restrict__ term,
                                                       const uint8_t *__restrict__ trunc,
                                                       const uint8_t *__restrict__ mask,
                                                       uint8_t *__restrict__ img) {
    extern __shared__ __align__(16) uint32_t lds_codes[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // (workgroup b runs on XCD b % 8: every XCD renders one contiguous eighth of the launch's pictures, as in mdpp_image.hip)
#ifndef MDPP_IMGC_XCD
#define MDPP_IMGC_XCD 1
#endif
    const uint32_t bx = (MDPP_IMGC_XCD && (gridDim.x & 7u) == 0u) ? (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3) : blockIdx.x;
    const long j = __builtin_amdgcn_readfirstlane((int)((long)bx * (kBlock / 64) + wave));
    if (j >= M) return;
    if (mask && !mask[j % a.N]) return;
    if (term && !(term[j] | trunc[j])) return;
    const int npix = a.n_sub * a.W * a.H, ngroup = (npix + 15) >> 4;   // (a ragged last group: the byte-store form below)
    const int ngpad = (ngroup + 3) & ~3;                                 // (the slab behind the codes stays 16-byte aligned)
    uint32_t *codes = lds_codes + (size_t)wave * (ngpad + 768);          // (+ the wave's 3 KiB store slab)
    uint32_t *slab = codes + ngpad;
    for (int g = lane; g < ngroup; g += 64) codes[g] = 0u;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    const float *st = (const float *)states_v + (size_t)j * a.D;
    const int32_t *cell = (const int32_t *)states_v + (size_t)j * a.G;
    auto paint = [&](int p, uint32_t code, bool clear) {           // pixel index p = (sub W + x) H + y
        uint32_t *w = codes + (p >> 4);
        const int sh = 2 * (p & 15);
        if (clear) atomicAnd(w, ~(3u << sh));
        atomicOr(w, code << sh);
    };
    // terminal hypercubes (relevant picture only), inclusive pixel rectangles
    for (int b = 0; b < a.n_boxes; b++) {
        int x0, y0, x1, y1;
        const float b0 = a.boxes[4 * b], b1 = a.boxes[4 * b + 1], b2 = a.boxes[4 * b + 2], b3 = a.boxes[4 * b + 3];     // (wave-uniform: scalar loads)
        if (GRID) {
            const int c0 = (int)b0, c1 = (int)b1;
            x0 = ig_px(c0, a.shape[0], a.W); y0 = ig_px(c1, a.shape[1], a.H);
            x1 = ig_px(c0 + 1.0, a.shape[0], a.W); y1 = ig_px(c1 + 1.0, a.shape[1], a.H);
        } else {
            x0 = ic_px(b0, a.smax, a.W); y0 = ic_px(b1, a.smax, a.H);
            x1 = ic_px(b2, a.smax, a.W); y1 = ic_px(b3, a.smax, a.H);
        }
        x0 = max(x0, 0); y0 = max(y0, 0); x1 = min(x1, a.W - 1); y1 = min(y1, a.H - 1);
        const int bw = x1 - x0 + 1, bh = y1 - y0 + 1;
        if (bw <= 0 || bh <= 0) continue;
        for (int k = lane; k < bw * bh; k += 64) {
            const int dx = k / bh, dy = k - dx * bh;
            paint((x0 + dx) * a.H + y0 + dy, 1u, false);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    const int T = 2 * a.R + 1;
    auto disc = [&](int sub, int cx, int cy, uint32_t code, bool clear) {
        for (int k = lane; k < T * T; k += 64) {
            const int dy = k / T, dx = k - dy * T;
            const int x = cx - a.R + dx, y = cy - a.R + dy;
            if (((a.disc_rows[dy] >> dx) & 1u) && x >= 0 && x < a.W && y >= 0 && y < a.H)
                paint((sub * a.W + x) * a.H + y, code, clear);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    };
    if (GRID) {     // discs at the cell centres; both sub-spaces scaled with the relevant grid's size (:261-262)
        disc(0, ig_px(a.gtarget[0] + 0.5, a.shape[0], a.W), ig_px(a.gtarget[1] + 0.5, a.shape[1], a.H), 2u, true);
        disc(0, ig_px(cell[0] + 0.5, a.shape[0], a.W), ig_px(cell[1] + 0.5, a.shape[1], a.H), 3u, false);
        if (a.n_sub > 1) disc(1, ig_px(cell[2] + 0.5, a.shape[0], a.W), ig_px(cell[3] + 0.5, a.shape[1], a.H), 3u, false);
    } else {
        disc(0, ic_px(a.target[0], a.smax, a.W), ic_px(a.target[1], a.smax, a.H), 2u, true);   // over the rectangles
        disc(0, ic_px(st[0], a.smax, a.W), ic_px(st[1], a.smax, a.H), 3u, false);              // 3 = 0b11: OR overrides
        if (a.n_sub > 1) disc(1, ic_px(st[2], a.smax, a.W), ic_px(st[3], a.smax, a.H), 3u, false);
    }

    // phase 2: 16 pixels -> 48 bytes per lane
    const size_t isz = (size_t)npix * 3;
    if (npix & 15) {                        // pictures whose pixel count is not a multiple of 16 (50 x 50 ...): pictures then start at
        uint8_t *o = img + (size_t)j * isz; // any byte address -- one pixel per lane, three byte stores (the reference has no such rule;
        for (int p = lane; p < npix; p += 64) {     // round 6: accepted, at the price of this form's store rate)
            const uint32_t code = (codes[p >> 4] >> (2 * (p & 15))) & 3u;
            const uint32_t bg = (GRID && ((a.lines[p >> 4] >> (p & 15)) & 1u)) ? 0xFFFFFFu : 0xD0D0D0u;
            const uint32_t px = code == 0 ? bg : code == 1 ? 0u : code == 2 ? 0x00FF00u : 0xFF0000u;
            o[3 * (size_t)p] = (uint8_t)px; o[3 * (size_t)p + 1] = (uint8_t)(px >> 8); o[3 * (size_t)p + 2] = (uint8_t)(px >> 16);
        }
        return;
    }
    const auto r_out = __builtin_amdgcn_make_buffer_rsrc((void *)(img + (size_t)j * isz), 0, (int)isz, 0x00020000);
#ifdef MDPP_IMGC_DIRECT
    for (int g = lane; g < ngroup; g += 64) {
#else
    for (int g0 = 0; g0 < ngroup; g0 += 64) {                      // (every lane takes part in every round's stores)
        const int g = min(g0 + lane, ngroup - 1);
#endif
        const uint32_t c = codes[g];
        const uint32_t ln = GRID ? (uint32_t)a.lines[g] : 0u;       // white where no shape covers the line
        u32x4 o0, o1, o2;
        if (__builtin_amdgcn_ballot_w64((c | ln) != 0u) == 0) {
            o0 = o1 = o2 = u32x4{0xD0D0D0D0u, 0xD0D0D0D0u, 0xD0D0D0D0u, 0xD0D0D0D0u};
        } else {
            uint32_t d[12];
#pragma unroll
            for (int q = 0; q < 4; q++) {                         // 4 pixels -> 3 dwords
                uint32_t px[4];
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const uint32_t code = (c >> (2 * (4 * q + e))) & 3u;
                    // bytes R, G, B as a little-endian 24-bit value
                    const uint32_t bg = (GRID && ((ln >> (4 * q + e)) & 1u)) ? 0xFFFFFFu : 0xD0D0D0u;
                    px[e] = code == 0 ? bg : code == 1 ? 0u : code == 2 ? 0x00FF00u : 0xFF0000u;
                }
                d[3 * q] = px[0] | (px[1] << 24);
                d[3 * q + 1] = (px[1] >> 8) | (px[2] << 16);
                d[3 * q + 2] = (px[2] >> 16) | (px[3] << 8);
            }
            o0 = u32x4{d[0], d[1], d[2], d[3]};
            o1 = u32x4{d[4], d[5], d[6], d[7]};
            o2 = u32x4{d[8], d[9], d[10], d[11]};
        }
#ifdef MDPP_IMGC_DIRECT
        __builtin_amdgcn_raw_buffer_store_b128(o0, r_out, g * 48, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b128(o1, r_out, g * 48 + 16, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b128(o2, r_out, g * 48 + 32, 0, 0);
#else
        // The lane's 48 bytes go through a 3 KiB LDS slab of the wave, so that every store instruction writes 1 KiB
        // CONTIGUOUS (lane l: bytes 16 l ..) instead of 16 bytes out of every 48
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");        // (the slab's previous round has been read)
        u32x4 *sl = (u32x4 *)slab;
        sl[lane * 3] = o0; sl[lane * 3 + 1] = o1; sl[lane * 3 + 2] = o2;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        const int base = g0 * 48;                                      // first byte of this round of 64 groups
        const int nb = min(64, ngroup - g0) * 48;                      // bytes of this round
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const int off = k * 1024 + lane * 16;
            if (off < nb) __builtin_amdgcn_raw_buffer_store_b128(sl[k * 64 + lane], r_out, base + off, 0, 0);
        }
#endif
    }
}

// K steps x N envs (time-major).  pass 0: states -> img_out for every image; pass 1 (img_final):
// final_states -> img_final for the steps that ended in a reset.
int launch_imagec_obs(mdpp_env *h, int K, const void *states, const void *final_states, const uint8_t *term,
                      const uint8_t *trunc, const uint8_t *mask, uint8_t *img_out, uint8_t *img_final, hipStream_t s) {
    const mdpp_config &c = h->cfg;
    const bool grid = c.kind == MDPP_KIND_GRID;
    ImageCArgs a;
    memset(&a, 0, sizeof(a));
    a.N = c.num_envs; a.W = c.img_w; a.H = c.img_h; a.R = c.img_r0;
    a.n_boxes = c.n_boxes; a.autoreset = c.autoreset;
    a.boxes = (const float *)h->d_imgc_boxes;
    if (grid) {
        a.G = c.grid_dims; a.D = c.grid_dims; a.n_sub = c.grid_dims / 2;
        for (int d = 0; d < c.grid_dims; d++) a.shape[d] = c.grid_shape[d];
        a.gtarget[0] = c.grid_target[0]; a.gtarget[1] = c.grid_target[1];
        a.lines = (const uint16_t *)h->d_img_tpl;
    } else {
        a.D = c.D; a.n_sub = c.D > 2 ? 2 : 1;
        a.smax = (float)c.state_space_max;
        a.target[0] = c.target[0]; a.target[1] = c.target[1];
    }
    for (int r = 0; r < 32; r++) a.disc_rows[r] = h->imgc_disc_rows[r];
    const long M = (long)K * a.N;
    const int per_block = kBlock / 64;
    const size_t lds = (size_t)per_block * (((((size_t)a.n_sub * a.W * a.H + 15) / 16 + 3) & ~(size_t)3) + 768) * 4;
    if (lds > 64 * 1024) { h->err = "k_imagec_obs: image too large for the LDS colour map"; return MDPP_EUNSUPPORTED; }
    const dim3 grd((unsigned)((M + per_block - 1) / per_block));
#define MDPP_IC_LAUNCH(GR, st, te, tr, im) hipLaunchKernelGGL((k_imagec_obs<GR>), grd, dim3(kBlock), lds, s, a, M, st, te, tr, mask, im)
    if (img_out) { if (grid) MDPP_IC_LAUNCH(true, states, nullptr, nullptr, img_out); else MDPP_IC_LAUNCH(false, states, nullptr, nullptr, img_out); }
    if (img_final && final_states && term) {
        if (grid) MDPP_IC_LAUNCH(true, final_states, term, trunc, img_final); else MDPP_IC_LAUNCH(false, final_states, term, trunc, img_final);
    }
#undef MDPP_IC_LAUNCH
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { h->err = std::string("k_imagec_obs launch: ") + hipGetErrorString(e); return MDPP_EHIP; }
    return MDPP_OK;
}

} // namespace mdpp
