// Batched GymEnvWrapper-style post-processor: what the reference's
// mdp_playground/envs/gym_env_wrapper.py does around ANY inner env, applied to caller-supplied
// (action | obs, reward, done) tensors of N independent env instances, one lane per instance:
//   mdpp_post_actions   discrete action noise                               gym_env_wrapper.py:354-366
//   mdpp_post_step_n    continuous observation noise :367-373, :400-402; image padding / shift
//                       :404-405, :523-618; reward delay, flush-on-done, terminal reward, noise, scale,
//                       shift :407-432
//   mdpp_post_reset     reset(): reward buffer refilled with zeros, image of the first observation :441-486
// Instance i owns ONE generator -- the wrapper's _np_random -- (numpy PCG64 state in HBM, or a Philox
// stream keyed per call) and a reward FIFO `ring[delay][N]` (float64, like the reference's list of
// Python floats) with a per-instance head, because a `done` step neither pushes nor pops (:407-414).
// Everything is HBM/latency-bound scalar work except the image path, which writes (W + 2 pad) x
// (H + 2 pad) x C bytes per instance and step: k_post_image moves dwords, transposing through the
// read side (the canvas is returned as [x][y][c], :616).
#include <math.h>
#include <string.h>

#include <string>
#include <vector>

#include "mdpp_internal.hpp"
#include "mdpp_rng.hpp"

using namespace mdpp;

struct mdpp_post {
    mdpp_post_config cfg;
    int device;
    std::string err;
    uint64_t tick, reset_tick;
    uint64_t action_tick;       // mdpp_post_actions calls so far (Philox counter of the action-noise stream)
    void *d_rng_s, *d_rng_inc, *d_half, *d_ring, *d_head, *d_noise_cdf, *d_shift, *d_xyc;
    int num_cus;
    size_t shift_cap;           // image placements held by d_shift (K * N of the largest call so far)
    bool seeded;
};

static std::string g_post_create_err;

namespace {

constexpr uint32_t kPhiloxPostAction = 6, kPhiloxPostStep = 7, kPhiloxPostReset = 8;

struct PostArgs {
    int32_t N, continuous, n_actions, obs_dim, obs_f64, delay, has_p, has_r, autoreset;
    int32_t image, H, W, C, pad, has_shift, sh_quant, philox;
    double p_noise, r_noise, scale, shift, term;
    uint64_t philox_seed, tick, action_tick;
    int64_t env_id_offset;
    ulonglong2 *rng_s, *rng_inc;
    uint2 *half;
    double *ring;               // [delay][N]
    uint32_t *head;             // [N] FIFO front
    const double *noise_cdf;    // [n][n]
    short2 *place;              // [K][N] (top, left) of the image inside the canvas
};

// get_transformed_image's placement (:560-611): draws the shift, returns (top, left)
template <class G>
__device__ __forceinline__ short2 post_place(const PostArgs &a, G &g, Half32 &h) {
    const int tot_w = a.W + 2 * a.pad, tot_h = a.H + 2 * a.pad;
    int shift_w = tot_w / 2, shift_h = tot_h / 2;
    if (a.has_shift) {
        const int max_w = (tot_w - a.W) / 2, max_h = (tot_h - a.W) / 2;       // R = width, :558
        int aw = np_integers(g, h, -max_w + 1, max_w);
        int ah = np_integers(g, h, -max_h + 1, max_h);
        aw = (aw / a.sh_quant) * a.sh_quant;                                  // int(a / q) * q: truncation
        ah = (ah / a.sh_quant) * a.sh_quant;
        shift_w += aw; shift_h += ah;
    }
    return make_short2((short)(shift_h - a.H / 2), (short)(shift_w - a.W / 2));
}

template <bool PHILOX>
__global__ __launch_bounds__(kBlock) void k_post_actions(PostArgs a, const int32_t *__restrict__ in,
                                                         int32_t *__restrict__ out) {
    const long i = (long)blockIdx.x * kBlock + threadIdx.x;
    if (i >= a.N) return;
    int act = in[i];
    if (act < 0 || act >= a.n_actions) { out[i] = act; return; }             // (the reference would raise IndexError)
    typename std::conditional<PHILOX, Philox, Pcg64>::type g;
    if constexpr (PHILOX) g.init(a.philox_seed, (uint64_t)(a.env_id_offset + i), a.action_tick, kPhiloxPostAction);
    else g.load(a.rng_s, a.rng_inc, i);
    const double u = np_random(g);                                            // choice(n, size=1, p=probs), :364
    out[i] = searchsorted_right(a.noise_cdf + (size_t)act * a.n_actions, a.n_actions, u);
    if constexpr (!PHILOX) g.store(a.rng_s, i);
}

// LDSRING: the instance's reward FIFO lives in LDS for the call (delay <= kPostLdsDelay: [delay][256] doubles),
// loaded once and written back at the end -- otherwise every step is a dependent HBM read-modify-write of its
// slot, and with one wave per SIMD that round trip IS the step time (2.1 us per step measured).
constexpr int kPostLdsDelay = 16;
constexpr int kPostPre = 8;              // reward / done inputs in flight per lane
constexpr int kPostUB = 8;               // k_post_image_lds: loads in flight per lane

// RING = 2: delay <= 8 -- the FIFO is eight registers in pay-out order (a shift per step, no memory round
// trip in the step at all: the LDS form spent most of a step waiting on its dependent ds_read / ds_write);
// RING = 1: the LDS form (delay <= 16); RING = 0: slots in HBM.
constexpr int kPostRegDelay = 8;
// DC > 0: the delay as a compile-time constant (register FIFO: the shift and the flush sum lose their selects)
template <bool PHILOX, int RING, int DC>
__global__ __launch_bounds__(kBlock) void k_post_step(PostArgs a, int K, const void *__restrict__ obs_in,
                                                      const double *__restrict__ reward_in,
                                                      const uint8_t *__restrict__ done_in, void *__restrict__ obs_out,
                                                      double *__restrict__ reward_out) {
    __shared__ uint64_t s_ki[256];
    __shared__ double s_wi[256], s_fi[256];
    constexpr bool LDSRING = RING == 1, REGRING = RING == 2;
    const int delay = DC > 0 ? DC : a.delay;
    __shared__ double s_ring[LDSRING ? kPostLdsDelay * kBlock : 1];
    const bool normals = !PHILOX && ((a.continuous && a.has_p) || a.has_r);
    if (normals) { zig_stage(s_ki, s_wi, s_fi, threadIdx.x, kBlock); __syncthreads(); }
    const ZigLds zig{s_ki, s_wi, s_fi};
    const long i = (long)blockIdx.x * kBlock + threadIdx.x;
    if (i >= a.N) return;
    const long N = a.N;
    typename std::conditional<PHILOX, Philox, Pcg64>::type g;
    Half32 hf{0, 0};
    const bool draws = (a.continuous && a.has_p) || a.has_r || (a.image && a.has_shift);
    if constexpr (!PHILOX) {
        if (draws) { g.load(a.rng_s, a.rng_inc, i); const uint2 hh = a.half[i]; hf = Half32{hh.x, hh.y}; }
    }
    uint32_t head = delay > 0 ? a.head[i] : 0u;
    // ring slot j of this lane: LDS column (conflict-free: consecutive lanes, consecutive banks) or HBM
    double *ringp = LDSRING ? s_ring + threadIdx.x : a.ring + i;
    const size_t rstride = LDSRING ? (size_t)kBlock : (size_t)N;
    if (LDSRING) for (int j = 0; j < delay; j++) s_ring[j * kBlock + threadIdx.x] = a.ring[(size_t)j * N + i];
    double rq[kPostRegDelay];            // REGRING: rq[0] pays out next
    if (REGRING) {
#pragma unroll
        for (int j = 0; j < kPostRegDelay; j++) {
            const uint32_t sl = head + (uint32_t)j < (uint32_t)delay ? head + (uint32_t)j : head + (uint32_t)j - (uint32_t)delay;
            rq[j] = j < delay ? a.ring[(size_t)sl * N + i] : 0.0;
        }
        head = 0;
    }
    double pre_r[kPostPre];
    uint8_t pre_d[kPostPre];
#pragma unroll
    for (int u = 0; u < kPostPre; u++) {
        const long oo = (long)(u < K ? u : K - 1) * N + i;
        pre_r[u] = reward_in[oo]; pre_d[u] = done_in[oo];
    }
    // one instance step; slot u of the prefetch buffers holds step k and is refilled with step k + kPostPre
    auto step = [&](const int k, const int u, const bool refill) __attribute__((always_inline)) {
        const long o = (long)k * N + i;
        double reward = pre_r[u];
        const bool done = pre_d[u] != 0;
        if (refill) {
            const int kn = k + kPostPre;
            const long oo = (long)(kn < K ? kn : K - 1) * N + i;
            pre_r[u] = reward_in[oo]; pre_d[u] = done_in[oo];
        }
        if constexpr (PHILOX) {
            g.init(a.philox_seed, (uint64_t)(a.env_id_offset + i), a.tick + (uint64_t)k, kPhiloxPostStep);
            hf = Half32{0, 0};
        }
        if (a.continuous) {                                                   // :367-373, :400-402
            for (int d = 0; d < a.obs_dim; d++) {
                const double nz = a.has_p ? 0.0 + a.p_noise * np_standard_normal_lds(g, zig) : 0.0;
                if (a.obs_f64) ((double *)obs_out)[o * a.obs_dim + d] = ((const double *)obs_in)[o * a.obs_dim + d] + nz;
                else ((float *)obs_out)[o * a.obs_dim + d] = (float)((double)((const float *)obs_in)[o * a.obs_dim + d] + nz);
            }
        }
        if (a.image) a.place[o] = post_place(a, g, hf);                      // :404-405 (pixels: k_post_image)
        if (done) {                                                           // :407-414
            // np.sum(buffer * scale + shift) in numpy's pairwise order, read straight from the FIFO (a private
            // array here would live in scratch memory, and some lane of a wave is done on most steps)
            auto val = [&](int j) __attribute__((always_inline)) -> double {
                const uint32_t sl = head + (uint32_t)j < (uint32_t)delay ? head + (uint32_t)j : head + (uint32_t)j - (uint32_t)delay;
                return ringp[(size_t)sl * rstride] * a.scale + a.shift;
            };
            double sum;
            if (REGRING) {
                if (delay < 8) {
                    sum = 0.;
#pragma unroll
                    for (int j = 0; j < kPostRegDelay - 1; j++) sum = j < delay ? sum + (rq[j] * a.scale + a.shift) : sum;
                } else {
                    double v[8];
#pragma unroll
                    for (int j = 0; j < 8; j++) v[j] = rq[j] * a.scale + a.shift;
                    sum = ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
                }
            } else if (delay < 8) {
                sum = 0.;
                for (int j = 0; j < delay; j++) sum += val(j);
            } else {
                double r0 = val(0), r1 = val(1), r2 = val(2), r3 = val(3), r4 = val(4), r5 = val(5), r6 = val(6), r7 = val(7);
                int j;
                for (j = 8; j < delay - (delay % 8); j += 8) {
                    r0 += val(j); r1 += val(j + 1); r2 += val(j + 2); r3 += val(j + 3);
                    r4 += val(j + 4); r5 += val(j + 5); r6 += val(j + 6); r7 += val(j + 7);
                }
                sum = ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7));
                for (; j < delay; j++) sum += val(j);
            }
            reward += sum;
            reward += a.term * a.scale;
            if (a.autoreset) {                                                // the caller's env reset itself: reset(), :456
                if (REGRING) {
#pragma unroll
                    for (int j = 0; j < kPostRegDelay; j++) rq[j] = 0.0;
                } else {
                    for (int j = 0; j < delay; j++) ringp[(size_t)j * rstride] = 0.0;
                }
                head = 0;
            }
        } else if (REGRING) {                                                 // :415-420, FIFO as a shift register
            const double out = rq[0];
#pragma unroll
            for (int j = 0; j < kPostRegDelay; j++) {
                const double nxt = j + 1 < kPostRegDelay ? rq[j + 1] : 0.0;
                rq[j] = j + 1 == delay ? reward : nxt;
            }
            reward = out;
        } else if (delay > 0) {                                             // :415-420
            double *slot = ringp + (size_t)head * rstride;
            const double out = *slot;
            *slot = reward;
            reward = out;
            head = head + 1u == (uint32_t)delay ? 0u : head + 1u;
        }
        const double nz = a.has_r ? 0.0 + a.r_noise * np_standard_normal_lds(g, zig) : 0.0;   // :426
        reward += nz;                                                         // :430-432
        reward *= a.scale;
        reward += a.shift;
        __builtin_nontemporal_store(reward, &reward_out[o]);                 // (written once, not read here)
    };
    // Single-exit loop over full groups, the ragged tail outside it: with a `break` in the unrolled body the
    // compiler waits for EVERY load in flight at the loop head (s_waitcnt vmcnt(0)) and the prefetch is void.
    const int kfull = K - K % kPostPre;
    for (int k0 = 0; k0 < kfull; k0 += kPostPre) {
#pragma unroll
        for (int u = 0; u < kPostPre; u++) step(k0 + u, u, true);
    }
#pragma unroll
    for (int u = 0; u < kPostPre - 1; u++)
        if (kfull + u < K) step(kfull + u, u, false);
    if (LDSRING) for (int j = 0; j < delay; j++) a.ring[(size_t)j * N + i] = s_ring[j * kBlock + threadIdx.x];
    if (REGRING) {
#pragma unroll
        for (int j = 0; j < kPostRegDelay; j++) if (j < delay) a.ring[(size_t)j * N + i] = rq[j];
    }
    if (delay > 0) a.head[i] = head;
    if constexpr (!PHILOX) {
        if (draws) { g.store(a.rng_s, i); a.half[i] = make_uint2(hf.has32, hf.u32); }
    }
}

template <bool PHILOX>
__global__ __launch_bounds__(kBlock) void k_post_reset(PostArgs a, uint64_t reset_tick, const uint8_t *__restrict__ mask) {
    const long i = (long)blockIdx.x * kBlock + threadIdx.x;
    if (i >= a.N) return;
    if (mask && !mask[i]) { if (a.image) a.place[i] = make_short2(-32768, 0); return; }
    for (int j = 0; j < a.delay; j++) a.ring[(size_t)j * a.N + i] = 0.0;      // :456
    if (a.delay > 0) a.head[i] = 0;
    if (a.image) {                                                            // :481-482
        typename std::conditional<PHILOX, Philox, Pcg64>::type g;
        Half32 hf{0, 0};
        if constexpr (PHILOX) g.init(a.philox_seed, (uint64_t)(a.env_id_offset + i), reset_tick, kPhiloxPostReset);
        else if (a.has_shift) { g.load(a.rng_s, a.rng_inc, i); const uint2 hh = a.half[i]; hf = Half32{hh.x, hh.y}; }
        a.place[i] = post_place(a, g, hf);
        if constexpr (!PHILOX) {
            if (a.has_shift) { g.store(a.rng_s, i); a.half[i] = make_uint2(hf.has32, hf.u32); }
        }
    }
}

// canvas[x][y][c] = image[y - top][x - left][c] inside the placed image, else 0.  One workgroup per image:
// the source image is staged in LDS TRANSPOSED ([sx][sy][c], rows at an odd dword pitch: byte scatter on the
// way in, 21 K byte writes for 84 x 84 x 3), so that a canvas row x is one LDS row shifted by top * C bytes
// between two runs of zeros: every canvas dword is two LDS dwords through a byte funnel shift
// (v_alignbyte) plus a range mask -- about 20 instructions, where gathering its four bytes one by one took
// about 100 and made the kernel issue-bound at 0.2 of the HBM roofline.  Stores go front to back, 1 KiB
// contiguous per wave instruction.  Which (x, byte of the row) a canvas dword starts at is the same for every
// image of a handle: table `xr` (x << 16 | row byte), made once at create().
__global__ __launch_bounds__(kBlock) void k_post_image_lds(PostArgs a, long M, const short2 *__restrict__ place,
                                                           const uint32_t *__restrict__ xr, const uint8_t *__restrict__ in,
                                                           uint8_t *__restrict__ out) {
    extern __shared__ __align__(16) uint8_t s_img[];
    const int tot_w = a.W + 2 * a.pad, tot_h = a.H + 2 * a.pad, C = a.C;
    const long out_bytes = (long)tot_w * tot_h * C, in_bytes = (long)a.H * a.W * C;
    const int dwords = (int)(out_bytes / 4), in_dw = (int)(in_bytes / 4);
    const int colb = a.H * C;                           // bytes of one transposed row (one source column)
    const int pitch = (((colb + 3) / 4 + 2) | 1) * 4;   // odd dword pitch, one spare dword behind the data (the funnel reads it)
    for (long img = blockIdx.x; img < M; img += gridDim.x) {
        const short2 pl = place[img];
        if (pl.x == -32768) continue;                  // (wave-uniform: one image per workgroup)
        const uint32_t *src = (const uint32_t *)(in + img * in_bytes);
        __syncthreads();                               // the previous image's readers are done
        // (kPostUB loads in flight per lane: one at a time, the loop is a chain of HBM round trips -- 50 us per image)
        for (int q0 = threadIdx.x; q0 < in_dw; q0 += kBlock * kPostUB) {
            uint32_t v[kPostUB], m[kPostUB];
#pragma unroll
            for (int u = 0; u < kPostUB; u++) {
                const int q = q0 + u * kBlock;
                v[u] = q < in_dw ? src[q] : 0u;
                m[u] = q < in_dw ? xr[dwords + q] : 0u; // (sx << 18 | sy << 4 | c) of the source dword's first byte
            }
#pragma unroll
            for (int u = 0; u < kPostUB; u++) {
                if (q0 + u * kBlock >= in_dw) break;
                const int sy = (int)((m[u] >> 4) & 0x3FFFu);
                int sx = (int)(m[u] >> 18), c = (int)(m[u] & 15u);
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    s_img[sx * pitch + sy * C + c] = (uint8_t)(v[u] >> (8 * j));
                    c += 1;
                    if (c == C) { c = 0; sx += 1; }    // (a dword never crosses a source row: row_bytes % 4 == 0)
                }
            }
        }
        // the spare dword behind every transposed row reads as zero
        for (int r = threadIdx.x; r < a.W; r += kBlock) {
            *(uint32_t *)(s_img + r * pitch + ((colb + 3) & ~3)) = 0u;
            if (colb & 3) for (int t = colb; t < ((colb + 3) & ~3); t++) s_img[r * pitch + t] = 0;
        }
        __syncthreads();
        uint32_t *dst = (uint32_t *)(out + img * out_bytes);
        const int shift_b = pl.x * C;                  // canvas row byte of the image's first byte
        for (int q0 = threadIdx.x; q0 < dwords; q0 += kBlock * kPostUB) {
            uint32_t t[kPostUB];
#pragma unroll
            for (int u = 0; u < kPostUB; u++) t[u] = q0 + u * kBlock < dwords ? xr[q0 + u * kBlock] : 0u;
#pragma unroll
            for (int u = 0; u < kPostUB; u++) {
                const int q = q0 + u * kBlock;
                if (q >= dwords) break;
                const int sx = (int)(t[u] >> 16) - pl.y, rel = (int)(t[u] & 0xFFFFu) - shift_b;   // byte of the transposed row this dword starts at
                uint32_t v = 0;
                if (sx >= 0 && sx < a.W && rel > -4 && rel < colb) {
                    const int base = rel >> 2;          // (arithmetic shift: -1 for rel in -3..-1)
                    const uint8_t *rowp = s_img + sx * pitch;
                    const uint32_t lo = base >= 0 ? *(const uint32_t *)(rowp + base * 4) : 0u;
                    const uint32_t hi = *(const uint32_t *)(rowp + (base + 1) * 4);
                    v = __builtin_amdgcn_alignbyte(hi, lo, (uint32_t)(rel & 3));
                    // bytes before the image (rel + j < 0) are zero through `lo`; bytes behind it through the zeroed spare
                }
                dst[q] = v;
            }
        }
    }
}

// general form (any size): one dword of the canvas per lane, source bytes gathered from HBM / L2
__global__ __launch_bounds__(kBlock) void k_post_image(PostArgs a, long M, const short2 *__restrict__ place,
                                                       const uint8_t *__restrict__ in, uint8_t *__restrict__ out) {
    const int tot_w = a.W + 2 * a.pad, tot_h = a.H + 2 * a.pad, C = a.C;
    const long out_bytes = (long)tot_w * tot_h * C, in_bytes = (long)a.H * a.W * C;
    const long dwords = out_bytes / 4;                 // (create() checks divisibility)
    const long total = M * dwords;
    for (long q = (long)blockIdx.x * kBlock + threadIdx.x; q < total; q += (long)gridDim.x * kBlock) {
        const long img = q / dwords;
        const short2 pl = place[img];
        if (pl.x == -32768) continue;                  // not reset: leave the caller's buffer alone
        const uint32_t b0 = (uint32_t)(q - img * dwords) * 4u;
        const uint8_t *src = in + img * in_bytes;
        uint32_t v = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint32_t b = b0 + (uint32_t)j;
            const uint32_t pix = b / (uint32_t)C, c = b - pix * (uint32_t)C;
            const int x = (int)(pix / (uint32_t)tot_h), y = (int)(pix - (uint32_t)x * (uint32_t)tot_h);
            const int sy = y - pl.x, sx = x - pl.y;
            uint32_t byte = 0;
            if (sy >= 0 && sy < a.H && sx >= 0 && sx < a.W) byte = src[((long)sy * a.W + sx) * C + c];
            v |= byte << (8 * j);
        }
        ((uint32_t *)(out + img * out_bytes))[q - img * dwords] = v;
    }
}

int pfail(mdpp_post *h, int code, const std::string &msg) {
    if (h) h->err = msg; else g_post_create_err = msg;
    return code;
}

#define PHIP(h, expr)                                                                    \
    do {                                                                                 \
        hipError_t e_ = (expr);                                                          \
        if (e_ != hipSuccess) return pfail(h, MDPP_EHIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

PostArgs make_args(const mdpp_post *h) {
    const mdpp_post_config &c = h->cfg;
    PostArgs a;
    memset(&a, 0, sizeof a);
    a.N = c.num_envs; a.continuous = c.continuous; a.n_actions = c.n_actions; a.obs_dim = c.obs_dim; a.obs_f64 = c.obs_f64;
    a.delay = c.delay; a.has_p = c.has_transition_noise; a.has_r = c.has_reward_noise; a.autoreset = c.autoreset;
    a.image = c.image; a.H = c.img_h; a.W = c.img_w; a.C = c.img_c; a.pad = c.img_pad; a.has_shift = c.img_has_shift;
    a.sh_quant = c.img_sh_quant > 0 ? c.img_sh_quant : 1;
    a.philox = c.rng_mode == MDPP_RNG_PHILOX;
    a.p_noise = c.transition_noise; a.r_noise = c.reward_noise; a.scale = c.reward_scale; a.shift = c.reward_shift;
    a.term = c.term_state_reward;
    a.philox_seed = c.philox_seed; a.tick = h->tick; a.action_tick = h->action_tick; a.env_id_offset = c.env_id_offset;
    a.rng_s = (ulonglong2 *)h->d_rng_s; a.rng_inc = (ulonglong2 *)h->d_rng_inc; a.half = (uint2 *)h->d_half;
    a.ring = (double *)h->d_ring; a.head = (uint32_t *)h->d_head; a.noise_cdf = (const double *)h->d_noise_cdf;
    a.place = (short2 *)h->d_shift;
    return a;
}

int ensure_place(mdpp_post *h, size_t count) {
    if (count <= h->shift_cap) return MDPP_OK;
    if (h->d_shift) (void)hipFree(h->d_shift);
    h->d_shift = nullptr; h->shift_cap = 0;
    PHIP(h, hipMalloc(&h->d_shift, count * sizeof(short2)));
    h->shift_cap = count;
    return MDPP_OK;
}

int launch_post_image(mdpp_post *h, const PostArgs &a, long M, const void *in, void *out, hipStream_t s) {
    if (h->d_xyc) {
        const size_t lds = ((size_t)a.W * ((((a.H * a.C + 3) / 4 + 2) | 1) * 4) + 15) & ~(size_t)15;
        static size_t allowed = 48 * 1024;
        if (lds > allowed) {
            (void)hipFuncSetAttribute((const void *)k_post_image_lds, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            allowed = lds;
        }
        const long resident = (long)h->num_cus * (lds > 40 * 1024 ? 2 : lds > 24 * 1024 ? 4 : 6);
        hipLaunchKernelGGL(k_post_image_lds, dim3((unsigned)(M < resident ? M : resident)), dim3(kBlock), lds, s, a, M, a.place,
                           (const uint32_t *)h->d_xyc, (const uint8_t *)in, (uint8_t *)out);
    } else {
        const long dwords = (long)(a.W + 2 * a.pad) * (a.H + 2 * a.pad) * a.C / 4;
        const long blocks = (M * dwords + kBlock - 1) / kBlock;
        hipLaunchKernelGGL(k_post_image, dim3((unsigned)(blocks < 65536 ? blocks : 65536)), dim3(kBlock), 0, s, a, M, a.place,
                           (const uint8_t *)in, (uint8_t *)out);
    }
    return MDPP_OK;
}

int post_ready(mdpp_post *h, const char *what) {
    if (h->cfg.rng_mode == MDPP_RNG_NUMPY_PCG64 && !h->seeded) return pfail(h, MDPP_ESTATE, std::string(what) + ": stream not seeded");
    return MDPP_OK;
}

} // namespace

extern "C" const char *mdpp_post_last_error(const mdpp_post *h) { return h ? h->err.c_str() : g_post_create_err.c_str(); }

extern "C" void mdpp_post_destroy(mdpp_post *h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    for (void *p : {h->d_rng_s, h->d_rng_inc, h->d_half, h->d_ring, h->d_head, h->d_noise_cdf, h->d_shift, h->d_xyc})
        if (p) (void)hipFree(p);
    delete h;
}

extern "C" int mdpp_post_create(const mdpp_post_config *cfg, int device, mdpp_post **out) {
    if (!cfg || !out) return pfail(nullptr, MDPP_EINVAL, "mdpp_post_create: null argument");
    if (cfg->abi_version != MDPP_ABI_VERSION) return pfail(nullptr, MDPP_EINVAL, "mdpp_post_create: abi_version mismatch");
    if (cfg->num_envs <= 0) return pfail(nullptr, MDPP_EINVAL, "mdpp_post_create: num_envs <= 0");
    if (cfg->delay < 0 || cfg->delay > 128)
        return pfail(nullptr, MDPP_EUNSUPPORTED, "mdpp_post_create: need 0 <= delay <= 128 (the flush on done sums the buffer like np.sum, blocked up to 128)");
    if (cfg->rng_mode != MDPP_RNG_NUMPY_PCG64 && cfg->rng_mode != MDPP_RNG_PHILOX)
        return pfail(nullptr, MDPP_EINVAL, "mdpp_post_create: unknown rng_mode");
    if (cfg->continuous && (cfg->obs_dim < 1 || cfg->image))
        return pfail(nullptr, MDPP_EINVAL, "mdpp_post_create: continuous needs obs_dim >= 1 and no image transforms (:135-138)");
    if (!cfg->continuous && cfg->has_transition_noise &&
        (cfg->n_actions < 2 || !(cfg->transition_noise >= 0.0 && cfg->transition_noise <= 1.0)))
        return pfail(nullptr, MDPP_EINVAL, "mdpp_post_create: discrete transition_noise must be in [0, 1] and n_actions >= 2 (:105-109)");
    if (cfg->image) {
        const long ob = (long)(cfg->img_w + 2 * cfg->img_pad) * (cfg->img_h + 2 * cfg->img_pad) * cfg->img_c;
        if (cfg->img_h < 2 || cfg->img_h != cfg->img_w || cfg->img_h % 2 || cfg->img_c < 1 || cfg->img_pad < 0 ||
            cfg->img_h + 2 * cfg->img_pad > 16384 || ob % 4)
            return pfail(nullptr, MDPP_EUNSUPPORTED, "mdpp_post_create: images must be square (:531) with an even side, and the "
                                                     "padded canvas a multiple of 4 bytes");
        if (cfg->img_has_shift && cfg->img_pad < 1)
            return pfail(nullptr, MDPP_EINVAL, "mdpp_post_create: shift needs image_padding >= 1");
    }
    hipError_t e = hipSetDevice(device);
    if (e != hipSuccess) return pfail(nullptr, MDPP_EHIP, std::string("hipSetDevice: ") + hipGetErrorString(e));
    mdpp_post *h = new mdpp_post();
    h->cfg = *cfg; h->device = device; h->tick = 0; h->reset_tick = 0; h->action_tick = 0; h->seeded = false;
    h->d_rng_s = h->d_rng_inc = h->d_half = h->d_ring = h->d_head = h->d_noise_cdf = h->d_shift = h->d_xyc = nullptr;
    h->num_cus = 256;
    { int v = 0; if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && v > 0) h->num_cus = v; }
    h->shift_cap = 0;
    const size_t N = (size_t)cfg->num_envs;
    auto alloc0 = [&](void **p, size_t bytes) {
        if (hipMalloc(p, bytes ? bytes : 16) != hipSuccess) return false;
        return hipMemset(*p, 0, bytes ? bytes : 16) == hipSuccess;
    };
    bool ok = true;
    if (cfg->rng_mode == MDPP_RNG_NUMPY_PCG64)
        ok = alloc0(&h->d_rng_s, N * 16) && alloc0(&h->d_rng_inc, N * 16) && alloc0(&h->d_half, N * 8);
    ok = ok && alloc0(&h->d_ring, (size_t)cfg->delay * N * 8) && alloc0(&h->d_head, N * 4);
    if (ok && !cfg->continuous && cfg->has_transition_noise && cfg->transition_noise != 0.0) {
        // row a = cumsum(probs) / cumsum(probs)[-1] of probs = noise / (n - 1), probs[a] = 1 - noise (:356-361), as numpy's choice forms it
        const int n = cfg->n_actions;
        std::vector<double> cdf((size_t)n * n);
        for (int a = 0; a < n; a++) {
            double acc = 0.0;
            for (int j = 0; j < n; j++) {
                const double p = (j == a) ? 1 - cfg->transition_noise : 1.0 * cfg->transition_noise / (double)(n - 1);
                acc += p; cdf[(size_t)a * n + j] = acc;
            }
            const double last = cdf[(size_t)a * n + n - 1];
            for (int j = 0; j < n; j++) cdf[(size_t)a * n + j] /= last;
        }
        ok = hipMalloc(&h->d_noise_cdf, cdf.size() * 8) == hipSuccess &&
             hipMemcpy(h->d_noise_cdf, cdf.data(), cdf.size() * 8, hipMemcpyHostToDevice) == hipSuccess;
    }
    if (ok && cfg->image && (cfg->img_w * cfg->img_c) % 4 == 0 && ((cfg->img_h + 2 * cfg->img_pad) * cfg->img_c) % 4 == 0 &&
        (size_t)cfg->img_w * ((((cfg->img_h * cfg->img_c + 3) / 4 + 2) | 1) * 4) <= 60 * 1024 &&
        (cfg->img_h + 2 * cfg->img_pad) * cfg->img_c < 65536 && cfg->img_w + 2 * cfg->img_pad < 16384 && cfg->img_c <= 16) {
        // where every canvas dword starts: (x << 16 | byte within the canvas row of th * C bytes), canvas [x][y][c].
        // (a dword may run over into the next row x + 1 only if th * C % 4 != 0: excluded below)
        const int tw = cfg->img_w + 2 * cfg->img_pad, th = cfg->img_h + 2 * cfg->img_pad, C = cfg->img_c;
        const size_t ndw = (size_t)tw * th * C / 4, sdw = (size_t)cfg->img_h * cfg->img_w * C / 4;
        std::vector<uint32_t> xyc(ndw + sdw);
        for (size_t q = 0; q < ndw; q++) {
            const size_t b = 4 * q;
            xyc[q] = ((uint32_t)(b / ((size_t)th * C)) << 16) | (uint32_t)(b % ((size_t)th * C));
        }
        // ... followed by where every SOURCE dword starts: (sx << 18 | sy << 4 | c), source [sy][sx][c]
        for (size_t q = 0; q < sdw; q++) {
            const size_t b = 4 * q, sy = b / ((size_t)cfg->img_w * C), rb = b % ((size_t)cfg->img_w * C);
            xyc[ndw + q] = ((uint32_t)(rb / C) << 18) | ((uint32_t)sy << 4) | (uint32_t)(rb % C);
        }
        ok = hipMalloc(&h->d_xyc, xyc.size() * 4) == hipSuccess &&
             hipMemcpy(h->d_xyc, xyc.data(), xyc.size() * 4, hipMemcpyHostToDevice) == hipSuccess;
    }
    if (!ok) { g_post_create_err = "mdpp_post_create: device allocation failed"; mdpp_post_destroy(h); return MDPP_ENOMEM; }
    *out = h;
    return MDPP_OK;
}

extern "C" int mdpp_post_seed_streams(mdpp_post *h, const uint64_t *words) {
    if (!h || !words) return MDPP_EINVAL;
    if (h->cfg.rng_mode != MDPP_RNG_NUMPY_PCG64) return pfail(h, MDPP_ESTATE, "post_seed_streams: handle is in Philox mode");
    PHIP(h, hipSetDevice(h->device));
    const size_t N = (size_t)h->cfg.num_envs;
    std::vector<uint64_t> st(2 * N), inc(2 * N);
    std::vector<uint32_t> half(2 * N);
    for (size_t i = 0; i < N; i++) {
        st[2 * i] = words[6 * i]; st[2 * i + 1] = words[6 * i + 1];
        inc[2 * i] = words[6 * i + 2]; inc[2 * i + 1] = words[6 * i + 3];
        half[2 * i] = (uint32_t)words[6 * i + 4]; half[2 * i + 1] = (uint32_t)words[6 * i + 5];
    }
    PHIP(h, hipDeviceSynchronize());
    PHIP(h, hipMemcpy(h->d_rng_s, st.data(), N * 16, hipMemcpyHostToDevice));
    PHIP(h, hipMemcpy(h->d_rng_inc, inc.data(), N * 16, hipMemcpyHostToDevice));
    PHIP(h, hipMemcpy(h->d_half, half.data(), N * 8, hipMemcpyHostToDevice));
    h->seeded = true;
    return MDPP_OK;
}

extern "C" int mdpp_post_get_streams(mdpp_post *h, uint64_t *words) {
    if (!h || !words) return MDPP_EINVAL;
    if (h->cfg.rng_mode != MDPP_RNG_NUMPY_PCG64) return pfail(h, MDPP_ESTATE, "post_get_streams: handle is in Philox mode");
    PHIP(h, hipSetDevice(h->device));
    PHIP(h, hipDeviceSynchronize());
    const size_t N = (size_t)h->cfg.num_envs;
    std::vector<uint64_t> st(2 * N), inc(2 * N);
    std::vector<uint32_t> half(2 * N);
    PHIP(h, hipMemcpy(st.data(), h->d_rng_s, N * 16, hipMemcpyDeviceToHost));
    PHIP(h, hipMemcpy(inc.data(), h->d_rng_inc, N * 16, hipMemcpyDeviceToHost));
    PHIP(h, hipMemcpy(half.data(), h->d_half, N * 8, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < N; i++) {
        words[6 * i] = st[2 * i]; words[6 * i + 1] = st[2 * i + 1];
        words[6 * i + 2] = inc[2 * i]; words[6 * i + 3] = inc[2 * i + 1];
        words[6 * i + 4] = half[2 * i]; words[6 * i + 5] = half[2 * i + 1];
    }
    return MDPP_OK;
}

extern "C" int mdpp_post_get_reward_buffer(mdpp_post *h, double *ring_host) {
    if (!h || !ring_host) return MDPP_EINVAL;
    PHIP(h, hipSetDevice(h->device));
    PHIP(h, hipDeviceSynchronize());
    const size_t N = (size_t)h->cfg.num_envs, d = (size_t)h->cfg.delay;
    if (d == 0) return MDPP_OK;
    std::vector<double> rg(d * N);
    std::vector<uint32_t> head(N);
    PHIP(h, hipMemcpy(rg.data(), h->d_ring, rg.size() * 8, hipMemcpyDeviceToHost));
    PHIP(h, hipMemcpy(head.data(), h->d_head, N * 4, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < N; i++)
        for (size_t j = 0; j < d; j++) ring_host[i * d + j] = rg[((head[i] + j) % d) * N + i];   // [0] pays out next
    return MDPP_OK;
}

extern "C" int mdpp_post_reset(mdpp_post *h, const uint8_t *mask_dev, const void *obs_in_dev, void *obs_out_dev, void *stream) {
    if (!h) return MDPP_EINVAL;
    int rc = post_ready(h, "mdpp_post_reset");
    if (rc) return rc;
    if (h->cfg.image && (!obs_in_dev || !obs_out_dev)) return pfail(h, MDPP_EINVAL, "mdpp_post_reset: image handles need the first observations");
    PHIP(h, hipSetDevice(h->device));
    hipStream_t s = (hipStream_t)stream;
    if (h->cfg.image) { rc = ensure_place(h, (size_t)h->cfg.num_envs); if (rc) return rc; }
    PostArgs a = make_args(h);
    const int grid = (a.N + kBlock - 1) / kBlock;
    if (a.philox) hipLaunchKernelGGL(k_post_reset<true>, dim3(grid), dim3(kBlock), 0, s, a, h->reset_tick, mask_dev);
    else hipLaunchKernelGGL(k_post_reset<false>, dim3(grid), dim3(kBlock), 0, s, a, h->reset_tick, mask_dev);
    if (h->cfg.image) launch_post_image(h, a, (long)a.N, obs_in_dev, obs_out_dev, s);
    PHIP(h, hipGetLastError());
    h->reset_tick += 1;
    return MDPP_OK;
}

extern "C" int mdpp_post_actions(mdpp_post *h, const int32_t *in_dev, int32_t *out_dev, void *stream) {
    if (!h || !in_dev || !out_dev) return MDPP_EINVAL;
    int rc = post_ready(h, "mdpp_post_actions");
    if (rc) return rc;
    if (h->cfg.continuous) return pfail(h, MDPP_EINVAL, "mdpp_post_actions: continuous actions pass through unchanged (:367-373 adds noise to observations)");
    PHIP(h, hipSetDevice(h->device));
    hipStream_t s = (hipStream_t)stream;
    const size_t N = (size_t)h->cfg.num_envs;
    // (Philox streams: the action noise is keyed by the number of mdpp_post_actions calls, not by the step counter --
    //  a block of K action rows processed before one fused step_n(K) gets K different draws per instance)
    if (!h->d_noise_cdf) {                                   // `if self.transition_noise:` false: identity
        if (in_dev != out_dev) PHIP(h, hipMemcpyAsync(out_dev, in_dev, N * 4, hipMemcpyDeviceToDevice, s));
        h->action_tick += 1;
        return MDPP_OK;
    }
    PostArgs a = make_args(h);
    h->action_tick += 1;
    const int grid = (a.N + kBlock - 1) / kBlock;
    if (a.philox) hipLaunchKernelGGL(k_post_actions<true>, dim3(grid), dim3(kBlock), 0, s, a, in_dev, out_dev);
    else hipLaunchKernelGGL(k_post_actions<false>, dim3(grid), dim3(kBlock), 0, s, a, in_dev, out_dev);
    PHIP(h, hipGetLastError());
    return MDPP_OK;
}

extern "C" int mdpp_post_step_n(mdpp_post *h, int K, const void *obs_in_dev, const double *reward_in_dev,
                                const uint8_t *done_dev, void *obs_out_dev, double *reward_out_dev, void *stream) {
    if (!h || K < 1 || !reward_in_dev || !done_dev || !reward_out_dev) return MDPP_EINVAL;
    int rc = post_ready(h, "mdpp_post_step");
    if (rc) return rc;
    if ((h->cfg.continuous || h->cfg.image) && (!obs_in_dev || !obs_out_dev))
        return pfail(h, MDPP_EINVAL, "mdpp_post_step: this handle transforms observations: obs_in / obs_out needed");
    PHIP(h, hipSetDevice(h->device));
    hipStream_t s = (hipStream_t)stream;
    if (h->cfg.image) { rc = ensure_place(h, (size_t)K * h->cfg.num_envs); if (rc) return rc; }
    PostArgs a = make_args(h);
    const int grid = (a.N + kBlock - 1) / kBlock;
    const int ring = a.delay >= 1 && a.delay <= kPostRegDelay ? 2 : (a.delay >= 1 && a.delay <= kPostLdsDelay ? 1 : 0);
#define MDPP_POST_LAUNCH(PH, LR, DC) hipLaunchKernelGGL((k_post_step<PH, LR, DC>), dim3(grid), dim3(kBlock), 0, s, a, K, obs_in_dev, \
                                                        reward_in_dev, done_dev, obs_out_dev, reward_out_dev)
#define MDPP_POST_REG(PH)                                                                        \
    switch (a.delay) {                                                                           \
    case 1: MDPP_POST_LAUNCH(PH, 2, 1); break; case 2: MDPP_POST_LAUNCH(PH, 2, 2); break;        \
    case 3: MDPP_POST_LAUNCH(PH, 2, 3); break; case 4: MDPP_POST_LAUNCH(PH, 2, 4); break;        \
    case 5: MDPP_POST_LAUNCH(PH, 2, 5); break; case 6: MDPP_POST_LAUNCH(PH, 2, 6); break;        \
    case 7: MDPP_POST_LAUNCH(PH, 2, 7); break; default: MDPP_POST_LAUNCH(PH, 2, 8); break;       \
    }
    if (a.philox) { if (ring == 2) { MDPP_POST_REG(true); } else if (ring == 1) MDPP_POST_LAUNCH(true, 1, 0); else MDPP_POST_LAUNCH(true, 0, 0); }
    else { if (ring == 2) { MDPP_POST_REG(false); } else if (ring == 1) MDPP_POST_LAUNCH(false, 1, 0); else MDPP_POST_LAUNCH(false, 0, 0); }
#undef MDPP_POST_REG
#undef MDPP_POST_LAUNCH
    if (h->cfg.image) launch_post_image(h, a, (long)K * a.N, obs_in_dev, obs_out_dev, s);
    PHIP(h, hipGetLastError());
    h->tick += (uint64_t)K;
    return MDPP_OK;
}

extern "C" int mdpp_post_step(mdpp_post *h, const void *obs_in_dev, const double *reward_in_dev, const uint8_t *done_dev,
                              void *obs_out_dev, double *reward_out_dev, void *stream) {
    return mdpp_post_step_n(h, 1, obs_in_dev, reward_in_dev, done_dev, obs_out_dev, reward_out_dev, stream);
}

// ---- episode statistics (episode_reward_mean / episode_len_mean of RLlib's result dict, config_processor.py:275-407) ----
// One lane per instance walks the K rows of a [K][N] block of per-step rewards and end flags: running return and
// length per instance, and over the episodes that ended inside the block the sum of returns, the sum of lengths
// and their number.  Two launches, no atomics: per-block partial sums in a fixed tree order, then one lane adds
// the partials in block order -- the result does not depend on scheduling.
namespace {
constexpr int kStatsPre = 8;
template <bool F64>
__global__ __launch_bounds__(kBlock) void k_episode_stats(int K, int N, const void *__restrict__ reward,
                                                          const uint8_t *__restrict__ e1, const uint8_t *__restrict__ e2,
                                                          double *__restrict__ ret, long long *__restrict__ len,
                                                          double *__restrict__ part_ret, long long *__restrict__ part_len,
                                                          long long *__restrict__ part_cnt) {
    __shared__ double s_r[kBlock];
    __shared__ long long s_l[kBlock], s_c[kBlock];
    const int i = blockIdx.x * kBlock + threadIdx.x;
    double sret = 0.0;
    long long slen = 0, cnt = 0;
    if (i < N) {
        double r = ret[i];
        long long l = len[i];
        auto rew = [&](int k) -> double {
            const size_t o = (size_t)k * N + i;
            return F64 ? ((const double *)reward)[o] : (double)((const float *)reward)[o];
        };
        auto fl = [&](int k) -> bool {
            const size_t o = (size_t)k * N + i;
            return (e1[o] | (e2 ? e2[o] : (uint8_t)0)) != 0;
        };
        int k = 0;
        for (; k + kStatsPre <= K; k += kStatsPre) {            // kStatsPre rows in flight per lane
            double rv[kStatsPre];
            bool ev[kStatsPre];
#pragma unroll
            for (int u = 0; u < kStatsPre; u++) { rv[u] = rew(k + u); ev[u] = fl(k + u); }
#pragma unroll
            for (int u = 0; u < kStatsPre; u++) {
                r += rv[u]; l += 1;
                sret += ev[u] ? r : 0.0; slen += ev[u] ? l : 0; cnt += ev[u] ? 1 : 0;
                r = ev[u] ? 0.0 : r; l = ev[u] ? 0 : l;
            }
        }
        for (; k < K; k++) {
            const bool e = fl(k);
            r += rew(k); l += 1;
            sret += e ? r : 0.0; slen += e ? l : 0; cnt += e ? 1 : 0;
            r = e ? 0.0 : r; l = e ? 0 : l;
        }
        ret[i] = r; len[i] = l;
    }
    s_r[threadIdx.x] = sret; s_l[threadIdx.x] = slen; s_c[threadIdx.x] = cnt;
    __syncthreads();
    for (int st = kBlock / 2; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) {
            s_r[threadIdx.x] += s_r[threadIdx.x + st]; s_l[threadIdx.x] += s_l[threadIdx.x + st]; s_c[threadIdx.x] += s_c[threadIdx.x + st];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) { part_ret[blockIdx.x] = s_r[0]; part_len[blockIdx.x] = s_l[0]; part_cnt[blockIdx.x] = s_c[0]; }
}

__global__ void k_episode_stats_final(int nblocks, const double *part_ret, const long long *part_len, const long long *part_cnt,
                                      double *sum_ret, long long *sum_len, long long *count) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double r = *sum_ret;
    long long l = *sum_len, c = *count;
    for (int b = 0; b < nblocks; b++) { r += part_ret[b]; l += part_len[b]; c += part_cnt[b]; }
    *sum_ret = r; *sum_len = l; *count = c;
}
} // namespace

extern "C" int mdpp_episode_stats(int32_t K, int32_t N, const void *reward_dev, int32_t reward_is_f64, const uint8_t *ended_dev,
                                  const uint8_t *ended2_dev, double *ret_dev, int64_t *len_dev, double *sum_ret_dev,
                                  int64_t *sum_len_dev, int64_t *count_dev, void *scratch_dev, void *stream) {
    if (K < 1 || N < 1 || !reward_dev || !ended_dev || !ret_dev || !len_dev || !sum_ret_dev || !sum_len_dev || !count_dev || !scratch_dev)
        return MDPP_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const int grid = (N + kBlock - 1) / kBlock;
    double *part_ret = (double *)scratch_dev;
    long long *part_len = (long long *)(part_ret + grid), *part_cnt = part_len + grid;
    if (reward_is_f64)
        hipLaunchKernelGGL(k_episode_stats<true>, dim3(grid), dim3(kBlock), 0, s, K, N, reward_dev, ended_dev, ended2_dev, ret_dev,
                           (long long *)len_dev, part_ret, part_len, part_cnt);
    else
        hipLaunchKernelGGL(k_episode_stats<false>, dim3(grid), dim3(kBlock), 0, s, K, N, reward_dev, ended_dev, ended2_dev, ret_dev,
                           (long long *)len_dev, part_ret, part_len, part_cnt);
    hipLaunchKernelGGL(k_episode_stats_final, dim3(1), dim3(64), 0, s, grid, part_ret, part_len, part_cnt, sum_ret_dev,
                       (long long *)sum_len_dev, (long long *)count_dev);
    return hipGetLastError() == hipSuccess ? MDPP_OK : MDPP_EHIP;
}
