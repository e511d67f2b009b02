// Cost of one Philox4x32-10 block and of one packed Box-Muller evaluation (4 normals) per wave64, in SIMD cycles:
//   hipcc --offload-arch=gfx950 -O3 -w -I include -I mdp_playground_amd/csrc tools/bench_philox.hip -o build/bench_philox && build/bench_philox
// (grid = 256 CUs x 4 SIMDs x W waves; every lane runs R dependent rounds; cycles = elapsed x 2.4 GHz / (R x W))
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include "mdpp.h"
#include "mdpp_rng.hpp"
using namespace mdpp;

template <int MODE>
__global__ __launch_bounds__(256) void k(uint32_t *out, int R, uint32_t seed) {
    uint32_t c0 = threadIdx.x + blockIdx.x * 256, acc = 0;
    float fz = 0.0f;
    for (int r = 0; r < R; r++) {
        uint32_t o[4];
        if (MODE == 0 || MODE == 2) philox4x32_10(c0, 1u, (uint32_t)r, 2u, seed, 77u, o);
        else { o[0] = c0 * 3u + r; o[1] = c0 ^ r; o[2] = c0 + 5u * r; o[3] = c0 - r; }
        if (MODE >= 1) {
            float z0, z1, z2, z3;
            philox_box_muller2(o, z0, z1, z2, z3);
            fz += z0 + z1 + z2 + z3;
        }
        acc ^= o[0] ^ o[1] ^ o[2] ^ o[3];
        c0 += acc & 1u;
    }
    out[threadIdx.x + blockIdx.x * 256] = acc + __float_as_uint(fz);
}

template <int MODE>
static void run(const char *what, int wavesPerSimd) {
    const int R = 4096, blocks = 256 * wavesPerSimd;       // 256 threads = 4 waves = one per SIMD of a CU
    uint32_t *d;
    hipMalloc(&d, (size_t)blocks * 256 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<blocks, 256>>>(d, R, 1u);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE><<<blocks, 256>>>(d, R, 2u);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%-28s waves/SIMD %d: %7.1f us, %6.1f SIMD cycles per wave-iteration at 2.4 GHz\n", what, wavesPerSimd, ms * 1e3,
           ms * 1e-3 * 2.4e9 / ((double)R * wavesPerSimd));
    hipFree(d);
}

int main() {
    for (int w : {1, 2, 4}) {
        run<0>("philox block", w);
        run<1>("box-muller x4 (packed)", w);
        run<2>("block + box-muller x4", w);
    }
    return 0;
}
