// coop_rows_check.hip -- standalone check of f32_rows_coop (search_kernels.hip) against a host loop in
// FullVec's order (vectors/src/full.rs:23-29): random tables, random wanted-lane patterns (0..64 lanes),
// every lane's sum compared bit for bit.  Diagnostic, not part of the library.
//   hipcc -O3 -std=c++17 -ffp-contract=off --offload-arch=gfx950 -Ihnsw_rs_amd/csrc -Iinclude \
//       scripts/micro/coop_rows_check.hip -o scripts/micro/bin/coop_rows_check
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "search_common.h"
namespace hx {


#include "coop_rows.inc"
template <int DS>
__global__ void __launch_bounds__(64) tk(const uint8_t *rows, const uint32_t *ids, const float *q, float *out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *yq = reinterpret_cast<float *>(smem);
    uint32_t *ids_s = reinterpret_cast<uint32_t *>(smem + DS * 4);
    unsigned char *img = smem + DS * 4 + 256;
    const int lane = threadIdx.x;
    for (int e = lane; e < DS; e += 64) yq[e] = q[e];
    wave_fence();
    const uint32_t id = ids[blockIdx.x * 64 + lane];
    const bool act = id != 0xFFFFFFFFu;
    const float s = f32_rows_coop<DS, HX_COOP_K>(rows, id, act, yq, ids_s, img, lane);
    out[blockIdx.x * 64 + lane] = s;
}
}  // namespace hx

template <int DS>
static int run(int nblocks) {
    const int N = 5000;
    std::vector<float> tab((size_t)N * DS), q(DS);
    srand(DS);
    for (auto &x : tab) x = (float)rand() / RAND_MAX - 0.5f;
    for (auto &x : q) x = (float)rand() / RAND_MAX - 0.5f;
    std::vector<uint32_t> ids((size_t)nblocks * 64);
    for (int b = 0; b < nblocks; b++) {
        const int density = b % 9;  // 0: none ... 8: all
        for (int l = 0; l < 64; l++) ids[b * 64 + l] = (rand() % 8 < density) ? (uint32_t)(rand() % N) : 0xFFFFFFFFu;
    }
    uint8_t *d_tab; uint32_t *d_ids; float *d_q, *d_out;
    hipMalloc(&d_tab, tab.size() * 4); hipMalloc(&d_ids, ids.size() * 4); hipMalloc(&d_q, DS * 4); hipMalloc(&d_out, ids.size() * 4);
    hipMemcpy(d_tab, tab.data(), tab.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(d_ids, ids.data(), ids.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(d_q, q.data(), DS * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(hx::tk<DS>, dim3(nblocks), dim3(64), DS * 4 + 256 + 4096, 0, d_tab, d_ids, d_q, d_out);
    std::vector<float> out(ids.size());
    if (hipMemcpy(out.data(), d_out, out.size() * 4, hipMemcpyDeviceToHost) != hipSuccess) { printf("HIP error\n"); return 1; }
    int bad = 0;
    for (size_t i = 0; i < ids.size(); i++) {
        float s = 0.0f;
        if (ids[i] != 0xFFFFFFFFu)
            for (int e = 0; e < DS; e++) { const float t = tab[(size_t)ids[i] * DS + e] - q[e]; s += t * t; }
        if (memcmp(&s, &out[i], 4) != 0) {
            if (bad < 8) printf("d=%d block %zu lane %zu id %u: got %.9g want %.9g\n", DS, i / 64, i % 64, ids[i], out[i], s);
            bad++;
        }
    }
    printf("d = %d: %d of %zu lane sums differ\n", DS, bad, ids.size());
    return bad != 0;
}

int main() {
    int rc = 0;
    rc |= run<128>(90);
    rc |= run<256>(90);
    rc |= run<768>(90);
    return rc;
}
