// Peer-copy gather of the observation shards (include/mdpp.h "mdpp_peer_*").
//
// The path's collective is ONE all-gather of the current observation shard per step / rollout.  RCCL's all-gather is a
// kernel, and the rollout kernels hold every compute unit of the chip (256 workgroups, all VGPRs of every SIMD): an RCCL
// gather can only run BETWEEN two launches.  This is the same exchange without a kernel: every rank owns a buffer
// [slots][world][shard_bytes] (+ one 64-bit flag per slot and rank), exports it with hipIpcGetMemHandle, opens the other
// ranks' buffers, and after a launch copies its shard into row `rank` of every rank's buffer with hipMemcpyAsync on a
// side stream (device-to-device copies between GPUs run on the SDMA engines; xGMI is point to point, so the world - 1
// copies go out on different links), followed by its sequence number into the flag.  A consumer makes its stream wait
// for the flags with a one-wave kernel that polls device memory (bounded; a timeout is reported, never a hang).
// No torch types, no RCCL: plain pointers and sizes; the handles travel through whatever channel the ranks already
// have (torch.distributed's all_gather_object in mdp_playground_amd/dist.py).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include <string>
#include <vector>

#include "../../include/mdpp.h"

struct mdpp_peer {
    int device, world, rank, slots;
    size_t shard;
    uint8_t *buf;                  // [slots][world][shard] then uint64 flags [slots][world]
    size_t flags_off;
    std::vector<uint8_t *> peer;   // peer[r] = rank r's buffer as mapped here (peer[rank] = buf)
    hipStream_t side;
    hipEvent_t ev;
    std::vector<hipEvent_t> ev_push;   // per slot: this rank's copies of the slot's latest push are done
    uint64_t *h_seq;               // pinned host ring [kSeqRing]: the sequence numbers in flight (sources of the flag copies --
                                   // a copy from host memory, not a kernel: a kernel could not start beside a rollout that holds every register)
    uint32_t *d_status;            // != 0: a wait ran into its bound
    bool opened, finegrained;
    std::string err;
};

namespace {
constexpr int kSeqRing = 256;
// lane r polls rank r's flag of the slot (system scope: the writers are other devices' copy engines)
__global__ void k_peer_wait(const uint64_t *flags, int world, uint64_t seq, uint32_t *status, uint32_t max_spins) {
    const int r = threadIdx.x;
    if (r >= world) return;
    uint32_t spins = 0;
    while (__hip_atomic_load(&flags[r], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < seq) {
        __builtin_amdgcn_s_sleep(8);
        if (++spins > max_spins) { atomicOr(status, 1u << (r & 31)); break; }
    }
}
int pfail(mdpp_peer *p, int code, const char *what, hipError_t e = hipSuccess) {
    if (p) p->err = std::string(what) + (e != hipSuccess ? std::string(": ") + hipGetErrorString(e) : std::string());
    return code;
}
}  // namespace

#define PCHK(p, call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return pfail(p, MDPP_EHIP, #call, e_); } while (0)

extern "C" int mdpp_peer_create(int device, int world, int rank, size_t shard_bytes, int slots, mdpp_peer **out) {
    if (!out || world < 1 || world > 64 || rank < 0 || rank >= world || shard_bytes == 0 || slots < 1) return MDPP_EINVAL;
    mdpp_peer *p = new mdpp_peer();
    p->device = device; p->world = world; p->rank = rank; p->slots = slots; p->shard = shard_bytes;
    p->buf = nullptr; p->side = nullptr; p->ev = nullptr; p->h_seq = nullptr; p->d_status = nullptr;
    p->opened = false; p->finegrained = false;
    p->peer.assign((size_t)world, nullptr);
    p->ev_push.assign((size_t)slots, nullptr);
    if (hipSetDevice(device) != hipSuccess) { delete p; return MDPP_EHIP; }
    p->flags_off = ((size_t)slots * world * shard_bytes + 255) & ~(size_t)255;
    const size_t bytes = p->flags_off + (size_t)slots * world * sizeof(uint64_t);
    // Fine-grained memory (coherent with the other devices' writes without cache maintenance).  With several ranks there
    // is NO fallback to coarse-grained memory: the buffer receives other devices' copy-engine writes, is polled by
    // k_peer_wait and read by consumers through the local L2, which is not guaranteed coherent with them -- stale flags and
    // stale rows, silently.  One rank (nothing remote ever writes the buffer) may take plain device memory.
    if (hipExtMallocWithFlags((void **)&p->buf, bytes, hipDeviceMallocFinegrained) == hipSuccess) p->finegrained = true;
    else {
        (void)hipGetLastError();
        if (world > 1) { delete p; return MDPP_EUNSUPPORTED; }
        if (hipMalloc((void **)&p->buf, bytes) != hipSuccess) { delete p; return MDPP_EHIP; }
    }
    bool ok = hipMemset(p->buf, 0, bytes) == hipSuccess &&
              hipStreamCreateWithFlags(&p->side, hipStreamNonBlocking) == hipSuccess &&
              hipEventCreateWithFlags(&p->ev, hipEventDisableTiming) == hipSuccess &&
              hipHostMalloc((void **)&p->h_seq, (size_t)kSeqRing * sizeof(uint64_t), hipHostMallocDefault) == hipSuccess &&
              hipMalloc((void **)&p->d_status, sizeof(uint32_t)) == hipSuccess &&
              hipMemset(p->d_status, 0, sizeof(uint32_t)) == hipSuccess && hipDeviceSynchronize() == hipSuccess;
    for (int k = 0; ok && k < slots; k++) ok = hipEventCreateWithFlags(&p->ev_push[(size_t)k], hipEventDisableTiming) == hipSuccess;
    if (!ok) { mdpp_peer_destroy(p); return MDPP_EHIP; }
    p->peer[(size_t)rank] = p->buf;
    *out = p;
    return MDPP_OK;
}

extern "C" int mdpp_peer_handle(mdpp_peer *p, void *handle_out) {
    if (!p || !handle_out) return MDPP_EINVAL;
    static_assert(sizeof(hipIpcMemHandle_t) == MDPP_PEER_HANDLE_BYTES, "hipIpcMemHandle_t is 64 bytes");
    hipIpcMemHandle_t h;
    PCHK(p, hipSetDevice(p->device));
    PCHK(p, hipIpcGetMemHandle(&h, p->buf));
    memcpy(handle_out, &h, sizeof h);
    return MDPP_OK;
}

extern "C" int mdpp_peer_open(mdpp_peer *p, const void *handles) {
    if (!p || !handles) return MDPP_EINVAL;
    if (p->opened) return pfail(p, MDPP_EINVAL, "mdpp_peer_open: already open");
    PCHK(p, hipSetDevice(p->device));
    for (int r = 0; r < p->world; r++) {
        if (r == p->rank) continue;
        hipIpcMemHandle_t h;
        memcpy(&h, (const uint8_t *)handles + (size_t)r * MDPP_PEER_HANDLE_BYTES, sizeof h);
        void *q = nullptr;
        PCHK(p, hipIpcOpenMemHandle(&q, h, hipIpcMemLazyEnablePeerAccess));
        p->peer[(size_t)r] = (uint8_t *)q;
    }
    p->opened = true;
    return MDPP_OK;
}

extern "C" int mdpp_peer_push(mdpp_peer *p, int slot, const void *shard_dev, uint64_t seq, void *stream) {
    if (!p || !shard_dev || slot < 0 || slot >= p->slots) return MDPP_EINVAL;
    if (!p->opened && p->world > 1) return pfail(p, MDPP_EINVAL, "mdpp_peer_push: mdpp_peer_open first");
    PCHK(p, hipSetDevice(p->device));
    // the side stream starts behind what `stream` has enqueued so far (the rollout that fills the shard)
    PCHK(p, hipEventRecord(p->ev, (hipStream_t)stream));
    PCHK(p, hipStreamWaitEvent(p->side, p->ev, 0));
    const size_t row = ((size_t)slot * p->world + (size_t)p->rank) * p->shard;
    for (int k = 0; k < p->world; k++) {
        const int r = (p->rank + 1 + k) % p->world;                   // (start with the neighbour: the ranks' copies fan out over different links)
        PCHK(p, hipMemcpyAsync(p->peer[(size_t)r] + row, shard_dev, p->shard, hipMemcpyDeviceToDevice, p->side));
    }
    // (the pinned word must stay stable until its flag copies have run: at most kSeqRing pushes in flight -- enforced by
    //  draining the side stream every kSeqRing / 2 pushes, which an open loop of launches otherwise never does)
    if (seq % (uint64_t)(kSeqRing / 2) == 0) PCHK(p, hipStreamSynchronize(p->side));
    uint64_t *src = p->h_seq + (seq % (uint64_t)kSeqRing);
    *src = seq;
    const size_t fo = p->flags_off + ((size_t)slot * p->world + (size_t)p->rank) * sizeof(uint64_t);
    for (int k = 0; k < p->world; k++) {                               // (same stream: a flag lands behind its data)
        const int r = (p->rank + 1 + k) % p->world;
        PCHK(p, hipMemcpyAsync(p->peer[(size_t)r] + fo, src, sizeof(uint64_t), hipMemcpyDefault, p->side));
    }
    PCHK(p, hipEventRecord(p->ev_push[(size_t)slot], p->side));
    return MDPP_OK;
}

// `stream` waits (an event, no kernel) until THIS rank's copies of the slot's latest push have left: the shard buffer may
// be overwritten after that.
extern "C" int mdpp_peer_fence(mdpp_peer *p, int slot, void *stream) {
    if (!p || slot < 0 || slot >= p->slots) return MDPP_EINVAL;
    PCHK(p, hipSetDevice(p->device));
    PCHK(p, hipStreamWaitEvent((hipStream_t)stream, p->ev_push[(size_t)slot], 0));
    return MDPP_OK;
}

extern "C" int mdpp_peer_wait(mdpp_peer *p, int slot, uint64_t seq, void *stream) {
    if (!p || slot < 0 || slot >= p->slots) return MDPP_EINVAL;
    PCHK(p, hipSetDevice(p->device));
    const uint64_t *flags = (const uint64_t *)(p->buf + p->flags_off) + (size_t)slot * p->world;
    hipLaunchKernelGGL(k_peer_wait, dim3(1), dim3(64), 0, (hipStream_t)stream, flags, p->world, seq, p->d_status, 1u << 22);
    PCHK(p, hipGetLastError());
    return MDPP_OK;
}

extern "C" void *mdpp_peer_buffer(mdpp_peer *p, int slot) {
    if (!p || slot < 0 || slot >= p->slots) return nullptr;
    return p->buf + (size_t)slot * p->world * p->shard;
}

extern "C" int mdpp_peer_status(mdpp_peer *p, uint32_t *status_out, int *finegrained_out) {
    if (!p || !status_out) return MDPP_EINVAL;
    PCHK(p, hipSetDevice(p->device));
    PCHK(p, hipMemcpy(status_out, p->d_status, sizeof(uint32_t), hipMemcpyDeviceToHost));
    if (finegrained_out) *finegrained_out = p->finegrained ? 1 : 0;
    return MDPP_OK;
}

extern "C" const char *mdpp_peer_last_error(mdpp_peer *p) { return p ? p->err.c_str() : "null handle"; }

extern "C" int mdpp_peer_destroy(mdpp_peer *p) {
    if (!p) return MDPP_EINVAL;
    (void)hipSetDevice(p->device);
    (void)hipDeviceSynchronize();
    for (int r = 0; r < p->world; r++)
        if (r != p->rank && p->peer[(size_t)r]) (void)hipIpcCloseMemHandle(p->peer[(size_t)r]);
    if (p->side) (void)hipStreamDestroy(p->side);
    if (p->ev) (void)hipEventDestroy(p->ev);
    for (hipEvent_t e : p->ev_push) if (e) (void)hipEventDestroy(e);
    if (p->h_seq) (void)hipHostFree(p->h_seq);
    if (p->d_status) (void)hipFree(p->d_status);
    if (p->buf) (void)hipFree(p->buf);
    delete p;
    return MDPP_OK;
}
