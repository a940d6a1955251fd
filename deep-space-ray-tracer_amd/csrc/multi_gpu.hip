// multi_gpu.hip -- one host process driving the GPUs of a node (include/dsrt.h: dsrt_multi_*).
//
// The reference is a single process calling gpu_render_scene once per frame on one device (src/main.cpp:310-431); it has no
// multi-GPU code at all.  This file is what lets that same single-process C host use every GPU of the node:
//
//   * dsrt_multi_render_frame   ONE frame sharded by interleaved screen tiles (tile g -> rank g mod N, include/dsrt.h
//     dsrt_shard_layout): every rank renders its tiles into a compact buffer on its own device and stream, then ONE RCCL gather
//     over xGMI (ncclGather, equal counts because every rank's buffer is padded to ceil(tiles / N) tiles) brings them to rank 0,
//     a small kernel restores image order there, and the image is copied to the caller's host buffer.
//   * dsrt_multi_render_sequence   MANY frames of one scene (the pose file): every rank renders its interleaved tiles of EVERY frame as
//     batch launches (dsrt_render_batch with a shard: up to 128 frames' tiles as one pool of work), one gather per launch for all its
//     frames, de-interleave on rank 0, images to pinned host memory.  In rng_mode 0 a pixel is a serial chain of spp samples whatever
//     the number of GPUs; in the pool the chains of all frames run under each other, which is what lets this split scale (DESIGN.md
//     section 5).
//
// All launches are asynchronous, so one host thread keeps N devices busy.  torch is not involved; torch.distributed (bench.py)
// is the one-process-per-GPU alternative over the same kernels and the same shard layout.
//
// Ranks that share a device (a one-GPU box: tests, rehearsals) cannot form an RCCL communicator -- RCCL refuses duplicate
// devices -- so for such a layout the gather is N device-to-device copies into the same receive layout; everything else (shard
// layout, padding, de-interleave, frame dealing) runs unchanged.  `dsrt_multi_uses_rccl` says which one a DsrtMulti got.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>
#include <set>
#include <string>
#include <vector>

#include "../../include/dsrt.h"
#include "../host/host_internal.hpp"

using dsrt::set_error;

namespace {

bool hip_ok(hipError_t e, const char* what) {
    if (e == hipSuccess) return true;
    set_error(std::string(what) + ": " + hipGetErrorString(e));
    return false;
}
bool nccl_ok(ncclResult_t r, const char* what) {
    if (r == ncclSuccess) return true;
    set_error(std::string(what) + ": " + ncclGetErrorString(r));
    return false;
}
#define HIP_TRY(expr) do { if (!hip_ok((expr), #expr)) return DSRT_ERR_HIP; } while (0)
#define NCCL_TRY(expr) do { if (!nccl_ok((expr), #expr)) return DSRT_ERR_COMM; } while (0)

struct Slot {                       // one frame in flight on one rank
    DsrtContext* ctx = nullptr;     // slot 0: the rank's own context; others: clones sharing its scene
    hipStream_t stream = nullptr;
    uint8_t* d_image = nullptr;     // W*H*3 (sequence mode)
    uint8_t* h_pinned = nullptr;
    size_t image_bytes = 0;
    int frame = -1;                 // frame index whose image is in h_pinned once the stream is idle
};

struct Rank {
    int device = 0;
    std::vector<Slot> slots;
    uint8_t* d_part = nullptr;      // compact tile buffer of the current sharded frame
    size_t part_bytes = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr, ev_sent = nullptr;   // render start / end; shared-device gather: this rank's copy has landed
};

}  // namespace

struct DsrtMulti {
    std::vector<Rank> ranks;
    std::vector<ncclComm_t> comms;  // empty when ranks share a device
    bool rccl = false;
    bool have_scene = false;
    uint8_t* d_gathered = nullptr;  // rank 0: N * padded bytes
    size_t gathered_bytes = 0;
    uint8_t* d_frame = nullptr;     // rank 0: W*H*3
    size_t frame_bytes = 0;
};

namespace {

int ensure(uint8_t** p, size_t* have, size_t need) {
    if (*have >= need) return DSRT_OK;
    if (*p) { (void)hipFree(*p); *p = nullptr; *have = 0; }
    HIP_TRY(hipMalloc((void**)p, need));
    *have = need;
    return DSRT_OK;
}

int ensure_slot_images(Slot& s, size_t bytes) {
    if (s.image_bytes >= bytes) return DSRT_OK;
    if (s.d_image) { (void)hipFree(s.d_image); s.d_image = nullptr; }
    if (s.h_pinned) { (void)hipHostFree(s.h_pinned); s.h_pinned = nullptr; }
    s.image_bytes = 0;
    HIP_TRY(hipMalloc((void**)&s.d_image, bytes));
    HIP_TRY(hipHostMalloc((void**)&s.h_pinned, bytes, hipHostMallocDefault));
    s.image_bytes = bytes;
    return DSRT_OK;
}

}  // namespace

extern "C" {

void dsrt_multi_destroy(DsrtMulti* m) {
    if (!m) return;
    for (size_t r = 0; r < m->ranks.size(); ++r) {
        Rank& k = m->ranks[r];
        (void)hipSetDevice(k.device);
        (void)hipDeviceSynchronize();
        for (Slot& s : k.slots) {
            if (s.d_image) (void)hipFree(s.d_image);
            if (s.h_pinned) (void)hipHostFree(s.h_pinned);
            if (s.stream) (void)hipStreamDestroy(s.stream);
            if (s.ctx) dsrt_ctx_destroy(s.ctx);
        }
        if (k.d_part) (void)hipFree(k.d_part);
        if (k.ev0) (void)hipEventDestroy(k.ev0);
        if (k.ev1) (void)hipEventDestroy(k.ev1);
        if (k.ev_sent) (void)hipEventDestroy(k.ev_sent);
    }
    if (!m->ranks.empty()) (void)hipSetDevice(m->ranks[0].device);
    if (m->d_gathered) (void)hipFree(m->d_gathered);
    if (m->d_frame) (void)hipFree(m->d_frame);
    for (ncclComm_t c : m->comms) (void)ncclCommDestroy(c);
    delete m;
}

int dsrt_multi_create(const int* devices, int n, int frames_in_flight, DsrtMulti** out) {
    if (!out) { set_error("dsrt_multi_create: null out"); return DSRT_ERR_INVALID; }
    *out = nullptr;
    if (!devices || n < 1 || n > 64 || frames_in_flight < 1 || frames_in_flight > 64) { set_error("dsrt_multi_create: bad argument"); return DSRT_ERR_INVALID; }
    return dsrt::guarded("dsrt_multi_create", [&]() -> int {
        DsrtMulti* m = new DsrtMulti();
        m->ranks.resize((size_t)n);
        auto fail = [&](int rc) { dsrt_multi_destroy(m); return rc; };
        std::set<int> distinct;
        for (int r = 0; r < n; ++r) {
            Rank& k = m->ranks[(size_t)r];
            k.device = devices[r];
            distinct.insert(devices[r]);
            k.slots.resize((size_t)frames_in_flight);
            int rc = dsrt_ctx_create(k.device, &k.slots[0].ctx);
            if (rc) return fail(rc);
            if (!hip_ok(hipSetDevice(k.device), "hipSetDevice")) return fail(DSRT_ERR_HIP);
            for (Slot& s : k.slots)
                if (!hip_ok(hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking), "hipStreamCreateWithFlags")) return fail(DSRT_ERR_HIP);
            if (!hip_ok(hipEventCreate(&k.ev0), "hipEventCreate") || !hip_ok(hipEventCreate(&k.ev1), "hipEventCreate") ||
                !hip_ok(hipEventCreateWithFlags(&k.ev_sent, hipEventDisableTiming), "hipEventCreateWithFlags")) return fail(DSRT_ERR_HIP);
        }
        if (n > 1 && (int)distinct.size() == n) {
            m->comms.resize((size_t)n);
            if (!nccl_ok(ncclCommInitAll(m->comms.data(), n, devices), "ncclCommInitAll")) { m->comms.clear(); return fail(DSRT_ERR_COMM); }
            m->rccl = true;
        }
        *out = m;
        return DSRT_OK;
    });
}

int dsrt_multi_count(const DsrtMulti* m) { return m ? (int)m->ranks.size() : 0; }
int dsrt_multi_uses_rccl(const DsrtMulti* m) { return m && m->rccl ? 1 : 0; }

// The smallest thing that exercises the collective on whatever is there: a one-rank RCCL communicator on `device` and one ncclGather of
// `bytes` bytes through it (rank 0's block lands at offset 0 of the receive buffer).  On a one-GPU box this is all of the RCCL path that
// can run -- library load, communicator, the call and its stream ordering; the N-rank gather itself needs N devices.
int dsrt_selftest_rccl_gather(int device, size_t bytes) {
    return dsrt::guarded("dsrt_selftest_rccl_gather", [&]() -> int {
        if (bytes == 0 || bytes > ((size_t)1 << 30)) { set_error("dsrt_selftest_rccl_gather: 1 byte to 1 GiB"); return DSRT_ERR_INVALID; }
        HIP_TRY(hipSetDevice(device));
        ncclComm_t comm = nullptr;
        const int devs[1] = {device};
        NCCL_TRY(ncclCommInitAll(&comm, 1, devs));
        uint8_t *send = nullptr, *recv = nullptr;
        hipStream_t stream = nullptr;
        int rc = DSRT_OK;
        std::vector<uint8_t> host(bytes), back(bytes);
        for (size_t i = 0; i < bytes; ++i) host[i] = (uint8_t)(i * 131u + 7u);
        if (!hip_ok(hipMalloc((void**)&send, bytes), "hipMalloc") || !hip_ok(hipMalloc((void**)&recv, bytes), "hipMalloc") ||
            !hip_ok(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking), "hipStreamCreateWithFlags") ||
            !hip_ok(hipMemcpyAsync(send, host.data(), bytes, hipMemcpyHostToDevice, stream), "hipMemcpyAsync") ||
            !hip_ok(hipMemsetAsync(recv, 0, bytes, stream), "hipMemsetAsync")) rc = DSRT_ERR_HIP;
        if (rc == DSRT_OK && !nccl_ok(ncclGather(send, recv, bytes, ncclUint8, 0, comm, stream), "ncclGather")) rc = DSRT_ERR_COMM;
        if (rc == DSRT_OK && (!hip_ok(hipMemcpyAsync(back.data(), recv, bytes, hipMemcpyDeviceToHost, stream), "hipMemcpyAsync") ||
                              !hip_ok(hipStreamSynchronize(stream), "hipStreamSynchronize"))) rc = DSRT_ERR_HIP;
        if (rc == DSRT_OK && std::memcmp(host.data(), back.data(), bytes) != 0) { set_error("dsrt_selftest_rccl_gather: gathered bytes differ"); rc = DSRT_ERR_COMM; }
        if (stream) (void)hipStreamDestroy(stream);
        if (send) (void)hipFree(send);
        if (recv) (void)hipFree(recv);
        (void)ncclCommDestroy(comm);
        return rc;
    });
}

int dsrt_multi_scene_upload(DsrtMulti* m, const GPUScene* host_scene) {
    if (!m || !host_scene) { set_error("dsrt_multi_scene_upload: null argument"); return DSRT_ERR_INVALID; }
    return dsrt::guarded("dsrt_multi_scene_upload", [&]() -> int {
        m->have_scene = false;
        for (Rank& k : m->ranks) {
            // one resident copy per DEVICE: ranks that share a device share the first one's scene too
            DsrtContext* same_device = nullptr;
            for (Rank& other : m->ranks) { if (&other == &k) break; if (other.device == k.device) { same_device = other.slots[0].ctx; break; } }
            for (size_t i = 1; i < k.slots.size(); ++i) if (k.slots[i].ctx) { dsrt_ctx_destroy(k.slots[i].ctx); k.slots[i].ctx = nullptr; }
            int rc;
            if (same_device) {
                dsrt_ctx_destroy(k.slots[0].ctx); k.slots[0].ctx = nullptr;
                rc = dsrt_ctx_clone(same_device, &k.slots[0].ctx);
            } else {
                rc = dsrt_scene_upload(k.slots[0].ctx, host_scene);
            }
            if (rc) return rc;
            for (size_t i = 1; i < k.slots.size(); ++i) if ((rc = dsrt_ctx_clone(k.slots[0].ctx, &k.slots[i].ctx))) return rc;
        }
        m->have_scene = true;
        return DSRT_OK;
    });
}

int dsrt_multi_render_frame(DsrtMulti* m, const DsrtRenderDesc* desc_in, const GPUCamera* cam, const float sun_dir_model[3], uint8_t* h_rgb8,
                            float* kernel_ms_per_rank, double* seconds) {
    if (!m || !desc_in || !cam || !h_rgb8) { set_error("dsrt_multi_render_frame: null argument"); return DSRT_ERR_INVALID; }
    if (!m->have_scene) { set_error("dsrt_multi_render_frame: no scene uploaded"); return DSRT_ERR_NO_SCENE; }
    return dsrt::guarded("dsrt_multi_render_frame", [&]() -> int {
        const int n = (int)m->ranks.size();
        const auto t0 = std::chrono::steady_clock::now();
        DsrtRenderDesc d = *desc_in;
        d.shard_count = n > 1 ? n : 0;
        d.shard_rank = 0;
        size_t padded_bytes = 0;
        int rc = dsrt_shard_layout(&d, nullptr, nullptr, nullptr, &padded_bytes);
        if (rc) return rc;
        const size_t image_bytes = (size_t)d.width * d.height * 3;
        Rank& root = m->ranks[0];
        HIP_TRY(hipSetDevice(root.device));
        if ((rc = ensure(&m->d_frame, &m->frame_bytes, image_bytes))) return rc;
        if (n > 1 && (rc = ensure(&m->d_gathered, &m->gathered_bytes, padded_bytes * (size_t)n))) return rc;
        // 1. every rank renders its tiles (asynchronous: the loop returns as soon as the launches are queued)
        for (int r = 0; r < n; ++r) {
            Rank& k = m->ranks[(size_t)r];
            Slot& s = k.slots[0];
            HIP_TRY(hipSetDevice(k.device));
            if ((rc = dsrt_scene_set_camera_sun(s.ctx, cam, sun_dir_model))) return rc;
            d.shard_rank = r;
            uint8_t* target = m->d_frame;
            if (n > 1) {
                if ((rc = ensure(&k.d_part, &k.part_bytes, padded_bytes))) return rc;
                target = k.d_part;
            }
            HIP_TRY(hipEventRecord(k.ev0, s.stream));
            if ((rc = dsrt_render(s.ctx, &d, target, nullptr, s.stream, nullptr))) return rc;
            HIP_TRY(hipEventRecord(k.ev1, s.stream));
        }
        // 2. the one collective of the frame: equal-sized compact buffers -> rank 0
        if (n > 1) {
            if (m->rccl) {
                NCCL_TRY(ncclGroupStart());
                for (int r = 0; r < n; ++r) {
                    Rank& k = m->ranks[(size_t)r];
                    if (!nccl_ok(ncclGather(k.d_part, m->d_gathered, padded_bytes, ncclUint8, 0, m->comms[(size_t)r], k.slots[0].stream), "ncclGather")) {
                        (void)ncclGroupEnd();
                        return DSRT_ERR_COMM;
                    }
                }
                NCCL_TRY(ncclGroupEnd());
            } else {
                // ranks share a device: the same receive layout filled by copies, each ordered behind its rank's render
                for (int r = 0; r < n; ++r) {
                    Rank& k = m->ranks[(size_t)r];
                    HIP_TRY(hipSetDevice(k.device));
                    HIP_TRY(hipMemcpyAsync(m->d_gathered + (size_t)r * padded_bytes, k.d_part, padded_bytes, hipMemcpyDeviceToDevice, k.slots[0].stream));
                    if (r > 0) {
                        HIP_TRY(hipEventRecord(k.ev_sent, k.slots[0].stream));
                        HIP_TRY(hipStreamWaitEvent(root.slots[0].stream, k.ev_sent, 0));
                    }
                }
            }
            HIP_TRY(hipSetDevice(root.device));
            d.shard_rank = 0;
            if ((rc = dsrt_deinterleave_tiles(root.slots[0].ctx, &d, m->d_gathered, m->d_frame, root.slots[0].stream))) return rc;
        }
        // 3. image to the caller
        HIP_TRY(hipSetDevice(root.device));
        HIP_TRY(hipMemcpyAsync(h_rgb8, m->d_frame, image_bytes, hipMemcpyDeviceToHost, root.slots[0].stream));
        for (int r = n - 1; r >= 0; --r) {
            Rank& k = m->ranks[(size_t)r];
            HIP_TRY(hipSetDevice(k.device));
            HIP_TRY(hipStreamSynchronize(k.slots[0].stream));
        }
        if (kernel_ms_per_rank)
            for (int r = 0; r < n; ++r) HIP_TRY(hipEventElapsedTime(&kernel_ms_per_rank[r], m->ranks[(size_t)r].ev0, m->ranks[(size_t)r].ev1));
        if (seconds) *seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        return DSRT_OK;
    });
}

int dsrt_multi_render_sequence(DsrtMulti* m, const DsrtRenderDesc* desc_in, const GPUCamera* cams, const float* sun_dirs, int n_frames,
                               uint8_t* const* h_images, double* seconds) {
    if (!m || !desc_in || !cams || !sun_dirs || n_frames < 0) { set_error("dsrt_multi_render_sequence: bad argument"); return DSRT_ERR_INVALID; }
    if (!m->have_scene) { set_error("dsrt_multi_render_sequence: no scene uploaded"); return DSRT_ERR_NO_SCENE; }
    return dsrt::guarded("dsrt_multi_render_sequence", [&]() -> int {
        const int n = (int)m->ranks.size();
        DsrtRenderDesc d = *desc_in;
        d.shard_count = n > 1 ? n : 0; d.shard_rank = 0;
        const size_t image_bytes = (size_t)d.width * d.height * 3;
        // EVERY rank renders its interleaved tiles of EVERY frame, as batch launches (dsrt_render_batch with a shard: the rank's tiles of all
        // the frames of a launch are one pool of work).  Loads are equal by construction (each rank has every 8th tile of every frame), every
        // frame's serial chains are spread over all GPUs AND run under the other frames' bulk, and the collective is one gather per launch for
        // all its frames.  PROJECTION, not a measurement of an N-GPU run: one rank's share of the 99-pose approach at 1080p x 250, timed alone
        // on one MI355X (no gather, no xGMI traffic, no root de-interleave), takes 0.29 s of 8 in rng_mode 0 -- which would be 5.8x the
        // single-GPU rate (whole frames dealt by cost: 5.0x, round-robin: 4.2x) -- and 0.24 s in rng_mode 1 (7.3x).  Frames go nearest
        // (costliest) first; as many per launch as its 32-bit work-item numbers allow, 128 at most.
        constexpr size_t kGroup = 128;
        int rc;
        size_t padded_bytes = image_bytes;
        if (n > 1 && (rc = dsrt_shard_layout(&d, nullptr, nullptr, nullptr, &padded_bytes))) return rc;
        std::vector<int> order((size_t)n_frames);
        {
            float lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
            if ((rc = dsrt_ctx_scene_bounds(m->ranks[0].slots[0].ctx, lo, hi))) return rc;
            const double centre[3] = {0.5 * ((double)lo[0] + hi[0]), 0.5 * ((double)lo[1] + hi[1]), 0.5 * ((double)lo[2] + hi[2])};
            std::vector<double> dist((size_t)n_frames);
            for (int i = 0; i < n_frames; ++i) {
                const double dx = cams[i].origin.x - centre[0], dy = cams[i].origin.y - centre[1], dz = cams[i].origin.z - centre[2];
                dist[(size_t)i] = dx * dx + dy * dy + dz * dz;
                order[(size_t)i] = i;
            }
            std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return dist[(size_t)a] < dist[(size_t)b]; });
        }
        size_t per_launch = std::max<size_t>(1, std::min<size_t>((size_t)n_frames, kGroup));
        while (per_launch > 1 && (unsigned long long)per_launch * (padded_bytes / 3) * (d.rng_mode == 1 ? 16ull : 1ull) >= (1ull << 32)) per_launch /= 2;
        Rank& root = m->ranks[0];
        for (Rank& k : m->ranks) {
            HIP_TRY(hipSetDevice(k.device));
            if (n > 1 && (rc = ensure(&k.d_part, &k.part_bytes, padded_bytes * per_launch))) return rc;
        }
        HIP_TRY(hipSetDevice(root.device));
        if ((rc = ensure_slot_images(root.slots[0], image_bytes * per_launch))) return rc;
        if (n > 1 && (rc = ensure(&m->d_gathered, &m->gathered_bytes, padded_bytes * per_launch * (size_t)n))) return rc;
        const auto t0 = std::chrono::steady_clock::now();
        std::vector<GPUCamera> gcams;
        std::vector<float> gsuns;
        for (size_t at = 0; at < (size_t)n_frames; at += per_launch) {
            const size_t count = std::min(per_launch, (size_t)n_frames - at);
            gcams.clear(); gsuns.clear();
            for (size_t q = 0; q < count; ++q) {
                const int fi = order[at + q];
                gcams.push_back(cams[fi]);
                gsuns.insert(gsuns.end(), sun_dirs + 3 * (size_t)fi, sun_dirs + 3 * (size_t)fi + 3);
            }
            // 1. every rank renders its tiles of these frames (asynchronous: the loop returns as soon as the launches are queued)
            for (int r = 0; r < n; ++r) {
                Rank& k = m->ranks[(size_t)r];
                HIP_TRY(hipSetDevice(k.device));
                d.shard_rank = r;
                if ((rc = dsrt_render_batch(k.slots[0].ctx, &d, (int)count, gcams.data(), gsuns.data(), n > 1 ? k.d_part : root.slots[0].d_image, nullptr,
                                            k.slots[0].stream, nullptr))) return rc;
            }
            // 2. the one collective of the launch: equal-sized buffers (count parts each) -> rank 0
            if (n > 1) {
                const size_t send = padded_bytes * count;
                if (m->rccl) {
                    NCCL_TRY(ncclGroupStart());
                    for (int r = 0; r < n; ++r) {
                        Rank& k = m->ranks[(size_t)r];
                        if (!nccl_ok(ncclGather(k.d_part, m->d_gathered, send, ncclUint8, 0, m->comms[(size_t)r], k.slots[0].stream), "ncclGather")) {
                            (void)ncclGroupEnd();
                            return DSRT_ERR_COMM;
                        }
                    }
                    NCCL_TRY(ncclGroupEnd());
                } else {
                    for (int r = 0; r < n; ++r) {                      // ranks share a device: the same receive layout filled by copies
                        Rank& k = m->ranks[(size_t)r];
                        HIP_TRY(hipSetDevice(k.device));
                        HIP_TRY(hipMemcpyAsync(m->d_gathered + (size_t)r * send, k.d_part, send, hipMemcpyDeviceToDevice, k.slots[0].stream));
                        if (r > 0) {
                            HIP_TRY(hipEventRecord(k.ev_sent, k.slots[0].stream));
                            HIP_TRY(hipStreamWaitEvent(root.slots[0].stream, k.ev_sent, 0));
                        }
                    }
                }
                HIP_TRY(hipSetDevice(root.device));
                d.shard_rank = 0;
                if ((rc = dsrt_deinterleave_batch(root.slots[0].ctx, &d, (int)count, m->d_gathered, root.slots[0].d_image, root.slots[0].stream))) return rc;
            }
            // 3. images to the caller
            HIP_TRY(hipSetDevice(root.device));
            HIP_TRY(hipMemcpyAsync(root.slots[0].h_pinned, root.slots[0].d_image, image_bytes * count, hipMemcpyDeviceToHost, root.slots[0].stream));
            for (int r = n - 1; r >= 0; --r) {
                Rank& k = m->ranks[(size_t)r];
                HIP_TRY(hipSetDevice(k.device));
                HIP_TRY(hipStreamSynchronize(k.slots[0].stream));
            }
            for (size_t q = 0; q < count; ++q) {
                const int fi = order[at + q];
                if (h_images && h_images[fi]) std::memcpy(h_images[fi], root.slots[0].h_pinned + q * image_bytes, image_bytes);
            }
        }
        if (seconds) *seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        return DSRT_OK;
    });
}

}  // extern "C"
