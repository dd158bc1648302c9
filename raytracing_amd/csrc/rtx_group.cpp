// rtx_group_*: one frame over several GPUs from one process, behind the C ABI (include/rtx.h).
//
// The reference spreads the rows of a frame over OpenMP threads (src/Scene.cpp:214-215, thread count from
// `-threads`, :232).  Here rank r of G renders rows r, r + G, ... on its own GPU (rtx_frame.row_first /
// row_step) into a compact device buffer; the buffers meet on rank 0's GPU in one exchange -- RCCL
// send/recv pairs inside one group call, point-to-point xGMI transfers into the root, no ring -- and a
// de-interleave kernel puts the rows in place.  One host thread drives all devices: every step is an
// asynchronous enqueue on the rank's own stream.
//
// Built on the public entry points only (rtx_render_device, rtx_deinterleave_rows_device, rtx_get_stats).
// RCCL is loaded at run time, and only when the group spans more than one GPU: a process that already holds
// an RCCL (PyTorch ships its own) must not get a second one, and a single-GPU user needs none.
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstddef>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/rtx.h"

namespace {

thread_local std::string g_group_err;
int groupFail(int code, const std::string &msg) {
    g_group_err = msg;
    return code;
}

// the handful of RCCL entry points the exchange needs (signatures as in rccl/rccl.h)
typedef void *ncclComm_t;
struct Rccl {
    void *lib = nullptr;
    int (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    int (*CommDestroy)(ncclComm_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Send)(const void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*Recv)(void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    bool load(std::string &why) {
        if (lib) return true;
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (lib) break;
        }
        if (!lib) {
            why = std::string("cannot load librccl: ") + dlerror();
            return false;
        }
        auto sym = [&](const char *n) { return dlsym(lib, n); };
        CommInitAll = reinterpret_cast<decltype(CommInitAll)>(sym("ncclCommInitAll"));
        CommDestroy = reinterpret_cast<decltype(CommDestroy)>(sym("ncclCommDestroy"));
        GroupStart = reinterpret_cast<decltype(GroupStart)>(sym("ncclGroupStart"));
        GroupEnd = reinterpret_cast<decltype(GroupEnd)>(sym("ncclGroupEnd"));
        Send = reinterpret_cast<decltype(Send)>(sym("ncclSend"));
        Recv = reinterpret_cast<decltype(Recv)>(sym("ncclRecv"));
        GetErrorString = reinterpret_cast<decltype(GetErrorString)>(sym("ncclGetErrorString"));
        if (!CommInitAll || !CommDestroy || !GroupStart || !GroupEnd || !Send || !Recv || !GetErrorString) {
            why = "librccl lacks an expected symbol";
            return false;
        }
        return true;
    }
};
Rccl g_rccl;
constexpr int kNcclUint8 = 1;  // ncclUint8

}  // namespace

struct rtx_group {
    int n = 0;
    std::vector<int> device;        // GPU of rank r
    std::vector<rtx_ctx *> ctx;     // its context
    std::vector<hipStream_t> stream;
    std::vector<hipEvent_t> done;   // recorded on the rank's stream after its render (and send)
    // the exchange: one RCCL rank per distinct GPU; comm_of[r] = index into comms of rank r's GPU
    std::vector<int> comm_devices;
    std::vector<ncclComm_t> comms;
    std::vector<int> comm_of;
    std::vector<hipStream_t> xstream;  // per RCCL rank: the stream its sends / receives go to (that of the first rank on the GPU)
    bool via_rccl_always = false;   // RTX_GROUP_FORCE_RCCL: ranks that share the root's GPU send to it as well
    // buffers, grown on demand
    std::vector<unsigned char *> local;  // rank r's rows when they have to travel (on its GPU)
    std::vector<size_t> local_bytes;
    unsigned char *gathered = nullptr, *frame = nullptr;  // on the root's GPU
    size_t gathered_bytes = 0, frame_bytes = 0;
};

#define GROUP_HIP(expr)                                                                              \
    do {                                                                                             \
        hipError_t e_ = (expr);                                                                      \
        if (e_ != hipSuccess) return groupFail(RTX_E_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)
#define GROUP_NCCL(expr)                                                                             \
    do {                                                                                             \
        int r_ = (expr);                                                                             \
        if (r_ != 0) return groupFail(RTX_E_DEVICE, std::string(#expr) + ": " + g_rccl.GetErrorString(r_)); \
    } while (0)

namespace {

int growBuffer(unsigned char **p, size_t *have, size_t need) {
    if (need <= *have) return RTX_OK;
    (void)hipFree(*p);
    *p = nullptr, *have = 0;
    GROUP_HIP(hipMalloc((void **)p, need));
    *have = need;
    return RTX_OK;
}

}  // namespace

extern "C" {

const char *rtx_group_last_error(void) { return g_group_err.c_str(); }

int rtx_group_create(rtx_group **out, const int *device_ids, int n_devices) {
    if (!out) return groupFail(RTX_E_INVALID, "rtx_group_create: out is null");
    *out = nullptr;
    if (!device_ids || n_devices <= 0 || n_devices > 64) return groupFail(RTX_E_INVALID, "rtx_group_create: 1 to 64 ranks");
    rtx_group *g = new rtx_group();
    g->n = n_devices;
    g->via_rccl_always = getenv("RTX_GROUP_FORCE_RCCL") != nullptr;
    for (int r = 0; r < n_devices; ++r) {
        rtx_ctx *c = nullptr;
        if (int e = rtx_create(&c, device_ids[r])) {
            g_group_err = std::string("rtx_group_create: rank ") + std::to_string(r) + ": " + rtx_last_error();
            rtx_group_destroy(g);
            return e;
        }
        g->device.push_back(device_ids[r]);
        g->ctx.push_back(c);
        hipStream_t s = nullptr;
        hipEvent_t ev = nullptr;
        if (hipSetDevice(device_ids[r]) != hipSuccess || hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) {
            rtx_group_destroy(g);
            return groupFail(RTX_E_DEVICE, "rtx_group_create: stream / event");
        }
        g->stream.push_back(s), g->done.push_back(ev);
        g->local.push_back(nullptr), g->local_bytes.push_back(0);
        int k = 0;
        while (k < (int)g->comm_devices.size() && g->comm_devices[k] != device_ids[r]) ++k;
        if (k == (int)g->comm_devices.size()) g->comm_devices.push_back(device_ids[r]), g->xstream.push_back(s);
        g->comm_of.push_back(k);
    }
    if (g->comm_devices.size() > 1 || g->via_rccl_always) {
        std::string why;
        if (!g_rccl.load(why)) {
            rtx_group_destroy(g);
            return groupFail(RTX_E_DEVICE, "rtx_group_create: " + why);
        }
        g->comms.resize(g->comm_devices.size());
        int rc = g_rccl.CommInitAll(g->comms.data(), (int)g->comm_devices.size(), g->comm_devices.data());
        if (rc != 0) {
            g->comms.clear();
            rtx_group_destroy(g);
            return groupFail(RTX_E_DEVICE, std::string("ncclCommInitAll: ") + g_rccl.GetErrorString(rc));
        }
    }
    *out = g;
    return RTX_OK;
}

void rtx_group_destroy(rtx_group *g) {
    if (!g) return;
    for (int r = 0; r < (int)g->ctx.size(); ++r) {
        (void)hipSetDevice(g->device[r]);
        if (r < (int)g->stream.size() && g->stream[r]) (void)hipStreamSynchronize(g->stream[r]);
    }
    for (ncclComm_t c : g->comms)
        if (c) (void)g_rccl.CommDestroy(c);
    for (int r = 0; r < (int)g->ctx.size(); ++r) {
        (void)hipSetDevice(g->device[r]);
        if (r < (int)g->local.size()) (void)hipFree(g->local[r]);
        if (r < (int)g->done.size() && g->done[r]) (void)hipEventDestroy(g->done[r]);
        if (r < (int)g->stream.size() && g->stream[r]) (void)hipStreamDestroy(g->stream[r]);
        rtx_destroy(g->ctx[r]);
    }
    if (!g->device.empty()) {
        (void)hipSetDevice(g->device[0]);
        (void)hipFree(g->gathered);
        (void)hipFree(g->frame);
    }
    delete g;
}

int rtx_group_size(const rtx_group *g) { return g ? g->n : 0; }

rtx_ctx *rtx_group_ctx(rtx_group *g, int rank) { return g && rank >= 0 && rank < g->n ? g->ctx[rank] : nullptr; }

int rtx_group_commit(rtx_group *g) {
    if (!g) return groupFail(RTX_E_INVALID, "null group");
    for (int r = 0; r < g->n; ++r)
        if (int e = rtx_commit(g->ctx[r])) return groupFail(e, std::string("rank ") + std::to_string(r) + ": " + rtx_last_error());
    return RTX_OK;
}

int rtx_group_render(rtx_group *g, const rtx_camera *cam, int width, int height, int depth, unsigned char *bgr_out, rtx_stats *stats) {
    if (!g || !cam) return groupFail(RTX_E_INVALID, "null argument");
    if (width <= 0 || height <= 0) return groupFail(RTX_E_INVALID, "bad frame size");
    if (!bgr_out) return groupFail(RTX_E_INVALID, "null output buffer");
    const int G = g->n, root_dev = g->device[0];
    const size_t row_bytes = (size_t)width * 3;
    const int rows_per_rank = (height + G - 1) / G;
    const size_t block = (size_t)rows_per_rank * row_bytes;
    GROUP_HIP(hipSetDevice(root_dev));
    if (int e = growBuffer(&g->gathered, &g->gathered_bytes, block * G)) return e;
    if (int e = growBuffer(&g->frame, &g->frame_bytes, (size_t)height * row_bytes)) return e;
    // ---- every rank renders its rows: in place into its block on the root's GPU when it shares that GPU,
    // into a buffer of its own otherwise ----
    std::vector<bool> travels(G, false);
    for (int r = 0; r < G; ++r) {
        travels[r] = r != 0 && (g->device[r] != root_dev || g->via_rccl_always);
        GROUP_HIP(hipSetDevice(g->device[r]));
        unsigned char *dst = g->gathered + (size_t)r * block;
        if (travels[r]) {
            if (int e = growBuffer(&g->local[r], &g->local_bytes[r], block)) return e;
            dst = g->local[r];
        }
        const rtx_frame fr = {width, height, depth, r, G};
        if (int e = rtx_render_device(g->ctx[r], cam, &fr, dst, g->stream[r]))
            return groupFail(e, std::string("rank ") + std::to_string(r) + ": " + rtx_last_error());
    }
    // ---- the exchange: one group of send / recv pairs, every transfer straight into the root.  All calls on
    // one communicator go to one stream (the GPU's first rank's); the other ranks of that GPU hand over by event.
    bool any = false;
    for (int r = 1; r < G; ++r) {
        any = any || travels[r];
        hipStream_t ordered_by = travels[r] ? g->xstream[g->comm_of[r]] : g->stream[0];  // whoever touches rank r's rows next
        if (ordered_by == g->stream[r]) continue;
        GROUP_HIP(hipSetDevice(g->device[r]));
        GROUP_HIP(hipEventRecord(g->done[r], g->stream[r]));
        GROUP_HIP(hipStreamWaitEvent(ordered_by, g->done[r], 0));
    }
    if (any) {
        GROUP_NCCL(g_rccl.GroupStart());
        for (int r = 1; r < G; ++r) {
            if (!travels[r]) continue;
            const rtx_frame fr = {width, height, depth, r, G};
            const size_t bytes = (size_t)rtx_frame_rows(&fr) * row_bytes;
            if (bytes == 0) continue;
            GROUP_NCCL(g_rccl.Send(g->local[r], bytes, kNcclUint8, g->comm_of[0], g->comms[g->comm_of[r]], g->xstream[g->comm_of[r]]));
            GROUP_NCCL(g_rccl.Recv(g->gathered + (size_t)r * block, bytes, kNcclUint8, g->comm_of[r], g->comms[g->comm_of[0]], g->stream[0]));
        }
        GROUP_NCCL(g_rccl.GroupEnd());
    }
    GROUP_HIP(hipSetDevice(root_dev));
    const unsigned char *whole = g->gathered;
    if (G > 1) {
        if (int e = rtx_deinterleave_rows_device(g->ctx[0], g->gathered, G, rows_per_rank, width, height, g->frame, g->stream[0]))
            return groupFail(e, rtx_last_error());
        whole = g->frame;
    }
    GROUP_HIP(hipMemcpyAsync(bgr_out, whole, (size_t)height * row_bytes, hipMemcpyDeviceToHost, g->stream[0]));
    GROUP_HIP(hipStreamSynchronize(g->stream[0]));
    for (int r = 1; r < G; ++r) {  // senders are done once the root has received, but their streams may still be draining
        GROUP_HIP(hipSetDevice(g->device[r]));
        GROUP_HIP(hipStreamSynchronize(g->stream[r]));
    }
    if (stats) {
        memset(stats, 0, sizeof *stats);
        for (int r = 0; r < G; ++r) {
            rtx_stats s;
            if (int e = rtx_get_stats(g->ctx[r], &s)) return groupFail(e, rtx_last_error());
            uint64_t *acc = reinterpret_cast<uint64_t *>(stats);
            const uint64_t *add = reinterpret_cast<const uint64_t *>(&s);
            for (size_t k = 0; k < offsetof(rtx_stats, kernel_ms) / sizeof(uint64_t); ++k) acc[k] += add[k];
            stats->kernel_ms = s.kernel_ms > stats->kernel_ms ? s.kernel_ms : stats->kernel_ms;
        }
    }
    return RTX_OK;
}

}  // extern "C"
