// libofdmtx, the optional exchange step: every rank's IQ shard into one buffer on a root GPU,
// point-to-point over RCCL (one send per peer, all receives of the root in one group, so the
// transfers of the peers run side by side on their own xGMI links).  Packets are independent and
// shard by index, so nothing on the transmit path itself needs this; it serves a consumer that
// wants one contiguous stream (the reference's single file sink / one radio per flowgraph,
// ftw_80211_ofdm_tx/src/examples/ftw_ofdm_tx_uhd.py:83-92).
//
// RCCL is not linked: librccl is looked up when the first communicator is asked for (in a process
// that already carries one -- PyTorch's -- that copy is found by its soname), so the transmit
// kernels load and run on machines and in processes without it.

#include <dlfcn.h>
#include <string.h>

#include <mutex>
#include <new>

#include <hip/hip_runtime.h>

// The seven entry points used, declared from RCCL's header where it is installed and by hand (the
// NCCL 2 ABI: opaque communicator, 128-byte id, int-sized enums) where it is not, so that the
// library builds on machines without the RCCL development files and answers OFDMTX_ECOMM there.
#if __has_include(<rccl/rccl.h>)
#include <rccl/rccl.h>
#else
extern "C" {
#define NCCL_UNIQUE_ID_BYTES 128
typedef struct ncclComm *ncclComm_t;
typedef struct { char internal[NCCL_UNIQUE_ID_BYTES]; } ncclUniqueId;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclChar = 0, ncclUint8 = 1, ncclInt32 = 2, ncclFloat32 = 7, ncclFloat = 7 } ncclDataType_t;
ncclResult_t ncclGetUniqueId(ncclUniqueId *id);
ncclResult_t ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId id, int rank);
ncclResult_t ncclCommDestroy(ncclComm_t comm);
ncclResult_t ncclSend(const void *buf, size_t count, ncclDataType_t type, int peer, ncclComm_t comm, hipStream_t stream);
ncclResult_t ncclRecv(void *buf, size_t count, ncclDataType_t type, int peer, ncclComm_t comm, hipStream_t stream);
ncclResult_t ncclGroupStart(void);
ncclResult_t ncclGroupEnd(void);
}
#endif

#include "ofdm_common.h"

namespace {

struct RcclApi {
    void *handle;
    decltype(&ncclGetUniqueId) get_unique_id;
    decltype(&ncclCommInitRank) comm_init_rank;
    decltype(&ncclCommDestroy) comm_destroy;
    decltype(&ncclSend) send;
    decltype(&ncclRecv) recv;
    decltype(&ncclGroupStart) group_start;
    decltype(&ncclGroupEnd) group_end;
    bool ok;
};

RcclApi g_rccl;
std::once_flag g_rccl_once;

const RcclApi *rccl()
{
    std::call_once(g_rccl_once, [] {
        memset(&g_rccl, 0, sizeof(g_rccl));
        const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char *n : names) {
            g_rccl.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (g_rccl.handle) break;
        }
        if (!g_rccl.handle) return;
        void *h = g_rccl.handle;
        g_rccl.get_unique_id = reinterpret_cast<decltype(&ncclGetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
        g_rccl.comm_init_rank = reinterpret_cast<decltype(&ncclCommInitRank)>(dlsym(h, "ncclCommInitRank"));
        g_rccl.comm_destroy = reinterpret_cast<decltype(&ncclCommDestroy)>(dlsym(h, "ncclCommDestroy"));
        g_rccl.send = reinterpret_cast<decltype(&ncclSend)>(dlsym(h, "ncclSend"));
        g_rccl.recv = reinterpret_cast<decltype(&ncclRecv)>(dlsym(h, "ncclRecv"));
        g_rccl.group_start = reinterpret_cast<decltype(&ncclGroupStart)>(dlsym(h, "ncclGroupStart"));
        g_rccl.group_end = reinterpret_cast<decltype(&ncclGroupEnd)>(dlsym(h, "ncclGroupEnd"));
        g_rccl.ok = g_rccl.get_unique_id && g_rccl.comm_init_rank && g_rccl.comm_destroy && g_rccl.send && g_rccl.recv &&
                    g_rccl.group_start && g_rccl.group_end;
    });
    return g_rccl.ok ? &g_rccl : nullptr;
}

thread_local int g_last_rccl = 0;

}  // namespace

#define RCCLCHK(expr)                      \
    do {                                   \
        const ncclResult_t r_ = (expr);    \
        if (r_ != ncclSuccess) {           \
            g_last_rccl = (int)r_;         \
            return OFDMTX_ECOMM;           \
        }                                  \
    } while (0)

struct ofdmtx_comm {
    ofdmtx_ctx *ctx;
    int rank, world;
    ncclComm_t comm;
};

extern "C" {

int ofdmtx_last_comm_error(void) { return g_last_rccl; }

int ofdmtx_comm_unique_id(void *id)
{
    static_assert(OFDMTX_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "id size");
    if (!id) return OFDMTX_EINVAL;
    const RcclApi *api = rccl();
    if (!api) return OFDMTX_ECOMM;
    ncclUniqueId u;
    RCCLCHK(api->get_unique_id(&u));
    memcpy(id, u.internal, NCCL_UNIQUE_ID_BYTES);
    return OFDMTX_OK;
}

int ofdmtx_comm_create(ofdmtx_ctx *ctx, int rank, int world, const void *id, ofdmtx_comm **out)
{
    if (!ctx || !id || !out || world < 1 || rank < 0 || rank >= world) return OFDMTX_EINVAL;
    *out = nullptr;
    const RcclApi *api = rccl();
    if (!api) return OFDMTX_ECOMM;
    HIPCHK(hipSetDevice(ctx->device));
    ofdmtx_comm *c = new (std::nothrow) ofdmtx_comm();
    if (!c) return OFDMTX_ENOMEM;
    c->ctx = ctx;
    c->rank = rank;
    c->world = world;
    ncclUniqueId u;
    memcpy(u.internal, id, NCCL_UNIQUE_ID_BYTES);
    const ncclResult_t r = api->comm_init_rank(&c->comm, world, u, rank);
    if (r != ncclSuccess) {
        g_last_rccl = (int)r;
        delete c;
        return OFDMTX_ECOMM;
    }
    *out = c;
    return OFDMTX_OK;
}

int ofdmtx_comm_destroy(ofdmtx_comm *c)
{
    if (!c) return OFDMTX_EINVAL;
    const RcclApi *api = rccl();
    (void)hipSetDevice(c->ctx->device);
    if (api && c->comm) (void)api->comm_destroy(c->comm);
    delete c;
    return OFDMTX_OK;
}

int ofdmtx_gather(ofdmtx_comm *c, const float *d_src, uint64_t n_samples, float *d_dst, const uint64_t *sample_off,
                  int root, void *stream)
{
    if (!c || root < 0 || root >= c->world || (n_samples && !d_src)) return OFDMTX_EINVAL;
    const RcclApi *api = rccl();
    if (!api) return OFDMTX_ECOMM;
    hipStream_t st = (hipStream_t)stream;
    HIPCHK(hipSetDevice(c->ctx->device));
    if (c->rank != root) {
        // 2 floats per sample; a rank with nothing to send posts nothing (the root skips it too)
        if (n_samples) RCCLCHK(api->send(d_src, (size_t)n_samples * 2u, ncclFloat32, root, c->comm, st));
        return OFDMTX_OK;
    }
    if (!sample_off || (!d_dst && sample_off[c->world])) return OFDMTX_EINVAL;
    for (int r = 0; r < c->world; r++)
        if (sample_off[r + 1] < sample_off[r]) return OFDMTX_EINVAL;
    if (sample_off[root + 1] - sample_off[root] != n_samples) return OFDMTX_EINVAL;
    RCCLCHK(api->group_start());
    ncclResult_t first_bad = ncclSuccess;
    for (int r = 0; r < c->world; r++) {
        const uint64_t cnt = sample_off[r + 1] - sample_off[r];
        if (r == root || cnt == 0) continue;
        const ncclResult_t rr = api->recv(d_dst + 2u * sample_off[r], (size_t)cnt * 2u, ncclFloat32, r, c->comm, st);
        if (rr != ncclSuccess && first_bad == ncclSuccess) first_bad = rr;
    }
    const ncclResult_t ge = api->group_end();                     // always closed, also after a failed post
    if (first_bad != ncclSuccess || ge != ncclSuccess) {
        g_last_rccl = (int)(first_bad != ncclSuccess ? first_bad : ge);
        return OFDMTX_ECOMM;
    }
    // the root's own shard: produced in place by a caller that hands d_dst + offset to its kernels,
    // copied otherwise
    float *own = d_dst + 2u * sample_off[root];
    if (n_samples && d_src != own)
        HIPCHK(hipMemcpyAsync(own, d_src, (size_t)n_samples * sizeof(float2), hipMemcpyDeviceToDevice, st));
    return OFDMTX_OK;
}

}  // extern "C"
