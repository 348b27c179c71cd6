// comm.hip -- RCCL behind the boundary: the exchange steps of the sharded fits as C-ABI calls, so that a
// Go process (one goroutine holding N handles, master/tasks.go:879-1034) or one process per GPU runs the multi-GPU path
// without any collective library of its own.  The collectives are enqueued on the handles' own HIP streams between the
// kernels that produce and consume their buffers: no host synchronisation anywhere in an exchange.
//
// librccl is opened on first use (dlopen), not linked: a single-GPU deployment never loads it.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <mutex>

#include "common.hpp"
#include "mf_internal.hpp"

using namespace gorse;

namespace {

struct Rccl {
    void *lib = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclBroadcast) Broadcast = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    std::string why;
};

std::string &rccl_why() {
    static std::string why;
    return why;
}

Rccl *rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            r.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (r.lib) break;
        }
        if (!r.lib) {
            const char *e = dlerror();
            r.why = e ? e : "librccl not found";
            return;
        }
        bool ok = true;
        auto sym = [&](const char *n) {
            void *p = dlsym(r.lib, n);
            if (!p) {
                ok = false;
                r.why = std::string("librccl lacks ") + n;
            }
            return p;
        };
        r.GetUniqueId = (decltype(r.GetUniqueId))sym("ncclGetUniqueId");
        r.CommInitRank = (decltype(r.CommInitRank))sym("ncclCommInitRank");
        r.CommInitAll = (decltype(r.CommInitAll))sym("ncclCommInitAll");
        r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
        r.AllReduce = (decltype(r.AllReduce))sym("ncclAllReduce");
        r.Broadcast = (decltype(r.Broadcast))sym("ncclBroadcast");
        r.GroupStart = (decltype(r.GroupStart))sym("ncclGroupStart");
        r.GroupEnd = (decltype(r.GroupEnd))sym("ncclGroupEnd");
        r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
        if (!ok) {
            dlclose(r.lib);
            r.lib = nullptr;
        }
    });
    if (!r.lib) rccl_why() = r.why;
    return r.lib ? &r : nullptr;
}

int32_t need_rccl(Rccl **out) {
    *out = rccl();
    if (!*out) {
        return fail(GORSE_ERR_HIP, "RCCL unavailable: %s", rccl_why().empty() ? "dlopen(librccl.so.1) failed" : rccl_why().c_str());
    }
    return GORSE_OK;
}

#define GORSE_RCCL_CHECK(R, expr)                                                                                  \
    do {                                                                                                           \
        ncclResult_t _e = (expr);                                                                                  \
        if (_e != ncclSuccess) return gorse::fail(GORSE_ERR_HIP, "%s:%d %s -> %s", __FILE__, __LINE__, #expr,     \
                                                  (R)->GetErrorString(_e));                                        \
    } while (0)

}  // namespace

struct gorse_comm {
    ncclComm_t comm = nullptr;
    int world = 1, rank = 0, device = 0;
    DevBuf<float> xbuf;  // the all-reduced item-factor delta (I * d) / staging of host all-reduces
};

extern "C" int32_t gorse_comm_unique_id(uint8_t *id) {
    if (!id) return fail(GORSE_ERR_INVALID, "id is NULL");
    Rccl *R;
    GORSE_TRY(need_rccl(&R));
    static_assert(GORSE_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "id size");
    ncclUniqueId u;
    GORSE_RCCL_CHECK(R, R->GetUniqueId(&u));
    memcpy(id, u.internal, GORSE_COMM_ID_BYTES);
    return GORSE_OK;
}

extern "C" int32_t gorse_comm_create(gorse_comm **out, const uint8_t *id, int32_t world, int32_t rank, int32_t device) {
    if (!out || !id) return fail(GORSE_ERR_INVALID, "NULL argument");
    *out = nullptr;
    if (world < 1 || rank < 0 || rank >= world) return fail(GORSE_ERR_INVALID, "rank %d of %d", rank, world);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return fail(GORSE_ERR_NO_DEVICE, "no HIP device visible (libgorse_hip needs an MI355X / gfx950)");
    if (device < 0 || device >= ndev) return fail(GORSE_ERR_INVALID, "device %d out of range [0,%d)", device, ndev);
    Rccl *R;
    GORSE_TRY(need_rccl(&R));
    GORSE_HIP_CHECK(hipSetDevice(device));
    gorse_comm *c = new (std::nothrow) gorse_comm();
    if (!c) return fail(GORSE_ERR_NOMEM, "out of host memory");
    c->world = world, c->rank = rank, c->device = device;
    ncclUniqueId u;
    memcpy(u.internal, id, GORSE_COMM_ID_BYTES);
    ncclResult_t e = R->CommInitRank(&c->comm, world, u, rank);
    if (e != ncclSuccess) {
        delete c;
        return fail(GORSE_ERR_HIP, "ncclCommInitRank(rank %d of %d): %s", rank, world, R->GetErrorString(e));
    }
    *out = c;
    return GORSE_OK;
}

extern "C" int32_t gorse_comm_create_local(gorse_comm **out, const int32_t *devices, int32_t n) {
    if (!out || !devices || n < 1 || n > 64) return fail(GORSE_ERR_INVALID, "bad arguments");
    for (int i = 0; i < n; i++) out[i] = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return fail(GORSE_ERR_NO_DEVICE, "no HIP device visible (libgorse_hip needs an MI355X / gfx950)");
    for (int i = 0; i < n; i++)
        if (devices[i] < 0 || devices[i] >= ndev) return fail(GORSE_ERR_INVALID, "device %d out of range [0,%d)", devices[i], ndev);
    Rccl *R;
    GORSE_TRY(need_rccl(&R));
    std::vector<ncclComm_t> comms((size_t)n);
    std::vector<int> devs(devices, devices + n);
    GORSE_RCCL_CHECK(R, R->CommInitAll(comms.data(), n, devs.data()));
    for (int i = 0; i < n; i++) {
        gorse_comm *c = new (std::nothrow) gorse_comm();
        if (!c) {  // nothing half-made is left behind: the wrappers made so far and every communicator go back
            for (int j = 0; j < i; j++) {
                delete out[j];
                out[j] = nullptr;
            }
            for (int j = 0; j < n; j++) (void)R->CommDestroy(comms[(size_t)j]);
            return fail(GORSE_ERR_NOMEM, "out of host memory");
        }
        c->comm = comms[(size_t)i], c->world = n, c->rank = i, c->device = devices[i];
        out[i] = c;
    }
    return GORSE_OK;
}

extern "C" int32_t gorse_comm_destroy(gorse_comm *c) {
    if (!c) return GORSE_OK;
    (void)hipSetDevice(c->device);
    Rccl *R = rccl();
    if (R && c->comm) (void)R->CommDestroy(c->comm);
    delete c;
    return GORSE_OK;
}

extern "C" int32_t gorse_comm_info(gorse_comm *c, int32_t *world, int32_t *rank) {
    if (!c) return fail(GORSE_ERR_INVALID, "communicator is NULL");
    if (world) *world = c->world;
    if (rank) *rank = c->rank;
    return GORSE_OK;
}

namespace {
int32_t check_pairs(gorse_mf *const *hs, gorse_comm *const *cs, int32_t n) {
    if (!hs || !cs || n < 1) return fail(GORSE_ERR_INVALID, "bad arguments");
    for (int i = 0; i < n; i++) {
        if (!hs[i] || !cs[i]) return fail(GORSE_ERR_INVALID, "NULL handle or communicator at %d", i);
        if (hs[i]->device != cs[i]->device)
            return fail(GORSE_ERR_INVALID, "handle %d lives on device %d, its communicator on %d", i, hs[i]->device, cs[i]->device);
    }
    return GORSE_OK;
}
}  // namespace

// BPR: Q <- Q_sync + sum over ranks of (Q - Q_sync), Q_sync <- Q.  n = the handles of THIS process (1 with one process per
// GPU; all of them for a single process driving N GPUs, whose collective calls are issued as one RCCL group).
extern "C" int32_t gorse_mf_item_allreduce(gorse_mf *const *hs, gorse_comm *const *cs, int32_t n) {
    GORSE_TRY(check_pairs(hs, cs, n));
    Rccl *R;
    GORSE_TRY(need_rccl(&R));
    std::vector<int> tok((size_t)n, -1);
    for (int i = 0; i < n; i++) {
        gorse_mf *h = hs[i];
        GORSE_TRY(h->use());
        GORSE_TRY(cs[i]->xbuf.ensure((size_t)h->I * h->d));
        GORSE_TRY(mf_delta_export_async(h, cs[i]->xbuf.p));
        tok[(size_t)i] = h->prof.begin(GORSE_PROF_COMM, h->stream);
    }
    GORSE_RCCL_CHECK(R, R->GroupStart());
    for (int i = 0; i < n; i++) {
        gorse_mf *h = hs[i];
        if (hipError_t he = hipSetDevice(h->device); he != hipSuccess) {
            (void)R->GroupEnd();  // the group never stays open on this thread
            return fail(GORSE_ERR_HIP, "hipSetDevice(%d): %s", h->device, hipGetErrorString(he));
        }
        ncclResult_t e = R->AllReduce(cs[i]->xbuf.p, cs[i]->xbuf.p, (size_t)h->I * h->d, ncclFloat32, ncclSum, cs[i]->comm, h->stream);
        if (e != ncclSuccess) {
            (void)R->GroupEnd();
            return fail(GORSE_ERR_HIP, "ncclAllReduce: %s", R->GetErrorString(e));
        }
    }
    GORSE_RCCL_CHECK(R, R->GroupEnd());
    for (int i = 0; i < n; i++) {
        gorse_mf *h = hs[i];
        GORSE_TRY(h->use());
        h->prof.end(tok[(size_t)i], h->stream);
        GORSE_TRY(mf_delta_import_async(h, cs[i]->xbuf.p));
    }
    return GORSE_OK;
}

// ALS: after a half-sweep every rank owns rows [row_splits[r], row_splits[r + 1]) of side 0 (P) / 1 (Q); each block is
// broadcast from its owner straight into the replicas' factor matrices (an all-gather with uneven blocks and no staging).
extern "C" int32_t gorse_mf_rows_allgather(gorse_mf *const *hs, gorse_comm *const *cs, int32_t n, int32_t side,
                                           const int64_t *row_splits) {
    GORSE_TRY(check_pairs(hs, cs, n));
    if (side != 0 && side != 1) return fail(GORSE_ERR_INVALID, "side must be 0 (users) or 1 (items)");
    if (!row_splits) return fail(GORSE_ERR_INVALID, "row_splits is NULL");
    const int world = cs[0]->world;
    const int64_t rows = side == 0 ? hs[0]->U : hs[0]->I;
    if (row_splits[0] != 0 || row_splits[world] != rows) return fail(GORSE_ERR_RANGE, "row_splits must run from 0 to %lld", (long long)rows);
    for (int r = 0; r < world; r++)
        if (row_splits[r + 1] < row_splits[r]) return fail(GORSE_ERR_RANGE, "row_splits decreases at %d", r);
    Rccl *R;
    GORSE_TRY(need_rccl(&R));
    std::vector<int> tok((size_t)n, -1);
    for (int i = 0; i < n; i++) {
        GORSE_TRY(hs[i]->use());
        tok[(size_t)i] = hs[i]->prof.begin(GORSE_PROF_COMM, hs[i]->stream);
    }
    GORSE_RCCL_CHECK(R, R->GroupStart());
    for (int i = 0; i < n; i++) {
        gorse_mf *h = hs[i];
        if (hipError_t he = hipSetDevice(h->device); he != hipSuccess) {
            (void)R->GroupEnd();
            return fail(GORSE_ERR_HIP, "hipSetDevice(%d): %s", h->device, hipGetErrorString(he));
        }
        float *base = side == 0 ? h->P.p : h->Q.p;
        for (int r = 0; r < world; r++) {
            const int64_t lo = row_splits[r], cnt = (row_splits[r + 1] - lo) * h->d;
            if (cnt == 0) continue;
            ncclResult_t e = R->Broadcast(base + lo * h->d, base + lo * h->d, (size_t)cnt, ncclFloat32, r, cs[i]->comm, h->stream);
            if (e != ncclSuccess) {
                (void)R->GroupEnd();
                return fail(GORSE_ERR_HIP, "ncclBroadcast: %s", R->GetErrorString(e));
            }
        }
    }
    GORSE_RCCL_CHECK(R, R->GroupEnd());
    for (int i = 0; i < n; i++) {
        GORSE_TRY(hs[i]->use());
        hs[i]->prof.end(tok[(size_t)i], hs[i]->stream);
    }
    return GORSE_OK;
}

// a few floats summed over the ranks, host to host (the metric partial sums of a sharded Evaluate, evaluator.go:56-70)
extern "C" int32_t gorse_comm_allreduce_f32(gorse_comm *c, float *buf, int64_t n) {
    if (!c || !buf || n < 0) return fail(GORSE_ERR_INVALID, "bad arguments");
    if (n == 0) return GORSE_OK;
    Rccl *R;
    GORSE_TRY(need_rccl(&R));
    GORSE_HIP_CHECK(hipSetDevice(c->device));
    DevBuf<float> tmp;
    GORSE_TRY(tmp.alloc((size_t)n));
    GORSE_HIP_CHECK(hipMemcpy(tmp.p, buf, (size_t)n * 4, hipMemcpyHostToDevice));
    GORSE_RCCL_CHECK(R, R->AllReduce(tmp.p, tmp.p, (size_t)n, ncclFloat32, ncclSum, c->comm, nullptr));
    GORSE_HIP_CHECK(hipStreamSynchronize(nullptr));
    GORSE_HIP_CHECK(hipMemcpy(buf, tmp.p, (size_t)n * 4, hipMemcpyDeviceToHost));
    return GORSE_OK;
}

// The same for ALL ranks of a single process (gorse_comm_create_local): the one-rank call above blocks until every rank has
// called it, so a process that owns several ranks must issue them as one group.  bufs[i] = rank i's n floats, summed in place.
extern "C" int32_t gorse_comm_allreduce_f32_local(gorse_comm *const *cs, int32_t n_comms, float *const *bufs, int64_t n) {
    if (!cs || !bufs || n_comms < 1 || n < 0) return fail(GORSE_ERR_INVALID, "bad arguments");
    for (int i = 0; i < n_comms; i++)
        if (!cs[i] || !bufs[i]) return fail(GORSE_ERR_INVALID, "NULL communicator or buffer at %d", i);
    if (n == 0) return GORSE_OK;
    Rccl *R;
    GORSE_TRY(need_rccl(&R));
    std::vector<DevBuf<float>> tmp((size_t)n_comms);
    for (int i = 0; i < n_comms; i++) {
        GORSE_HIP_CHECK(hipSetDevice(cs[i]->device));
        GORSE_TRY(tmp[(size_t)i].alloc((size_t)n));
        GORSE_HIP_CHECK(hipMemcpy(tmp[(size_t)i].p, bufs[i], (size_t)n * 4, hipMemcpyHostToDevice));
    }
    GORSE_RCCL_CHECK(R, R->GroupStart());
    for (int i = 0; i < n_comms; i++) {
        ncclResult_t e = hipSetDevice(cs[i]->device) == hipSuccess
                             ? R->AllReduce(tmp[(size_t)i].p, tmp[(size_t)i].p, (size_t)n, ncclFloat32, ncclSum, cs[i]->comm, nullptr)
                             : ncclUnhandledCudaError;
        if (e != ncclSuccess) {
            (void)R->GroupEnd();
            return fail(GORSE_ERR_HIP, "ncclAllReduce: %s", R->GetErrorString(e));
        }
    }
    GORSE_RCCL_CHECK(R, R->GroupEnd());
    for (int i = 0; i < n_comms; i++) {
        GORSE_HIP_CHECK(hipSetDevice(cs[i]->device));
        GORSE_HIP_CHECK(hipStreamSynchronize(nullptr));
        GORSE_HIP_CHECK(hipMemcpy(bufs[i], tmp[(size_t)i].p, (size_t)n * 4, hipMemcpyDeviceToHost));
    }
    return GORSE_OK;
}

// can this process open RCCL at all?  Every rank asks BEFORE the ranks meet inside gorse_comm_create (a collective
// initialisation: a rank that cannot load the library would leave the others waiting there) and the answers are agreed on by
// whatever carried the unique id.
extern "C" int32_t gorse_comm_available(void) {
    Rccl *R;
    return need_rccl(&R);
}
