// pa_comm: the one collective of the path -- the final all-gather of per-realisation observables over
// RCCL (xGMI inside a node), one communicator rank per GPU / process.
//
// The reference has no multi-process level: its outer rayon loop runs the realisations of one process
// (spin-sim/src/simulation/mod.rs:887-903) and aggregates them serially (statistics/results.rs:165-282).
// Here rank g of G owns a contiguous block of realisations (seeds by global index, pa_sim_desc), runs with
// no data-path collective, and this file gathers the fixed-size result blocks so that every rank can
// apply the reference's aggregation in realisation order -- bit-identical to a single-GPU run.
//
// RCCL is bound at run time (dlopen), so a single-GPU installation without librccl still loads the
// library; every entry point then fails with PA_ERR_DEVICE and a message.  No torch, no MPI: the 128-byte
// unique id is created by one rank (pa_comm_unique_id) and handed to the others by the caller (any
// channel: a file, a socket, an environment variable).
#include "../../include/peapods_amd.h"

#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

namespace {

struct Rccl {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Broadcast)(const void *, void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    std::string error;
};

std::mutex g_rccl_mutex;
Rccl g_rccl;

// Resolves the RCCL entry points once per process.
const Rccl *rccl(std::string &err) {
    std::lock_guard<std::mutex> lock(g_rccl_mutex);
    if (g_rccl.handle) return &g_rccl;
    if (!g_rccl.error.empty()) {
        err = g_rccl.error;
        return nullptr;
    }
    for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
        g_rccl.handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
        if (g_rccl.handle) break;
    }
    if (!g_rccl.handle) {
        g_rccl.error = std::string("librccl.so not found: ") + dlerror();
        err = g_rccl.error;
        return nullptr;
    }
    auto sym = [&](const char *name) {
        void *p = dlsym(g_rccl.handle, name);
        if (!p && g_rccl.error.empty()) g_rccl.error = std::string("librccl.so lacks ") + name;
        return p;
    };
    g_rccl.GetUniqueId = reinterpret_cast<decltype(g_rccl.GetUniqueId)>(sym("ncclGetUniqueId"));
    g_rccl.CommInitRank = reinterpret_cast<decltype(g_rccl.CommInitRank)>(sym("ncclCommInitRank"));
    g_rccl.CommDestroy = reinterpret_cast<decltype(g_rccl.CommDestroy)>(sym("ncclCommDestroy"));
    g_rccl.AllGather = reinterpret_cast<decltype(g_rccl.AllGather)>(sym("ncclAllGather"));
    g_rccl.AllReduce = reinterpret_cast<decltype(g_rccl.AllReduce)>(sym("ncclAllReduce"));
    g_rccl.Broadcast = reinterpret_cast<decltype(g_rccl.Broadcast)>(sym("ncclBroadcast"));
    g_rccl.Send = reinterpret_cast<decltype(g_rccl.Send)>(sym("ncclSend"));
    g_rccl.Recv = reinterpret_cast<decltype(g_rccl.Recv)>(sym("ncclRecv"));
    g_rccl.GroupStart = reinterpret_cast<decltype(g_rccl.GroupStart)>(sym("ncclGroupStart"));
    g_rccl.GroupEnd = reinterpret_cast<decltype(g_rccl.GroupEnd)>(sym("ncclGroupEnd"));
    g_rccl.GetErrorString = reinterpret_cast<decltype(g_rccl.GetErrorString)>(sym("ncclGetErrorString"));
    if (!g_rccl.error.empty()) {
        dlclose(g_rccl.handle);
        g_rccl.handle = nullptr;
        err = g_rccl.error;
        return nullptr;
    }
    return &g_rccl;
}

void write_err(char *err, size_t len, const std::string &msg) {
    if (err && len) std::snprintf(err, len, "%s", msg.c_str());
}

} // namespace

struct pa_comm {
    const Rccl *api = nullptr;
    ncclComm_t comm = nullptr;
    int n_ranks = 1, rank = 0, device = 0;
    hipStream_t stream = nullptr;
    void *send = nullptr, *recv = nullptr; // device staging buffers, grown on demand
    size_t send_cap = 0, recv_cap = 0;
    std::string error;
};

namespace {

int comm_fail(pa_comm *c, const std::string &msg) {
    c->error = msg;
    return PA_ERR_DEVICE;
}

#define PA_COMM_HIP(c, expr)                                                                       \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess) return comm_fail(c, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)
#define PA_COMM_NCCL(c, expr)                                                                      \
    do {                                                                                           \
        ncclResult_t r_ = (expr);                                                                  \
        if (r_ != ncclSuccess) return comm_fail(c, std::string(#expr) + ": " + (c)->api->GetErrorString(r_)); \
    } while (0)

int reserve(pa_comm *c, size_t send_bytes, size_t recv_bytes) {
    if (send_bytes > c->send_cap) {
        if (c->send) (void)hipFree(c->send);
        c->send = nullptr;
        c->send_cap = 0;
        PA_COMM_HIP(c, hipMalloc(&c->send, send_bytes));
        c->send_cap = send_bytes;
    }
    if (recv_bytes > c->recv_cap) {
        if (c->recv) (void)hipFree(c->recv);
        c->recv = nullptr;
        c->recv_cap = 0;
        PA_COMM_HIP(c, hipMalloc(&c->recv, recv_bytes));
        c->recv_cap = recv_bytes;
    }
    return PA_OK;
}

} // namespace

extern "C" {

int pa_comm_unique_id(uint8_t id_out[PA_COMM_ID_BYTES], char *err, size_t err_len) {
    static_assert(PA_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "id size");
    std::string msg;
    const Rccl *api = rccl(msg);
    if (!api) {
        write_err(err, err_len, msg);
        return PA_ERR_DEVICE;
    }
    ncclUniqueId id;
    const ncclResult_t r = api->GetUniqueId(&id);
    if (r != ncclSuccess) {
        write_err(err, err_len, std::string("ncclGetUniqueId: ") + api->GetErrorString(r));
        return PA_ERR_DEVICE;
    }
    std::memcpy(id_out, id.internal, PA_COMM_ID_BYTES);
    return PA_OK;
}

int pa_comm_init_rank(const uint8_t id[PA_COMM_ID_BYTES], int32_t n_ranks, int32_t rank, int32_t device,
                      pa_comm **out, char *err, size_t err_len) {
    *out = nullptr;
    if (n_ranks < 1 || rank < 0 || rank >= n_ranks) {
        write_err(err, err_len, "rank out of range");
        return PA_ERR_INVALID;
    }
    std::string msg;
    const Rccl *api = rccl(msg);
    if (!api) {
        write_err(err, err_len, msg);
        return PA_ERR_DEVICE;
    }
    int n_devices = 0;
    if (hipGetDeviceCount(&n_devices) != hipSuccess || device < 0 || device >= n_devices) {
        write_err(err, err_len, "HIP device ordinal out of range");
        return PA_ERR_DEVICE;
    }
    pa_comm *c = new pa_comm();
    c->api = api;
    c->n_ranks = n_ranks;
    c->rank = rank;
    c->device = device;
    auto bail = [&](const std::string &m) {
        write_err(err, err_len, m);
        pa_comm_destroy(c);
        return PA_ERR_DEVICE;
    };
    if (hipSetDevice(device) != hipSuccess) return bail("hipSetDevice failed");
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) return bail("hipStreamCreate failed");
    ncclUniqueId uid;
    std::memcpy(uid.internal, id, PA_COMM_ID_BYTES);
    const ncclResult_t r = api->CommInitRank(&c->comm, n_ranks, uid, rank);
    if (r != ncclSuccess) {
        c->comm = nullptr;
        return bail(std::string("ncclCommInitRank: ") + api->GetErrorString(r));
    }
    *out = c;
    return PA_OK;
}

void pa_comm_destroy(pa_comm *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->comm) (void)c->api->CommDestroy(c->comm);
    if (c->send) (void)hipFree(c->send);
    if (c->recv) (void)hipFree(c->recv);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

const char *pa_comm_last_error(const pa_comm *c) { return c ? c->error.c_str() : ""; }
int32_t pa_comm_n_ranks(const pa_comm *c) { return c->n_ranks; }
int32_t pa_comm_rank(const pa_comm *c) { return c->rank; }

int pa_comm_allgather(pa_comm *c, const void *send_host, void *recv_host, uint64_t bytes_per_rank) {
    if (bytes_per_rank == 0) return PA_OK;
    PA_COMM_HIP(c, hipSetDevice(c->device));
    if (int rc = reserve(c, bytes_per_rank, bytes_per_rank * c->n_ranks)) return rc;
    PA_COMM_HIP(c, hipMemcpyAsync(c->send, send_host, bytes_per_rank, hipMemcpyHostToDevice, c->stream));
    PA_COMM_NCCL(c, c->api->AllGather(c->send, c->recv, bytes_per_rank, ncclUint8, c->comm, c->stream));
    PA_COMM_HIP(c, hipMemcpyAsync(recv_host, c->recv, bytes_per_rank * c->n_ranks, hipMemcpyDeviceToHost, c->stream));
    PA_COMM_HIP(c, hipStreamSynchronize(c->stream));
    return PA_OK;
}

static int allreduce(pa_comm *c, void *inout_host, uint64_t count, size_t elem, ncclDataType_t type, ncclRedOp_t op) {
    if (count == 0) return PA_OK;
    PA_COMM_HIP(c, hipSetDevice(c->device));
    if (int rc = reserve(c, count * elem, count * elem)) return rc;
    PA_COMM_HIP(c, hipMemcpyAsync(c->send, inout_host, count * elem, hipMemcpyHostToDevice, c->stream));
    PA_COMM_NCCL(c, c->api->AllReduce(c->send, c->recv, count, type, op, c->comm, c->stream));
    PA_COMM_HIP(c, hipMemcpyAsync(inout_host, c->recv, count * elem, hipMemcpyDeviceToHost, c->stream));
    PA_COMM_HIP(c, hipStreamSynchronize(c->stream));
    return PA_OK;
}

int pa_comm_allreduce_max_f64(pa_comm *c, double *inout, uint64_t count) {
    return allreduce(c, inout, count, sizeof(double), ncclFloat64, ncclMax);
}

int pa_comm_allreduce_sum_u64(pa_comm *c, uint64_t *inout, uint64_t count) {
    return allreduce(c, inout, count, sizeof(uint64_t), ncclUint64, ncclSum);
}

int pa_comm_barrier(pa_comm *c) {
    uint64_t one = 1;
    return pa_comm_allreduce_sum_u64(c, &one, 1);
}

// Variable-length gather of host bytes in rank order.  Every rank learns all sizes (one 8-byte all-gather), then
// exactly the bytes each rank holds travel once: to `root` by ncclSend / ncclRecv, or to everybody by one ncclBroadcast
// per contributing rank, all inside one group call (xGMI links are point to point, so the per-peer transfers of a
// group run side by side).  The result arrays of a sharded run -- P(q) histograms above all: 2.4 MB per realisation
// at 16^3 x 24 temperatures -- go through here (distributed.sample).
int pa_comm_gatherv(pa_comm *c, const void *send_host, uint64_t send_bytes, void *recv_host, uint64_t recv_capacity,
                    int32_t root, uint64_t *bytes_per_rank_out) {
    if (root >= c->n_ranks) return comm_fail(c, "gather root out of range");
    // every rank learns every rank's size AND receive capacity, so the decision to go on is the same everywhere: a
    // receiver whose buffer is too small fails the call on all ranks before anyone posts a send
    const uint64_t mine[2] = {send_bytes, recv_capacity};
    std::vector<uint64_t> all(2 * static_cast<size_t>(c->n_ranks), 0);
    if (int rc = pa_comm_allgather(c, mine, all.data(), sizeof(mine))) return rc;
    std::vector<uint64_t> sizes(c->n_ranks, 0);
    for (int r = 0; r < c->n_ranks; ++r) sizes[r] = all[2 * r];
    if (bytes_per_rank_out) std::memcpy(bytes_per_rank_out, sizes.data(), sizes.size() * sizeof(uint64_t));
    uint64_t total = 0;
    std::vector<uint64_t> offset(c->n_ranks, 0);
    for (int r = 0; r < c->n_ranks; ++r) {
        offset[r] = total;
        total += sizes[r];
    }
    const bool receives = root < 0 || root == c->rank;
    for (int r = 0; r < c->n_ranks; ++r)
        if ((root < 0 || root == r) && total > all[2 * r + 1])
            return comm_fail(c, "receive buffer of rank " + std::to_string(r) + " smaller than the gathered bytes");
    if (total == 0) return PA_OK;
    PA_COMM_HIP(c, hipSetDevice(c->device));
    if (int rc = reserve(c, std::max<uint64_t>(send_bytes, 1), receives ? total : 1)) return rc;
    if (send_bytes) PA_COMM_HIP(c, hipMemcpyAsync(c->send, send_host, send_bytes, hipMemcpyHostToDevice, c->stream));
    char *recv = static_cast<char *>(c->recv);
    PA_COMM_NCCL(c, c->api->GroupStart());
    ncclResult_t first_error = ncclSuccess;
    auto note = [&](ncclResult_t r) { if (r != ncclSuccess && first_error == ncclSuccess) first_error = r; };
    for (int r = 0; r < c->n_ranks; ++r) {
        if (sizes[r] == 0) continue;
        if (root < 0) {
            note(c->api->Broadcast(c->send, recv + offset[r], sizes[r], ncclUint8, r, c->comm, c->stream));
        } else if (c->rank == root) {
            if (r != root) note(c->api->Recv(recv + offset[r], sizes[r], ncclUint8, r, c->comm, c->stream));
        } else if (r == c->rank) {
            note(c->api->Send(c->send, sizes[r], ncclUint8, root, c->comm, c->stream));
        }
    }
    PA_COMM_NCCL(c, c->api->GroupEnd()); // (always closed, also after a failed call inside the group)
    PA_COMM_NCCL(c, first_error);
    if (root >= 0 && c->rank == root && send_bytes)
        PA_COMM_HIP(c, hipMemcpyAsync(recv + offset[root], c->send, send_bytes, hipMemcpyDeviceToDevice, c->stream));
    if (receives) PA_COMM_HIP(c, hipMemcpyAsync(recv_host, c->recv, total, hipMemcpyDeviceToHost, c->stream));
    PA_COMM_HIP(c, hipStreamSynchronize(c->stream));
    return PA_OK;
}

int pa_comm_gather_result_blocks(pa_comm *c, pa_sim *sim, double *out, uint64_t out_capacity,
                                 uint64_t *n_total_out) {
    const uint64_t n_local = pa_sim_n_realizations(sim);
    // ranks may own different numbers of realisations: exchange the counts, gather blocks padded to the
    // largest, then compact in rank order (= global realisation order for contiguous shards)
    const uint64_t len = pa_sim_result_block_len(sim);
    std::vector<uint64_t> counts(c->n_ranks, 0);
    if (int rc = pa_comm_allgather(c, &n_local, counts.data(), sizeof(uint64_t))) return rc;
    const uint64_t n_max = *std::max_element(counts.begin(), counts.end());
    uint64_t n_total = 0;
    for (uint64_t k : counts) n_total += k;
    if (n_total_out) *n_total_out = n_total;
    if (n_total > out_capacity) return comm_fail(c, "output buffer holds fewer blocks than the communicator gathers");
    std::vector<double> mine(n_max * len, 0.0), all(n_max * len * c->n_ranks);
    if (n_local) {
        const int rc = pa_sim_get_result_blocks(sim, mine.data());
        if (rc) return comm_fail(c, std::string("pa_sim_get_result_blocks: ") + pa_sim_last_error(sim));
    }
    if (int rc = pa_comm_allgather(c, mine.data(), all.data(), n_max * len * sizeof(double))) return rc;
    uint64_t at = 0;
    for (int r = 0; r < c->n_ranks; ++r) {
        std::memcpy(out + at * len, all.data() + static_cast<size_t>(r) * n_max * len, counts[r] * len * sizeof(double));
        at += counts[r];
    }
    return PA_OK;
}

} // extern "C"
