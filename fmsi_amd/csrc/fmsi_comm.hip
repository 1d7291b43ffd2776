// Multi-GPU result gather over RCCL: one process per GPU, each rank answers a contiguous,
// order-preserving range of the windows, and the per-rank result buffers go to the root over the direct xGMI link of
// each peer -- rank order = output order, so the concatenation is the single-GPU byte stream.  There is no exchange
// step inside the search; this is the only collective of the path.
//
// RCCL is opened with dlopen() when the first communicator is made: the library proper (and the single-GPU CLI) do not
// link against it, and a process that already holds an RCCL (torch.distributed's, same soname) shares that copy.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "index_priv.h"

namespace {

struct Rccl {
    void* handle = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
};

Rccl g_rccl;
std::mutex g_rccl_mutex;

template <typename F>
bool resolve(void* h, const char* name, F& fn) {
    fn = reinterpret_cast<F>(dlsym(h, name));
    return fn != nullptr;
}

int load_rccl() {
    std::lock_guard<std::mutex> lock(g_rccl_mutex);
    if (g_rccl.handle) return 0;
    void* h = nullptr;
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (h) break;
    }
    if (!h) {
        fmsi::set_last_error(std::string("cannot open librccl: ") + dlerror());
        return FMSI_HIP_ECOMM;
    }
    Rccl r;
    if (!(resolve(h, "ncclGetUniqueId", r.GetUniqueId) && resolve(h, "ncclCommInitRank", r.CommInitRank) &&
          resolve(h, "ncclCommDestroy", r.CommDestroy) && resolve(h, "ncclGroupStart", r.GroupStart) &&
          resolve(h, "ncclGroupEnd", r.GroupEnd) && resolve(h, "ncclSend", r.Send) && resolve(h, "ncclRecv", r.Recv) &&
          resolve(h, "ncclGetErrorString", r.GetErrorString))) {
        fmsi::set_last_error("librccl lacks an expected symbol");
        dlclose(h);
        return FMSI_HIP_ECOMM;
    }
    r.handle = h;
    g_rccl = r;
    return 0;
}

#define FMSI_RCCL_TRY(expr)                                                                       \
    do {                                                                                          \
        ncclResult_t r_ = (expr);                                                                 \
        if (r_ != ncclSuccess) {                                                                  \
            fmsi::set_last_error(std::string(#expr) + ": " + g_rccl.GetErrorString(r_));          \
            return FMSI_HIP_ECOMM;                                                                \
        }                                                                                         \
    } while (0)

}  // namespace

struct fmsi_hip_comm {
    ncclComm_t comm = nullptr;
    int world = 0;
    int rank = 0;
    int device = 0;
};

extern "C" {

int fmsi_hip_comm_unique_id(uint8_t* id) {
    if (!id) {
        fmsi::set_last_error("null argument");
        return FMSI_HIP_EINVAL;
    }
    int rc = load_rccl();
    if (rc) return rc;
    static_assert(sizeof(ncclUniqueId) == FMSI_HIP_COMM_ID_BYTES, "RCCL unique id size");
    ncclUniqueId u;
    FMSI_RCCL_TRY(g_rccl.GetUniqueId(&u));
    std::memcpy(id, &u, sizeof(u));
    return 0;
}

int fmsi_hip_comm_create(const uint8_t* id, int world, int rank, int device, fmsi_hip_comm** out) {
    if (!id || !out || world < 1 || rank < 0 || rank >= world) {
        fmsi::set_last_error("fmsi_hip_comm_create: bad argument");
        return FMSI_HIP_EINVAL;
    }
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || device < 0 || device >= n_dev) {
        fmsi::set_last_error("no such HIP device");
        return FMSI_HIP_ENODEV;
    }
    int rc = load_rccl();
    if (rc) return rc;
    FMSI_HIP_ON_DEVICE(device);
    ncclUniqueId u;
    std::memcpy(&u, id, sizeof(u));
    ncclComm_t comm = nullptr;
    FMSI_RCCL_TRY(g_rccl.CommInitRank(&comm, world, u, rank));
    auto* c = new fmsi_hip_comm;
    c->comm = comm;
    c->world = world;
    c->rank = rank;
    c->device = device;
    *out = c;
    return 0;
}

void fmsi_hip_comm_destroy(fmsi_hip_comm* comm) {
    if (!comm) return;
    if (comm->comm && g_rccl.handle) {
        ::fmsi::DeviceGuard device_guard_(comm->device);
        g_rccl.CommDestroy(comm->comm);
    }
    delete comm;
}

int fmsi_hip_gather(fmsi_hip_comm* comm, const void* d_send, uint64_t send_bytes, void* d_recv, const uint64_t* bytes_per_rank,
                    int root, void* stream) {
    if (!comm || !bytes_per_rank || (send_bytes && !d_send)) {
        fmsi::set_last_error("fmsi_hip_gather: bad argument");
        return FMSI_HIP_EINVAL;
    }
    if (root < 0 || root >= comm->world) {
        fmsi::set_last_error("fmsi_hip_gather: root is not a rank of the communicator");
        return FMSI_HIP_EINVAL;
    }
    if (bytes_per_rank[comm->rank] != send_bytes) {
        fmsi::set_last_error("fmsi_hip_gather: send_bytes differs from this rank's entry of bytes_per_rank");
        return FMSI_HIP_EINVAL;
    }
    FMSI_HIP_ON_DEVICE(comm->device);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (comm->rank != root) {
        if (send_bytes) FMSI_RCCL_TRY(g_rccl.Send(d_send, send_bytes, ncclUint8, root, comm->comm, s));
        return 0;
    }
    uint64_t total = 0;
    for (int r = 0; r < comm->world; ++r) total += bytes_per_rank[r];
    if (total && !d_recv) {
        fmsi::set_last_error("fmsi_hip_gather: the root needs a receive buffer");
        return FMSI_HIP_EINVAL;
    }
    // one grouped round of receives, each peer's bytes straight to its place in the output order
    char* dst = static_cast<char*>(d_recv);
    FMSI_RCCL_TRY(g_rccl.GroupStart());
    uint64_t off = 0;
    for (int r = 0; r < comm->world; ++r) {
        if (r != root && bytes_per_rank[r]) {
            ncclResult_t e = g_rccl.Recv(dst + off, bytes_per_rank[r], ncclUint8, r, comm->comm, s);
            if (e != ncclSuccess) {
                g_rccl.GroupEnd();
                fmsi::set_last_error(std::string("ncclRecv: ") + g_rccl.GetErrorString(e));
                return FMSI_HIP_ECOMM;
            }
        }
        off += bytes_per_rank[r];
    }
    FMSI_RCCL_TRY(g_rccl.GroupEnd());
    // the root's own results: a device-to-device copy on the same stream
    off = 0;
    for (int r = 0; r < root; ++r) off += bytes_per_rank[r];
    if (send_bytes && dst + off != d_send)
        FMSI_HIP_TRY(hipMemcpyAsync(dst + off, d_send, send_bytes, hipMemcpyDeviceToDevice, s));
    return 0;
}

}  // extern "C"
