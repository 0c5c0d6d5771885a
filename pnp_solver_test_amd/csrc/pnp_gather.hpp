// Optional exchange of result arrays between the ranks of a sharded solve, over RCCL.  The path
// itself has no collective (problems are independent, every rank solves its own slice); this is
// for consumers that want the poses of all ranks in one place.  RCCL is opened at run time, on the
// first call that needs it, so that the library loads on hosts without it.
// Included by pnp_hip.hip inside its unnamed namespace (uses fail, HIP_TRY, g_err).
#pragma once

#include <dlfcn.h>

#include <type_traits>

// Where rccl.h is installed (pnp_hip.hip includes it at file scope and defines PNP_HAVE_RCCL_H) the
// table below is typed with its own typedefs and every entry is checked, at compile time, against
// the prototype the header declares (argument order and types); nothing of RCCL is linked either
// way.  Without the header the same table is typed by hand: ncclResult_t and ncclDataType_t are
// int-sized enums (0 = success, ncclChar = 0), ncclComm_t an opaque pointer, ncclUniqueId 128
// opaque bytes passed by value.
struct Rccl {
#ifdef PNP_HAVE_RCCL_H
    using UniqueId = ncclUniqueId;
    using Result = ncclResult_t;
    using Comm = ncclComm_t;
    using DataType = ncclDataType_t;
    static constexpr DataType kChar = ncclChar;
#else
    struct UniqueId {
        char internal[128];
    };
    using Result = int;
    using Comm = void *;
    using DataType = int;
    static constexpr DataType kChar = 0;
#endif
    Result (*GetUniqueId)(UniqueId *) = nullptr;
    Result (*CommInitRank)(Comm *, int, UniqueId, int) = nullptr;
    Result (*CommDestroy)(Comm) = nullptr;
    Result (*AllGather)(const void *, void *, size_t, DataType, Comm, hipStream_t) = nullptr;
    Result (*Send)(const void *, size_t, DataType, int, Comm, hipStream_t) = nullptr;
    Result (*Recv)(void *, size_t, DataType, int, Comm, hipStream_t) = nullptr;
    Result (*GroupStart)() = nullptr;
    Result (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(Result) = nullptr;
    bool ok = false;
    char why[256] = "";
};

static_assert(sizeof(Rccl::UniqueId) == PNP_COMM_ID_BYTES, "PNP_COMM_ID_BYTES must be the size of ncclUniqueId");
static_assert(sizeof(Rccl::Result) == sizeof(int) && sizeof(Rccl::DataType) == sizeof(int) &&
                  sizeof(Rccl::Comm) == sizeof(void *) && static_cast<int>(Rccl::kChar) == 0,
              "the C ABI passes RCCL's status as int, its communicator as void * and counts bytes as ncclChar");
#ifdef PNP_HAVE_RCCL_H
static_assert(std::is_same<decltype(&ncclGetUniqueId), decltype(Rccl::GetUniqueId)>::value, "ncclGetUniqueId");
static_assert(std::is_same<decltype(&ncclCommInitRank), decltype(Rccl::CommInitRank)>::value, "ncclCommInitRank");
static_assert(std::is_same<decltype(&ncclCommDestroy), decltype(Rccl::CommDestroy)>::value, "ncclCommDestroy");
static_assert(std::is_same<decltype(&ncclAllGather), decltype(Rccl::AllGather)>::value, "ncclAllGather");
static_assert(std::is_same<decltype(&ncclSend), decltype(Rccl::Send)>::value, "ncclSend");
static_assert(std::is_same<decltype(&ncclRecv), decltype(Rccl::Recv)>::value, "ncclRecv");
static_assert(std::is_same<decltype(&ncclGroupStart), decltype(Rccl::GroupStart)>::value, "ncclGroupStart");
static_assert(std::is_same<decltype(&ncclGroupEnd), decltype(Rccl::GroupEnd)>::value, "ncclGroupEnd");
static_assert(std::is_same<decltype(&ncclGetErrorString), decltype(Rccl::GetErrorString)>::value, "ncclGetErrorString");
#endif

// Opens RCCL once per process.  A copy already mapped into the process (PyTorch-ROCm brings its
// own) is the one found, since the lookup is by soname.
inline const Rccl &rccl()
{
    static const Rccl table = [] {
        Rccl r;
        void *h = nullptr;
        for (const char *name : {"librccl.so.1", "librccl.so"}) {
            h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (h) break;
        }
        if (!h) {
            snprintf(r.why, sizeof r.why, "RCCL could not be opened: %s", dlerror());
            return r;
        }
        bool all = true;
        auto sym = [&](const char *name) {
            void *p = dlsym(h, name);
            if (!p) {
                all = false;
                snprintf(r.why, sizeof r.why, "RCCL has no symbol %s", name);
            }
            return p;
        };
        r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
        r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
        r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
        r.AllGather = reinterpret_cast<decltype(r.AllGather)>(sym("ncclAllGather"));
        r.Send = reinterpret_cast<decltype(r.Send)>(sym("ncclSend"));
        r.Recv = reinterpret_cast<decltype(r.Recv)>(sym("ncclRecv"));
        r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(sym("ncclGroupStart"));
        r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(sym("ncclGroupEnd"));
        r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
        r.ok = all;
        return r;
    }();
    return table;
}

#define RCCL_TRY(expr)                                                                                    \
    do {                                                                                                  \
        const Rccl::Result e_ = (expr);                                                                   \
        if (e_ != 0) return fail(PNP_E_RCCL, "%s failed: %s", #expr, rccl().GetErrorString(e_));          \
    } while (0)

inline int rccl_ready()
{
    if (!rccl().ok) return fail(PNP_E_RCCL, "%s", rccl().why);
    return PNP_OK;
}

inline int comm_unique_id_impl(void *id_out)
{
    g_err[0] = 0;
    if (!id_out) return fail(PNP_E_ARG, "id_out is NULL");
    int rc = rccl_ready();
    if (rc) return rc;
    Rccl::UniqueId id;
    RCCL_TRY(rccl().GetUniqueId(&id));
    std::memcpy(id_out, id.internal, sizeof id.internal);
    return PNP_OK;
}

inline int comm_init_impl(const void *id_in, int32_t rank, int32_t world, void **comm_out)
{
    g_err[0] = 0;
    if (!id_in || !comm_out) return fail(PNP_E_ARG, "id and comm_out must not be NULL");
    if (world < 1 || rank < 0 || rank >= world) return fail(PNP_E_ARG, "rank %d outside [0, %d)", rank, world);
    int rc = rccl_ready();
    if (rc) return rc;
    Rccl::UniqueId id;
    std::memcpy(id.internal, id_in, sizeof id.internal);
    Rccl::Comm comm = nullptr;
    RCCL_TRY(rccl().CommInitRank(&comm, world, id, rank));
    *comm_out = comm;
    return PNP_OK;
}

inline int comm_destroy_impl(void *comm)
{
    g_err[0] = 0;
    if (!comm) return PNP_OK;
    int rc = rccl_ready();
    if (rc) return rc;
    RCCL_TRY(rccl().CommDestroy(static_cast<Rccl::Comm>(comm)));
    return PNP_OK;
}

// send: this rank's bytes_per_rank bytes (device); recv: world * bytes_per_rank bytes (device), in
// rank order, needed on every rank (root < 0: all-gather) or on `root` only (gather: the other
// ranks' slices travel over world - 1 different xGMI links into one device, the root copies its
// own slice on the device).
inline int gather_impl(const void *send, void *recv, int64_t bytes_per_rank, int32_t root, int32_t rank, int32_t world,
                       void *comm_handle, hipStream_t s)
{
    const Rccl::Comm comm = static_cast<Rccl::Comm>(comm_handle);
    g_err[0] = 0;
    if (bytes_per_rank < 0) return fail(PNP_E_ARG, "bytes_per_rank = %lld is negative", static_cast<long long>(bytes_per_rank));
    if (world < 1 || rank < 0 || rank >= world) return fail(PNP_E_ARG, "rank %d outside [0, %d)", rank, world);
    if (root >= world) return fail(PNP_E_ARG, "root %d outside [0, %d)", root, world);
    if (!comm) return fail(PNP_E_ARG, "comm is NULL");
    if (bytes_per_rank == 0) return PNP_OK;
    const bool receives = root < 0 || root == rank;
    if (!send || (receives && !recv)) return fail(PNP_E_ARG, "send (and recv on a receiving rank) must not be NULL");
    int rc = rccl_ready();
    if (rc) return rc;
    const size_t n = static_cast<size_t>(bytes_per_rank);
    if (root < 0) {
        RCCL_TRY(rccl().AllGather(send, recv, n, Rccl::kChar, comm, s));
        return PNP_OK;
    }
    RCCL_TRY(rccl().GroupStart());
    if (rank == root) {
        for (int r = 0; r < world; ++r) {
            if (r == rank) continue;
            const Rccl::Result e = rccl().Recv(static_cast<char *>(recv) + n * r, n, Rccl::kChar, r, comm, s);
            if (e != 0) {
                (void)rccl().GroupEnd();
                return fail(PNP_E_RCCL, "ncclRecv from rank %d failed: %s", r, rccl().GetErrorString(e));
            }
        }
    } else {
        const Rccl::Result e = rccl().Send(send, n, Rccl::kChar, root, comm, s);
        if (e != 0) {
            (void)rccl().GroupEnd();
            return fail(PNP_E_RCCL, "ncclSend to rank %d failed: %s", root, rccl().GetErrorString(e));
        }
    }
    RCCL_TRY(rccl().GroupEnd());
    if (rank == root)
        HIP_TRY(hipMemcpyAsync(static_cast<char *>(recv) + n * rank, send, n, hipMemcpyDeviceToDevice, s));
    return PNP_OK;
}
