// fs_comm.h: the replica fan-out on librccl directly (ncclBroadcast / ncclAllGather / ncclAllReduce), bound lazily with dlopen.
#include "fs_comm.h"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "fs_common.h"
#include "lm_engine.h"

namespace fs {

static_assert(COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "fishrt.h's FS_COMM_ID_BYTES must match rccl.h");

namespace {

struct Rccl {
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclBroadcast) Broadcast = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    void* handle = nullptr;
};

const Rccl& rccl() {
    static Rccl R;
    static std::once_flag once;
    static std::string err;
    std::call_once(once, [] {
        // a librccl the process already holds (e.g. the one a PyTorch-ROCm host brought) is returned by SONAME; otherwise the system one
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* n : names) {
            R.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (R.handle) break;
        }
        if (!R.handle) { err = std::string("librccl not found (dlopen: ") + dlerror() + ")"; return; }
        auto sym = [&](const char* n) -> void* {
            void* p = dlsym(R.handle, n);
            if (!p && err.empty()) err = std::string("librccl lacks ") + n;
            return p;
        };
        R.GetUniqueId = reinterpret_cast<decltype(R.GetUniqueId)>(sym("ncclGetUniqueId"));
        R.CommInitRank = reinterpret_cast<decltype(R.CommInitRank)>(sym("ncclCommInitRank"));
        R.CommDestroy = reinterpret_cast<decltype(R.CommDestroy)>(sym("ncclCommDestroy"));
        R.Broadcast = reinterpret_cast<decltype(R.Broadcast)>(sym("ncclBroadcast"));
        R.AllGather = reinterpret_cast<decltype(R.AllGather)>(sym("ncclAllGather"));
        R.AllReduce = reinterpret_cast<decltype(R.AllReduce)>(sym("ncclAllReduce"));
        R.GetErrorString = reinterpret_cast<decltype(R.GetErrorString)>(sym("ncclGetErrorString"));
    });
    if (!err.empty()) throw Error("fs_comm: " + err);
    return R;
}

void check(ncclResult_t r, const char* what) {
    if (r != ncclSuccess) throw Error(std::string("fs_comm: ") + what + " failed: " + rccl().GetErrorString(r));
}

}  // namespace

void Comm::unique_id(uint8_t out[COMM_ID_BYTES]) {
    ncclUniqueId id;
    check(rccl().GetUniqueId(&id), "ncclGetUniqueId");
    std::memcpy(out, id.internal, COMM_ID_BYTES);
}

Comm::Comm(const uint8_t id[COMM_ID_BYTES], int rank, int world, int device) : rank_(rank), world_(world), device_(device) {
    FS_REQUIRE(world >= 1 && rank >= 0 && rank < world, "fs_comm: rank outside [0, world)");
    FS_HIP(hipSetDevice(device));
    FS_HIP(hipStreamCreateWithFlags(&st_, hipStreamNonBlocking));
    ncclUniqueId uid;
    std::memcpy(uid.internal, id, COMM_ID_BYTES);
    ncclComm_t c = nullptr;
    check(rccl().CommInitRank(&c, world, uid, rank), "ncclCommInitRank");
    comm_ = c;
}

Comm::~Comm() {
    (void)hipSetDevice(device_);
    if (st_) (void)hipStreamSynchronize(st_);
    if (comm_) (void)rccl().CommDestroy(static_cast<ncclComm_t>(comm_));
    if (dbuf_) (void)hipFree(dbuf_);
    if (st_) (void)hipStreamDestroy(st_);
}

void* Comm::stage(size_t bytes) {
    if (bytes > dcap_) {
        if (dbuf_) FS_HIP(hipFree(dbuf_));
        dbuf_ = nullptr;
        dcap_ = (bytes + 4095) & ~(size_t)4095;
        FS_HIP(hipMalloc(&dbuf_, dcap_));
    }
    return dbuf_;
}
void Comm::sync() { FS_HIP(hipStreamSynchronize(st_)); }

void Comm::all_reduce_f64(double* vals, int n, int op) {
    FS_REQUIRE(vals && n >= 1 && n <= 4096 && op >= 0 && op <= 2, "fs_comm: bad all-reduce arguments");
    FS_HIP(hipSetDevice(device_));
    void* d = stage(sizeof(double) * n);
    FS_HIP(hipMemcpyAsync(d, vals, sizeof(double) * n, hipMemcpyHostToDevice, st_));
    const ncclRedOp_t ops[3] = {ncclSum, ncclMax, ncclMin};
    check(rccl().AllReduce(d, d, (size_t)n, ncclFloat64, ops[op], static_cast<ncclComm_t>(comm_), st_), "ncclAllReduce");
    FS_HIP(hipMemcpyAsync(vals, d, sizeof(double) * n, hipMemcpyDeviceToHost, st_));
    sync();
}
void Comm::barrier() {  // an all-reduce every rank has to enter + the stream sync behind it
    double one = 1.0;
    all_reduce_f64(&one, 1, 0);
    if ((int)(one + 0.5) != world_) throw Error("fs_comm: barrier counted " + std::to_string(one) + " ranks of " + std::to_string(world_));
}

void Comm::broadcast_host(void* buf, size_t bytes, int src) {
    FS_REQUIRE(src >= 0 && src < world_, "fs_comm: broadcast root outside the communicator");
    if (bytes == 0) return;
    FS_REQUIRE(buf, "fs_comm: null buffer");
    FS_HIP(hipSetDevice(device_));
    void* d = stage(bytes);
    if (rank_ == src) FS_HIP(hipMemcpyAsync(d, buf, bytes, hipMemcpyHostToDevice, st_));
    check(rccl().Broadcast(d, d, bytes, ncclUint8, src, static_cast<ncclComm_t>(comm_), st_), "ncclBroadcast");
    if (rank_ != src) FS_HIP(hipMemcpyAsync(buf, d, bytes, hipMemcpyDeviceToHost, st_));
    sync();
}

void Comm::all_gather_host(const void* send, void* recv, size_t bytes_per_rank) {
    if (bytes_per_rank == 0) return;
    FS_REQUIRE(send && recv, "fs_comm: null buffer");
    FS_HIP(hipSetDevice(device_));
    // [world + 1] slots: slot `world` holds this rank's contribution (ncclAllGather's in-place form wants it AT its slot; a separate
    // send buffer keeps the call valid for every rank without aliasing rules)
    unsigned char* d = static_cast<unsigned char*>(stage(bytes_per_rank * (size_t)(world_ + 1)));
    unsigned char* ds = d + bytes_per_rank * (size_t)world_;
    FS_HIP(hipMemcpyAsync(ds, send, bytes_per_rank, hipMemcpyHostToDevice, st_));
    check(rccl().AllGather(ds, d, bytes_per_rank, ncclUint8, static_cast<ncclComm_t>(comm_), st_), "ncclAllGather");
    FS_HIP(hipMemcpyAsync(recv, d, bytes_per_rank * (size_t)world_, hipMemcpyDeviceToHost, st_));
    sync();
}

size_t Comm::broadcast_device(void* dev, size_t bytes, int src, size_t chunk) {
    FS_REQUIRE(src >= 0 && src < world_, "fs_comm: broadcast root outside the communicator");
    FS_REQUIRE(dev || bytes == 0, "fs_comm: null device buffer");
    FS_REQUIRE(chunk >= 1, "fs_comm: zero chunk");
    FS_HIP(hipSetDevice(device_));
    unsigned char* p = static_cast<unsigned char*>(dev);
    // pieces: bounded channel staging inside RCCL, and the launches of consecutive pieces overlap on the stream
    for (size_t o = 0; o < bytes; o += chunk) {
        const size_t n = bytes - o < chunk ? bytes - o : chunk;
        check(rccl().Broadcast(p + o, p + o, n, ncclUint8, src, static_cast<ncclComm_t>(comm_), st_), "ncclBroadcast (weights)");
    }
    sync();
    return bytes;
}

size_t Comm::broadcast_weights(LMBase* lm, int src) {
    FS_REQUIRE(lm, "fs_comm: null handle");
    void* ptr = nullptr;
    size_t n = 0;
    lm->weights_arena(&ptr, &n);
    // max and -min in one reduction: EVERY rank sees a mismatch (and throws) before any byte moves
    double mm[2] = {(double)n, -(double)n};
    all_reduce_f64(mm, 2, 1);
    if (mm[0] != -mm[1])
        throw Error("fs_comm: weight arenas differ across ranks (" + std::to_string((size_t)-mm[1]) + " .. " + std::to_string((size_t)mm[0]) + " bytes, " +
                    std::to_string(n) + " here): same model args and dtype on every rank");
    FS_HIP(hipDeviceSynchronize());  // the source's loader / the receivers' allocation ran on other streams
    broadcast_device(ptr, n, src);
    if (rank_ != src) lm->weights_adopt();
    return n;
}

}  // namespace fs
