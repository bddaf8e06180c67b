// Replica fan-out over RCCL (BASELINE.json north_star: "batch-sharded across the 8 GPUs of one node with RCCL over
// xGMI only for multi-request fan-out").  One communicator per process (= per GPU); nothing on the per-token path crosses GPUs, so the
// collectives here are the start-up weight broadcast (1.28 GB, once), the packed prompt batch (<= 3.5 MB), the end-of-run code
// all-gather (<= 0.3 MB per rank) and scalar reductions for the job's clock.  librccl is bound with dlopen at the first fs_comm_* call:
// libfishrt.so keeps no link-time dependency on it (single-GPU hosts never load it).  The reference has no distributed layer (SURVEY.md
// section 2a): this is the C-ABI form of what a fish_speech_core-shaped Rust host needs in place of its `Arc<Mutex<model>>` state
// (server/lib/state.rs:12-29) when it runs one replica per GPU.
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

namespace fs {

class LMBase;

constexpr int COMM_ID_BYTES = 128;  // == NCCL_UNIQUE_ID_BYTES (checked against rccl.h in fs_comm.cpp)

class Comm {
  public:
    // rank 0 makes the id (ncclGetUniqueId) and hands it to the other ranks by whatever channel the host has (env, file, TCP store)
    static void unique_id(uint8_t out[COMM_ID_BYTES]);
    Comm(const uint8_t id[COMM_ID_BYTES], int rank, int world, int device);
    ~Comm();
    Comm(const Comm&) = delete;
    Comm& operator=(const Comm&) = delete;

    int rank() const { return rank_; }
    int world() const { return world_; }
    void barrier();
    // op: 0 sum, 1 max, 2 min; in place on a host array (staged through the device: the values are a handful of doubles)
    void all_reduce_f64(double* vals, int n, int op);
    // host buffers, staged through a device buffer owned by the communicator
    void broadcast_host(void* buf, size_t bytes, int src);
    void all_gather_host(const void* send, void* recv, size_t bytes_per_rank);
    // device memory in place (the weight arena), in pieces of `chunk` bytes; returns the bytes moved
    size_t broadcast_device(void* dev, size_t bytes, int src, size_t chunk = (size_t)256 << 20);
    // rank `src` holds a loaded handle; every other rank's handle (same model args / dtype, not loaded) receives the
    // arena and adopts it.  Throws on every rank when the arenas differ in size.
    size_t broadcast_weights(LMBase* lm, int src);

  private:
    void* stage(size_t bytes);
    void sync();
    void* comm_ = nullptr;  // ncclComm_t
    hipStream_t st_ = nullptr;
    void* dbuf_ = nullptr;
    size_t dcap_ = 0;
    int rank_ = 0, world_ = 1, device_ = 0;
};

}  // namespace fs
