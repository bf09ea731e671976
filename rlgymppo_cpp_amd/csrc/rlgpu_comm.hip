// rlgpu_comm.hip — the one exchange step of the multi-GPU path, on RCCL directly (include/rlgpu.h "multi-GPU").
//
// The path shards by env: one process per GPU, every rank owns its own arenas, experience and shuffle; parameters and Adam state
// are replicated.  Per optimizer step the ranks exchange ONE all-reduce(sum) of the flat gradient buffer [policy | critic]
// (332 635 fp32 = 1.33 MB at the headline shape) on the learner's stream, then every rank scales by 1 / world inside
// rlgpu_clip_adam_step, clips by the global norm and steps Adam -- so clipping sees what a single learner on the union would.
// A 1.33 MB ring all-reduce over xGMI is latency-bound (7 point-to-point links per GPU, ~10-20 us per step of the ring); RCCL
// picks its low-latency protocol for this size.  The rendezvous (handing rank 0's ncclUniqueId to the other ranks) is a file
// next to the launcher's MASTER_PORT -- control plane only, 128 bytes once.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <unistd.h>
#include "../../include/rlgpu.h"

struct rlgpu_comm {
    ncclComm_t comm = nullptr;
    int device = 0, rank = 0, world = 1;
    std::string err;
};

namespace {
std::string g_comm_err;
int fail(rlgpu_comm* c, const std::string& m) { if (c) c->err = m; g_comm_err = m; return RLGPU_ERR_HIP; }
}  // namespace

extern "C" {

const char* rlgpu_comm_last_error(const rlgpu_comm* c) { return c ? c->err.c_str() : g_comm_err.c_str(); }

int rlgpu_comm_unique_id(void* id_out) {
    static_assert(sizeof(ncclUniqueId) == RLGPU_COMM_ID_BYTES, "RLGPU_COMM_ID_BYTES must hold an ncclUniqueId");
    ncclUniqueId id;
    ncclResult_t r = ncclGetUniqueId(&id);
    if (r != ncclSuccess) return fail(nullptr, std::string("ncclGetUniqueId: ") + ncclGetErrorString(r));
    memcpy(id_out, &id, sizeof(id));
    return RLGPU_OK;
}

int rlgpu_comm_init(rlgpu_comm** out, int device, int rank, int world, const void* id_bytes) {
    if (!out || !id_bytes || world < 1 || rank < 0 || rank >= world) return RLGPU_ERR_ARG;
    rlgpu_comm* c = new rlgpu_comm();
    c->device = device; c->rank = rank; c->world = world;
    if (hipSetDevice(device) != hipSuccess) { delete c; return fail(nullptr, "hipSetDevice failed"); }
    ncclUniqueId id; memcpy(&id, id_bytes, sizeof(id));
    ncclResult_t r = ncclCommInitRank(&c->comm, world, id, rank);
    if (r != ncclSuccess) { std::string m = std::string("ncclCommInitRank: ") + ncclGetErrorString(r); delete c; return fail(nullptr, m); }
    *out = c;
    return RLGPU_OK;
}

// rank / world / rendezvous from the launcher's environment (torchrun or any launcher that sets RANK, WORLD_SIZE, LOCAL_RANK,
// MASTER_PORT): rank 0 writes its id to $RLGPU_COMM_DIR (default /tmp)/rlgpu_comm_<MASTER_PORT>_<RLGPU_COMM_TAG or the launcher's pid>.id, the others wait for it
// Where the ranks of one launch meet: <RLGPU_COMM_DIR or /tmp>/rlgpu_comm_<MASTER_PORT>_<tag>.id.  The tag keeps a stale file of an earlier
// launch on the same port apart: RLGPU_COMM_TAG when the launcher of the ranks sets one (bench.py does: its ranks' programs are children of
// one Python process EACH), else the parent's pid -- the ranks of `torch.distributed.run my_program` are children of the same agent.
static std::string rendezvous_path() {
    const char* dir = getenv("RLGPU_COMM_DIR"); const char* port = getenv("MASTER_PORT"); const char* tag = getenv("RLGPU_COMM_TAG");
    return std::string(dir ? dir : "/tmp") + "/rlgpu_comm_" + (port ? port : "0") + "_" + (tag ? tag : std::to_string((long)getppid())) + ".id";
}
int rlgpu_comm_rendezvous_path(char* buf, int cap) {
    const std::string p = rendezvous_path();
    if (!buf || cap <= (int)p.size()) return RLGPU_ERR_ARG;
    memcpy(buf, p.c_str(), p.size() + 1);
    return RLGPU_OK;
}

int rlgpu_comm_init_env(rlgpu_comm** out, int* rank_out, int* world_out) {
    auto env_i = [](const char* k, int d) { const char* v = getenv(k); return v ? atoi(v) : d; };
    const int rank = env_i("RANK", 0), world = env_i("WORLD_SIZE", 1), local = env_i("LOCAL_RANK", rank);
    if (rank_out) *rank_out = rank;
    if (world_out) *world_out = world;
    const std::string path = rendezvous_path();
    unsigned char id[RLGPU_COMM_ID_BYTES];
    if (rank == 0) {
        int rc = rlgpu_comm_unique_id(id);
        if (rc != RLGPU_OK) return rc;
        std::string tmp = path + ".tmp";
        FILE* f = fopen(tmp.c_str(), "wb");
        if (!f) return fail(nullptr, "cannot write " + tmp);
        fwrite(id, 1, sizeof(id), f); fclose(f);
        rename(tmp.c_str(), path.c_str());
    } else {
        bool ok = false;
        for (int tries = 0; tries < 6000 && !ok; tries++) {     // up to 60 s
            FILE* f = fopen(path.c_str(), "rb");
            if (f) { ok = fread(id, 1, sizeof(id), f) == sizeof(id); fclose(f); }
            if (!ok) std::this_thread::sleep_for(std::chrono::milliseconds(10));
        }
        if (!ok) return fail(nullptr, "timed out waiting for " + path);
    }
    int rc = rlgpu_comm_init(out, local, rank, world, id);
    if (rc == RLGPU_OK && rank == 0 && world > 1) {
        // everybody has read the file once the communicator exists (ncclCommInitRank is collective)
        remove(path.c_str());
    } else if (rc == RLGPU_OK && world == 1) remove(path.c_str());
    return rc;
}

int rlgpu_comm_destroy(rlgpu_comm* c) {
    if (!c) return RLGPU_OK;
    if (c->comm) ncclCommDestroy(c->comm);
    delete c;
    return RLGPU_OK;
}
int rlgpu_comm_rank(const rlgpu_comm* c) { return c ? c->rank : 0; }
int rlgpu_comm_world(const rlgpu_comm* c) { return c ? c->world : 1; }

int rlgpu_comm_allreduce_f32(rlgpu_comm* c, float* dev_ptr, int64_t n, void* stream) {
    if (!c || !dev_ptr || n < 0) return RLGPU_ERR_ARG;
    ncclResult_t r = ncclAllReduce(dev_ptr, dev_ptr, (size_t)n, ncclFloat, ncclSum, c->comm, (hipStream_t)stream);
    if (r != ncclSuccess) return fail(c, std::string("ncclAllReduce: ") + ncclGetErrorString(r));
    return RLGPU_OK;
}
int rlgpu_comm_broadcast(rlgpu_comm* c, void* dev_ptr, int64_t bytes, int root, void* stream) {
    if (!c || !dev_ptr || bytes < 0) return RLGPU_ERR_ARG;
    ncclResult_t r = ncclBroadcast(dev_ptr, dev_ptr, (size_t)bytes, ncclChar, root, c->comm, (hipStream_t)stream);
    if (r != ncclSuccess) return fail(c, std::string("ncclBroadcast: ") + ncclGetErrorString(r));
    return RLGPU_OK;
}

}  // extern "C"
