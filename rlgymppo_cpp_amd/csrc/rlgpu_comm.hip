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
#include <mutex>
#include <condition_variable>
#include <memory>
#include <ctime>
#include <cerrno>
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>
#include <sys/mman.h>
#include <vector>
#include <algorithm>
#include "../../include/rlgpu.h"

// Test-only second transport (RLGPU_COMM_TRANSPORT=shm): the same three collectives staged through a POSIX shared-memory segment, so that the
// ranks of a launch can share ONE device -- RCCL refuses two ranks on a device, and this pool's boxes have one GPU: with it every line of the
// hosts' N > 1 path (sharded envs, rank-keyed sampler, all-reduce per optimizer step, rank-0 broadcasts, replica checks, fail-fast) executes
// before the first real multi-GPU run.  Sums are taken in rank order by every rank itself: identical bits everywhere, run to run.
struct ShmSeg {
    uint64_t magic; int32_t world; int32_t pad;
    int32_t arrive[2];              // sense-reversing barrier: arrivals of the current phase (all flags below: __atomic accesses only)
    int32_t phase;
    int32_t dead;                   // a rank that failed sets it: everybody else stops waiting
    // then world slots of SLOT bytes
};
constexpr size_t SHM_SLOT = 8u << 20;
constexpr uint64_t SHM_MAGIC = 0x314d48535f4c52ull;

struct rlgpu_comm {
    ncclComm_t comm = nullptr;
    int device = 0, rank = 0, world = 1;
    // shm transport
    ShmSeg* seg = nullptr; size_t seg_bytes = 0; std::string seg_name; int my_phase = 0; int timeout_s = 300;
    std::string err;
};

namespace {
std::string g_comm_err;
int fail(rlgpu_comm* c, const std::string& m) { if (c) c->err = m; g_comm_err = m; if (c && c->seg) __atomic_store_n(&c->seg->dead, 1, __ATOMIC_RELEASE); return RLGPU_ERR_HIP; }

unsigned char* shm_slot(rlgpu_comm* c, int r) { return reinterpret_cast<unsigned char*>(c->seg) + 4096 + (size_t)r * SHM_SLOT; }
// everybody arrives, or the wait ends with an error: a peer that died (its process gone or its `dead` mark) must not hang the others
int shm_barrier(rlgpu_comm* c, const char* what) {
    ShmSeg* s = c->seg;
    const int ph = c->my_phase & 1;
    // (acquire / release on every flag: the slots a rank wrote before it arrived must be visible to whoever sees its arrival -- x86 gave that for
    // free with plain volatile accesses, other hosts do not: ADVICE r04)
    auto ld = [](const int32_t* p) { return __atomic_load_n(p, __ATOMIC_ACQUIRE); };
    __atomic_fetch_add(&s->arrive[ph], 1, __ATOMIC_ACQ_REL);
    const auto t0 = std::chrono::steady_clock::now();
    while (ld(&s->arrive[ph]) < c->world) {
        if (ld(&s->dead)) return fail(c, std::string(what) + ": a peer rank failed");
        if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(c->timeout_s))
            return fail(c, std::string(what) + ": rank " + std::to_string(c->rank) + " waited " + std::to_string(c->timeout_s) + " s for its peers (a rank died?)");
        std::this_thread::sleep_for(std::chrono::microseconds(50));
    }
    // the last one to leave re-arms the OTHER phase's counter: nobody can be waiting on it (they are all here)
    c->my_phase++;
    if (c->rank == 0) __atomic_store_n(&s->arrive[(c->my_phase) & 1], 0, __ATOMIC_RELEASE);
    // second half: nobody proceeds to the next barrier before rank 0 has re-armed it
    __atomic_fetch_add(&s->phase, 1, __ATOMIC_ACQ_REL);
    const int want = c->my_phase * c->world;
    while (ld(&s->phase) < want) {
        if (ld(&s->dead)) return fail(c, std::string(what) + ": a peer rank failed");
        if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(c->timeout_s)) return fail(c, std::string(what) + ": rank " + std::to_string(c->rank) + " waited " + std::to_string(c->timeout_s) + " s for its peers to leave the barrier (a rank died?)");
        std::this_thread::sleep_for(std::chrono::microseconds(50));
    }
    return RLGPU_OK;
}
// RCCL reports failures of earlier, asynchronous work (a peer that went away, a link error) through the communicator's async error
int rccl_async(rlgpu_comm* c, const char* what) {
    ncclResult_t st = ncclSuccess;
    ncclResult_t r = ncclCommGetAsyncError(c->comm, &st);
    if (r != ncclSuccess) return fail(c, std::string(what) + ": ncclCommGetAsyncError: " + ncclGetErrorString(r));
    if (st != ncclSuccess && st != ncclInProgress) return fail(c, std::string(what) + ": asynchronous RCCL error: " + ncclGetErrorString(st));
    return RLGPU_OK;
}
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

// ncclCommInitRank is collective and has no timeout of its own: a rank that never arrives (a crashed peer, a stale rendezvous id) would
// hang the others for ever.  It runs on a helper thread; past RLGPU_COMM_TIMEOUT_S (default 300 s: the first init of an 8-GPU node builds
// its topology and rings for tens of seconds) the caller gets an error back and is
// expected to exit (the helper thread is left behind: a process in that state cannot use the communicator anyway).
int rlgpu_comm_init(rlgpu_comm** out, int device, int rank, int world, const void* id_bytes) {
    if (!out || !id_bytes || world < 1 || rank < 0 || rank >= world) return RLGPU_ERR_ARG;
    if (hipSetDevice(device) != hipSuccess) return fail(nullptr, "hipSetDevice(" + std::to_string(device) + ") failed");
    struct Job { std::mutex mu; std::condition_variable cv; bool done = false; ncclResult_t r = ncclSuccess; ncclComm_t comm = nullptr; };
    auto job = std::make_shared<Job>();
    ncclUniqueId id; memcpy(&id, id_bytes, sizeof(id));
    std::thread([job, id, device, rank, world]() {
        (void)hipSetDevice(device);
        ncclComm_t comm = nullptr;
        ncclResult_t r = ncclCommInitRank(&comm, world, id, rank);
        std::lock_guard<std::mutex> lk(job->mu);
        job->r = r; job->comm = comm; job->done = true; job->cv.notify_all();
    }).detach();
    const char* tv = getenv("RLGPU_COMM_TIMEOUT_S");
    const int timeout_s = tv && atoi(tv) > 0 ? atoi(tv) : 300;
    {
        std::unique_lock<std::mutex> lk(job->mu);
        if (!job->cv.wait_for(lk, std::chrono::seconds(timeout_s), [&] { return job->done; }))
            return fail(nullptr, "ncclCommInitRank: rank " + std::to_string(rank) + " of " + std::to_string(world) + " timed out after " + std::to_string(timeout_s) + " s (a peer never arrived, or a stale rendezvous id)");
    }
    if (job->r != ncclSuccess) return fail(nullptr, std::string("ncclCommInitRank: ") + ncclGetErrorString(job->r));
    rlgpu_comm* c = new rlgpu_comm();
    c->device = device; c->rank = rank; c->world = world; c->comm = job->comm;
    *out = c;
    return RLGPU_OK;
}

// Where the ranks of one launch meet: <dir>/rlgpu_comm_<MASTER_PORT>_<tag>.id.  dir = RLGPU_COMM_DIR, else a directory of the user's own
// (/tmp/rlgpu_comm_<uid>, mode 0700, checked to be a real directory owned by the user) -- not a predictable name in world-writable /tmp.
// The tag keeps launches on the same port apart: RLGPU_COMM_TAG when the launcher of the ranks sets one (bench.py does: its ranks' programs
// are children of one Python process EACH), else the parent's pid -- the ranks of `torch.distributed.run my_program` are children of the
// same agent.  Node-local: a multi-node launch needs RLGPU_COMM_DIR on a shared file system.
static std::string rendezvous_dir() {
    const char* dir = getenv("RLGPU_COMM_DIR");
    if (dir && *dir) return dir;
    const std::string d = "/tmp/rlgpu_comm_" + std::to_string((long)getuid());
    (void)mkdir(d.c_str(), 0700);
    struct stat st;
    if (lstat(d.c_str(), &st) != 0 || !S_ISDIR(st.st_mode) || st.st_uid != getuid() || (st.st_mode & 077) != 0) return std::string();   // somebody else's: refuse
    return d;
}
// the launching agent as a name no later launch can have: its pid AND the kernel's start time of that pid (/proc/<pid>/stat field 22) -- a pid
// alone comes round again, and a file a crashed launch left under it would be taken for this launch's id
static std::string parent_identity() {
    const long pp = (long)getppid();
    std::string id = std::to_string(pp);
    if (FILE* f = fopen(("/proc/" + id + "/stat").c_str(), "r")) {
        char buf[1024]; const size_t n = fread(buf, 1, sizeof(buf) - 1, f); fclose(f); buf[n] = 0;
        if (const char* close = strrchr(buf, ')')) {     // fields after "(comm)": state is field 3, starttime field 22
            int field = 2; const char* q = close + 1;
            while (*q && field < 21) { while (*q == ' ') q++; while (*q && *q != ' ') q++; field++; }
            while (*q == ' ') q++;
            if (*q) { std::string st; while (*q && *q != ' ') st += *q++; id += "_" + st; }
        }
    }
    return id;
}
static std::string rendezvous_path() {
    const char* port = getenv("MASTER_PORT"); const char* tag = getenv("RLGPU_COMM_TAG"); const char* run = getenv("TORCHELASTIC_RUN_ID");
    const std::string dir = rendezvous_dir();
    std::string name = std::string("/rlgpu_comm_") + (port ? port : "0") + "_" + (tag ? std::string(tag) : parent_identity());
    if (run && *run && strcmp(run, "none") != 0) name += std::string("_") + run;      // torchrun's own id of the launch, when it has one
    return (dir.empty() ? std::string("/nonexistent") : dir) + name + ".id";
}
int rlgpu_comm_rendezvous_path(char* buf, int cap) {
    const std::string p = rendezvous_path();
    if (!buf || cap <= (int)p.size()) return RLGPU_ERR_ARG;
    memcpy(buf, p.c_str(), p.size() + 1);
    return RLGPU_OK;
}

// The file: 8 bytes magic, 8 bytes wall-clock seconds at which rank 0 wrote it, then the id.  A reader only takes a file written after
// its own process started minus RLGPU_COMM_STALE_S (default 300 s: ranks of one launch start within that of each other) -- what an
// earlier launch with the same port and tag left behind (a crash between write and remove) is not this launch's id.
namespace {
constexpr uint64_t RDV_MAGIC = 0x31444950475f4c52ull;   // "RL_GPID1"
const int64_t g_process_start = (int64_t)time(nullptr);
}

// rank / world / rendezvous from the launcher's environment (torchrun or any launcher that sets RANK, WORLD_SIZE, LOCAL_RANK, MASTER_PORT)
int rlgpu_comm_init_env(rlgpu_comm** out, int* rank_out, int* world_out) {
    auto env_i = [](const char* k, int d) { const char* v = getenv(k); return v ? atoi(v) : d; };
    // this pool's host driver only supports dmabuf IPC; RCCL's peer mappings fail with the legacy mode.  Only effective when the HSA runtime
    // has not been initialised yet (the hosts call this before their first HIP call); an exported value is left alone.
    (void)setenv("HSA_ENABLE_IPC_MODE_LEGACY", "0", 0);
    const int rank = env_i("RANK", 0), world = env_i("WORLD_SIZE", 1), local_rank = env_i("LOCAL_RANK", rank);
    if (rank_out) *rank_out = rank;
    if (world_out) *world_out = world;
    if (rendezvous_dir().empty()) return fail(nullptr, "rendezvous directory /tmp/rlgpu_comm_<uid> is not a private directory of this user (set RLGPU_COMM_DIR)");
    const std::string path = rendezvous_path();
    const int timeout_s = env_i("RLGPU_COMM_TIMEOUT_S", 300) > 0 ? env_i("RLGPU_COMM_TIMEOUT_S", 300) : 300;
    // rank 0 publishes RLGPU_COMM_ID_BYTES of payload through the rendezvous file (tmp file + rename: a reader never sees half of it), the others
    // wait for a FRESH file (written after their own start minus RLGPU_COMM_STALE_S: what a crashed earlier launch left under the name is refused)
    auto publish = [&](const unsigned char* payload) -> int {
        (void)unlink(path.c_str());                          // whatever an earlier launch left under this name
        const std::string tmp = path + ".tmp." + std::to_string((long)getpid());
        (void)unlink(tmp.c_str());
        const int fd = open(tmp.c_str(), O_WRONLY | O_CREAT | O_EXCL | O_NOFOLLOW, 0600);
        if (fd < 0) return fail(nullptr, "cannot create " + tmp + ": " + strerror(errno));
        const uint64_t hdr[2] = {RDV_MAGIC, (uint64_t)time(nullptr)};
        const bool ok = write(fd, hdr, sizeof(hdr)) == (ssize_t)sizeof(hdr) && write(fd, payload, RLGPU_COMM_ID_BYTES) == (ssize_t)RLGPU_COMM_ID_BYTES;
        close(fd);
        if (!ok || rename(tmp.c_str(), path.c_str()) != 0) { (void)unlink(tmp.c_str()); return fail(nullptr, "cannot write " + path); }
        return RLGPU_OK;
    };
    auto await = [&](unsigned char* payload) -> int {
        const int64_t stale_s = env_i("RLGPU_COMM_STALE_S", 300);
        bool ok = false;
        for (int tries = 0; tries < timeout_s * 100 && !ok; tries++) {
            const int fd = open(path.c_str(), O_RDONLY | O_NOFOLLOW);
            if (fd >= 0) {
                uint64_t hdr[2] = {0, 0};
                const bool whole = read(fd, hdr, sizeof(hdr)) == (ssize_t)sizeof(hdr) && read(fd, payload, RLGPU_COMM_ID_BYTES) == (ssize_t)RLGPU_COMM_ID_BYTES;
                close(fd);
                ok = whole && hdr[0] == RDV_MAGIC && (int64_t)hdr[1] >= g_process_start - stale_s;
            }
            if (!ok) std::this_thread::sleep_for(std::chrono::milliseconds(10));
        }
        return ok ? RLGPU_OK : fail(nullptr, "timed out waiting for a fresh " + path);
    };
    unsigned char id[RLGPU_COMM_ID_BYTES];
    const char* transport = getenv("RLGPU_COMM_TRANSPORT");
    if (transport && !strcmp(transport, "shm")) {
        // The segment's name is this launch's own -- rank 0's pid and clock -- and travels through the rendezvous file like an RCCL id does: a segment
        // a crashed earlier launch left behind (right size, magic and world: it used to be accepted when it was opened before rank 0 had replaced
        // it) has another name and is never looked at.
        const size_t bytes = 4096 + (size_t)world * SHM_SLOT;
        rlgpu_comm* c = new rlgpu_comm();
        // (the test transport exists so that the ranks of a launch can share ONE device: RLGPU_SHM_DEVICE puts every rank there whatever the
        // launcher's LOCAL_RANK says -- `torch.distributed.run --nproc-per-node 2 bench.py --gpus 2` on a one-GPU box, tests/test_host_cpp.py)
        const int local = env_i("RLGPU_SHM_DEVICE", local_rank);
        c->device = local; c->rank = rank; c->world = world; c->timeout_s = timeout_s; c->seg_bytes = bytes;
        int fd = -1;
        memset(id, 0, sizeof(id));
        if (rank == 0) {
            const auto now = std::chrono::steady_clock::now().time_since_epoch();
            snprintf(reinterpret_cast<char*>(id), sizeof(id), "/rlgpu_shm_%ld_%lld", (long)getpid(), (long long)std::chrono::duration_cast<std::chrono::nanoseconds>(now).count());
            c->seg_name = reinterpret_cast<const char*>(id);
            fd = shm_open(c->seg_name.c_str(), O_RDWR | O_CREAT | O_EXCL, 0600);
            if (fd < 0 || ftruncate(fd, (off_t)bytes) != 0) { const std::string why = strerror(errno); if (fd >= 0) { close(fd); (void)shm_unlink(c->seg_name.c_str()); } const std::string n = c->seg_name; delete c; return fail(nullptr, "shm transport: cannot create " + n + ": " + why); }
        } else {
            if (await(id) != RLGPU_OK) { delete c; return RLGPU_ERR_HIP; }
            id[sizeof(id) - 1] = 0;
            c->seg_name = reinterpret_cast<const char*>(id);
            fd = shm_open(c->seg_name.c_str(), O_RDWR, 0600);
            struct stat st;
            if (fd >= 0 && (fstat(fd, &st) != 0 || (size_t)st.st_size < bytes)) { close(fd); fd = -1; }
            if (fd < 0) { const std::string n = c->seg_name; delete c; return fail(nullptr, "shm transport: cannot open this launch's segment " + n); }
        }
        void* m = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        close(fd);
        if (m == MAP_FAILED) { if (rank == 0) (void)shm_unlink(c->seg_name.c_str()); delete c; return fail(nullptr, std::string("shm transport: mmap: ") + strerror(errno)); }
        c->seg = reinterpret_cast<ShmSeg*>(m);
        if (rank == 0) {
            // the header is complete BEFORE the name is published: a reader that can open the segment sees this launch's magic and world
            c->seg->world = world; c->seg->arrive[0] = c->seg->arrive[1] = 0; c->seg->phase = 0; c->seg->dead = 0;
            __atomic_store_n(&c->seg->magic, SHM_MAGIC, __ATOMIC_RELEASE);
            if (publish(id) != RLGPU_OK) { (void)shm_unlink(c->seg_name.c_str()); munmap(m, bytes); delete c; return RLGPU_ERR_HIP; }
        } else if (__atomic_load_n(&c->seg->magic, __ATOMIC_ACQUIRE) != SHM_MAGIC || c->seg->world != world) {
            const std::string n = c->seg_name; munmap(m, bytes); delete c; return fail(nullptr, "shm transport: " + n + " is not this launch's segment");
        }
        int rc = shm_barrier(c, "shm transport: first barrier");
        if (rank == 0) { (void)shm_unlink(c->seg_name.c_str()); (void)unlink(path.c_str()); }   // everybody has it mapped (or the wait ended): nothing is left behind
        if (rc != RLGPU_OK) { const std::string e = c->err; munmap(m, bytes); delete c; return fail(nullptr, e); }
        *out = c;
        return RLGPU_OK;
    }
    if (rank == 0) {
        int rc = rlgpu_comm_unique_id(id);
        if (rc != RLGPU_OK) return rc;
        rc = publish(id);
        if (rc != RLGPU_OK) return rc;
    } else {
        int rc = await(id);
        if (rc != RLGPU_OK) return rc;
    }
    int rc = rlgpu_comm_init(out, local_rank, rank, world, id);
    // everybody has read the file once ncclCommInitRank has returned (it is collective), whatever it returned
    if (rank == 0) (void)unlink(path.c_str());
    return rc;
}

int rlgpu_comm_destroy(rlgpu_comm* c) {
    if (!c) return RLGPU_OK;
    if (c->seg) munmap(c->seg, c->seg_bytes);
    if (c->comm) ncclCommDestroy(c->comm);
    delete c;
    return RLGPU_OK;
}
int rlgpu_comm_rank(const rlgpu_comm* c) { return c ? c->rank : 0; }
int rlgpu_comm_world(const rlgpu_comm* c) { return c ? c->world : 1; }
int rlgpu_comm_device(const rlgpu_comm* c) { return c ? c->device : -1; }

int rlgpu_comm_allreduce_f32(rlgpu_comm* c, float* dev_ptr, int64_t n, void* stream) {
    if (!c || !dev_ptr || n < 0) return RLGPU_ERR_ARG;
    if (c->seg) {   // host-staged, in chunks of a slot: D2H into the own slot | barrier | sum of all slots in rank order | H2D | barrier
        const int64_t per = (int64_t)(SHM_SLOT / 4);
        std::vector<float> sum;
        for (int64_t o = 0; o < n; o += per) {
            const int64_t k = std::min(per, n - o);
            if (hipMemcpyAsync(shm_slot(c, c->rank), dev_ptr + o, (size_t)k * 4, hipMemcpyDeviceToHost, (hipStream_t)stream) != hipSuccess || hipStreamSynchronize((hipStream_t)stream) != hipSuccess)
                return fail(c, "shm all-reduce: device -> host copy failed");
            int rc = shm_barrier(c, "shm all-reduce");
            if (rc) return rc;
            sum.assign((size_t)k, 0.f);
            for (int r = 0; r < c->world; r++) { const float* p = reinterpret_cast<const float*>(shm_slot(c, r)); for (int64_t i = 0; i < k; i++) sum[(size_t)i] += p[i]; }
            rc = shm_barrier(c, "shm all-reduce");   // (everybody has read every slot before anybody overwrites its own)
            if (rc) return rc;
            if (hipMemcpyAsync(dev_ptr + o, sum.data(), (size_t)k * 4, hipMemcpyHostToDevice, (hipStream_t)stream) != hipSuccess || hipStreamSynchronize((hipStream_t)stream) != hipSuccess)
                return fail(c, "shm all-reduce: host -> device copy failed");
        }
        return RLGPU_OK;
    }
    int rc = rccl_async(c, "before ncclAllReduce");
    if (rc) return rc;
    ncclResult_t r = ncclAllReduce(dev_ptr, dev_ptr, (size_t)n, ncclFloat, ncclSum, c->comm, (hipStream_t)stream);
    if (r != ncclSuccess) return fail(c, std::string("ncclAllReduce: ") + ncclGetErrorString(r));
    return rccl_async(c, "after ncclAllReduce");
}
int rlgpu_comm_broadcast(rlgpu_comm* c, void* dev_ptr, int64_t bytes, int root, void* stream) {
    if (!c || !dev_ptr || bytes < 0 || root < 0 || root >= c->world) return RLGPU_ERR_ARG;
    if (c->seg) {
        for (int64_t o = 0; o < bytes; o += (int64_t)SHM_SLOT) {
            const int64_t k = std::min<int64_t>((int64_t)SHM_SLOT, bytes - o);
            if (c->rank == root && (hipMemcpyAsync(shm_slot(c, root), (char*)dev_ptr + o, (size_t)k, hipMemcpyDeviceToHost, (hipStream_t)stream) != hipSuccess || hipStreamSynchronize((hipStream_t)stream) != hipSuccess))
                return fail(c, "shm broadcast: device -> host copy failed");
            int rc = shm_barrier(c, "shm broadcast");
            if (rc) return rc;
            if (c->rank != root && (hipMemcpyAsync((char*)dev_ptr + o, shm_slot(c, root), (size_t)k, hipMemcpyHostToDevice, (hipStream_t)stream) != hipSuccess || hipStreamSynchronize((hipStream_t)stream) != hipSuccess))
                return fail(c, "shm broadcast: host -> device copy failed");
            rc = shm_barrier(c, "shm broadcast");
            if (rc) return rc;
        }
        return RLGPU_OK;
    }
    int rc = rccl_async(c, "before ncclBroadcast");
    if (rc) return rc;
    ncclResult_t r = ncclBroadcast(dev_ptr, dev_ptr, (size_t)bytes, ncclChar, root, c->comm, (hipStream_t)stream);
    if (r != ncclSuccess) return fail(c, std::string("ncclBroadcast: ") + ncclGetErrorString(r));
    return rccl_async(c, "after ncclBroadcast");
}
// Has anything gone wrong with the ranks' exchange since the last look?  RCCL: the communicator's asynchronous error; shm: a peer's failure mark.
// The hosts call it once per iteration and end the process with the text (a restart is a fresh launch of every rank).
int rlgpu_comm_check(rlgpu_comm* c) {
    if (!c) return RLGPU_OK;
    if (c->seg) return c->seg->dead ? fail(c, "a peer rank failed") : RLGPU_OK;
    return rccl_async(c, "rlgpu_comm_check");
}

}  // extern "C"
