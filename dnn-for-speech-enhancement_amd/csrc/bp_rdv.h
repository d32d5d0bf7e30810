// bp_rdv.h -- host-side rendezvous of the data-parallel ranks (one process per GPU): a POSIX shared-memory block
// "/bpdp-<key>" that carries the hipIpc handles of every rank, a host barrier and a small all-gather.  No HIP calls
// in here: the bp_rdv_* entry points (include/bp_c_api.h) run without a GPU, which is how the N > 1 launch paths
// (bench.py --gpus N, bptrain gpu_used=N) are covered by CPU tests.
//
// The reference has no counterpart (its multi-GPU path is one process driving devices 0..G-1, BP_GPU.cu:29-36,80-111);
// this replaces what an MPI/NCCL bootstrap would do for one node.
//
// Lifetime rules (a crashed job must never poison the next one that reuses its key):
//   * rank 0 CREATES the block: it unlinks whatever carries the name, creates with O_EXCL, initialises the fields and
//     publishes `magic` last.  Other ranks only ever open an existing name, and ignore a block whose magic is missing,
//     whose world differs or whose creator process is dead (a stale block of a crashed job) until the real one shows up;
//   * as soon as every rank has joined (first barrier), rank 0 unlinks the NAME: the mapping lives on, but nothing is
//     left in /dev/shm whatever happens later;
//   * a rank that times out in a barrier raises abort_flag, so its peers fail at once instead of after their own timeout.
#pragma once
#include <atomic>
#include <chrono>
#include <errno.h>
#include <fcntl.h>
#include <signal.h>
#include <stdint.h>
#include <string.h>
#include <string>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

enum { BP_RDV_MAXRANKS = 8, BP_RDV_SLOT_BYTES = 64, BP_RDV_BLOB_BYTES = 640 };
static const uint64_t BP_RDV_MAGIC = 0x62706470'72647633ull;     // "bpdprdv3"

struct RdvShm {
    std::atomic<uint64_t> magic;
    uint64_t nonce;                       // creator pid << 32 | creation time: identifies this job's block
    int creator_pid, world;
    std::atomic<int> joined, bar_count, bar_gen, abort_flag;
    int pid[BP_RDV_MAXRANKS];
    unsigned char slot[BP_RDV_MAXRANKS][BP_RDV_SLOT_BYTES];   // bp_rdv_allgather payloads
    unsigned char blob[BP_RDV_MAXRANKS][BP_RDV_BLOB_BYTES];   // per-rank data of the GPU layer (hipIpc handles, device, bus id)
    unsigned char shared[256];                                  // one record written by rank 0 (e.g. an RCCL unique id)
};

struct bp_rdv {
    RdvShm *shm;
    std::string name;
    int world, rank;
    double timeout_s;
    bool name_live;                       // rank 0: the name still exists in /dev/shm
};

static thread_local std::string g_rdv_err;

static inline double rdv_now()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
static inline bool rdv_pid_alive(int pid) { return pid > 0 && (kill((pid_t)pid, 0) == 0 || errno == EPERM); }

static int rdv_barrier(bp_rdv *r)
{
    RdvShm *s = r->shm;
    const int gen = s->bar_gen.load();
    if (s->bar_count.fetch_add(1) + 1 == r->world) { s->bar_count.store(0); s->bar_gen.fetch_add(1); return 0; }
    const double t0 = rdv_now();
    while (s->bar_gen.load() == gen) {
        if (s->abort_flag.load()) { g_rdv_err = "data-parallel group: a peer rank failed"; return -3; }
        if (rdv_now() - t0 > r->timeout_s) {
            s->abort_flag.store(1);
            g_rdv_err = "data-parallel group: host barrier timed out (a rank is missing)";
            return -3;
        }
        usleep(50);
    }
    return 0;
}

static void rdv_close(bp_rdv *r, bool failed)
{
    if (!r) return;
    if (r->shm) {
        if (failed) r->shm->abort_flag.store(1);
        munmap((void *)r->shm, sizeof(RdvShm));
    }
    if (r->rank == 0 && r->name_live) shm_unlink(r->name.c_str());
    delete r;
}

// Returns 0 and *out on success; a negative bp_status and g_rdv_err otherwise.
static int rdv_open(const char *key, int world, int rank, double timeout_s, bp_rdv **out)
{
    if (!key || !*key || !out) { g_rdv_err = "rendezvous: null argument"; return -1; }
    if (world < 1 || world > BP_RDV_MAXRANKS || rank < 0 || rank >= world) { g_rdv_err = "rendezvous: world must be 1..8 and 0 <= rank < world"; return -1; }
    for (const char *c = key; *c; ++c)
        if (*c == '/') { g_rdv_err = "rendezvous: the key must not contain '/'"; return -1; }
    bp_rdv *r = new bp_rdv();
    r->shm = nullptr; r->name = std::string("/bpdp-") + key; r->world = world; r->rank = rank;
    r->timeout_s = timeout_s > 0.5 ? timeout_s : 0.5; r->name_live = false;
    const double t0 = rdv_now();
    if (rank == 0) {
        // Whatever carries the name is either the stale block of a crashed job (replace it) or the block of a LIVE job that is
        // still joining under the same key (unlinking that one would send its late ranks into OUR block): look first.
        {
            const int fd0 = shm_open(r->name.c_str(), O_RDWR, 0600);
            if (fd0 >= 0) {
                struct stat st0;
                bool live = false;
                if (fstat(fd0, &st0) == 0 && (size_t)st0.st_size >= sizeof(RdvShm)) {
                    void *p0 = mmap(nullptr, sizeof(RdvShm), PROT_READ | PROT_WRITE, MAP_SHARED, fd0, 0);
                    if (p0 != MAP_FAILED) {
                        const RdvShm *s0 = (const RdvShm *)p0;
                        live = s0->magic.load(std::memory_order_acquire) == BP_RDV_MAGIC && s0->creator_pid != (int)getpid() &&
                               rdv_pid_alive(s0->creator_pid) && !s0->abort_flag.load();
                        munmap(p0, sizeof(RdvShm));
                    }
                }
                close(fd0);
                if (live) { g_rdv_err = "rendezvous: a live job is already joining under the key of " + r->name + " (keys must be unique per job)"; delete r; return -3; }
            }
        }
        shm_unlink(r->name.c_str());                              // a stale block of a crashed job with this key, if any
        const int fd = shm_open(r->name.c_str(), O_CREAT | O_EXCL | O_RDWR, 0600);
        if (fd < 0 || ftruncate(fd, sizeof(RdvShm)) != 0) {
            if (fd >= 0) { close(fd); shm_unlink(r->name.c_str()); }
            g_rdv_err = "rendezvous: cannot create the shared block " + r->name + " (" + strerror(errno) + "; is another job using this key?)";
            delete r; return -3;
        }
        r->name_live = true;
        void *p = mmap(nullptr, sizeof(RdvShm), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        close(fd);
        if (p == MAP_FAILED) { g_rdv_err = "rendezvous: mmap failed"; rdv_close(r, true); return -3; }
        r->shm = (RdvShm *)p;                                     // (a fresh shared-memory object is zero-filled)
        r->shm->creator_pid = (int)getpid(); r->shm->world = world;
        r->shm->nonce = ((uint64_t)getpid() << 32) ^ (uint64_t)std::chrono::system_clock::now().time_since_epoch().count();
        r->shm->pid[0] = (int)getpid();
        r->shm->magic.store(BP_RDV_MAGIC, std::memory_order_release);
    } else {
        for (;;) {
            if (rdv_now() - t0 > r->timeout_s) { g_rdv_err = "rendezvous: timed out waiting for rank 0 to create " + r->name; delete r; return -3; }
            const int fd = shm_open(r->name.c_str(), O_RDWR, 0600);
            if (fd < 0) { usleep(200); continue; }
            struct stat st;
            if (fstat(fd, &st) != 0 || (size_t)st.st_size < sizeof(RdvShm)) { close(fd); usleep(200); continue; }   // (still being sized)
            void *p = mmap(nullptr, sizeof(RdvShm), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
            close(fd);
            if (p == MAP_FAILED) { usleep(200); continue; }
            RdvShm *s = (RdvShm *)p;
            const bool ready = s->magic.load(std::memory_order_acquire) == BP_RDV_MAGIC;
            if (ready && s->world != world && rdv_pid_alive(s->creator_pid)) {
                munmap(p, sizeof(RdvShm));
                g_rdv_err = "rendezvous: ranks disagree on the world size";
                delete r; return -1;
            }
            // not published yet, or left behind by a dead job: look again until rank 0's block is there
            if (!ready || !rdv_pid_alive(s->creator_pid) || s->abort_flag.load()) { munmap(p, sizeof(RdvShm)); usleep(200); continue; }
            // claim the rank's slot: a second process that arrives with the same rank is an error, not a silent overwrite
            int expected = 0;
            if (!__atomic_compare_exchange_n(&s->pid[rank], &expected, (int)getpid(), false, __ATOMIC_ACQ_REL, __ATOMIC_ACQUIRE) && expected != (int)getpid()) {
                if (!rdv_pid_alive(expected)) { munmap(p, sizeof(RdvShm)); usleep(200); continue; }       // (a stale block whose creator pid was reused)
                munmap(p, sizeof(RdvShm));
                g_rdv_err = "rendezvous: rank " + std::to_string(rank) + " of " + r->name + " is already taken by process " + std::to_string(expected);
                delete r; return -1;
            }
            if (s->joined.load() >= world) {                      // a complete group's block that outlived its name: not ours
                __atomic_store_n(&s->pid[rank], 0, __ATOMIC_RELEASE);
                munmap(p, sizeof(RdvShm)); usleep(200); continue;
            }
            r->shm = s;
            break;
        }
    }
    r->shm->joined.fetch_add(1);
    int rc = rdv_barrier(r);                                      // everyone has mapped THIS block ...
    if (rc != 0) { rdv_close(r, true); return rc; }
    if (rank == 0) { shm_unlink(r->name.c_str()); r->name_live = false; }   // ... so the name can go
    *out = r;
    return 0;
}

// All-gather of one small record per rank (<= BP_RDV_SLOT_BYTES): all[p*bytes ..] = rank p's record.
static int rdv_allgather(bp_rdv *r, const void *mine, size_t bytes, void *all)
{
    if (bytes > BP_RDV_SLOT_BYTES) { g_rdv_err = "rendezvous: all-gather record too large"; return -1; }
    memcpy(r->shm->slot[r->rank], mine, bytes);
    int rc = rdv_barrier(r);
    if (rc != 0) return rc;
    for (int p = 0; p < r->world; ++p) memcpy((char *)all + (size_t)p * bytes, r->shm->slot[p], bytes);
    return rdv_barrier(r);                                        // nobody overwrites a slot before everyone has read it
}
