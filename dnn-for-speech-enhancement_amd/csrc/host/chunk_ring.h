// chunk_ring.h -- ONE reader per node for data-parallel training (bptrain gpu_used=N).
//
// The reference has one host reader feeding its G devices (Interface::Readchunk fills one host buffer, BP_GPU::train
// splits it, Interface.cc:689-861, BP_GPU.cu:269-277).  With one process per GPU the naive equivalent makes every rank
// read, byte-swap, normalise and window-plan the WHOLE chunk and then keep 1/N of its rows -- N times the host work, on
// the cores that already limit a single GPU.  Here the ranks (forked from one parent) share a two-slot ring in anonymous
// shared memory and split the work so that every Pfile byte is read ONCE per node:
//     rank 0      builds the chunk's tables (window starts, target frames, noise-aware rows; consumes the lrand48
//                 shuffle stream exactly as the single-process reader does), later its noise-aware block
//     every rank  converts 1/N of the chunk's frames (positioned reads, byte swap, mean/variance normalisation) straight
//                 into the shared slot
//     every rank  then takes ITS rows of every global minibatch out of the shared tables and uploads
// The result in the slot is bit-identical to PfileReader::read_chunk_windows (same pieces, tests/test_pfile_reader.py).
// Slot i&1 is refilled while the other one trains.  A rank that leaves through exit() raises `abort` (atexit hook of the
// caller); a rank that dies WITHOUT running it (SIGKILL, a crash, _exit) is noticed by its peers' waits, which poll the
// registered pids a few times a second; and a wait that outlasts the group's time budget (BP_DP_TIMEOUT_S, the bound of
// the device-side waits and of the rendezvous barriers, x2 + 30 s: a chunk's training may legitimately sit between two
// waits) aborts as well.  A producer that hits a reader error leaves the text in the header for the main threads.
#pragma once
#include <atomic>
#include <new>
#include <errno.h>
#include <signal.h>
#include <string>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <time.h>
#include <unistd.h>
#include <vector>

#include "pfile_reader.h"

namespace bp {

// Rows of a chunk that rank `rank` of `world` trains on: its B = Bg/world frames of every full global minibatch
// (rows i*Bg + rank*B ... of minibatch i; the partial last minibatch is dropped as in BP_GPU.cu:315-318).
inline std::vector<int> shard_rows(int n_samples, int global_bunch, int world, int rank)
{
    const int B = global_bunch / world, nb = n_samples / global_bunch;
    std::vector<int> idx((size_t)nb * B);
    for (int i = 0; i < nb; ++i)
        for (int j = 0; j < B; ++j) idx[(size_t)i * B + j] = i * global_bunch + rank * B + j;
    return idx;
}

class ChunkRing {
public:
    struct View {                      // one ready chunk in a slot (valid until done(seq))
        int n_samples, n_frames, n_nat;
        const float *fea, *targ, *nat;
        const int *win_start, *targ_frame, *nat_row;
    };
    // Call in the parent BEFORE forking the ranks: the mapping is inherited.  frames_cap / samples_cap: the largest chunk
    // of the plan; nat_cap: an upper bound of noise-aware rows per chunk.
    ChunkRing(int world, int frames_cap, int samples_cap, int nat_cap, int D, int OD, bool nat)
        : world_(world), fcap_(frames_cap), scap_(samples_cap), ncap_(nat ? nat_cap : 0), D_(D), OD_(OD)
    {
        slot_floats_ = (size_t)fcap_ * D_ + (size_t)fcap_ * OD_ + (size_t)ncap_ * D_;
        slot_bytes_ = ((slot_floats_ + 3 * (size_t)scap_) * 4 + 4095) & ~(size_t)4095;
        bytes_ = 4096 + 2 * slot_bytes_;
        void *p = mmap(nullptr, bytes_, PROT_READ | PROT_WRITE, MAP_SHARED | MAP_ANONYMOUS, -1, 0);
        if (p == MAP_FAILED) die("cannot map %zu bytes of shared memory for the chunk ring", bytes_);
        base_ = (char *)p;
        hdr_ = new (base_) Hdr();
        hdr_->abort.store(0);
        hdr_->err[0] = 0;
        for (int r = 0; r < 8; ++r) hdr_->pid[r].store(0);
        {
            const char *e = getenv("BP_DP_TIMEOUT_S");
            const double v = e ? atof(e) : 60.0;
            timeout_s_ = 2.0 * (v > 0.5 ? v : 0.5) + 30.0;
        }
        for (int s = 0; s < 2; ++s) { hdr_->slot[s].tables_seq.store(-1); hdr_->slot[s].ready_seq.store(-1); hdr_->slot[s].converted.store(0); hdr_->slot[s].consumed.store(world); }
    }
    ~ChunkRing() { if (base_) munmap(base_, bytes_); }
    void abort() { if (hdr_) hdr_->abort.store(1); }
    // every rank, once, after the fork: lets the peers' waits notice a process that died without raising `abort`
    void register_rank(int rank) { if (rank >= 0 && rank < 8) hdr_->pid[rank].store((int)getpid()); }
    // first reader error a producer thread ran into (empty: none)
    std::string error() const { char b[sizeof(hdr_->err)]; memcpy(b, hdr_->err, sizeof(b)); b[sizeof(b) - 1] = 0; return std::string(b); }
    void fail(const std::string &msg)
    {
        int expected = 0;
        if (hdr_->err_set.compare_exchange_strong(expected, 1)) { strncpy(hdr_->err, msg.c_str(), sizeof(hdr_->err) - 1); hdr_->err[sizeof(hdr_->err) - 1] = 0; }
        hdr_->abort.store(1);
    }
    bool aborted() const { return hdr_->abort.load() != 0; }
    // where this rank's producer thread spent its time, seconds (tools/reader_ring_bench.cc): waiting for the slot to be consumed |
    // building the tables (rank 0) | waiting for the tables | converting its slice | waiting for the other slices (rank 0) | noise-aware rows
    struct Times { double wait_slot = 0, tables = 0, wait_tables = 0, convert = 0, wait_converted = 0, nat = 0; };
    const Times &times() const { return times_; }

    // The producer side of chunk number `seq` of the epoch (plan chunk `chunk_index`), run by EVERY rank in order
    // seq = 0, 1, ... on a helper thread: rank 0 publishes the tables, everyone converts a slice, rank 0 finishes.
    // Returns false when the group was aborted.
    bool produce(PfileReader &r, const PfileReader::Plan &p, int seq, int chunk_index, bool shuffle, int rank)
    {
        Slot &s = hdr_->slot[seq & 1];
        float *fea = slot_fea(seq & 1), *targ = fea + (size_t)fcap_ * D_, *nat = targ + (size_t)fcap_ * OD_;
        int *ws = (int *)(nat + (size_t)ncap_ * D_), *tf = ws + scap_, *nr = tf + scap_;
        const PfileReader::ChunkShape c = r.chunk_shape(p, chunk_index);
        if (c.n_frames > fcap_ || c.n_samples > scap_) { fail("chunk ring: chunk " + std::to_string(chunk_index) + " exceeds the planned capacity"); return false; }
        double tk = clock_s();
        auto lap = [&](double &acc) { const double n = clock_s(); acc += n - tk; tk = n; };
        if (rank == 0) {
            if (!wait([&] { return s.consumed.load() == world_; })) return false;       // previous tenant fully consumed
            lap(times_.wait_slot);
            s.converted.store(0); s.consumed.store(0);
            std::vector<int> seg_start, seg_sent;
            r.build_tables(p, chunk_index, shuffle, ws, tf, r.nat() ? nr : nullptr, seg_start, seg_sent);
            lap(times_.tables);
            if ((int)seg_start.size() > ncap_ && r.nat()) { fail("chunk ring: more noise-aware rows than planned"); return false; }
            s.n_samples = c.n_samples; s.n_frames = c.n_frames; s.n_nat = r.nat() ? (int)seg_start.size() : 0;
            seg_start_ = seg_start; seg_sent_ = seg_sent;
            s.tables_seq.store(seq);
        }
        if (!wait([&] { return s.tables_seq.load() == seq; })) return false;
        lap(times_.wait_tables);
        const int lo = (int)((long)c.n_frames * rank / world_), hi = (int)((long)c.n_frames * (rank + 1) / world_);
        {   // (this runs on a helper thread: no print-and-exit in here, the text goes to the main threads through the header)
            const std::string e = r.try_convert_frames(p, chunk_index, c.frame_st, lo, hi, fea, targ);
            if (!e.empty()) { fail(e); return false; }
        }
        lap(times_.convert);
        s.converted.fetch_add(1);
        if (rank == 0) {
            if (!wait([&] { return s.converted.load() == world_; })) return false;
            lap(times_.wait_converted);
            if (r.nat() && c.n_frames > 0) {
                const std::string e = r.try_nat_rows(p, chunk_index, fea, seg_start_, seg_sent_, nat);
                if (!e.empty()) { fail(e); return false; }
            }
            lap(times_.nat);
            s.ready_seq.store(seq);
        }
        return true;
    }
    // consumer side (every rank's main thread): wait for chunk `seq`, use it, then done(seq)
    bool acquire(int seq, View &v)
    {
        Slot &s = hdr_->slot[seq & 1];
        if (!wait([&] { return s.ready_seq.load() == seq; })) return false;
        float *fea = slot_fea(seq & 1), *targ = fea + (size_t)fcap_ * D_, *nat = targ + (size_t)fcap_ * OD_;
        int *ws = (int *)(nat + (size_t)ncap_ * D_);
        v.n_samples = s.n_samples; v.n_frames = s.n_frames; v.n_nat = s.n_nat;
        v.fea = fea; v.targ = targ; v.nat = s.n_nat ? nat : nullptr;
        v.win_start = ws; v.targ_frame = ws + scap_; v.nat_row = s.n_nat ? ws + 2 * scap_ : nullptr;
        return true;
    }
    void done(int seq) { hdr_->slot[seq & 1].consumed.fetch_add(1); }
    void *base() const { return base_; }        // the whole mapping, e.g. to pin it for DMA uploads (after the fork, per process)
    size_t bytes() const { return bytes_; }

private:
    struct Slot { std::atomic<int> tables_seq, converted, ready_seq, consumed; int n_samples, n_frames, n_nat; };
    struct Hdr { std::atomic<int> abort, err_set; std::atomic<int> pid[8]; Slot slot[2]; char err[256]; };
    float *slot_fea(int s) const { return (float *)(base_ + 4096 + (size_t)s * slot_bytes_); }
    // A rank is gone when its pid no longer exists -- or exists only as a ZOMBIE: the ranks are children of rank 0, which
    // reaps them at the very end, so a crashed child stays in the process table (kill(pid, 0) still succeeds) until then.
    static bool pid_gone(int pid)
    {
        if (kill((pid_t)pid, 0) != 0 && errno == ESRCH) return true;
        char path[64], buf[512];
        snprintf(path, sizeof path, "/proc/%d/stat", pid);
        FILE *f = fopen(path, "r");
        if (!f) return false;                                    // (no procfs: fall back to the time-out)
        const size_t n = fread(buf, 1, sizeof buf - 1, f);
        fclose(f);
        buf[n] = 0;
        const char *p = strrchr(buf, ')');                      // "pid (comm) S ...": the state letter follows the last ')'
        return p && p[1] == ' ' && (p[2] == 'Z' || p[2] == 'X');
    }
    template <class F> bool wait(F cond)
    {
        struct timespec t0; clock_gettime(CLOCK_MONOTONIC, &t0);
        for (unsigned spins = 0; !cond(); ++spins) {
            if (hdr_->abort.load()) return false;
            if ((spins & 1023) == 1023) {                        // every ~0.2 s
                for (int r = 0; r < world_ && r < 8; ++r) {      // a registered peer that no longer exists: nobody will wake us
                    const int pid = hdr_->pid[r].load();
                    if (pid > 0 && pid_gone(pid)) { fail("chunk ring: rank " + std::to_string(r) + " (process " + std::to_string(pid) + ") is gone"); return false; }
                }
                struct timespec t1; clock_gettime(CLOCK_MONOTONIC, &t1);
                if ((double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec) > timeout_s_) { fail("chunk ring: timed out waiting for a peer"); return false; }
            }
            usleep(spins < 64 ? 20 : 200);
        }
        return true;
    }
    static double clock_s() { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec; }
    Times times_;
    int world_, fcap_, scap_, ncap_, D_, OD_;
    double timeout_s_ = 150.0;
    size_t slot_floats_, slot_bytes_, bytes_;
    char *base_ = nullptr;
    Hdr *hdr_ = nullptr;
    std::vector<int> seg_start_, seg_sent_;       // rank 0's helper only
};

}  // namespace bp
